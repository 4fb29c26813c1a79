"""Kinematic character model: MJCF -> joint tree, dof <-> quaternion, forward kinematics.

Host-side mirror of the reference's ``anim/kin_char_model.py`` (KinCharModel :142-449, Joint :17-100):
same constructor, method names and attribute names (``_body_names``, ``_parent_indices``,
``_local_translation``, ``_local_rotation``, ``_joints``, ``_lower_dof_limits`` ...), because the env,
agent and recorder reach into them.  The batch math (dof_to_rot / rot_to_dof / forward_kinematics,
reference :478-541) runs in the HIP kernels of parc_amd/csrc/parc_kin.hip; there is no CPU fallback.

The parser additionally keeps what the simulator needs from the MJCF and the reference only hands to
Isaac Gym: geoms with densities, joint stiffness / damping / armature, motor gears.
"""
import enum
import os
import xml.etree.ElementTree as ET

import numpy as np
import torch

from .. import _hip
from ..util import torch_util


class JointType(enum.Enum):
    ROOT = 0
    HINGE = 1
    SPHERICAL = 2
    FIXED = 3


class GeomType(enum.Enum):
    BOX = 0
    SPHERE = 1
    CAPSULE = 2
    CYLINDER = 3
    MESH = 4


_DOF_DIM = {JointType.ROOT: 0, JointType.HINGE: 1, JointType.SPHERICAL: 3, JointType.FIXED: 0}


class Joint:
    def __init__(self, name, joint_type, axis, limits=None, stiffness=None, damping=None, armature=None):
        self.name = name
        self.joint_type = joint_type
        self.axis = axis
        self.dof_idx = -1
        self.limits = limits
        # per-dof actuator parameters straight from the MJCF (numpy, length = dof dim)
        self.stiffness = stiffness
        self.damping = damping
        self.armature = armature

    def get_dof_dim(self):
        return _DOF_DIM[self.joint_type]

    def get_joint_dof(self, dof):
        return dof[..., self.dof_idx:self.dof_idx + self.get_dof_dim()]

    def set_joint_dof(self, j_dof, out_dof):
        out_dof[..., self.dof_idx:self.dof_idx + self.get_dof_dim()] = j_dof


class Geom:
    """One collision primitive of a body, in the body frame."""

    def __init__(self, shape_type, offset, dims, quat=None, radius=None, density=1000.0, name=None):
        self._shape_type = shape_type
        self._offset = np.asarray(offset, dtype=np.float64)   # sphere/box centre, capsule start point
        self._dims = np.asarray(dims, dtype=np.float64)       # sphere [r], box half-extents, capsule end-start
        self._quat = quat
        self._radius = radius
        self._density = float(density)
        self._name = name


def _floats(text, default=None):
    if text is None:
        return default
    return np.array([float(x) for x in text.split()], dtype=np.float64)


class KinCharModel:
    def __init__(self, device):
        self._device = torch.device(device)

    # ------------------------------------------------------------------ construction
    def init(self, body_names, parent_indices, local_translation, local_rotation, joints, geoms=None):
        nb = len(body_names)
        assert len(parent_indices) == nb and len(local_translation) == nb and len(local_rotation) == nb and len(joints) == nb
        assert nb <= _hip.MAX_BODIES, "the HIP kernels hold one body per lane of a 16-lane group"
        self._body_names = list(body_names)
        self._parent_indices = torch.as_tensor(np.asarray(parent_indices), dtype=torch.long, device=self._device)
        self._local_translation = torch.as_tensor(np.asarray(local_translation), dtype=torch.float32, device=self._device)
        self._original_local_translation = self._local_translation.clone()
        self._local_rotation = torch.as_tensor(np.asarray(local_rotation), dtype=torch.float32, device=self._device)
        self._joints = joints
        d = 0
        for j in joints:
            j.dof_idx = d
            d += j.get_dof_dim()
        self._dof_size = d
        assert d <= _hip.MAX_DOFS
        self._name_body_map = {n: i for i, n in enumerate(self._body_names)}
        lo, hi = [], []
        for j in joints:
            if j.limits is None:
                continue
            lim = (j.limits.detach().cpu().numpy() if isinstance(j.limits, torch.Tensor) else np.asarray(j.limits)).astype(np.float32).reshape(-1, 2)
            lo.append(lim[:, 0])
            hi.append(lim[:, 1])
        self._lower_dof_limits = torch.as_tensor(np.concatenate(lo), device=self._device) if lo else []
        self._upper_dof_limits = torch.as_tensor(np.concatenate(hi), device=self._device) if hi else []
        self._geoms = geoms
        self._c_struct = None

    def load_char_file(self, char_file):
        """Parse an MJCF humanoid (reference: load_char_file :206-449).  Bodies are numbered depth-first in
        document order -- the order Isaac Gym uses too (reference check: envs/ig_char_env.py:219-227)."""
        root = ET.parse(char_file).getroot()
        world = root.find("worldbody")
        assert world is not None
        top = world.find("body")
        assert top is not None
        defaults = self._parse_defaults(root)
        self._char_file = char_file
        self._motor_gears = {m.attrib.get("joint"): float(m.attrib.get("gear", "1").split()[0])
                             for m in root.iter("motor")}

        names, parents, trans, rots, joints, geoms = [], [], [], [], [], []

        def visit(node, parent):
            idx = len(names)
            names.append(node.attrib.get("name"))
            parents.append(parent)
            trans.append(_floats(node.attrib.get("pos"), np.zeros(3)))
            quat = node.attrib.get("quat")
            if quat is None:
                rots.append(np.array([0.0, 0.0, 0.0, 1.0]))
            else:
                w, x, y, z = _floats(quat)          # MJCF stores w first
                rots.append(np.array([x, y, z, w]))
            joints.append(self._make_joint(node, idx, defaults))
            geoms.append([self._make_geom(g, defaults) for g in node.findall("geom")])
            for child in node.findall("body"):
                visit(child, idx)

        visit(top, -1)
        self.init(names, parents, trans, rots, joints, geoms)

    @staticmethod
    def _parse_defaults(root):
        out = {"joint": {}, "geom": {}}
        top = root.find("default")
        if top is None:
            return out
        for scope in [top] + top.findall("default"):
            for kind in ("joint", "geom"):
                el = scope.find(kind)
                if el is not None:
                    out[kind].update(el.attrib)
        return out

    def _make_joint(self, node, body_index, defaults):
        if body_index == 0:
            return Joint("root", JointType.ROOT, None)
        els = node.findall("joint")
        dj = defaults["joint"]

        def attr(el, key, fallback):
            return float(el.attrib.get(key, dj.get(key, fallback)))

        for el in els:
            jtype = el.attrib.get("type", dj.get("type", "hinge"))
            assert jtype == "hinge", "Unsupported joint type: {}".format(jtype)
            pos = _floats(el.attrib.get("pos"))
            assert pos is None or not np.any(pos), "Joint offsets are not supported"
            assert el.attrib.get("range") is not None, "Need joint limits"
        if len(els) == 0:
            return Joint(node.attrib.get("name"), JointType.FIXED, None)
        limits = np.stack([_floats(el.attrib["range"]) for el in els]).astype(np.float32)
        limits = limits * np.float32(np.pi / 180.0)
        kp = np.array([attr(el, "stiffness", 0.0) for el in els])
        kd = np.array([attr(el, "damping", 0.0) for el in els])
        arm = np.array([attr(el, "armature", 0.0) for el in els])
        if len(els) == 1:
            axis = torch.tensor(_floats(els[0].attrib["axis"]), dtype=torch.float32, device=self._device)
            j = Joint(els[0].attrib.get("name"), JointType.HINGE, axis,
                      limits=torch.as_tensor(limits[0], device=self._device), stiffness=kp, damping=kd, armature=arm)
            j.dof_names = [els[0].attrib.get("name")]
            return j
        assert len(els) == 3, "Series joints are not supported."
        name = els[0].attrib.get("name")
        name = name[:name.rfind("_")]
        # three orthogonal hinges = one spherical joint whose dofs are an exponential map (:654-695)
        j = Joint(name, JointType.SPHERICAL, None, limits=torch.as_tensor(limits, device=self._device),
                  stiffness=kp, damping=kd, armature=arm)
        j.dof_names = [el.attrib.get("name") for el in els]
        return j

    @staticmethod
    def _make_geom(el, defaults):
        dg = defaults["geom"]
        gtype = el.attrib.get("type", dg.get("type", "sphere"))
        density = float(el.attrib.get("density", dg.get("density", 1000.0)))
        quat = _floats(el.attrib.get("quat"), np.array([1.0, 0.0, 0.0, 0.0]))
        quat = np.array([quat[1], quat[2], quat[3], quat[0]])
        name = el.attrib.get("name")
        if gtype == "sphere":
            return Geom(GeomType.SPHERE, _floats(el.attrib.get("pos"), np.zeros(3)), _floats(el.attrib.get("size"), np.array([0.1])),
                        quat=quat, density=density, name=name)
        if gtype == "box":
            return Geom(GeomType.BOX, _floats(el.attrib.get("pos"), np.zeros(3)), _floats(el.attrib.get("size")),
                        quat=quat, density=density, name=name)
        if gtype == "capsule":
            ft = _floats(el.attrib.get("fromto"))
            return Geom(GeomType.CAPSULE, ft[0:3], ft[3:6] - ft[0:3], quat=quat, radius=float(el.attrib.get("size")),
                        density=density, name=name)
        raise AssertionError("unsupported geom type {}".format(gtype))

    # ------------------------------------------------------------------ queries
    def get_body_names(self):
        return self._body_names

    def get_joint(self, j):
        assert j > 0
        return self._joints[j]

    def get_parent_id(self, j):
        return self._parent_indices[j]

    def get_dof_size(self):
        return self._dof_size

    def get_joint_dof_idx(self, j):
        return self.get_joint(j).dof_idx

    def get_joint_dof_dim(self, j):
        return self.get_joint(j).get_dof_dim()

    def get_num_joints(self):
        return len(self._joints)

    def get_num_non_root_joints(self):
        return len(self._joints) - 1

    def get_body_name(self, body_id):
        return self._body_names[body_id]

    def get_body_id(self, body_name):
        assert body_name in self._name_body_map
        return self._name_body_map[body_name]

    def get_joint_id(self, body_name):
        return self.get_body_id(body_name) - 1

    def get_geoms(self, body_id):
        return self._geoms[body_id]

    # ------------------------------------------------------------------ C-ABI view
    def c_struct(self):
        """parc_char_model_t (include/parc_hip.h) of this tree."""
        if self._c_struct is None:
            s = _hip.CharModelS()
            nb = self.get_num_joints()
            s.num_bodies = nb
            s.dof_size = self._dof_size
            par = self._parent_indices.cpu().numpy()
            lt = self._local_translation.cpu().numpy()
            lr = self._local_rotation.cpu().numpy()
            depth = np.zeros(nb, dtype=np.int32)
            for b in range(1, nb):
                depth[b] = depth[par[b]] + 1
            s.max_depth = int(depth.max())
            for b in range(nb):
                s.parent[b] = int(par[b])
                s.joint_type[b] = self._joints[b].joint_type.value
                s.dof_idx[b] = int(self._joints[b].dof_idx)
                s.depth[b] = int(depth[b])
                for k in range(3):
                    s.local_translation[b][k] = float(lt[b, k])
                for k in range(4):
                    s.local_rotation[b][k] = float(lr[b, k])
                ax = self._joints[b].axis
                if ax is not None:
                    axn = ax.cpu().numpy()
                    for k in range(3):
                        s.joint_axis[b][k] = float(axn[k])
            self._c_struct = s
        return self._c_struct

    # ------------------------------------------------------------------ batch math (HIP)
    def dof_to_rot(self, dof):
        """[..., D] -> [..., J, 4]   (reference :478-491)"""
        lead = list(dof.shape[:-1])
        flat = dof.reshape(-1, self._dof_size).contiguous().float()
        n = flat.shape[0]
        out = torch.empty((n, self.get_num_joints() - 1, 4), dtype=torch.float32, device=dof.device)
        _hip.check(_hip.lib().parc_dof_to_rot(_hip.stream(), self.c_struct(), n, _hip.ptr(flat), _hip.ptr(out)), "parc_dof_to_rot")
        return out.reshape(lead + [self.get_num_joints() - 1, 4])

    def rot_to_dof(self, rot):
        """[..., J, 4] -> [..., D]   (reference :493-507)"""
        lead = list(rot.shape[:-2])
        J = self.get_num_joints() - 1
        flat = rot.reshape(-1, J, 4).contiguous().float()
        n = flat.shape[0]
        out = torch.empty((n, self._dof_size), dtype=torch.float32, device=rot.device)
        _hip.check(_hip.lib().parc_rot_to_dof(_hip.stream(), self.c_struct(), n, _hip.ptr(flat), _hip.ptr(out)), "parc_rot_to_dof")
        return out.reshape(lead + [self._dof_size])

    def forward_kinematics(self, root_pos, root_rot, joint_rot):
        """-> body_pos [..., B, 3], body_rot [..., B, 4]   (reference :509-541)"""
        lead = list(root_pos.shape[:-1])
        B = self.get_num_joints()
        rp = root_pos.reshape(-1, 3).contiguous().float()
        rr = root_rot.reshape(-1, 4).contiguous().float()
        jr = joint_rot.reshape(-1, B - 1, 4).contiguous().float()
        n = rp.shape[0]
        bp = torch.empty((n, B, 3), dtype=torch.float32, device=rp.device)
        br = torch.empty((n, B, 4), dtype=torch.float32, device=rp.device)
        _hip.check(_hip.lib().parc_forward_kinematics(_hip.stream(), self.c_struct(), n, _hip.ptr(rp), _hip.ptr(rr), _hip.ptr(jr),
                                                      _hip.ptr(bp), _hip.ptr(br)), "parc_forward_kinematics")
        return bp.reshape(lead + [B, 3]), br.reshape(lead + [B, 4])

    # ------------------------------------------------------------------ batch math (torch, differentiable)
    # The same maps as the kernels above, written with torch ops for callers that need gradients with respect to the pose (the
    # terrain-penetration loss of the motion optimiser).  Level-batched: bodies of one tree depth are composed in one shot.
    def _torch_tables(self):
        if getattr(self, "_tt", None) is None:
            J = self.get_num_joints() - 1
            kind = torch.zeros(J, dtype=torch.long)                 # 0 fixed, 1 hinge, 3 spherical (= dof count)
            first = torch.zeros(J, dtype=torch.long)
            axis = torch.zeros((J, 3), dtype=torch.float32)
            for j in range(1, J + 1):
                jt = self._joints[j]
                n = jt.get_dof_dim()
                kind[j - 1], first[j - 1] = n, jt.dof_idx
                if n == 1:
                    ax = jt.axis.detach().cpu().numpy() if torch.is_tensor(jt.axis) else np.asarray(jt.axis)
                    axis[j - 1] = torch.as_tensor(np.asarray(ax, dtype=np.float32).reshape(3))
            par = self._parent_indices.cpu().tolist()
            depth = [0] * (J + 1)
            for b in range(1, J + 1):
                depth[b] = depth[par[b]] + 1
            levels = [[b for b in range(1, J + 1) if depth[b] == d] for d in range(1, max(depth) + 1)]
            dev = self._device
            self._tt = {"hinge": (kind == 1).nonzero().flatten().to(dev), "sph": (kind == 3).nonzero().flatten().to(dev),
                        "first": first.to(dev), "axis": axis.to(dev),
                        # per tree level: body indices (tensor and list) and their parents' indices
                        "levels": [(torch.tensor(l, device=dev), list(l), torch.tensor([par[b] for b in l], device=dev)) for l in levels]}
        return self._tt

    def dof_to_rot_torch(self, dof):
        """[..., D] -> [..., J, 4] with autograd (hinge: axis-angle, spherical: exponential map, fixed: identity)"""
        t = self._torch_tables()
        J = self.get_num_joints() - 1
        rot = torch.zeros(dof.shape[:-1] + (J, 4), dtype=dof.dtype, device=dof.device)
        ident = torch.zeros_like(rot)
        ident[..., 3] = 1.0
        parts = []
        if t["hinge"].numel():
            ang = dof[..., t["first"][t["hinge"]]] / 2                       # [..., H]
            ax = t["axis"][t["hinge"]]
            ax = ax / torch.linalg.vector_norm(ax, dim=-1, keepdim=True).clamp(min=1e-9)
            q = torch.cat([ax * ang.sin().unsqueeze(-1), ang.cos().unsqueeze(-1)], dim=-1)
            parts.append((t["hinge"], q / torch.linalg.vector_norm(q, dim=-1, keepdim=True).clamp(min=1e-9)))
        if t["sph"].numel():
            idx = t["first"][t["sph"]].unsqueeze(-1) + torch.arange(3, device=dof.device)          # [S, 3]
            parts.append((t["sph"], self._exp_map_to_quat_torch(dof[..., idx])))
        out = ident
        for where, q in parts:
            out = out.index_copy(-2, where, q)
        return out

    @staticmethod
    def _exp_map_to_quat_torch(e):
        ang = torch.linalg.vector_norm(e, dim=-1)
        safe = ang.clamp(min=1e-20)
        axis = e / safe.unsqueeze(-1)
        ang = torch.atan2(torch.sin(ang), torch.cos(ang))
        big = ang.abs() > 1e-5
        zaxis = torch.zeros_like(e)
        zaxis[..., 2] = 1
        axis = torch.where(big.unsqueeze(-1), axis, zaxis)
        half = (torch.where(big, ang, torch.zeros_like(ang)) / 2).unsqueeze(-1)
        q = torch.cat([axis / torch.linalg.vector_norm(axis, dim=-1, keepdim=True).clamp(min=1e-9) * half.sin(), half.cos()], dim=-1)
        return q / torch.linalg.vector_norm(q, dim=-1, keepdim=True).clamp(min=1e-9)

    def forward_kinematics_torch(self, root_pos, root_rot, joint_rot):
        """-> body_pos [..., B, 3], body_rot [..., B, 4] with autograd; one composition per tree level"""
        t = self._torch_tables()
        B = self.get_num_joints()
        lead = root_pos.shape[:-1]
        local = torch_util.quat_mul_compact(self._local_rotation[1:].expand(lead + (B - 1, 4)), joint_rot)       # local_rot * joint_rot
        pos = root_pos.unsqueeze(-2).expand(lead + (B, 3))
        rot = root_rot.unsqueeze(-2).expand(lead + (B, 4))
        for bodies, _, parents in t["levels"]:
            ppos, prot = pos[..., parents, :], rot[..., parents, :]
            lt = self._local_translation[bodies].expand(ppos.shape)
            pos = pos.index_copy(-2, bodies, ppos + torch_util.quat_rotate(prot, lt))
            rot = rot.index_copy(-2, bodies, torch_util.quat_mul_compact(prot, local[..., bodies - 1, :]))
        return pos, rot

    def pose_chain(self, root_pos, root_exp, dof):
        """(root position [T, 3], root exponential map [T, 3], joint dofs [T, D]) -> (root quaternion [T, 4], joint rotations [T, J, 4],
        body positions [T, B, 3], body rotations [T, B, 4]), differentiable: forward and vector-Jacobian product are one HIP launch each
        (parc_pose_chain_forward / _backward) instead of the ~190 autograd nodes of exp_map_to_quat + dof_to_rot_torch +
        forward_kinematics_torch, whose values and gradients they reproduce."""
        return _PoseChain.apply(self, root_pos, root_exp, dof)

    def apply_joint_dof_limits(self, joint_dofs):
        return torch.minimum(torch.maximum(joint_dofs, self._lower_dof_limits), self._upper_dof_limits)


class _PoseChain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, km, root_pos, root_exp, dof):
        B, D = km.get_num_joints(), km.get_dof_size()
        n = int(root_pos.shape[0])
        assert root_pos.shape == (n, 3) and root_exp.shape == (n, 3) and dof.shape == (n, D)
        rp, re, dd = (t.detach().to(torch.float32).contiguous() for t in (root_pos, root_exp, dof))
        dev = rp.device
        rq = torch.empty((n, 4), dtype=torch.float32, device=dev)
        jr = torch.empty((n, B - 1, 4), dtype=torch.float32, device=dev)
        bp = torch.empty((n, B, 3), dtype=torch.float32, device=dev)
        br = torch.empty((n, B, 4), dtype=torch.float32, device=dev)
        _hip.check(_hip.lib().parc_pose_chain_forward(_hip.stream(), km.c_struct(), n, _hip.ptr(rp), _hip.ptr(re), _hip.ptr(dd), _hip.ptr(rq),
                                                      _hip.ptr(jr), _hip.ptr(bp), _hip.ptr(br)), "parc_pose_chain_forward")
        ctx.km = km
        ctx.save_for_backward(re, dd)
        return rq, jr, bp, br

    @staticmethod
    def backward(ctx, g_rq, g_jr, g_bp, g_br):
        km = ctx.km
        re, dd = ctx.saved_tensors
        n = int(re.shape[0])
        g = [x.to(torch.float32).contiguous() for x in (g_rq, g_jr, g_bp, g_br)]
        g_rp, g_re, g_dd = torch.empty_like(re), torch.empty_like(re), torch.empty_like(dd)
        _hip.check(_hip.lib().parc_pose_chain_backward(_hip.stream(), km.c_struct(), n, _hip.ptr(re), _hip.ptr(dd), _hip.ptr(g[0]), _hip.ptr(g[1]),
                                                       _hip.ptr(g[2]), _hip.ptr(g[3]), _hip.ptr(g_rp), _hip.ptr(g_re), _hip.ptr(g_dd)),
                   "parc_pose_chain_backward")
        return None, g_rp, g_re, g_dd


def default_char_file():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "humanoid.xml")
