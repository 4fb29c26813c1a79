"""Dynamics model of the character for the HIP simulator (include/parc_sim.h parc_sim_model_t).

Everything Isaac Gym derives from the MJCF when the reference calls ``gym.load_asset`` / ``create_actor``
(envs/ig_char_env.py:92-137) is computed here on the host: link masses and inertias from geom volumes and
densities, PD gains from joint stiffness / damping, armature, joint ranges, motor gears as torque limits,
and a set of sample spheres per geom as collision geometry.
"""
import ctypes

import numpy as np

from . import _hip
from .anim.kin_char_model import GeomType, JointType

MAX_BODIES = 16
MAX_DOFS = 64
MAX_SPHERES = 64

c_i32 = ctypes.c_int32
c_f = ctypes.c_float


class SimModelS(ctypes.Structure):
    _fields_ = [("num_bodies", c_i32), ("dof_size", c_i32), ("num_spheres", c_i32), ("_pad", c_i32),
                ("parent", c_i32 * MAX_BODIES), ("joint_type", c_i32 * MAX_BODIES), ("dof_idx", c_i32 * MAX_BODIES),
                ("local_translation", (c_f * 3) * MAX_BODIES), ("local_rotation", (c_f * 4) * MAX_BODIES),
                ("joint_axis", (c_f * 3) * MAX_BODIES),
                ("mass", c_f * MAX_BODIES), ("com", (c_f * 3) * MAX_BODIES), ("inertia_o", (c_f * 6) * MAX_BODIES),
                ("kp", c_f * MAX_DOFS), ("kd", c_f * MAX_DOFS), ("armature", c_f * MAX_DOFS),
                ("limit_lo", c_f * MAX_DOFS), ("limit_hi", c_f * MAX_DOFS), ("effort", c_f * MAX_DOFS),
                ("sph_body", c_i32 * MAX_SPHERES), ("sph_pos", (c_f * 3) * MAX_SPHERES), ("sph_radius", c_f * MAX_SPHERES),
                ("gravity", c_f),
                ("contact_kn", c_f), ("contact_cn", c_f), ("contact_ct", c_f), ("friction_mu", c_f), ("contact_max_pen", c_f),
                ("limit_kp", c_f), ("limit_kd", c_f), ("max_angular_velocity", c_f), ("angular_damping", c_f),
                ("cap_p0", (c_f * 3) * MAX_BODIES), ("cap_p1", (c_f * 3) * MAX_BODIES), ("cap_radius", c_f * MAX_BODIES),
                ("self_mask", ctypes.c_uint32 * MAX_BODIES)]


def _quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _axis_frame(d):
    """Rotation whose z axis is the unit vector d."""
    d = d / np.linalg.norm(d)
    a = np.array([1.0, 0, 0]) if abs(d[0]) < 0.9 else np.array([0, 1.0, 0])
    x = np.cross(a, d)
    x /= np.linalg.norm(x)
    y = np.cross(d, x)
    return np.stack([x, y, d], axis=1)


def geom_mass_properties(g):
    """(mass, com[3], inertia about com [3,3]) of one geom in the body frame."""
    rho = g._density
    if g._shape_type == GeomType.SPHERE:
        r = float(np.atleast_1d(g._dims)[0])
        m = rho * 4.0 / 3.0 * np.pi * r ** 3
        return m, g._offset.copy(), np.eye(3) * (0.4 * m * r * r)
    if g._shape_type == GeomType.BOX:
        a, b, c = g._dims
        m = rho * 8.0 * a * b * c
        I = np.diag([m / 3.0 * (b * b + c * c), m / 3.0 * (a * a + c * c), m / 3.0 * (a * a + b * b)])
        R = _quat_to_mat(g._quat) if g._quat is not None else np.eye(3)
        return m, g._offset.copy(), R @ I @ R.T
    if g._shape_type == GeomType.CAPSULE:
        r = float(g._radius)
        L = float(np.linalg.norm(g._dims))
        mc = rho * np.pi * r * r * L
        ms = rho * 4.0 / 3.0 * np.pi * r ** 3
        m = mc + ms
        Ia = 0.5 * mc * r * r + 0.4 * ms * r * r
        It = mc * (L * L / 12.0 + r * r / 4.0) + ms * (0.4 * r * r + L * L / 4.0 + 3.0 * L * r / 8.0)
        F = _axis_frame(g._dims)
        I = F @ np.diag([It, It, Ia]) @ F.T
        return m, g._offset + 0.5 * g._dims, I
    raise NotImplementedError(g._shape_type)


def geom_sample_spheres(g):
    """Collision proxy of a geom: list of (centre[3], radius)."""
    if g._shape_type == GeomType.SPHERE:
        return [(g._offset.copy(), float(np.atleast_1d(g._dims)[0]))]
    if g._shape_type == GeomType.CAPSULE:
        L = float(np.linalg.norm(g._dims))
        n = 3 if L > 0.15 else 2
        return [(g._offset + t * g._dims, float(g._radius)) for t in np.linspace(0.0, 1.0, n)]
    if g._shape_type == GeomType.BOX:
        R = _quat_to_mat(g._quat) if g._quat is not None else np.eye(3)
        a, b, c = g._dims
        pts = []
        for sx in (-1, 1):
            for sy in (-1, 1):
                for sz in (-1, 1):
                    pts.append((g._offset + R @ np.array([sx * a, sy * b, sz * c]), 0.0))
        return pts
    raise NotImplementedError(g._shape_type)


def body_capsule(geoms):
    """One collision capsule per body for link-link contact: (p0[3], p1[3], radius) in the body frame.  A single capsule / sphere geom
    is taken as it is; a box becomes the capsule along its longest axis; a body made of several geoms (pelvis: two spheres, torso:
    chest sphere + clavicles) gets the segment from its largest sphere to the centroid of the other geoms, with that sphere's radius."""
    def ends(g):
        if g._shape_type == GeomType.SPHERE:
            return g._offset.copy(), g._offset.copy(), float(np.atleast_1d(g._dims)[0])
        if g._shape_type == GeomType.CAPSULE:
            return g._offset.copy(), g._offset + g._dims, float(g._radius)
        if g._shape_type == GeomType.BOX:
            R = _quat_to_mat(g._quat) if g._quat is not None else np.eye(3)
            half = np.asarray(g._dims, np.float64)
            k = int(np.argmax(half))
            r = float(np.sort(half)[1])                   # the middle half extent: covers the box's width, overshoots its height a little
            d = R[:, k] * max(half[k] - r, 0.0)
            return g._offset - d, g._offset + d, r
        raise NotImplementedError(g._shape_type)
    if len(geoms) == 0:
        return np.zeros(3), np.zeros(3), 0.0
    if len(geoms) == 1:
        return ends(geoms[0])
    spheres = [g for g in geoms if g._shape_type == GeomType.SPHERE]
    if spheres:
        big = max(spheres, key=lambda g: float(np.atleast_1d(g._dims)[0]))
        rest = [g for g in geoms if g is not big]
        cen = np.mean([0.5 * (ends(g)[0] + ends(g)[1]) for g in rest], axis=0)
        return big._offset.copy(), cen, float(np.atleast_1d(big._dims)[0])
    pts = [e for g in geoms for e in ends(g)[:2]]
    i, j = max(((i, j) for i in range(len(pts)) for j in range(i + 1, len(pts))), key=lambda ij: np.linalg.norm(pts[ij[0]] - pts[ij[1]]))
    return pts[i], pts[j], max(ends(g)[2] for g in geoms)


class SimModel:
    def __init__(self, kin_char_model, gravity=9.81, contact_kn=4.0e4, contact_cn=1.0e3, contact_ct=3.0e3, friction_mu=1.0,
                 contact_max_pen=0.04, limit_kp=2.0e3, limit_kd=50.0, max_angular_velocity=100.0, angular_damping=0.01, self_collision=True):
        km = kin_char_model
        B, D = km.get_num_joints(), km.get_dof_size()
        assert B <= MAX_BODIES and D <= MAX_DOFS
        s = SimModelS()
        s.num_bodies, s.dof_size = B, D
        par = km._parent_indices.cpu().numpy()
        lt = km._local_translation.cpu().numpy()
        lr = km._local_rotation.cpu().numpy()
        self.body_mass = np.zeros(B)
        self.body_com = np.zeros((B, 3))
        self.body_inertia_com = np.zeros((B, 3, 3))
        spheres = []
        gears = getattr(km, "_motor_gears", {})
        for b in range(B):
            s.parent[b] = int(par[b])
            jt = km._joints[b]
            s.joint_type[b] = jt.joint_type.value
            s.dof_idx[b] = int(jt.dof_idx)
            for k in range(3):
                s.local_translation[b][k] = float(lt[b, k])
            for k in range(4):
                s.local_rotation[b][k] = float(lr[b, k])
            if jt.axis is not None:
                ax = jt.axis.cpu().numpy()
                for k in range(3):
                    s.joint_axis[b][k] = float(ax[k])
            # mass properties
            m_tot, mc = 0.0, np.zeros(3)
            parts = [geom_mass_properties(g) for g in km.get_geoms(b)]
            for m, c, _ in parts:
                m_tot += m
                mc += m * c
            com = mc / m_tot
            I_com = np.zeros((3, 3))
            for m, c, I in parts:
                d = c - com
                I_com += I + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            I_o = I_com + m_tot * (np.dot(com, com) * np.eye(3) - np.outer(com, com))
            self.body_mass[b], self.body_com[b], self.body_inertia_com[b] = m_tot, com, I_com
            s.mass[b] = m_tot
            for k in range(3):
                s.com[b][k] = float(com[k])
            for k, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
                s.inertia_o[b][k] = float(I_o[i, j])
            # drives
            dd = jt.get_dof_dim()
            if dd > 0:
                lim = jt.limits.detach().cpu().numpy().reshape(-1, 2)
                names = getattr(jt, "dof_names", [None] * dd)
                for k in range(dd):
                    d = jt.dof_idx + k
                    s.kp[d] = float(jt.stiffness[k])
                    s.kd[d] = float(jt.damping[k])
                    s.armature[d] = float(jt.armature[k])
                    s.limit_lo[d] = float(lim[k, 0])
                    s.limit_hi[d] = float(lim[k, 1])
                    s.effort[d] = float(gears.get(names[k], 0.0))
            for g in km.get_geoms(b):
                for c, r in geom_sample_spheres(g):
                    spheres.append((b, c, r))
        # link-link contact: every pair of bodies that is not joined by a joint (collision filter 0, envs/ig_char_env.py:105-113)
        self.capsules = []
        for b in range(B):
            p0, p1, r = body_capsule(km.get_geoms(b))
            self.capsules.append((p0, p1, r))
            for k in range(3):
                s.cap_p0[b][k], s.cap_p1[b][k] = float(p0[k]), float(p1[k])
            s.cap_radius[b] = float(r)
            mask = 0
            for j in range(B):
                if self_collision and j != b and int(par[b]) != j and int(par[j]) != b:
                    mask |= 1 << j
            s.self_mask[b] = mask
        assert len(spheres) <= MAX_SPHERES, len(spheres)
        s.num_spheres = len(spheres)
        for k, (b, c, r) in enumerate(spheres):
            s.sph_body[k] = b
            for a in range(3):
                s.sph_pos[k][a] = float(c[a])
            s.sph_radius[k] = float(r)
        s.gravity = gravity
        s.contact_kn, s.contact_cn, s.contact_ct = contact_kn, contact_cn, contact_ct
        s.friction_mu, s.contact_max_pen = friction_mu, contact_max_pen
        s.limit_kp, s.limit_kd, s.max_angular_velocity = limit_kp, limit_kd, max_angular_velocity
        s.angular_damping = angular_damping          # envs/ig_char_env.py:141-142: angular_damping 0.01, max_angular_velocity 100
        self.struct = s
        self.total_mass = float(self.body_mass.sum())
        self._device_copy = None

    def device_ptr(self, device):
        """Device copy of the struct (uint8 tensor kept alive by this object)."""
        import torch
        if self._device_copy is None or str(self._device_copy.device) != str(device):
            raw = bytes(self.struct)
            self._device_copy = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        return _hip.c_vp(self._device_copy.data_ptr())

    def invalidate(self):
        self._device_copy = None


def action_bounds_pd(kin_char_model):
    """PD action bounds from the joint ranges (reference: IGCharEnv._build_action_bounds_pd, envs/ig_char_env.py:308-348;
    the reference reads the ranges back from Isaac Gym, here they come from the MJCF directly)."""
    km = kin_char_model
    D = km.get_dof_size()
    lo_lim = km._lower_dof_limits.cpu().numpy().astype(np.float64)
    hi_lim = km._upper_dof_limits.cpu().numpy().astype(np.float64)
    low, high = np.zeros(D), np.zeros(D)
    for j in range(1, km.get_num_joints()):
        jt = km.get_joint(j)
        dd = jt.get_dof_dim()
        if dd == 0:
            continue
        d0 = jt.dof_idx
        jl, jh = lo_lim[d0:d0 + dd], hi_lim[d0:d0 + dd]
        if dd == 3:
            scale = 1.2 * max(np.max(np.abs(jl)), np.max(np.abs(jh)))
            low[d0:d0 + dd], high[d0:d0 + dd] = -scale, scale
        else:
            mid = 0.5 * (jh + jl)
            scale = 0.7 * (jh - jl)
            low[d0:d0 + dd], high[d0:d0 + dd] = mid - scale, mid + scale
    return low, high
