"""Stage 2's contact-aware motion optimiser on MI355X.

Mirror of the reference's ``tools/motion_opt/motion_optimization.py``: ``LossType``, ``BodyConstraint``,
``compute_approx_body_constraints`` (:34-181), ``motion_terrain_contact_loss`` (:183-395) and ``motion_contact_optimization``
(:404-500) with the same arguments, loss definitions and Adam descent, so ``parc_2_kin_gen.py:445`` / ``optimize_motions.py:157``
call it unchanged.  What differs is the mechanics:

* all 15 bodies' sample points go through the terrain query together: two launches of ``parc_points_hf_sdf`` per evaluation (inside /
  outside distance) over ``[frames x points]`` instead of 30 calls that each materialise ``[points, cells, 3]`` temporaries; the
  kernel reports the arg-min column, and autograd sees the distance to that one column only (terrain_util.points_hf_sdf);
* the pose chain (exponential maps -> quaternions -> forward kinematics) and its adjoint are one HIP launch each
  (KinCharModel.pose_chain: parc_pose_chain_forward / _backward) instead of ~190 autograd nodes per evaluation;
* nothing in an iteration reads the device back (the reference takes nine ``.item()`` per iteration for its loss dict); the terms
  are read at the logging stride only, so a whole iteration (forward, backward, Adam) is captured ONCE in a hipGraph and replayed.

Checked against fixture G20 (the reference's loss terms, gradient, body constraints and a 40-iteration descent).
"""
import enum
import sys
import time

import numpy as np
import torch

from ...util import geom_util, terrain_util, torch_util
from ...util import logger as logger_mod
from ...anim import kin_char_model


class LossType(enum.Enum):
    ROOT_POS_LOSS = 0
    ROOT_ROT_LOSS = 1
    JOINT_ROT_LOSS = 2
    SMOOTHNESS_LOSS = 3
    PENETRATION_LOSS = 4
    CONTACT_LOSS = 5
    SLIDING_LOSS = 6
    BODY_CONSTRAINT_LOSS = 7
    JERK_LOSS = 8
    LOOPING_LOSS = 9


# order of the stacked term tensor the loss evaluation returns (= the keys of the reference's loss dict, in its insertion order)
_TERM_ORDER = (LossType.ROOT_POS_LOSS, LossType.ROOT_ROT_LOSS, LossType.JOINT_ROT_LOSS, LossType.SMOOTHNESS_LOSS, LossType.PENETRATION_LOSS,
               LossType.CONTACT_LOSS, LossType.SLIDING_LOSS, LossType.JERK_LOSS, LossType.BODY_CONSTRAINT_LOSS)


class BodyConstraint:
    """A body (its sole / its sphere) should touch `constraint_point` over frames [start_frame_idx, end_frame_idx]."""
    start_frame_idx = 0
    end_frame_idx = 0
    constraint_point = None      # 3D torch vector


# motion files carry these objects under the key "opt:body_constraints" (optimize_motions.py:189-190); the reference's pickle.load
# resolves the class by this path
REFERENCE_MODULE = "tools.motion_opt.motion_optimization"
BodyConstraint.__module__ = REFERENCE_MODULE
terrain_util.register_reference_pickle_module(REFERENCE_MODULE, sys.modules[__name__])


def _f32(x, device):
    if torch.is_tensor(x):
        return x.detach().to(device=device, dtype=torch.float32)
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(device)


# ---------------------------------------------------------------------------------------------------------------------
# approximate contact constraints from the contact labels (reference :34-181)
# ---------------------------------------------------------------------------------------------------------------------
def _consecutive_true_runs(flags):
    """index tensors of the maximal runs of True in a 1-D bool tensor (reference extract_consecutive_trues :43-73, including its
    rule that a trailing run of ONE frame after another run is dropped)"""
    idx = torch.nonzero(flags.flatten(), as_tuple=True)[0]
    if idx.numel() == 0:
        return []
    breaks = [0] + (torch.nonzero(idx[1:] - idx[:-1] > 1, as_tuple=True)[0] + 1).tolist()
    runs = [idx[breaks[i]:breaks[i + 1]].clone() for i in range(len(breaks) - 1)]
    if breaks[-1] < idx.shape[0] - 1:
        runs.append(idx[breaks[-1]:].clone())
    return runs


def _project_points_to_surface(points, terrain, num_iters=1000, lr=0.01):
    """Every point descends 0.5 * d(sd^2) for `num_iters` SGD steps, sd = distance to the terrain's column set (reference
    optimize_contact_points :93-115, one point at a time there; all points of a body advance together here - the objective is
    a sum of independent terms, so each point follows the same trajectory)."""
    if points.shape[0] == 0:
        return points
    base_z = terrain.hf.min().item() - 10.0
    hf, mp = terrain.hf.unsqueeze(0), terrain.min_point.unsqueeze(0)
    grid = terrain_util.HfGrid(hf, terrain.dxdy, points.device)
    p = points.clone().requires_grad_(True)
    for _ in range(num_iters):
        sd = terrain_util.points_hf_sdf(p.unsqueeze(0), hf, mp, terrain.dxdy, base_z=base_z, inverted=False, grid=grid)
        g, = torch.autograd.grad(torch.sum(torch.square(sd)), p)
        with torch.no_grad():
            p -= lr * g
    return p.detach()


def compute_approx_body_constraints(root_pos, root_rot, joint_rot, contacts, char_model, terrain):
    """Per body a list of BodyConstraint: for feet and hands, every run of frames labelled "in contact" (> 0.9) becomes one
    constraint at the run's mean body position, projected onto the terrain surface."""
    body_pos, body_rot = char_model.forward_kinematics(root_pos, root_rot, joint_rot)
    ids = {n: char_model.get_body_id(n) for n in ("left_foot", "right_foot", "left_hand", "right_hand")}
    pos = {}
    for n, b in ids.items():
        pos[n] = body_pos[:, b]
        if n.endswith("foot"):       # the centre of the foot box, not the ankle (:136-142)
            off = _f32(char_model.get_geoms(b)[0]._offset, body_pos.device)
            pos[n] = pos[n] + torch_util.quat_rotate(body_rot[:, b], off.unsqueeze(0).expand(body_pos.shape[0], 3))
    out = [[] for _ in range(char_model.get_num_joints())]
    # every run of every body becomes one point; all of them descend together (independent terms of one objective)
    todo = [(b, r, pos[n][r].mean(dim=0)) for n, b in ids.items() for r in _consecutive_true_runs(contacts[:, b] > 0.9)]
    if todo:
        pts = _project_points_to_surface(torch.stack([p for _, _, p in todo]), terrain)
        for k, (b, r, _) in enumerate(todo):
            c = BodyConstraint()
            c.start_frame_idx = r[0].item()
            c.end_frame_idx = r[-1].item()
            c.constraint_point = pts[k].clone()
            out[b].append(c)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the loss (reference :183-395)
# ---------------------------------------------------------------------------------------------------------------------
class _Problem:
    """Everything of one optimisation problem that does not change between evaluations, laid out once on the device: the sample
    points of all bodies as one [P, 3] table with their owner, the heightfield grid, contact weights, and the body constraints
    as index ranges.  evaluate() touches the device only through launches (capturable)."""

    def __init__(self, src_root_pos, src_root_rot_quat, src_joint_rot, src_body_vels, src_body_rot_vels, contacts, terrain, body_points,
                 char_model, body_constraints):
        dev = src_root_pos.device
        self.km = char_model
        self.src = (src_root_pos, src_root_rot_quat, src_joint_rot, src_body_vels, src_body_rot_vels)
        self.contacts = contacts
        B = char_model.get_num_joints()
        assert len(body_points) == B
        self.points = terrain_util.BodyPoints(body_points, dev)
        counts = self.points.counts
        self.local, self.owner = self.points.local, self.points.owner
        self.start = self.points.start[:B]
        self.counts = counts
        self.hf = terrain.hf.to(dev, torch.float32).unsqueeze(0)
        self.min_point = terrain.min_point.to(dev, torch.float32).unsqueeze(0)
        self.dxdy = terrain.dxdy
        self.grid = terrain_util.HfGrid(self.hf, terrain.dxdy, dev)
        # contact present in both frames of a pair, negative labels (generator artefacts) clipped (:232-234)
        self.pair_contact = torch.clamp(torch.minimum(contacts[1:], contacts[:-1]), min=0.0)
        T = int(src_root_pos.shape[0])
        # per-body closest point: the bodies' point ranges padded to one width (a repeated point does not change a minimum)
        pmax = max(counts)
        assert min(counts) > 0, "every body needs at least one sample point"
        self.body_pts = torch.tensor([[self.start[b] + min(k, counts[b] - 1) for k in range(pmax)] for b in range(B)], dtype=torch.int64, device=dev)
        # body constraints as flat row tables, one row per (constraint, frame[, sole point]): frame index, body / point index,
        # constraint point, radius[, geom offset].  The frame pairs a constraint covers are exempt from the sliding term (:328-333).
        sph = {"f": [], "b": [], "pt": [], "r": [], "off": []}
        box = {"f": [], "p": [], "pt": [], "r": []}
        keep = torch.ones((T - 1, B), dtype=torch.float32, device=dev)
        if body_constraints is not None:
            for b in range(B):
                for c in body_constraints[b]:
                    geom = char_model.get_geoms(b)[0]
                    s, e = int(c.start_frame_idx), int(c.end_frame_idx)
                    point = _f32(c.constraint_point, "cpu").reshape(3).tolist()
                    fr = list(range(s, min(e, T - 1) + 1))
                    if geom._shape_type == kin_char_model.GeomType.SPHERE:
                        radius = float(_f32(geom._dims, "cpu").reshape(-1)[0])
                        off = _f32(geom._offset, "cpu").reshape(3).tolist()
                        sph["f"] += fr
                        sph["b"] += [b] * len(fr)
                        sph["pt"] += [point] * len(fr)
                        sph["r"] += [radius] * len(fr)
                        sph["off"] += [off] * len(fr)
                    elif geom._shape_type == kin_char_model.GeomType.BOX:
                        radius = float(torch.linalg.vector_norm(_f32(geom._dims, "cpu"))) * 1.25
                        assert counts[b] >= 18, "the sole of a box body is its first 18 sample points"
                        for f in fr:
                            box["f"] += [f] * 18
                            box["p"] += list(range(self.start[b], self.start[b] + 18))
                        box["pt"] += [point] * (18 * len(fr))
                        box["r"] += [radius] * (18 * len(fr))
                    else:
                        continue
                    keep[s:e + 1, b] = 0.0
        i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
        f32 = lambda v, w_: torch.tensor(v, dtype=torch.float32, device=dev).reshape(-1, w_) if w_ else torch.tensor(v, dtype=torch.float32, device=dev)
        self.sph = (i64(sph["f"]), i64(sph["b"]), f32(sph["pt"], 3), f32(sph["r"], 0), f32(sph["off"], 3)) if sph["f"] else None
        self.box = (i64(box["f"]), i64(box["p"]), f32(box["pt"], 3), f32(box["r"], 0)) if box["f"] else None
        self.pair_keep = keep
        self.has_constraints = body_constraints is not None

    def evaluate(self, tgt_root_pos, tgt_root_rot, tgt_joint_dof, w, max_jerk):
        """-> (weighted total, the nine terms stacked in _TERM_ORDER)"""
        km = self.km
        s_rp, s_rq, s_jr, s_bv, s_brv = self.src
        zero = tgt_root_pos.new_zeros(())
        root_pos_loss = torch.sum(torch.square(tgt_root_pos - s_rp))
        # root quaternion, joint rotations and every body's pose in one launch, with a one-launch adjoint (KinCharModel.pose_chain)
        tgt_rq, tgt_jr, body_pos, body_rot = km.pose_chain(tgt_root_pos, tgt_root_rot, tgt_joint_dof)
        root_rot_loss = torch.sum(torch.square(_diff_angle(tgt_rq, s_rq)))
        joint_rot_loss = torch.sum(torch.square(_diff_angle(tgt_jr, s_jr)))
        rot_vel_err_sq = torch.square(_diff_angle(body_rot[1:], body_rot[:-1]) - s_brv)
        dt = 1.0 / 30.0                 # the reference hard-codes the frame time (:360)
        fused_terms = body_pos.is_cuda and int(body_pos.shape[0]) >= 4
        if fused_terms:
            # smoothness, sliding and jerk sums of the frame-to-frame errors: one launch + one reduction (and one launch back)
            tt = _TemporalTerms.apply(body_pos, rot_vel_err_sq, s_bv.contiguous(), self.pair_keep, self.pair_contact, max_jerk * dt ** 3)
            smoothness_loss = tt[0]
        else:
            body_vels = body_pos[1:] - body_pos[:-1]
            vel_err_sq = torch.square(body_vels - s_bv)
            smoothness_loss = torch.sum(vel_err_sq) + torch.sum(rot_vel_err_sq)

        T, P = int(tgt_root_pos.shape[0]), int(self.local.shape[0])
        world = self.points.world(body_pos, body_rot)                      # [T, P, 3], one launch (and one for its adjoint)
        flat = world.reshape(1, T * P, 3)
        inside = terrain_util.points_hf_sdf(flat, self.hf, self.min_point, self.dxdy, base_z=-10.0, inverted=True, grid=self.grid)
        penetration_loss = torch.sum(-torch.clamp(inside, max=0.0))
        if w["w_contact"] != 0.0:
            outside = terrain_util.points_hf_sdf(flat, self.hf, self.min_point, self.dxdy, base_z=-10.0, inverted=False, grid=self.grid)
            outside = torch.clamp(outside, min=0.0).reshape(T, P)
            closest = outside[:, self.body_pts].min(dim=-1)[0]                     # [T, B]: per body its closest sample point
            contact_loss = torch.sum(closest * self.contacts)
        else:
            contact_loss = zero
        body_constraint_loss = zero
        if self.sph is not None:        # sphere bodies: |distance of the constraint point to the sphere| (:295-304)
            f, b, pt, r, off = self.sph
            centre = torch_util.quat_rotate(body_rot[f, b], off) + body_pos[f, b]
            body_constraint_loss = body_constraint_loss + torch.sum(torch.abs(geom_util.sdSphere(pt, centre, r)))
        if self.box is not None:        # box bodies (feet): every sole point within 1.25 box diagonals of the point (:306-322)
            f, p, pt, r = self.box
            body_constraint_loss = body_constraint_loss + torch.sum(torch.clamp(geom_util.sdSphere(pt, world[f, p], r), min=0.0))
        if fused_terms:
            sliding_loss = tt[1] if w["w_sliding"] != 0.0 else zero
            jerk_loss = tt[2]
        else:
            if w["w_sliding"] != 0.0:
                c, c2 = 0.03, 0.0009        # pseudo-Huber
                k = self.pair_keep
                sliding_loss = torch.sum((torch.sqrt(torch.sum(vel_err_sq * k.unsqueeze(-1), dim=-1) + c2) - c) * self.pair_contact) \
                    + torch.sum((torch.sqrt(rot_vel_err_sq * k + c2) - c) * self.pair_contact)
            else:
                sliding_loss = zero
            acc = body_vels[1:] - body_vels[:-1]
            jerk = torch.linalg.vector_norm(acc[1:] - acc[:-1], dim=-1)
            jerk_loss = torch.sum(torch.clamp(jerk - max_jerk * dt ** 3, min=0.0))

        terms = torch.stack([root_pos_loss, root_rot_loss, joint_rot_loss, smoothness_loss, penetration_loss, contact_loss, sliding_loss, jerk_loss,
                             body_constraint_loss])
        loss = w["w_root_pos"] * root_pos_loss + w["w_root_rot"] * root_rot_loss + w["w_joint_rot"] * joint_rot_loss \
            + w["w_smoothness"] * smoothness_loss + w["w_penetration"] * penetration_loss + w["w_contact"] * contact_loss \
            + w["w_sliding"] * sliding_loss + w["w_body_constraints"] * body_constraint_loss + w["w_jerk"] * jerk_loss
        return loss, terms


class _TemporalTerms(torch.autograd.Function):
    """(smoothness, sliding, jerk) sums of the frame-to-frame errors: per-(frame, body) partials in one launch + one reduction, the adjoint
    in one launch (parc_temporal_terms / _grad).  rot_err_sq carries its own gradient (it comes from the rotation-angle kernel)."""

    @staticmethod
    def forward(ctx, body_pos, rot_err_sq, src_vel, keep, pair_contact, jerk_limit):
        from ... import _hip
        T, B = int(body_pos.shape[0]), int(body_pos.shape[1])
        pos = body_pos.detach().to(torch.float32).contiguous()
        r = rot_err_sq.detach().to(torch.float32).contiguous()
        partial = torch.empty((3, T, B), dtype=torch.float32, device=pos.device)
        ctx.args = (T, B, 0.03, 0.0009, float(jerk_limit))          # pseudo-Huber constants of the reference (:349-350)
        _hip.check(_hip.lib().parc_temporal_terms(_hip.stream(), T, B, _hip.ptr(pos), _hip.ptr(r), _hip.ptr(src_vel), _hip.ptr(keep),
                                                  _hip.ptr(pair_contact), ctx.args[2], ctx.args[3], ctx.args[4], _hip.ptr(partial)), "parc_temporal_terms")
        ctx.save_for_backward(pos, r, src_vel, keep, pair_contact)
        return partial.sum(dim=(1, 2))

    @staticmethod
    def backward(ctx, g):
        from ... import _hip
        pos, r, src_vel, keep, pair_contact = ctx.saved_tensors
        T, B, c, c2, lim = ctx.args
        g_pos, g_r = torch.empty_like(pos), torch.empty_like(r)
        gg = g.to(torch.float32).contiguous()
        _hip.check(_hip.lib().parc_temporal_terms_grad(_hip.stream(), T, B, _hip.ptr(pos), _hip.ptr(r), _hip.ptr(src_vel), _hip.ptr(keep),
                                                       _hip.ptr(pair_contact), c, c2, lim, _hip.ptr(gg), _hip.ptr(g_pos), _hip.ptr(g_r)),
                   "parc_temporal_terms_grad")
        return g_pos, g_r, None, None, None, None


def _diff_angle(q0, q1):
    """torch_util.quat_diff_angle: one launch (and one for its adjoint) on the GPU, torch ops elsewhere"""
    if q0.is_cuda:
        return torch_util.quat_diff_angle_fused(q0, q1)
    return torch_util.quat_to_axis_angle(torch_util.quat_mul_compact(q1, torch_util.quat_conjugate(q0)))[1]


def _weights(**kw):
    names = ("w_root_pos", "w_root_rot", "w_joint_rot", "w_smoothness", "w_penetration", "w_contact", "w_sliding", "w_body_constraints", "w_jerk")
    return {n: float(kw[n]) for n in names}


def motion_terrain_contact_loss(tgt_root_pos, tgt_root_rot, tgt_joint_dof, src_root_pos, src_root_rot_quat, src_joint_rot, src_body_vels,
                                src_body_rot_vels, contacts, terrain, body_points, char_model, w_root_pos, w_root_rot, w_joint_rot, w_smoothness,
                                w_penetration, w_contact, w_sliding, w_body_constraints, w_jerk, body_constraints, max_jerk):
    """-> (weighted loss tensor, {LossType: float}).  One read-back for the whole dict."""
    prob = _Problem(src_root_pos, src_root_rot_quat, src_joint_rot, src_body_vels, src_body_rot_vels, contacts, terrain, body_points, char_model,
                    body_constraints)
    w = _weights(w_root_pos=w_root_pos, w_root_rot=w_root_rot, w_joint_rot=w_joint_rot, w_smoothness=w_smoothness, w_penetration=w_penetration,
                 w_contact=w_contact, w_sliding=w_sliding, w_body_constraints=w_body_constraints, w_jerk=w_jerk)
    loss, terms = prob.evaluate(tgt_root_pos, tgt_root_rot, tgt_joint_dof, w, max_jerk)
    return loss, dict(zip(_TERM_ORDER, terms.detach().tolist()))


# ---------------------------------------------------------------------------------------------------------------------
# the descent (reference :404-500)
# ---------------------------------------------------------------------------------------------------------------------
def build_logger(log_file, exp_name=None, use_wandb=True):
    """Text logger with the reference's keys (its wandb upload has no counterpart: there is no network on the target)."""
    log = logger_mod.Logger()
    log.set_step_key("Iteration")
    if log_file is not None:
        log.configure_output_file(str(log_file))
    return log


def motion_contact_optimization(src_frames, contacts, body_points, terrain, char_model, num_iters, step_size, w_root_pos, w_root_rot, w_joint_rot,
                                w_smoothness, w_penetration, w_contact, w_sliding, w_body_constraints, w_jerk, body_constraints, max_jerk, exp_name,
                                use_wandb, log_file, use_graph=None, verbose=True, loss_trace=None):
    """Adam on (root position, root exponential map, joint dofs) of every frame; returns the optimised frames [T, 34].
    use_graph (default: on a GPU): capture one iteration in a hipGraph after three eager ones and replay it.
    loss_trace: optional list that receives the total loss of every iteration as ONE device tensor at the end (tests)."""
    start_time = time.time()
    dev = src_frames.device
    src_frames = src_frames.to(torch.float32)
    contacts = contacts.to(torch.float32)
    with torch.no_grad():
        src_root_pos = src_frames[:, 0:3]
        src_root_rot_quat = torch_util.exp_map_to_quat(src_frames[:, 3:6])
        src_joint_rot = char_model.dof_to_rot(src_frames[:, 6:34].contiguous())
        src_body_pos, src_body_rot = char_model.forward_kinematics(src_root_pos.contiguous(), src_root_rot_quat, src_joint_rot)
        src_body_vels = src_body_pos[1:] - src_body_pos[:-1]
        src_body_rot_vels = torch_util.quat_diff_angle(src_body_rot[1:], src_body_rot[:-1])
    prob = _Problem(src_root_pos, src_root_rot_quat, src_joint_rot, src_body_vels, src_body_rot_vels, contacts, terrain, body_points, char_model,
                    body_constraints)
    w = _weights(w_root_pos=w_root_pos, w_root_rot=w_root_rot, w_joint_rot=w_joint_rot, w_smoothness=w_smoothness, w_penetration=w_penetration,
                 w_contact=w_contact, w_sliding=w_sliding, w_body_constraints=w_body_constraints, w_jerk=w_jerk)
    params = [src_frames[:, 0:3].clone().requires_grad_(True), src_frames[:, 3:6].clone().requires_grad_(True),
              src_frames[:, 6:34].clone().requires_grad_(True)]
    on_gpu = src_frames.is_cuda
    use_graph = on_gpu if use_graph is None else use_graph
    # one multi-tensor launch per step on the GPU (same update rule)
    optimizer = torch.optim.Adam(params, lr=step_size, capturable=bool(use_graph), **({"fused": True} if on_gpu else {}))
    log_iter_stride = 25
    logger = build_logger(log_file=log_file, exp_name=exp_name, use_wandb=use_wandb)
    # the iteration's outputs live in fixed buffers: [total, nine terms]
    record = torch.zeros(1 + len(_TERM_ORDER), dtype=torch.float32, device=dev)
    trace = torch.zeros(max(num_iters, 1), dtype=torch.float32, device=dev) if loss_trace is not None else None
    it_idx = torch.zeros((), dtype=torch.int64, device=dev)

    def iteration():
        optimizer.zero_grad(set_to_none=True)
        loss, terms = prob.evaluate(params[0], params[1], params[2], w, max_jerk)
        loss.backward()
        optimizer.step()
        with torch.no_grad():
            record[0] = loss.detach()
            record[1:] = terms.detach()
            if trace is not None:
                trace.index_copy_(0, it_idx.reshape(1), loss.detach().reshape(1))
                it_idx.add_(1)

    def log_now(it):
        vals = record.tolist()
        logger.log("Iteration", it)
        logger.log("Time (min)", (time.time() - start_time) / 60.0)
        logger.log("TOTAL WEIGHTED LOSS", vals[0])
        for key, val in zip(_TERM_ORDER, vals[1:]):
            logger.log(key.name, val)
        if verbose:
            logger.print_log()
        logger.write_log()

    graph = None
    warm = 3
    for it in range(num_iters):
        if use_graph and it == warm:
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                iteration()
            # capture records, it does not run: this iteration is the first replay below
        if graph is not None:
            graph.replay()
        elif use_graph:
            # warm-up iterations of the capture protocol run on a side stream (they are ordinary iterations of the descent)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                iteration()
            torch.cuda.current_stream().wait_stream(side)
        else:
            iteration()
        if it % log_iter_stride == 0:
            log_now(it)
    if getattr(logger, "_file", None) is not None:
        logger._file.close()
        logger._file = None
    if loss_trace is not None:
        loss_trace.append(trace[:num_iters].clone())
    return torch.cat([p.detach() for p in params], dim=-1)
