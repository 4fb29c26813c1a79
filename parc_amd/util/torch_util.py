"""Quaternion / heading helpers on torch tensors (quaternions are xyzw, last axis).

Host-side mirror of the part of the reference's ``util/torch_util.py`` that the tracker path and its
callers use (function names and edge-case rules as there: quat_mul :41-58, quat_rotate :61-66,
quat_to_axis_angle :68-88, axis_angle_to_quat :311-317, quat_to_exp_map :346-351, quat_to_tan_norm :361-373,
exp_map_to_axis_angle :394-411, exp_map_to_quat :414-419, quat_diff_angle :427-431, slerp :443-468,
calc_heading* :470-499, rotate_2d_vec :619-631).  The per-step work of the tracker does NOT go through these:
it runs in the HIP kernels (parc_math.h); this module serves tools, dataset code, tests and drop-in callers
that import ``util.torch_util``.  Checked against fixture G1 (tests/test_host_logic.py).
"""
import numpy as np
import torch


def normalize_angle(x):
    return torch.atan2(torch.sin(x), torch.cos(x))


def normalize(x, eps: float = 1e-9):
    return x / torch.linalg.vector_norm(x, dim=-1, keepdim=True).clamp_min(eps)


def quat_unit(a):
    return normalize(a)


def quat_conjugate(q):
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


quat_inv = quat_conjugate


def quat_pos(q):
    return torch.where(q[..., 3:4] < 0, -q, q)


def quat_abs(q):
    return torch.linalg.vector_norm(q, dim=-1)


def quat_normalize(q):
    return quat_unit(quat_pos(q))


def quat_mul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw,
                        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


quat_multiply = quat_mul


def quat_mul_compact(a, b):
    """a * b in vector form (w = aw bw - av.bv, v = aw bv + bw av + av x bv): the same product in 10 launches instead of 29 and with
    a matching short backward, for differentiable batch code that evaluates it thousands of times (the motion optimiser)."""
    av, aw = a[..., :3], a[..., 3:4]
    bv, bw = b[..., :3], b[..., 3:4]
    w = aw * bw - (av * bv).sum(dim=-1, keepdim=True)
    v = aw * bv + bw * av + torch.cross(av.expand_as(bv) if av.shape != bv.shape else av, bv, dim=-1)
    return torch.cat([v, w], dim=-1)


def quat_rotate(q, v):
    u = q[..., :3]
    t = 2.0 * torch.cross(u, v, dim=-1)
    return v + q[..., 3:4] * t + torch.cross(u, t, dim=-1)


def _z_axis_like(x3):
    ax = torch.zeros_like(x3)
    ax[..., 2] = 1.0
    return ax


def quat_to_axis_angle(q, eps: float = 1e-5):
    """Axis (unit, +z when the rotation is below eps) and angle in [0, 2pi) of the w >= 0 representative."""
    q = quat_pos(q)
    s = torch.linalg.vector_norm(q[..., :3], dim=-1)
    ok = s > eps
    angle = torch.where(ok, 2.0 * torch.atan2(s, q[..., 3]), torch.zeros_like(s))
    axis = torch.where(ok.unsqueeze(-1), q[..., :3] / s.unsqueeze(-1), _z_axis_like(q[..., :3]))
    return axis, angle


def axis_angle_to_quat(axis, angle):
    h = (0.5 * angle).unsqueeze(-1)
    return quat_unit(torch.cat([normalize(axis) * torch.sin(h), torch.cos(h)], dim=-1))


def heading_to_quat(heading):
    axis = torch.zeros(heading.shape + (3,), dtype=torch.float32, device=heading.device)
    axis[..., 2] = 1.0
    return axis_angle_to_quat(axis, heading)


def axis_angle_to_exp_map(axis, angle):
    return axis * angle.unsqueeze(-1)


def quat_to_exp_map(q):
    return axis_angle_to_exp_map(*quat_to_axis_angle(q))


def exp_map_to_axis_angle(exp_map, min_theta: float = 1e-5):
    raw = torch.linalg.vector_norm(exp_map, dim=-1)
    angle = normalize_angle(raw)
    ok = angle.abs() > min_theta
    axis = torch.where(ok.unsqueeze(-1), exp_map / raw.unsqueeze(-1), _z_axis_like(exp_map))
    return axis, torch.where(ok, angle, torch.zeros_like(angle))


def exp_map_to_quat(exp_map):
    return axis_angle_to_quat(*exp_map_to_axis_angle(exp_map))


def normalize_exp_map(exp_map):
    raw = torch.linalg.vector_norm(exp_map, dim=-1).clamp_min(1e-9)
    return exp_map * (normalize_angle(raw) / raw).unsqueeze(-1)


def quat_to_tan_norm(q):
    """Rotated x axis (tangent) followed by rotated z axis (normal): the 6-vector the observations use."""
    ex = torch.zeros_like(q[..., :3])
    ex[..., 0] = 1.0
    return torch.cat([quat_rotate(q, ex), quat_rotate(q, _z_axis_like(q[..., :3]))], dim=-1)


def quat_diff(q0, q1):
    return quat_mul(q1, quat_conjugate(q0))


def quat_diff_angle(q0, q1):
    return quat_to_axis_angle(quat_diff(q0, q1))[1]


class _QuatDiffAngle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q0, q1):
        from .. import _hip
        shape = torch.broadcast_shapes(q0.shape, q1.shape)
        a = q0.detach().to(torch.float32).expand(shape).contiguous()
        b = q1.detach().to(torch.float32).expand(shape).contiguous()
        n = a.numel() // 4
        out = torch.empty(shape[:-1], dtype=torch.float32, device=a.device)
        _hip.check(_hip.lib().parc_quat_diff_angle(_hip.stream(), n, _hip.ptr(a), _hip.ptr(b), _hip.ptr(out)), "parc_quat_diff_angle")
        ctx.save_for_backward(a, b)
        ctx.shapes = (q0.shape, q1.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        from .. import _hip
        a, b = ctx.saved_tensors
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        gg = g.to(torch.float32).contiguous()
        _hip.check(_hip.lib().parc_quat_diff_angle_grad(_hip.stream(), a.numel() // 4, _hip.ptr(a), _hip.ptr(b), _hip.ptr(gg), _hip.ptr(ga), _hip.ptr(gb)),
                   "parc_quat_diff_angle_grad")
        s0, s1 = ctx.shapes
        return ga.sum_to_size(s0) if tuple(s0) != tuple(ga.shape) else ga, gb.sum_to_size(s1) if tuple(s1) != tuple(gb.shape) else gb


def quat_diff_angle_fused(q0, q1):
    """quat_diff_angle on the GPU as one launch, with a one-launch adjoint (parc_quat_diff_angle / _grad): for differentiable batch
    code that evaluates it thousands of times (the motion optimiser); same values and gradients as quat_diff_angle."""
    return _QuatDiffAngle.apply(q0, q1)


def slerp(q0, q1, t):
    """Shortest-arc interpolation with the reference's two fall-backs: plain average when sin(half angle) < 1e-3,
    q0 when |cos| >= 1."""
    c = torch.sum(q0 * q1, dim=-1, keepdim=True)
    q1 = torch.where(c < 0, -q1, q1)
    c = c.abs()
    half = torch.acos(c)
    s = torch.sqrt(1.0 - c * c)
    if t.dim() == q0.dim() - 1:
        t = t.unsqueeze(-1)
    out = (torch.sin((1.0 - t) * half) / s) * q0 + (torch.sin(t * half) / s) * q1
    out = torch.where(s.abs() < 1e-3, 0.5 * q0 + 0.5 * q1, out)
    return torch.where(c >= 1.0, q0, out)


def calc_heading(q):
    assert q.shape[-1] == 4
    ex = torch.zeros_like(q[..., :3])
    ex[..., 0] = 1.0
    d = quat_rotate(q, ex)
    return torch.atan2(d[..., 1], d[..., 0])


def calc_heading_quat(q):
    return axis_angle_to_quat(_z_axis_like(q[..., :3]), calc_heading(q))


def calc_heading_quat_inv(q):
    return axis_angle_to_quat(_z_axis_like(q[..., :3]), -calc_heading(q))


def rotate_2d_vec(vec, angle):
    c, s = torch.cos(angle), torch.sin(angle)
    x, y = vec[..., 0], vec[..., 1]
    return torch.stack([x * c - y * s, x * s + y * c], dim=-1)


def heading_angle_from_xy(x, y):
    return torch.atan2(y, x)


def quat_differentiate_angular_velocity(next_q, curr_q, dt):
    return quat_to_exp_map(quat_normalize(quat_mul(next_q, quat_conjugate(curr_q)))) / dt


def rotate_quat_by_heading(heading, quat):
    if not isinstance(heading, torch.Tensor):
        heading = torch.tensor([heading], dtype=torch.float32, device=quat.device)
    return quat_mul(heading_to_quat(heading).expand_as(quat), quat)


def rotate_exp_map_by_heading(heading, exp_map):
    return quat_to_exp_map(rotate_quat_by_heading(heading, exp_map_to_quat(exp_map)))


_NP_OF_TORCH = {torch.float32: np.float32, torch.float64: np.float64, torch.uint8: np.uint8, torch.int32: np.int32,
                torch.int64: np.int64, torch.bool: np.bool_}


def torch_dtype_to_numpy(torch_dtype):
    return _NP_OF_TORCH[torch_dtype]


def numpy_dtype_to_torch(numpy_dtype):
    nd = np.dtype(numpy_dtype)
    for t, n in _NP_OF_TORCH.items():
        if np.dtype(n) == nd:
            return t
    raise KeyError(numpy_dtype)
