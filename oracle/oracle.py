"""ctypes binding of oracle/parc_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libparc_oracle.so")
_lib = None

c_int = ctypes.c_int
c_float = ctypes.c_float
c_double = ctypes.c_double


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "parc_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _l(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Char:
    """Character arrays exactly as KinCharModel holds them (anim/kin_char_model.py:147-178)."""

    def __init__(self, parent, local_translation, local_rotation, joint_type, joint_axis, dof_idx):
        self.parent = _i(parent)
        self.ltrans = _f(local_translation)
        self.lrot = _f(local_rotation)
        self.jtype = _i(joint_type)
        self.jaxis = _f(joint_axis)
        self.dof_idx = _i(dof_idx)
        self.nb = int(self.parent.shape[0])
        dims = {1: 1, 2: 3}
        self.dof_size = int(sum(dims.get(int(t), 0) for t in self.jtype))

    @classmethod
    def from_npz(cls, z):
        return cls(z["parent"], z["local_translation"], z["local_rotation"], z["joint_type"], z["joint_axis"], z["dof_idx"])

    def args(self):
        return [c_int(self.nb), _p(self.parent), _p(self.ltrans), _p(self.lrot), _p(self.jtype), _p(self.jaxis), _p(self.dof_idx)]


class MotionLib:
    """Flat clip database (anim/motion_lib.py:204-380), built with orc_motion_derive."""

    def __init__(self, char, clips, fps, loop_modes, weights=None, contacts=None):
        self.char = char
        J, D, B = char.nb - 1, char.dof_size, char.nb
        self.J, self.D, self.B = J, D, B
        rp, rr, jr, rv, rav, dv, co = [], [], [], [], [], [], []
        nf, ln, delta = [], [], []
        L = lib()
        for k, fr in enumerate(clips):
            fr = _f(fr)
            F = fr.shape[0]
            a_rp = np.zeros((F, 3), np.float32)
            a_rr = np.zeros((F, 4), np.float32)
            a_jr = np.zeros((F, J, 4), np.float32)
            a_rv = np.zeros((F, 3), np.float32)
            a_rav = np.zeros((F, 3), np.float32)
            a_dv = np.zeros((F, D), np.float32)
            L.orc_motion_derive(*char.args(), c_int(F), _p(fr), c_double(float(fps[k])), _p(a_rp), _p(a_rr), _p(a_jr),
                                _p(a_rv), _p(a_rav), _p(a_dv))
            rp.append(a_rp); rr.append(a_rr); jr.append(a_jr); rv.append(a_rv); rav.append(a_rav); dv.append(a_dv)
            co.append(_f(contacts[k]) if contacts is not None else np.zeros((F, B), np.float32))
            nf.append(F)
            ln.append(1.0 / float(fps[k]) * (F - 1))
            d = a_rp[-1] - a_rp[0]
            d[2] = 0.0
            delta.append(d)
        self.M = len(clips)
        self.num_frames = _l(nf)
        start = np.roll(self.num_frames, 1)
        start[0] = 0
        self.start_idx = _l(np.cumsum(start))
        self.length = _f(ln)
        self.loop_mode = _i(loop_modes)
        self.pos_delta = _f(np.stack(delta))
        self.root_pos = _f(np.concatenate(rp)); self.root_rot = _f(np.concatenate(rr))
        self.joint_rot = _f(np.concatenate(jr)); self.root_vel = _f(np.concatenate(rv))
        self.root_ang_vel = _f(np.concatenate(rav)); self.dof_vel = _f(np.concatenate(dv))
        self.contacts = _f(np.concatenate(co))
        w = _f(weights if weights is not None else np.ones(self.M))
        self.weights = w / w.sum()

    def args(self):
        return [c_int(self.M), c_int(self.J), c_int(self.D), c_int(self.B), _p(self.num_frames), _p(self.start_idx),
                _p(self.length), _p(self.loop_mode), _p(self.pos_delta), _p(self.root_pos), _p(self.root_rot),
                _p(self.joint_rot), _p(self.root_vel), _p(self.root_ang_vel), _p(self.dof_vel), _p(self.contacts)]

    def calc_motion_frame(self, ids, times):
        ids = _l(ids); times = _f(times)
        Q = ids.shape[0]
        out = dict(root_pos=np.zeros((Q, 3), np.float32), root_rot=np.zeros((Q, 4), np.float32),
                   root_vel=np.zeros((Q, 3), np.float32), root_ang_vel=np.zeros((Q, 3), np.float32),
                   joint_rot=np.zeros((Q, self.J, 4), np.float32), dof_vel=np.zeros((Q, self.D), np.float32),
                   contacts=np.zeros((Q, self.B), np.float32))
        lib().orc_calc_motion_frame(*self.args(), c_int(Q), _p(ids), _p(times), _p(out["root_pos"]), _p(out["root_rot"]),
                                    _p(out["root_vel"]), _p(out["root_ang_vel"]), _p(out["joint_rot"]), _p(out["dof_vel"]),
                                    _p(out["contacts"]))
        return out


    def slerp_cosines(self, ids, times):
        """-> (|cos half angle| [Q, 1 + J] as slerp evaluates it for the frame pair of each query, blend [Q]); column 0 = root rotation"""
        ids = _l(ids); times = _f(times)
        Q = ids.shape[0]
        cos = np.zeros((Q, self.J + 1), np.float32)
        blend = np.zeros(Q, np.float32)
        lib().orc_slerp_cosines(*self.args(), c_int(Q), _p(ids), _p(times), _p(cos), _p(blend))
        return cos, blend


def _batch(name, n, ins, out_shape):
    out = np.zeros(out_shape, np.float32)
    getattr(lib(), name)(c_int(n), *[_p(x) for x in ins], _p(out))
    return out


def quat_mul(a, b): a, b = _f(a), _f(b); return _batch("orc_quat_mul", len(a), [a, b], a.shape)
def quat_rotate(q, v): q, v = _f(q), _f(v); return _batch("orc_quat_rotate", len(q), [q, v], v.shape)
def exp_map_to_quat(e): e = _f(e); return _batch("orc_exp_map_to_quat", len(e), [e], (len(e), 4))
def quat_to_exp_map(q): q = _f(q); return _batch("orc_quat_to_exp_map", len(q), [q], (len(q), 3))
def axis_angle_to_quat(ax, an): ax, an = _f(ax), _f(an); return _batch("orc_axis_angle_to_quat", len(ax), [ax, an], (len(ax), 4))
def quat_to_tan_norm(q): q = _f(q); return _batch("orc_quat_to_tan_norm", len(q), [q], (len(q), 6))
def slerp(a, b, t): a, b, t = _f(a), _f(b), _f(t); return _batch("orc_slerp", len(a), [a, b, t], a.shape)
def calc_heading(q): q = _f(q); return _batch("orc_calc_heading", len(q), [q], (len(q),))
def calc_heading_quat_inv(q): q = _f(q); return _batch("orc_calc_heading_quat_inv", len(q), [q], (len(q), 4))
def quat_diff_angle(a, b): a, b = _f(a), _f(b); return _batch("orc_quat_diff_angle", len(a), [a, b], (len(a),))


def dof_to_rot(char, dof):
    dof = _f(dof); n = dof.shape[0]
    out = np.zeros((n, char.nb - 1, 4), np.float32)
    lib().orc_dof_to_rot(*char.args(), c_int(n), _p(dof), _p(out))
    return out


def rot_to_dof(char, jrot):
    jrot = _f(jrot); n = jrot.shape[0]
    out = np.zeros((n, char.dof_size), np.float32)
    lib().orc_rot_to_dof(*char.args(), c_int(n), _p(jrot), _p(out))
    return out


def forward_kinematics(char, root_pos, root_rot, jrot):
    root_pos, root_rot, jrot = _f(root_pos), _f(root_rot), _f(jrot)
    n = root_pos.shape[0]
    bp = np.zeros((n, char.nb, 3), np.float32); br = np.zeros((n, char.nb, 4), np.float32)
    lib().orc_forward_kinematics(*char.args(), c_int(n), _p(root_pos), _p(root_rot), _p(jrot), _p(bp), _p(br))
    return bp, br


def xy_points_cone(dx, num_neg, num_pos, rays_neg, rays_pos, angle):
    n = (rays_neg + 1 + rays_pos) * (num_neg + num_pos + 1)
    out = np.zeros((n, 2), np.float32)
    lib().orc_xy_points_cone(c_float(dx), c_int(num_neg), c_int(num_pos), c_int(rays_neg), c_int(rays_pos), c_float(angle), _p(out))
    return out


def refresh_ray_obs_hfs(ray_xy, root_pos, heading, hf, min_point, dxdy, min_h=-3.0, max_h=3.0):
    ray_xy, root_pos, heading, hf = _f(ray_xy), _f(root_pos), _f(heading), _f(hf)
    N, P = root_pos.shape[0], ray_xy.shape[0]
    out = np.zeros((N, P), np.float32)
    lib().orc_refresh_ray_obs_hfs(c_int(N), c_int(P), _p(ray_xy), _p(root_pos), _p(heading), _p(hf), c_int(hf.shape[0]),
                                  c_int(hf.shape[1]), c_float(min_point[0]), c_float(min_point[1]), c_float(dxdy[0]),
                                  c_float(dxdy[1]), c_float(min_h), c_float(max_h), _p(out))
    return out


def obs_width(mlib, S, K, P, global_root_height_obs=False, enable_tar_obs=True, use_contact_info=True, has_target_xy_obs=False,
              replan_timer=False):
    """columns of IGParkourEnv._compute_obs's row for one setting of its switches (ig_parkour_env.py:1150-1244)"""
    J, D, B = mlib.J, mlib.D, mlib.B
    w = int(global_root_height_obs) + (12 + 6 * J + D + 3 * K) + P
    if enable_tar_obs:
        w += S * (9 + 6 * J + 3 * K)
    if use_contact_info:
        w += B + (S * B if enable_tar_obs else 0)
    return w + 2 * int(has_target_xy_obs) + int(replan_timer)


def compute_obs(char, mlib, tar_steps_dt, key_body_ids, motion_ids, motion_times, motion_xy_offset, char_root_pos,
                char_root_rot, char_root_vel, char_root_ang_vel, char_dof_pos, char_dof_vel, contact_forces, ray_hfs,
                contact_eps=1e-5, global_obs=False, global_root_height_obs=False, enable_tar_obs=True, use_contact_info=True,
                target_xy=None, replan_t=None):
    """target_xy [N,2] = has_target_xy_obs; replan_t (a float) = enable_replan_timer_obs on an env with motion-generator rows"""
    N = char_root_pos.shape[0]
    S = len(tar_steps_dt); K = len(key_body_ids); P = ray_hfs.shape[1]
    obs_dim = obs_width(mlib, S, K, P, global_root_height_obs, enable_tar_obs, use_contact_info, target_xy is not None, replan_t is not None)
    obs = np.zeros((N, obs_dim), np.float32)
    a = [_f(tar_steps_dt), _l(key_body_ids), _l(motion_ids), _f(motion_times), _f(motion_xy_offset), _f(char_root_pos),
         _f(char_root_rot), _f(char_root_vel), _f(char_root_ang_vel), _f(char_dof_pos), _f(char_dof_vel),
         _f(contact_forces), _f(ray_hfs)]
    txy = None if target_xy is None else _f(target_xy)
    assert txy is None or txy.shape == (N, 2)
    lib().orc_compute_obs_ex(*char.args(), *mlib.args(), c_int(N), c_int(S), _p(a[0]), c_int(K), _p(a[1]), _p(a[2]), _p(a[3]),
                             _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]), _p(a[8]), _p(a[9]), _p(a[10]), _p(a[11]), _p(a[12]),
                             c_int(P), c_float(contact_eps), c_int(bool(global_obs)), c_int(bool(global_root_height_obs)),
                             c_int(bool(enable_tar_obs)), c_int(bool(use_contact_info)), None if txy is None else _p(txy),
                             c_int(replan_t is not None), c_float(0.0 if replan_t is None else float(replan_t)), _p(obs), c_int(obs_dim))
    return obs


def compute_reward(char, key_body_ids, st, ref, joint_err_w, dof_err_w, contact_w, w5, rel_dm_w=1.0, track_root=True, track_root_h=True,
                   use_contact_info=True, target_xy=None, task1_w=0.7, task2_w=0.3, target_radius=1.0, rel_task_w=0.0, all_terms=False):
    """-> reward [N], terms [N,6] (pose, vel, root_pose, root_vel, key_pos, contact_penalty); all_terms: [N,9] with task_r1, task_r2 and
    the total task reward behind them"""
    N = st["char_root_pos"].shape[0]
    K = len(key_body_ids)
    reward = np.zeros(N, np.float32); terms = np.zeros((N, 9), np.float32)
    arrs = [_l(key_body_ids)] + [_f(st[k]) for k in ("char_root_pos", "char_root_rot", "char_root_vel", "char_root_ang_vel",
                                                      "char_dof_pos", "char_dof_vel", "char_rigid_body_pos")] + \
           [_f(ref[k]) for k in ("ref_root_pos", "ref_root_rot", "ref_root_vel", "ref_root_ang_vel", "ref_joint_rot",
                                 "ref_dof_vel", "ref_body_pos", "ref_contacts")] + \
           [_f(st["contact_forces"]), _f(joint_err_w), _f(dof_err_w), _f(contact_w), _f(w5)]
    txy = None if target_xy is None else _f(target_xy)
    assert txy is None or txy.shape == (N, 2)
    lib().orc_compute_reward_ex(*char.args(), c_int(N), c_int(K), *[_p(x) for x in arrs], c_float(rel_dm_w), c_int(bool(track_root)),
                                c_int(bool(track_root_h)), c_int(bool(use_contact_info)), None if txy is None else _p(txy),
                                c_float(task1_w), c_float(task2_w), c_float(target_radius), c_float(rel_task_w), _p(reward), _p(terms))
    return reward, (terms if all_terms else np.ascontiguousarray(terms[:, :6]))


def update_done(time_buf, ep_len, char_root_rot, body_pos, ref_root_rot, ref_body_pos, contact_forces, contact_body_ids,
                env_offsets, hf, min_point, dxdy, termination_height, pose_termination, pose_termination_dist,
                enable_early_termination, track_root, root_pos_term_dist, root_rot_term_angle, motion_ids, motion_times,
                motion_len, motion_loop_mode, fail_rates, ema_w=0.01):
    N, B = body_pos.shape[0], body_pos.shape[1]
    mask = np.zeros(B, np.int32)
    for b in contact_body_ids:
        mask[int(b)] = 1
    hf = _f(hf)
    done_pre = np.zeros(N, np.int32); done = np.zeros(N, np.int32)
    fr = _f(fail_rates).copy()
    a = [_f(time_buf), _f(char_root_rot), _f(body_pos), _f(ref_root_rot), _f(ref_body_pos), _f(contact_forces), mask,
         _f(env_offsets), hf, _f(pose_termination_dist), _l(motion_ids), _f(motion_times), _f(motion_len), _i(motion_loop_mode)]
    lib().orc_update_done(c_int(N), c_int(B), c_int(len(fr)), _p(a[0]), c_float(ep_len), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]),
                          _p(a[5]), c_int(len(contact_body_ids)), _p(a[6]), _p(a[7]), _p(a[8]), c_int(hf.shape[0]),
                          c_int(hf.shape[1]), c_float(min_point[0]), c_float(min_point[1]), c_float(dxdy[0]), c_float(dxdy[1]),
                          c_float(termination_height), c_int(int(pose_termination)), _p(a[9]), c_int(int(enable_early_termination)),
                          c_int(int(track_root)), c_float(root_pos_term_dist), c_float(root_rot_term_angle), _p(a[10]), _p(a[11]),
                          _p(a[12]), _p(a[13]), c_float(ema_w), _p(done_pre), _p(done), _p(fr))
    return done_pre, done, fr


def tracking_error(root_pos, root_rot, body_rot, body_pos, tar_root_pos, tar_root_rot, tar_body_rot, tar_body_pos,
                   root_vel, root_ang_vel, dof_vel, tar_root_vel, tar_root_ang_vel, tar_dof_vel):
    N, B = body_pos.shape[0], body_pos.shape[1]
    D = dof_vel.shape[1]
    out = np.zeros((N, 7), np.float32)
    a = [_f(x) for x in (root_pos, root_rot, body_rot, body_pos, tar_root_pos, tar_root_rot, tar_body_rot, tar_body_pos,
                         root_vel, root_ang_vel, dof_vel, tar_root_vel, tar_root_ang_vel, tar_dof_vel)]
    lib().orc_tracking_error(c_int(N), c_int(B), c_int(D), *[_p(x) for x in a], _p(out))
    return out


def td_lambda_return(r, next_vals, done, discount, td_lambda):
    r, next_vals, done = _f(r), _f(next_vals), _i(done)
    T, N = r.shape
    ret = np.zeros((T, N), np.float32)
    lib().orc_td_lambda_return(c_int(T), c_int(N), _p(r), _p(next_vals), _p(done), c_float(discount), c_float(td_lambda), _p(ret))
    return ret


def adv_normalize(ret, vals, rand_mask, clip):
    ret, vals, rand_mask = _f(ret), _f(vals), _f(rand_mask)
    out = np.zeros(ret.shape, np.float32)
    mean = c_float(0); std = c_float(0)
    lib().orc_adv_normalize(c_int(ret.size), _p(ret), _p(vals), _p(rand_mask), c_float(clip), _p(out), ctypes.byref(mean), ctypes.byref(std))
    return out, mean.value, std.value


def update_ref_motion(char, mlib, motion_ids, motion_times, motion_xy_offset):
    N = len(motion_ids)
    J, D, B = mlib.J, mlib.D, mlib.B
    o = dict(ref_root_pos=np.zeros((N, 3), np.float32), ref_root_rot=np.zeros((N, 4), np.float32),
             ref_root_vel=np.zeros((N, 3), np.float32), ref_root_ang_vel=np.zeros((N, 3), np.float32),
             ref_joint_rot=np.zeros((N, J, 4), np.float32), ref_dof_vel=np.zeros((N, D), np.float32),
             ref_contacts=np.zeros((N, B), np.float32), ref_body_pos=np.zeros((N, B, 3), np.float32),
             ref_dof_pos=np.zeros((N, D), np.float32))
    a = [_l(motion_ids), _f(motion_times), _f(motion_xy_offset)]
    lib().orc_update_ref_motion(*char.args(), *mlib.args(), c_int(N), _p(a[0]), _p(a[1]), _p(a[2]), *[_p(o[k]) for k in (
        "ref_root_pos", "ref_root_rot", "ref_root_vel", "ref_root_ang_vel", "ref_joint_rot", "ref_dof_vel", "ref_contacts",
        "ref_body_pos", "ref_dof_pos")])
    return o


def num_threads():
    return int(lib().orc_num_threads())


def points_hf_sdf(points, hf, min_box_center, dxdy, base_z=-10.0, inverted=True, radius=None):
    """numpy restatement of terrain_util.points_hf_sdf (util/terrain_util.py:1835-1893, with points_boxes_sdf :1774-1804 and
    geom_util.sdBox util/geom_util.py:122-143): fp32 throughout, min over all cells taken batch by batch, 64 points at a time."""
    points, hf, mbc, dxdy = _f(points), _f(hf), _f(min_box_center), _f(dxdy)
    B, N = points.shape[:2]
    X, Y = hf.shape[1:]

    def linspace32(end, n):          # torch.linspace(0, end, n) in fp32: forward steps up to the middle, backward from the end after it
        if n == 1:
            return np.zeros(1, np.float32)
        step = np.float32(np.float32(end) / np.float32(n - 1))
        i = np.arange(n)
        return np.where(i < n // 2, np.float32(0) + step * i.astype(np.float32), np.float32(end) - step * (n - 1 - i).astype(np.float32)).astype(np.float32)

    xs = linspace32((X - 1.0) * float(dxdy[0]), X)
    ys = linspace32((Y - 1.0) * float(dxdy[1]), Y)
    out = np.zeros((B, N), np.float32)
    for b in range(B):
        cx = np.repeat(xs + mbc[b, 0], Y)
        cy = np.tile(ys + mbc[b, 1], X)
        h = hf[b].reshape(-1)
        if inverted:
            top = np.float32(-base_z)
            cz, hz = (h + top) / np.float32(2), (top - h) / np.float32(2)
        else:
            cz, hz = (h + np.float32(base_z)) / np.float32(2), (h - np.float32(base_z)) / np.float32(2)
        c = np.stack([cx, cy, cz], -1).astype(np.float32)
        hd = np.stack([np.full_like(cx, dxdy[0] / np.float32(2)), np.full_like(cx, dxdy[1] / np.float32(2)), hz], -1).astype(np.float32)
        for s in range(0, N, 64):
            q = np.abs(points[b, s:s + 64, None, :] - c[None]) - hd[None]
            pos = np.maximum(q, np.float32(0))
            sd = np.sqrt((pos * pos).sum(-1, dtype=np.float32)) + np.minimum(q.max(-1), np.float32(0))
            if radius is not None:
                sd = sd - np.float32(radius)
            out[b, s:s + 64] = sd.min(-1)
    return -out if inverted else out
