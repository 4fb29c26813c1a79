#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own Python (CPU, torch).

Runs ONLY in the build container (needs /root/reference).  The reference's torch code for the
kinematic / observation / reward / termination / TD(lambda) half of the tracker imports on CPU once
empty stub modules are registered for the third-party packages it names but never calls on this
path (isaacgym, gym, trimesh, wandb, tensorboardX ...).  This script calls those functions on seeded
inputs and stores inputs + outputs as small .npz files.  Nothing from the reference is copied: the
fixtures are data.

The two motion clips that ship with the reference (data/terrains/*.pkl) are read with the
non-executing reader parc_amd.util.safe_pickle (they are never unpickled); temporary motion files that
the reference's MotionLib then loads are written by this script itself.

usage:  python tests/golden/gen_golden.py            # every stage, rewrites tests/golden/*.npz
        python tests/golden/gen_golden.py --check    # regenerate into a scratch dir and compare with the committed fixtures
        python tests/golden/gen_golden.py --only-<stage>
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PARC_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)


def _stub(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


for _n in ["trimesh", "gym", "gym.spaces", "isaacgym", "isaacgym.gymapi", "isaacgym.gymtorch",
           "isaacgym.gymutil", "wandb", "tensorboardX", "polyscope", "cv2", "embreex"]:
    _stub(_n)
sys.modules["gym"].spaces = sys.modules["gym.spaces"]


class _Box:                       # what the reference's agent checks with isinstance(space, gym.spaces.Box)
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, shape if shape is not None else low.shape, dtype


class _Discrete:
    pass


sys.modules["gym.spaces"].Box = _Box
sys.modules["gym.spaces"].Discrete = _Discrete
sys.path.insert(0, REF)
os.chdir(REF)  # the reference opens data/assets/humanoid.xml relative to its root

import torch  # noqa: E402

torch.set_num_threads(4)

import util.torch_util as torch_util  # noqa: E402
import anim.kin_char_model as kin_char_model  # noqa: E402
import anim.motion_lib as motion_lib  # noqa: E402
import util.terrain_util as terrain_util  # noqa: E402
import util.geom_util as geom_util  # noqa: E402
import learning.rl_util as rl_util  # noqa: E402
import envs.ig_parkour.mgdm_dm_util as dmu  # noqa: E402
import envs.ig_char_env as ig_char_env  # noqa: E402

from parc_amd.util.safe_pickle import load_motion_file_safe  # noqa: E402

OUT = HERE
CHAR_FILE = "data/assets/humanoid.xml"


def t(x, dtype=torch.float32):
    return torch.tensor(np.asarray(x), dtype=dtype)


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def rand_quat(rng, n):
    q = rng.standard_normal((n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    return q


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print("wrote", path, {k: npy(v).shape for k, v in arrs.items()})


# --------------------------------------------------------------------------------------------
def gen_quat(rng):
    n = 192
    a = rand_quat(rng, n)
    b = rand_quat(rng, n)
    # edge cases: identity, w<0, antipodal pairs, nearly equal, tiny angle
    a[0] = [0, 0, 0, 1]
    b[0] = [0, 0, 0, 1]
    a[1] = [0, 0, 0, -1]
    b[2] = -a[2]
    b[3] = a[3]
    b[4] = a[4] + 1e-4 * rng.standard_normal(4).astype(np.float32)
    b[4] /= np.linalg.norm(b[4])
    a[5] = [1e-7, 0, 0, 1]
    a[5] /= np.linalg.norm(a[5])
    a[6] = [0, 0, 1, 0]  # 180 deg about z
    v = rng.standard_normal((n, 3)).astype(np.float32)
    em = (rng.standard_normal((n, 3)) * 1.5).astype(np.float32)
    em[0] = 0.0
    em[1] = [1e-6, 0, 0]
    em[2] = [0, 0, 3.5]       # |theta| > pi  -> normalize_angle wraps
    em[3] = [4.0, 4.0, 0.0]
    ang = (rng.standard_normal(n) * 2.0).astype(np.float32)
    axis = rng.standard_normal((n, 3)).astype(np.float32)
    tt = rng.random(n).astype(np.float32)
    tt[0:3] = [0.0, 1.0, 0.5]
    A, B, V, EM = t(a), t(b), t(v), t(em)
    save("g1_quat",
         a=a, b=b, v=v, exp_map=em, angle=ang, axis=axis, blend=tt,
         quat_mul=torch_util.quat_mul(A, B),
         quat_rotate=torch_util.quat_rotate(A, V),
         exp_map_to_quat=torch_util.exp_map_to_quat(EM),
         quat_to_exp_map=torch_util.quat_to_exp_map(A),
         axis_angle_to_quat=torch_util.axis_angle_to_quat(t(axis), t(ang)),
         quat_to_tan_norm=torch_util.quat_to_tan_norm(A),
         slerp=torch_util.slerp(A, B, t(tt)),
         calc_heading=torch_util.calc_heading(A),
         calc_heading_quat_inv=torch_util.calc_heading_quat_inv(A),
         quat_diff_angle=torch_util.quat_diff_angle(A, B),
         )


def load_char():
    km = kin_char_model.KinCharModel("cpu")
    km.load_char_file(CHAR_FILE)
    return km


def gen_char(km):
    """Character description the build's own MJCF parser must reproduce (a3)."""
    joint_type = []
    axes = []
    dof_idx = []
    for j in range(km.get_num_joints()):
        jt = km._joints[j]
        joint_type.append(jt.joint_type.value)
        axes.append(npy(jt.axis) if jt.axis is not None else np.zeros(3, np.float32))
        dof_idx.append(jt.dof_idx)
    save("g2_char",
         parent=npy(km._parent_indices), local_translation=km._local_translation,
         local_rotation=km._local_rotation, joint_type=np.array(joint_type, np.int32),
         joint_axis=np.stack(axes).astype(np.float32), dof_idx=np.array(dof_idx, np.int32),
         lower=km._lower_dof_limits, upper=km._upper_dof_limits,
         body_names=np.array(km.get_body_names()))


def gen_kin(rng, km, clip_frames):
    n = 96
    dof = (rng.standard_normal((n, 28)) * 0.8).astype(np.float32)
    dof[0] = 0.0
    dof[1] = clip_frames[10, 6:]
    dof[2] = clip_frames[100, 6:]
    dof[3, 0:3] = [0.0, 0.0, 3.3]   # spherical exp-map beyond pi
    root_pos = (rng.standard_normal((n, 3)) * 2.0).astype(np.float32)
    root_rot = rand_quat(rng, n)
    D = t(dof)
    jr = km.dof_to_rot(D)
    back = km.rot_to_dof(jr)
    bp, br = km.forward_kinematics(t(root_pos), t(root_rot), jr)
    # arbitrary unit quats through rot_to_dof (hinge sign branch, w<0)
    q = rand_quat(rng, n * 14).reshape(n, 14, 4)
    dof_from_q = km.rot_to_dof(t(q))
    save("g2_kin", dof=dof, root_pos=root_pos, root_rot=root_rot, joint_rot=jr, dof_back=back,
         body_pos=bp, body_rot=br, rand_joint_rot=q, dof_from_rand=dof_from_q)


def make_synth_clip(rng, km, num_frames, fps, seed_frames):
    """Smooth random clip: low-frequency sinusoids around a seed pose."""
    tt = np.arange(num_frames, dtype=np.float32) / fps
    base = seed_frames[rng.integers(0, seed_frames.shape[0])].copy()
    frames = np.tile(base[None], (num_frames, 1)).astype(np.float32)
    for d in range(34):
        amp = 0.3 if d >= 3 else 0.5
        frames[:, d] += amp * np.sin(2 * np.pi * (0.2 + rng.random()) * tt + rng.random() * 6.28).astype(np.float32)
    frames[:, 0] += 1.2 * tt  # walk along +x
    frames[:, 2] = np.abs(frames[:, 2]) + 0.8
    contacts = (rng.random((num_frames, 15)) > 0.7).astype(np.float32)
    return frames, contacts


def write_motion_file(path, frames, contacts, fps, loop_mode, terrain):
    data = {"fps": fps, "loop_mode": loop_mode, "frames": frames, "contacts": contacts, "terrain": terrain}
    with open(path, "wb") as f:
        pickle.dump(data, f)


def ref_terrain_from_dict(d):
    hf = d["hf"]
    ter = terrain_util.SubTerrain("terrain", hf.shape[0], hf.shape[1], float(d["dxdy"][0]), float(d["dxdy"][1]),
                                  float(d["min_point"][0]), float(d["min_point"][1]), device="cpu")
    ter.hf[:] = t(hf)
    ter.hf_mask[:] = torch.tensor(d["hf_mask"])
    ter.hf_maxmin[:] = t(d["hf_maxmin"])
    return ter


def gen_motion(rng, km, civ, teaser):
    tmp = tempfile.mkdtemp(prefix="parc_golden_")
    clips = []
    # clip 0: the real shipped clip (CLAMP, 30 fps); clip 1: the other real clip
    clips.append(("civ", civ["frames"], civ["contacts"], 30, "CLAMP"))
    clips.append(("teaser", teaser["frames"], teaser["contacts"], 30, "CLAMP"))
    f2, c2 = make_synth_clip(rng, km, 45, 30, civ["frames"])
    clips.append(("synth_wrap", f2, c2, 30, "WRAP"))
    f3, c3 = make_synth_clip(rng, km, 5, 20, civ["frames"])
    clips.append(("synth_short", f3, c3, 20, "CLAMP"))
    ter = ref_terrain_from_dict(civ["terrain"]).numpy_copy()
    entries = []
    for name, fr, co, fps, lm in clips:
        p = os.path.join(tmp, name + ".pkl")
        write_motion_file(p, fr.astype(np.float32), co.astype(np.float32), fps, lm, ter)
        entries.append({"file": p, "weight": float(1.0 + len(entries))})
    import yaml
    ypath = os.path.join(tmp, "motions.yaml")
    with open(ypath, "w") as f:
        yaml.safe_dump({"motions": entries}, f)
    mlib = motion_lib.MotionLib(ypath, km, "cpu", contact_info=True)

    nq = 400
    ids = rng.integers(0, len(clips), nq).astype(np.int64)
    lens = npy(mlib._motion_lengths)[ids]
    times = (rng.random(nq).astype(np.float32) * 1.4 - 0.15) * lens   # incl. <0 and > len
    # exact frame times and the clip end
    times[0:4] = [0.0, lens[1], 1.0 / 30.0, 2.0 / 30.0]
    ids[0:4] = [0, 1, 0, 0]
    times[4] = npy(mlib._motion_lengths)[2] * 2.5
    ids[4] = 2  # wrap clip, several loops
    out = mlib.calc_motion_frame(t(ids, torch.int64), t(times))
    root_pos, root_rot, root_vel, root_ang_vel, joint_rot, dof_vel, contacts = out
    dof_pos = mlib.joint_rot_to_dof(joint_rot)
    save("g3_motion",
         clip_names=np.array([c[0] for c in clips]),
         clip_fps=np.array([c[3] for c in clips], np.float32),
         clip_loop=np.array([motion_lib.LoopMode[c[4]].value for c in clips], np.int32),
         clip_weights_in=np.array([e["weight"] for e in entries], np.float32),
         frames_0=clips[0][1], frames_1=clips[1][1], frames_2=clips[2][1], frames_3=clips[3][1],
         contacts_0=clips[0][2], contacts_1=clips[1][2], contacts_2=clips[2][2], contacts_3=clips[3][2],
         motion_weights=mlib._motion_weights, motion_num_frames=mlib._motion_num_frames,
         motion_lengths=mlib._motion_lengths, motion_start_idx=mlib._motion_start_idx,
         motion_root_pos_delta=mlib._motion_root_pos_delta,
         frame_root_pos=mlib._frame_root_pos, frame_root_rot=mlib._frame_root_rot,
         frame_root_vel=mlib._frame_root_vel, frame_root_ang_vel=mlib._frame_root_ang_vel,
         frame_joint_rot=mlib._frame_joint_rot, frame_dof_vel=mlib._frame_dof_vel,
         frame_contacts=mlib._frame_contacts,
         q_ids=ids, q_times=times,
         q_root_pos=root_pos, q_root_rot=root_rot, q_root_vel=root_vel, q_root_ang_vel=root_ang_vel,
         q_joint_rot=joint_rot, q_dof_vel=dof_vel, q_contacts=contacts, q_dof_pos=dof_pos)
    return mlib, clips


def gen_rays():
    pts = geom_util.get_xy_points_cone(center=torch.zeros(2), dx=0.05, num_neg=2, num_pos=60,
                                       num_rays_neg=3, num_rays_pos=3, angle_between_rays=0.26179938779)
    save("g4_rays", ray_xy_points=pts, params=np.array([0.05, 2, 60, 3, 3, 0.26179938779], np.float64))
    return pts


def gen_heightmap(rng, civ, teaser, rays):
    for name, d in (("civ", civ), ("teaser", teaser)):
        ter = ref_terrain_from_dict(d["terrain"])
        n = 128
        lo = npy(ter.min_point)
        hi = lo + npy(ter.dims).astype(np.float32) * npy(ter.dxdy)
        xy = (lo + (hi - lo) * rng.random((n, 2))).astype(np.float32)
        # some roots outside the grid -> index clamping; some high/low -> +-3 clamp
        xy[0] = lo - 2.5
        xy[1] = hi + 1.0
        z = (rng.random(n) * 4.0 - 0.5).astype(np.float32)
        z[2] = 9.0
        z[3] = -7.0
        heading = ((rng.random(n) * 2 - 1) * np.pi).astype(np.float32)
        heading[4] = 0.0
        root = np.concatenate([xy, z[:, None]], -1)
        # what RefCharEnv._refresh_ray_obs_hfs does (mgdm_dm_util.py:158-179)
        R = t(rays).unsqueeze(0).expand(n, -1, -1)
        H = t(heading).unsqueeze(-1).expand(-1, R.shape[1])
        pts = torch_util.rotate_2d_vec(R, H) + t(xy).unsqueeze(1)
        pts_flat = pts.reshape(-1, 2)
        hf = terrain_util.get_local_hf_from_terrain(pts_flat, ter).view(n, -1)
        out = torch.clamp(hf - t(z).unsqueeze(-1), min=-3.0, max=3.0)
        # distance of every query to the nearest rounding boundary (for tolerance-aware tests)
        u = (pts_flat - ter.min_point) / ter.dxdy
        frac = torch.abs(u - torch.floor(u) - 0.5).min(dim=-1)[0].view(n, -1)
        save("g5_hf_" + name, hf=ter.hf, min_point=ter.min_point, dxdy=ter.dxdy, dims=npy(ter.dims),
             root_pos=root, heading=heading, ray_hfs=out, raw_hf=hf, boundary_dist=frac,
             grid_index=ter.get_grid_index(pts_flat).view(n, -1, 2))


def gen_obs_reward_done(rng, km, mlib, civ, rays):
    """One full post-physics pass of the tracker on a made-up simulator state (K3,K2,K4,K5..K10)."""
    n = 64
    device = "cpu"
    M = mlib.num_motions()
    ter = ref_terrain_from_dict(civ["terrain"])
    motion_ids = t(rng.integers(0, M, n), torch.int64)
    time_buf = t(rng.integers(0, 200, n).astype(np.float32) / 30.0)
    time_buf[0] = 0.0       # first step -> no fail
    time_buf[1] = 10.0      # timeout
    motion_time_offsets = t(rng.random(n).astype(np.float32)) * mlib._motion_lengths[motion_ids]
    motion_offsets = t((rng.standard_normal((M, 1, 2)) * 3.0).astype(np.float32))
    env_offsets = torch.zeros(n, 3)
    num_env_per_row = int(np.sqrt(n))
    for i in range(n):
        env_offsets[i, 0] = 2.0 * 2 * (i % num_env_per_row)
        env_offsets[i, 1] = 2.0 * 2 * (i // num_env_per_row)
    terrain_ids = torch.zeros(n, dtype=torch.int64)
    key_body_ids = t([km.get_body_id(b) for b in ["right_hand", "left_hand", "right_foot", "left_foot"]], torch.int64)
    tar_obs_steps = t([1, 2, 3, 10, 20, 30], torch.int)
    timestep = 1.0 / 30.0

    def move_to_terrain(pos, ids=None):
        off = motion_offsets[motion_ids, terrain_ids] - env_offsets[:, 0:2]
        while len(off.shape) < len(pos.shape):
            off = off.unsqueeze(1)
        return pos + off

    # ---- reference state (DeepMimicEnv._update_ref_motion, dm_env.py:570-595)
    motion_times = time_buf + motion_time_offsets
    rp, rr, rv, rav, jr, dv, cont = mlib.calc_motion_frame(motion_ids, motion_times)
    rp[..., 0:2] = move_to_terrain(rp[..., 0:2])
    ref_body_pos, _ = km.forward_kinematics(rp, rr, jr)
    ref_dof_pos = mlib.joint_rot_to_dof(jr)

    # ---- simulated character = reference + perturbation
    char_root_pos = rp + t(rng.standard_normal((n, 3)).astype(np.float32)) * 0.15
    char_root_pos[2] = rp[2] + t([1.0, 0.0, 0.0])          # root pos fail
    dq = torch_util.exp_map_to_quat(t(rng.standard_normal((n, 3)).astype(np.float32)) * 0.3)
    dq[3] = torch_util.exp_map_to_quat(t([[0.0, 0.0, 1.5]]))[0]  # root rot fail
    char_root_rot = torch_util.quat_mul(dq, rr)
    char_root_vel = rv + t(rng.standard_normal((n, 3)).astype(np.float32)) * 0.5
    char_root_ang_vel = rav + t(rng.standard_normal((n, 3)).astype(np.float32)) * 0.5
    char_dof_pos = ref_dof_pos + t(rng.standard_normal((n, 28)).astype(np.float32)) * 0.2
    char_dof_pos[4, 17:20] += 2.5                           # big pose error (right hip) -> pose fail
    char_dof_vel = dv + t(rng.standard_normal((n, 28)).astype(np.float32)) * 1.0
    char_joint_rot = km.dof_to_rot(char_dof_pos)
    sim_body_pos, sim_body_rot = km.forward_kinematics(char_root_pos, char_root_rot, char_joint_rot)
    # the simulator's rigid-body positions differ slightly from kinematic FK in general
    char_rigid_body_pos = sim_body_pos + t(rng.standard_normal((n, 15, 3)).astype(np.float32)) * 0.01
    char_rigid_body_pos[:, 0] = char_root_pos
    contact_forces = t(rng.standard_normal((n, 15, 3)).astype(np.float32)) * 0.6
    contact_forces[rng.random((n, 15)) > 0.4] = 0.0
    contact_forces[5, 14] = t([0.0, 0.0, 250.0])

    # ---- heightmap (IGParkourEnv._refresh_obs_hfs, ig_parkour_env.py:636-656)
    glob = char_root_pos + env_offsets
    heading = torch_util.calc_heading(char_root_rot)
    R = rays.unsqueeze(0).expand(n, -1, -1)
    H = heading.unsqueeze(-1).expand(-1, R.shape[1])
    pts = (torch_util.rotate_2d_vec(R, H) + glob[:, 0:2].unsqueeze(1)).reshape(-1, 2)
    ray_hfs = terrain_util.get_local_hf_from_terrain(pts, ter).view(n, -1) - glob[:, 2].unsqueeze(-1)
    ray_hfs = torch.clamp(ray_hfs, min=-3.0, max=3.0)
    u = (pts - ter.min_point) / ter.dxdy
    hf_boundary = torch.abs(u - torch.floor(u) - 0.5).min(dim=-1)[0].view(n, -1)

    # ---- target obs (DeepMimicEnv.compute_tar_obs, dm_env.py:686-718)
    tp, tr, tj, tc = dmu.fetch_tar_obs_data(motion_ids, motion_times, mlib, timestep, tar_obs_steps)
    tp[..., 0:2] = move_to_terrain(tp[..., 0:2])
    tbp, _ = km.forward_kinematics(tp.reshape(-1, 3), tr.reshape(-1, 4), tj.reshape(-1, 14, 4))
    tbp = tbp.reshape(n, 6, 15, 3)
    tar_key_pos = tbp[..., key_body_ids, :]

    # ---- obs (IGParkourEnv._compute_obs, ig_parkour_env.py:1054-1244)
    key_pos = sim_body_pos[..., key_body_ids, :]
    obs_dict = dmu.compute_deepmimic_obs(root_pos=char_root_pos, root_rot=char_root_rot, root_vel=char_root_vel,
                                         root_ang_vel=char_root_ang_vel, joint_rot=char_joint_rot,
                                         dof_vel=char_dof_vel, key_pos=key_pos, global_obs=False,
                                         root_height_obs=False, enable_tar_obs=True, tar_root_pos=tp,
                                         tar_root_rot=tr, tar_joint_rot=tj, tar_key_pos=tar_key_pos)
    char_obs = obs_dict["char_obs"]
    tar_obs = obs_dict["tar_obs"].reshape(n, -1)
    char_contacts = (torch.norm(contact_forces, dim=-1) > 1e-5).float()
    obs = torch.cat([char_obs, tar_obs, tc.reshape(n, -1), char_contacts, ray_hfs], dim=-1)
    assert obs.shape[1] == 1312, obs.shape

    # ---- reward (IGParkourEnv._update_reward, ig_parkour_env.py:1275-1339,1399-1404)
    joint_err_w = t([1.0, 0.6, 0.6, 0.4, 0.0, 0.6, 0.4, 0.0, 1.0, 0.6, 0.4, 1.0, 0.6, 0.4])
    dof_err_w = torch.zeros(28)
    for j in range(1, 15):
        dd = km.get_joint_dof_dim(j)
        if dd > 0:
            di = km.get_joint_dof_idx(j)
            dof_err_w[di:di + dd] = joint_err_w[j - 1]
    comp = dmu.compute_deepmimic_reward(
        root_pos=char_root_pos, root_rot=char_root_rot, root_vel=char_root_vel, root_ang_vel=char_root_ang_vel,
        joint_rot=char_joint_rot, dof_vel=char_dof_vel, key_pos=char_rigid_body_pos[..., key_body_ids, :],
        tar_root_pos=rp, tar_root_rot=rr, tar_root_vel=rv, tar_root_ang_vel=rav, tar_joint_rot=jr,
        tar_dof_vel=dv, tar_key_pos=ref_body_pos[..., key_body_ids, :],
        joint_rot_err_w=joint_err_w, dof_err_w=dof_err_w, track_root_h=True, track_root=True)
    w = np.array([0.5, 0.1, 0.15, 0.1, 0.15])
    w = w / w.sum()
    contact_w = torch.full((15,), 5.0)
    contact_pen = torch.mean(dmu.compute_contact_reward(cont, contact_forces, contact_w), dim=-1)
    reward = (float(w[0]) * comp[:, 0] + float(w[1]) * comp[:, 1] + float(w[2]) * comp[:, 2]
              + float(w[3]) * comp[:, 3] + float(w[4]) * comp[:, 4]) + contact_pen
    reward = 1.0 * reward

    # ---- done (RefCharEnv.update_done + DeepMimicEnv.update_done, mgdm_dm_util.py:205-230, dm_env.py:720-783)
    pose_term = t([0.7, 1.0, 0.7, 0.7, 0.7, 0.7, 0.7, 0.7, 1.0, 1.2, 10.0, 1.0, 1.2, 10.0])
    global_body_pos = char_rigid_body_pos[..., 0:2] + env_offsets[:, 0:2].unsqueeze(1)
    bgi = ter.get_grid_index(global_body_pos)
    term_h = ter.hf[bgi[..., 0], bgi[..., 1]] + 0.15
    done_buf = torch.zeros(n, dtype=torch.int)
    outs = {}
    for tag, cb in (("nocontact", torch.zeros(0, dtype=torch.int64)),
                    ("feet", t([km.get_body_id("right_foot"), km.get_body_id("left_foot")], torch.int64))):
        d = dmu.compute_done(done_buf=done_buf, time=time_buf, ep_len=10.0, root_rot=char_root_rot,
                             body_pos=char_rigid_body_pos, char_root_pos=char_root_pos, tar_root_rot=rr,
                             tar_body_pos=ref_body_pos, contact_force=contact_forces, contact_body_ids=cb,
                             termination_heights=term_h, pose_termination=True, pose_termination_dist=pose_term,
                             global_obs=False, enable_early_termination=True, track_root=True,
                             root_pos_termination_dist=0.6, root_rot_termination_angle=1.309)
        outs[tag] = d.clone()
    motion_len = mlib.get_motion_length(motion_ids)
    motion_end = torch.logical_and(motion_times >= motion_len,
                                   mlib.get_motion_loop_mode(motion_ids) != motion_lib.LoopMode.WRAP.value)
    done_final = outs["nocontact"].clone()
    # fail-rate EMA (sequential over done envs, dm_env.py:758-772)
    fail_rates = torch.ones(M)
    done_any = torch.logical_or(done_final != 0, motion_end)
    for e in torch.nonzero(done_any).flatten().tolist():
        m = motion_ids[e]
        if done_final[e] == 1:
            fail_rates[m] = fail_rates[m] * (1.0 - 0.01) + 0.01
        else:
            fail_rates[m] = fail_rates[m] * (1.0 - 0.01)
    done_final[motion_end] = 1

    # tracking error (mgdm_dm_util.py:578-611)
    rb_pos, rb_rot = km.forward_kinematics(rp, rr, jr)
    terr = dmu.compute_tracking_error(char_root_pos, char_root_rot, sim_body_rot, sim_body_pos, rp, rr, rb_rot,
                                      rb_pos, char_root_vel, char_root_ang_vel, char_dof_vel, rv, rav, dv)

    save("g6_step",
         motion_ids=motion_ids, time_buf=time_buf, motion_time_offsets=motion_time_offsets,
         motion_offsets=motion_offsets, env_offsets=env_offsets, key_body_ids=key_body_ids,
         tar_obs_steps=tar_obs_steps, hf=ter.hf, min_point=ter.min_point, dxdy=ter.dxdy,
         rays=rays,
         ref_root_pos=rp, ref_root_rot=rr, ref_root_vel=rv, ref_root_ang_vel=rav, ref_joint_rot=jr,
         ref_dof_vel=dv, ref_contacts=cont, ref_body_pos=ref_body_pos, ref_dof_pos=ref_dof_pos,
         char_root_pos=char_root_pos, char_root_rot=char_root_rot, char_root_vel=char_root_vel,
         char_root_ang_vel=char_root_ang_vel, char_dof_pos=char_dof_pos, char_dof_vel=char_dof_vel,
         char_rigid_body_pos=char_rigid_body_pos, contact_forces=contact_forces,
         ray_hfs=ray_hfs, hf_boundary=hf_boundary,
         tar_root_pos=tp, tar_root_rot=tr, tar_joint_rot=tj, tar_contacts=tc, tar_key_pos=tar_key_pos,
         char_obs=char_obs, tar_obs=tar_obs, obs=obs,
         joint_err_w=joint_err_w, dof_err_w=dof_err_w, reward_terms=comp, contact_penalty=contact_pen,
         reward=reward, reward_w=w.astype(np.float32),
         pose_termination_dist=pose_term, termination_heights=term_h,
         done_nocontact=outs["nocontact"], done_feet=outs["feet"], motion_end=motion_end,
         done_final=done_final, fail_rates=fail_rates, tracking_error=terr)


def gen_td_lambda(rng):
    T, n = 32, 96
    r = rng.random((T, n)).astype(np.float32)
    v = (rng.random((T, n)) * 100.0).astype(np.float32)
    done = rng.choice([0, 0, 0, 0, 0, 0, 0, 1, 2, 3], size=(T, n)).astype(np.int32)
    ret = rl_util.compute_td_lambda_return(t(r), t(v), t(done, torch.int), 0.99, 0.95)
    # advantage normalisation (dm_ppo_agent.py:393-403)
    vals = (rng.random((T, n)) * 100.0).astype(np.float32)
    mask = (rng.random((T, n)) > 0.1).astype(np.float32)
    adv = ret - t(vals)
    ra = adv.flatten()[(t(mask) == 1.0).flatten()]
    std, mean = torch.std_mean(ra)
    norm_adv = torch.clamp((adv - mean) / torch.clamp_min(std, 1e-5), -4.0, 4.0)
    # T=1 edge case
    ret1 = rl_util.compute_td_lambda_return(t(r[:1]), t(v[:1]), t(done[:1], torch.int), 0.99, 0.95)
    save("g9_td_lambda", r=r, next_vals=v, done=done, ret=ret, vals=vals, rand_action_mask=mask,
         adv_mean=mean, adv_std=std, norm_adv=norm_adv, ret_T1=ret1,
         params=np.array([0.99, 0.95, 4.0], np.float64))


def gen_voxel_mesh(rng):
    """G10: heightfield -> voxelised triangle mesh (util/terrain_util.py:1099-1251).  Case A is the authors' own output:
    data/terrains/civilization_ig.pkl holds the terrain together with the verts / tris their build produced from it
    (dm_env.py:143-148, padding 0).  Case B runs the reference function on a small random field with padding 0.8."""
    ig = load_motion_file_safe(os.path.join(REF, "data/terrains/civilization_ig.pkl"))
    # the terrain inside the _ig file is stored as torch tensors (not materialised by the non-executing reader); it was built
    # from civilization.pkl (data/terrains/civilization_motions.yaml, terrain_build_mode "file"), whose arrays are numpy
    ter = load_motion_file_safe(os.path.join(REF, "data/terrains/civilization.pkl"))["terrain"]
    hf = np.asarray(ter["hf"], np.float32)
    a_v = np.asarray(ig["all_terrain_verts"][0]).reshape(-1, 3)
    a_t = np.asarray(ig["all_terrain_tris"][0]).reshape(-1, 3)
    # the reference function reproduces the shipped arrays (checked here so the fixture is known to be that function's output)
    # (the reference passes min_point / dxdy as .item() of float32 tensors: dx is the double value of float32(0.4))
    mn = np.asarray(ter["min_point"], np.float32).astype(np.float64)
    dx = float(np.float32(ter["dxdy"][0]))
    v, tr = terrain_util.convert_heightfield_to_voxelized_trimesh(torch.tensor(hf), float(mn[0]), float(mn[1]), dx, padding=0)
    assert np.array_equal(v, a_v) and np.array_equal(tr, a_t)
    hf_b = (rng.integers(-2, 4, size=(7, 5)) * 0.3).astype(np.float32)
    b_v, b_t = terrain_util.convert_heightfield_to_voxelized_trimesh(torch.tensor(hf_b), -1.3, 0.7, 0.4, padding=0.8)
    save("g10_voxel_mesh", a_hf=hf, a_min_point=mn, a_dx=np.float64(dx), a_verts=a_v,
         a_tris=a_t, b_hf=hf_b, b_min_point=np.array([-1.3, 0.7]), b_dx=np.float64(0.4), b_padding=np.float64(0.8), b_verts=b_v, b_tris=b_t)


def gen_dataset_yaml():
    """G11: PARC/util/create_dataset.py on the folder tree of tests/golden/dataset_tree.py (class-balanced weights, class
    halving, large-terrain and bad-loss exclusion, ignore folders)."""
    import json
    import yaml
    from pathlib import Path
    sys.path.insert(0, HERE)
    import dataset_tree
    import PARC.util.create_dataset as ref_cd
    # the function samples character surface points through trimesh (not installed) before it knows whether the
    # preprocessing step is requested; with compute_preprocessing_data=False the points are never used
    # (patched for the duration of this call only: gen_terrain_geometry needs the real function afterwards)
    real_samples = ref_cd.geom_util.get_char_point_samples
    ref_cd.geom_util.get_char_point_samples = lambda char_model: None

    def make_terrain(hf):
        t = terrain_util.SubTerrain("t", hf.shape[0], hf.shape[1], 0.4, 0.4, 0.0, 0.0, device="cpu")
        return t.numpy_copy()
    root = tempfile.mkdtemp(prefix="parc_ds_")
    folders = dataset_tree.build(root, make_terrain)
    out = os.path.join(root, "out.yaml")
    try:
        ref_cd.create_dataset_yaml([Path(f) for f in folders], Path(out), os.path.join(REF, "data/assets/humanoid.xml"), False, True,
                                   ["running"], 45, 45)
    finally:
        ref_cd.geom_util.get_char_point_samples = real_samples
    with open(out) as f:
        y = yaml.safe_load(f)
    entries = [{"file": os.path.relpath(m["file"], root), "weight": float(m["weight"])} for m in y["motions"]]
    with open(os.path.join(OUT, "g11_dataset_yaml.json"), "w") as f:
        json.dump({"cut_classes": ["running"], "max_dim": [45, 45], "motions": entries}, f, indent=1)
    print("wrote g11_dataset_yaml.json", len(entries), "entries")


def _seed_all(k):
    import random
    torch.manual_seed(k)
    random.seed(k)
    np.random.seed(k)


def gen_procgen():
    """G12: the procedural terrain generators (util/terrain_util.py:320-470,544-595,864-1043) under fixed torch / random /
    numpy seeds: boxes with the kin-gen parameters (parc_2_kin_gen.py:36-43), stairs, curvy paths, gap / vault course."""
    out = {}
    _seed_all(12)
    hf = torch.zeros((16, 16), dtype=torch.float32)
    terrain_util.add_boxes_to_hf2(hf, box_max_height=3.0, box_min_height=-3.0, num_boxes=10, box_max_len=10, box_min_len=5)
    out["boxes_hf"] = hf.numpy().copy()
    _seed_all(13)
    t = terrain_util.SubTerrain("s", 24, 20, 0.4, 0.4, -1.0, 0.5, device="cpu")
    terrain_util.add_stairs_to_hf(t, num_stairs=2)
    out["stairs_hf"] = t.hf.numpy().copy()
    _seed_all(14)
    t = terrain_util.SubTerrain("p", 30, 28, 0.4, 0.4, -2.0, -3.0, device="cpu")
    terrain_util.gen_paths_hf(t, num_paths=3)
    out["paths_hf"] = t.hf.numpy().copy()
    _seed_all(15)
    # (numpy terrain: with torch fields the reference's round(y / dy) raises under torch 2.10)
    t = terrain_util.SubTerrain("c", 6, 400, 0.1, 0.1, 0.0, 0.0, device="cpu").numpy_copy()
    t2, v, tr = terrain_util.random_linear_parkour_course(t, gap_width=11, gap_height=-1.0, vault_width=1, vault_height=1.0,
                                                          num_padding_cells=4)
    out["course_hf"] = np.asarray(t2.hf).copy()
    out["course_verts"] = v
    out["course_tris"] = tr
    save("g12_procgen", **out)


def gen_terrain_geometry():
    """G13: the terrain / body-geometry functions either side of the tracker, outputs of the reference's
    util/geom_util.get_char_point_samples, util/terrain_util.points_hf_sdf (:1835), motion_frames_hf_sdf_loss (:1895),
    compute_hf_extra_vals (:2017) and slice_terrain_around_motion (:1675).  trimesh is absent, so the character's sphere
    geoms (which the reference samples through trimesh.creation.icosphere) are removed from the reference model's geom lists
    before sampling: the point sets cover its capsules and boxes."""
    rng = np.random.default_rng(13)
    km = load_char()
    for b in range(km.get_num_joints()):
        km._geoms[b] = [g for g in km._geoms[b] if g._shape_type != kin_char_model.GeomType.SPHERE]
    pts = geom_util.get_char_point_samples(km)
    out = {"pts_count": np.array([p.shape[0] for p in pts]), "pts": torch.cat(pts, dim=0)}
    out["box_pts"] = geom_util.get_box_point_surface_samples(t([0.0885, 0.045, 0.0275]), "cpu", num_slices=3, dim_x=4, dim_y=5)
    out["capsule_pts"] = geom_util.get_capsule_point_surface_samples(0.31, 0.055, "cpu", num_cylinder_slices=5, num_circle_points=6)
    # points_hf_sdf on random fields
    B, N, X, Y = 3, 700, 9, 7
    hf = (rng.integers(-3, 4, size=(B, X, Y)) * 0.35).astype(np.float32)
    mbc = rng.uniform(-1, 1, size=(B, 2)).astype(np.float32)
    dxdy = np.array([0.4, 0.3], np.float32)
    p = np.concatenate([rng.uniform(-1.5, 4.5, size=(B, N, 1)), rng.uniform(-1.5, 3.0, size=(B, N, 1)), rng.uniform(-2.0, 2.0, size=(B, N, 1))],
                       axis=-1).astype(np.float32)
    out.update(sdf_points=p, sdf_hf=hf, sdf_mbc=mbc, sdf_dxdy=dxdy)
    out["sdf_inverted"] = terrain_util.points_hf_sdf(t(p), t(hf), t(mbc), t(dxdy))
    out["sdf_plain"] = terrain_util.points_hf_sdf(t(p), t(hf), t(mbc), t(dxdy), base_z=-5.0, inverted=False)
    out["sdf_round"] = terrain_util.points_hf_sdf(t(p), t(hf), t(mbc), t(dxdy), inverted=False, radius=0.07)
    # penetration loss of two clips' worth of frames on the civilization terrain (lowered into the ground to get hits)
    civ = load_motion_file_safe(os.path.join(REF, "data/terrains/civilization.pkl"))
    frames = np.asarray(civ["frames"], np.float32)
    ter = ref_terrain_from_dict(civ["terrain"])
    mf = np.stack([frames[0:24], frames[100:124]], axis=0).copy()
    mf[1, :, 2] -= 0.25
    mbc2 = np.stack([npy(ter.min_point)] * 2, axis=0)
    hf2 = np.stack([npy(ter.hf)] * 2, axis=0)
    loss, lp, lsdf = terrain_util.motion_frames_hf_sdf_loss(t(mf), pts, t(hf2), t(mbc2), ter.dxdy, km, ret_vis_info=True)
    out.update(loss_frames=mf, loss=loss, loss_points=lp, loss_sdf=lsdf, civ_hf=npy(ter.hf), civ_min_point=npy(ter.min_point), civ_dxdy=npy(ter.dxdy))
    # gradient of the summed loss with respect to the frames (what the motion optimiser descends along), by the reference's autograd
    mf_g = t(mf).requires_grad_(True)
    terrain_util.motion_frames_hf_sdf_loss(mf_g, pts, t(hf2), t(mbc2), ter.dxdy, km).sum().backward()
    out["loss_grad"] = mf_g.grad
    # ... and of points_hf_sdf itself with respect to the points
    p_g = t(p).requires_grad_(True)
    terrain_util.points_hf_sdf(p_g, t(hf), t(mbc), t(dxdy)).sum().backward()
    out["sdf_inverted_grad"] = p_g.grad
    # compute_hf_extra_vals on the first 60 frames (lifted over part of the clip so the jump branch fires)
    clip = frames[0:60].copy()
    clip[20:40, 2] += 1.0
    inds = terrain_util.compute_hf_extra_vals(t(clip), ter, km, pts)
    out.update(extra_frames=clip, extra_mask=ter.hf_mask, extra_maxmin=ter.hf_maxmin, extra_inds_count=np.array([i.shape[0] for i in inds]),
               extra_inds=torch.cat(inds, dim=0))
    # slice_terrain_around_motion
    ter2 = ref_terrain_from_dict(civ["terrain"])
    sl, lf = terrain_util.slice_terrain_around_motion(t(frames[30:90]), ter2, padding=1.0)
    out.update(slice_frames_in=frames[30:90], slice_hf=sl.hf, slice_min_point=sl.min_point, slice_dims=np.array(sl.hf.shape), slice_frames_out=lf)
    save("g13_terrain_geometry", **out)


# --------------------------------------------------------------------------------------------
# Learner side (K12, K14, K19, K21, K22) and the widened termination fixture.  Each stage seeds its own generator, so
# adding a stage never changes the fixtures of the others.
# --------------------------------------------------------------------------------------------
def mlib_from_fixture(km):
    """The 4-clip MotionLib of G3 rebuilt from the frames stored in g3_motion.npz (independent of the main rng stream)."""
    import yaml
    z = np.load(os.path.join(OUT, "g3_motion.npz"))
    civ = load_motion_file_safe(os.path.join(REF, "data/terrains/civilization.pkl"))
    ter = ref_terrain_from_dict(civ["terrain"]).numpy_copy()
    tmp = tempfile.mkdtemp(prefix="parc_golden_")
    entries = []
    for k in range(4):
        pth = os.path.join(tmp, str(z["clip_names"][k]) + ".pkl")
        write_motion_file(pth, z["frames_%d" % k].astype(np.float32), z["contacts_%d" % k].astype(np.float32), int(z["clip_fps"][k]),
                          motion_lib.LoopMode(int(z["clip_loop"][k])).name, ter)
        entries.append({"file": pth, "weight": float(z["clip_weights_in"][k])})
    ypath = os.path.join(tmp, "motions.yaml")
    with open(ypath, "w") as f:
        yaml.safe_dump({"motions": entries}, f)
    return motion_lib.MotionLib(ypath, km, "cpu", contact_info=True)


def gen_done_branches():
    """G8b: RefCharEnv.update_done / compute_done (mgdm_dm_util.py:205-230,392-460) + the motion-end override of
    DeepMimicEnv.update_done (dm_env.py:746-783), every branch tripped on its own.  Base state of every env = the reference
    pose exactly (no failure); each env then gets ONE perturbation.  `case` names the branch, `expect_fail` what the
    perturbation is meant to do (the stored flags are what the reference returned, not this intent)."""
    rng = np.random.default_rng(81)
    km = load_char()
    mlib = mlib_from_fixture(km)
    pose_term = np.array([0.7, 1.0, 0.7, 0.7, 0.7, 0.7, 0.7, 0.7, 1.0, 1.2, 10.0, 1.0, 1.2, 10.0], np.float32)
    feet = [km.get_body_id("right_foot"), km.get_body_id("left_foot")]
    cases = []          # (name, expect_fail(None = n/a), fn(env state dict))
    for b in range(1, 15):
        thr = float(pose_term[b - 1])
        for axis, scale in ((0, 1.05), (1, 1.05), (2, -1.05), (0, 0.95)):
            cases.append(("body%d_%s" % (b, "over" if abs(scale) > 1 else "under"), abs(scale) > 1, ("body", b, axis, scale * thr)))
    for d, over in (([0.61, 0, 0], True), ([0, -0.45, 0.45], True), ([0.59, 0, 0], False), ([0.3, 0.3, 0.3], False)):
        cases.append(("root_pos_" + ("over" if over else "under"), over, ("root_pos", d)))
    for e, over in (([0, 0, 1.33], True), ([1.0, 0.9, 0], True), ([0, 0, 1.29], False), ([0.7, -0.7, 0.7], False)):
        cases.append(("root_rot_" + ("over" if over else "under"), over, ("root_rot", e)))
    # fall branch (only evaluated with contact bodies = feet): body below the termination height AND a contact force on a non-foot body
    cases += [("fall_height_and_force_same_body", True, ("fall", "low", "force_same")),
              ("fall_height_and_force_other_body", True, ("fall", "low", "force_other")),
              ("fall_height_only", False, ("fall", "low", None)),
              ("fall_force_only", False, ("fall", None, "force_same")),
              ("fall_force_below_threshold", False, ("fall", "low", "force_small")),
              ("fall_foot_low_foot_force", False, ("fall", "foot_low", "force_foot")),
              ("fall_height_and_foot_force", False, ("fall", "low", "force_foot")),
              ("fall_negative_force_component", True, ("fall", "low", "force_neg"))]
    cases += [("first_step_body", False, ("first", "body")), ("first_step_root_pos", False, ("first", "root_pos")),
              ("first_step_root_rot", False, ("first", "root_rot")), ("first_step_time_2e-5", True, ("first", "late"))]
    cases += [("timeout", None, ("time", 10.5, False)), ("timeout_exact", None, ("time", 10.0, False)),
              ("just_before_timeout", None, ("time", 9.99, False)), ("timeout_and_fail", True, ("time", 11.0, True))]
    cases += [("motion_end_clamp_past", None, ("mend", 0, 0.2)), ("motion_end_clamp_before", None, ("mend", 0, -0.2)),
              ("motion_end_clamp_short_clip", None, ("mend", 3, 0.05)), ("motion_end_wrap_past", None, ("mend", 2, 0.7)),
              ("motion_end_wrap_far_past", None, ("mend", 2, 3.1)), ("motion_end_clamp_teaser_past", None, ("mend", 1, 1.0))]
    n = len(cases)
    M = mlib.num_motions()
    lens = npy(mlib._motion_lengths)
    motion_ids = rng.integers(0, M, n).astype(np.int64)
    time_buf = np.full(n, 1.0, np.float32)
    mto = (rng.random(n).astype(np.float32) * np.maximum(lens[motion_ids] - 1.3, 0.0)).astype(np.float32)
    mto[lens[motion_ids] < 1.3] = 0.0
    for i, (name, _, spec) in enumerate(cases):
        if spec[0] == "mend":
            motion_ids[i] = spec[1]
            mto[i] = np.float32(lens[spec[1]] + spec[2]) - time_buf[i]
        elif lens[motion_ids[i]] < 1.3:           # the 4-frame clip ends before t = 1 s: keep it for the motion-end cases only
            motion_ids[i] = 0
            mto[i] = np.float32(rng.random() * (lens[0] - 1.3))
        if spec[0] == "time":
            time_buf[i] = spec[1]
            mto[i] = 0.0
            motion_ids[i] = 2                      # WRAP clip: no motion end however late
        if spec[0] == "first":
            time_buf[i] = 2e-5 if spec[1] == "late" else 0.0
    nonfoot = [b for b in range(15) if b not in feet]
    for sweep in range(2):
        motion_ids_t, times = t(motion_ids, torch.int64), t(time_buf + mto)
        rp, rr, rv, rav, jr, dv, cont = mlib.calc_motion_frame(motion_ids_t, times)
        ref_body_pos, _ = km.forward_kinematics(rp, rr, jr)
        if sweep == 0:
            # the fall cases lower a body to just under the termination height; that must stay far below the pose thresholds,
            # so they all use the pose whose lowest non-foot body is the lowest of the whole fixture
            star = int(torch.argmin(ref_body_pos[:, nonfoot, 2].min(dim=-1)[0]))
            for i, (name, _, spec) in enumerate(cases):
                if spec[0] == "fall":
                    motion_ids[i], mto[i], time_buf[i] = motion_ids[star], mto[star], time_buf[star]
    ref_dof_pos = mlib.joint_rot_to_dof(jr)
    # flat terrain 0.25 m under the lowest non-foot body of any env: nobody is below the termination height (0.15) unperturbed
    h0 = float(ref_body_pos[:, nonfoot, 2].min()) - 0.25
    hf = np.full((40, 40), h0, np.float32)
    min_point = np.array([-8.0, -8.0], np.float32)
    dxdy = np.array([0.4, 0.4], np.float32)
    ter = terrain_util.SubTerrain("flat", 40, 40, 0.4, 0.4, -8.0, -8.0, device="cpu")
    ter.hf[:] = t(hf)
    char_root_pos, char_root_rot = rp.clone(), rr.clone()
    body_pos = ref_body_pos.clone()
    contact_forces = torch.zeros(n, 15, 3)
    for i, (name, _, spec) in enumerate(cases):
        kind = spec[0]
        if kind == "body" or (kind == "first" and spec[1] in ("body", "late")) or (kind == "time" and spec[2]):
            b, axis, d = (spec[1], spec[2], spec[3]) if kind == "body" else (4, 0, 1.5)
            body_pos[i, b, axis] += d
        elif kind == "root_pos" or (kind == "first" and spec[1] == "root_pos"):
            d = t(spec[1]) if kind == "root_pos" else t([0.9, 0.0, 0.0])
            char_root_pos[i] += d
            body_pos[i] += d                          # the whole character moves: relative body positions stay equal
        elif kind == "root_rot" or (kind == "first" and spec[1] == "root_rot"):
            e = t([spec[1]]) if kind == "root_rot" else t([[0.0, 0.0, 2.0]])
            char_root_rot[i] = torch_util.quat_mul(torch_util.exp_map_to_quat(e), rr[i:i + 1])[0]
        elif kind == "fall":
            low_b = nonfoot[int(torch.argmin(ref_body_pos[i, nonfoot, 2]))]
            other_b = 2 if low_b != 2 else 1
            if spec[1] == "low":
                body_pos[i, low_b, 2] = h0 + 0.15 - 0.02
            elif spec[1] == "foot_low":
                body_pos[i, feet[0], 2] = h0 + 0.15 - 0.10
            f = {"force_same": (low_b, [0.0, 0.0, 30.0]), "force_other": (other_b, [0.15, 0.0, 0.0]), "force_small": (low_b, [0.09, -0.09, 0.09]),
                 "force_foot": (feet[0], [0.0, 0.0, 400.0]), "force_neg": (other_b, [0.0, -0.2, 0.0]), None: None}[spec[2]]
            if f is not None:
                contact_forces[i, f[0]] = t(f[1])
    margin = float((body_pos[..., nonfoot, 2] - (h0 + 0.15)).abs().min())
    assert margin > 2e-3, margin                   # no body sits on the termination height itself: the flags are not roundoff
    env_offsets = torch.zeros(n, 3)
    gbi = ter.get_grid_index(body_pos[..., 0:2] + env_offsets[:, 0:2].unsqueeze(1))
    term_h = ter.hf[gbi[..., 0], gbi[..., 1]] + 0.15
    outs = {}
    for tag, cb in (("nocontact", torch.zeros(0, dtype=torch.int64)), ("feet", t(feet, torch.int64))):
        outs[tag] = dmu.compute_done(done_buf=torch.zeros(n, dtype=torch.int), time=t(time_buf), ep_len=10.0, root_rot=char_root_rot,
                                     body_pos=body_pos, char_root_pos=char_root_pos, tar_root_rot=rr, tar_body_pos=ref_body_pos,
                                     contact_force=contact_forces, contact_body_ids=cb, termination_heights=term_h, pose_termination=True,
                                     pose_termination_dist=t(pose_term), global_obs=False, enable_early_termination=True, track_root=True,
                                     root_pos_termination_dist=0.6, root_rot_termination_angle=1.309).clone()
    motion_len = mlib.get_motion_length(motion_ids_t)
    motion_end = torch.logical_and(times >= motion_len, mlib.get_motion_loop_mode(motion_ids_t) != motion_lib.LoopMode.WRAP.value)
    finals, frs = {}, {}
    for tag in outs:                                   # fail-rate EMA in env order, then the motion-end override (dm_env.py:746-783)
        fr = torch.ones(M)
        done_any = torch.logical_or(outs[tag] != 0, motion_end)
        for e in torch.nonzero(done_any).flatten().tolist():
            fr[motion_ids[e]] = fr[motion_ids[e]] * (1.0 - 0.01) + (0.01 if outs[tag][e] == 1 else 0.0)
        fin = outs[tag].clone()
        fin[motion_end] = 1
        finals[tag], frs[tag] = fin, fr
    # the fixture must actually exercise what it claims: report every case whose reference flag differs from the intent
    for i, (name, expect, spec) in enumerate(cases):
        tag = "feet" if spec[0] == "fall" else "nocontact"
        if expect is not None:
            assert bool(outs[tag][i] == 1) == expect, (name, int(outs[tag][i]))
    char_dof_pos = ref_dof_pos.clone()
    save("g8b_done_branches", case=np.array([c[0] for c in cases]), expect_fail=np.array([-1 if c[1] is None else int(c[1]) for c in cases]),
         motion_ids=motion_ids, time_buf=time_buf, motion_time_offsets=mto, motion_offsets=np.zeros((M, 1, 2), np.float32),
         env_offsets=env_offsets, hf=hf, min_point=min_point, dxdy=dxdy, rays=gen_rays_points(),
         char_root_pos=char_root_pos, char_root_rot=char_root_rot, char_root_vel=rv, char_root_ang_vel=rav, char_dof_pos=char_dof_pos,
         char_dof_vel=dv, char_rigid_body_pos=body_pos, contact_forces=contact_forces, ref_root_pos=rp, ref_root_rot=rr,
         ref_body_pos=ref_body_pos, pose_termination_dist=pose_term, termination_heights=term_h, motion_end=motion_end,
         done_nocontact=outs["nocontact"], done_feet=outs["feet"], done_final_nocontact=finals["nocontact"], done_final_feet=finals["feet"],
         fail_rates_nocontact=frs["nocontact"], fail_rates_feet=frs["feet"], feet=np.array(feet))


def gen_rays_points():
    return geom_util.get_xy_points_cone(center=torch.zeros(2), dx=0.05, num_neg=2, num_pos=60, num_rays_neg=3, num_rays_pos=3,
                                        angle_between_rays=0.26179938779)


class _Stub:
    pass


def _import_learning():
    import learning.base_agent as base_agent
    import learning.distribution_gaussian_diag as dgd
    import learning.dm_ppo_return_tracker as rt
    import learning.normalizer as normalizer
    import learning.ppo_agent as ppo_agent
    import learning.tracking_error_tracker as tet
    return base_agent, dgd, rt, normalizer, ppo_agent, tet


def gen_ppo_loss():
    """G14: PPOAgent._compute_loss / _compute_actor_loss / _compute_critic_loss (learning/ppo_agent.py:212-330) and
    BaseAgent._compute_action_bound_loss (learning/base_agent.py:456-475), called as unbound methods on a stub agent whose model
    returns the reference's own DistributionGaussianDiag over given means.  Stored per case: loss, every info term, and the
    autograd gradients with respect to mean, logstd and the critic's prediction."""
    base_agent, dgd, _, normalizer, ppo_agent, _ = _import_learning()
    g = torch.Generator().manual_seed(14)
    B, A = 1500, 28
    mean0 = torch.randn(B, A, generator=g) * 0.7            # some |mean| > 1: the action-bound term is active
    logstd0 = torch.randn(A, generator=g) * 0.2 - 1.5
    pred0 = torch.randn(B, generator=g)
    norm_a = mean0 + torch.exp(logstd0) * torch.randn(B, A, generator=g)
    adv = torch.randn(B, generator=g)
    tar = torch.randn(B, generator=g)
    old_noise = 0.3 * torch.randn(B, generator=g)
    masks = {"most": (torch.rand(B, generator=g) < 0.8).float(), "all": torch.ones(B), "one": torch.zeros(B), "none": torch.zeros(B)}
    masks["one"][1234 % B] = 1.0
    a_low, a_high = -np.ones(A, np.float32) * 2.0, np.ones(A, np.float32) * 3.0
    cases = {"default": dict(mask="most"), "all_terms_l1": dict(mask="most", ew=0.01, rw=0.003, l1=True),
             "critic_gate": dict(mask="most", tar_shift=9.0), "mask_all": dict(mask="all"), "mask_one": dict(mask="one", ew=0.01, rw=0.003),
             "mask_none": dict(mask="none"), "no_bound_term": dict(mask="most", bw=0.0)}
    out = dict(mean=mean0, logstd=logstd0, pred=pred0, norm_action=norm_a, adv=adv, tar_val=tar, a_low=a_low, a_high=a_high,
               case_names=np.array(list(cases.keys())), info_names=np.array(["loss", "critic_loss", "actor_loss", "clip_frac", "imp_ratio",
                                                                             "action_bound_loss", "action_entropy", "action_reg_loss"]))
    old_logp = dgd.DistributionGaussianDiag(mean0, logstd0.expand(B, A)).log_prob(norm_a) + old_noise
    out["a_logp"] = old_logp
    for name, c in cases.items():
        mean, logstd, pred = mean0.clone().requires_grad_(True), logstd0.clone().requires_grad_(True), pred0.clone().requires_grad_(True)
        ag = _Stub()
        ag._ppo_clip_ratio, ag._critic_loss_weight = 0.2, 0.5
        ag._action_bound_weight, ag._action_entropy_weight, ag._action_reg_weight = c.get("bw", 10.0), c.get("ew", 0.0), c.get("rw", 0.0)
        ag._critic_loss_type = "L1" if c.get("l1") else "L2"
        ident = _Stub()
        ident.normalize = lambda x: x
        ag._obs_norm = ag._a_norm = ident
        ag._env = _Stub()
        ag._env.get_action_space = lambda: _Box(a_low, a_high)
        model = _Stub()
        # "observations" are row numbers, so that the rows the reference selects by mask reach the right means
        model.eval_actor = lambda o: dgd.DistributionGaussianDiag(mean[o[:, 0].long()], torch.broadcast_to(logstd, (o.shape[0], A)))
        model.eval_critic = lambda o: pred[o[:, 0].long()].unsqueeze(-1)
        ag._model = model
        for fn in ("_compute_critic_loss", "_compute_actor_loss", "_compute_action_bound_loss"):
            setattr(ag, fn, getattr(ppo_agent.PPOAgent, fn).__get__(ag))
        tar_c = tar + c.get("tar_shift", 0.0)
        batch = {"obs": torch.arange(B, dtype=torch.float32).unsqueeze(-1), "action": norm_a, "tar_val": tar_c, "a_logp": old_logp,
                 "adv": adv, "rand_action_mask": masks[c["mask"]]}
        if c["mask"] == "none":
            # no random action in the batch: the reference's mean over an empty selection is NaN, its NaN trap then dumps the
            # batch and exits the program (ppo_agent.py:242-252).  Only the terms are recorded, the trap is not run.
            info = {**ppo_agent.PPOAgent._compute_critic_loss(ag, {"norm_obs": batch["obs"], "tar_val": tar_c}),
                    **ppo_agent.PPOAgent._compute_actor_loss(ag, {"norm_obs": batch["obs"], "norm_action": norm_a, "a_logp": old_logp,
                                                                   "adv": adv, "rand_action_mask": masks["none"]})}
            assert torch.isnan(info["actor_loss"])
            info["loss"] = info["actor_loss"] + 0.5 * info["critic_loss"]
            grads = [torch.zeros_like(mean), torch.zeros_like(logstd), torch.zeros_like(pred)]
        else:
            info = ppo_agent.PPOAgent._compute_loss(ag, batch)
            grads = torch.autograd.grad(info["loss"], [mean, logstd, pred], allow_unused=True)
            grads = [torch.zeros_like(p_) if g_ is None else g_ for g_, p_ in zip(grads, (mean, logstd, pred))]
        vals = [float(info[k]) if k in info else np.nan for k in out["info_names"]]
        out[name + "_info"] = np.array(vals, np.float64)
        out[name + "_mask"] = masks[c["mask"]]
        out[name + "_tar_val"] = tar_c
        out[name + "_params"] = np.array([0.2, ag._action_bound_weight, ag._action_entropy_weight, ag._action_reg_weight, 0.5, 20.0,
                                          1.0 if c.get("l1") else 0.0], np.float64)
        out[name + "_grad_mean"], out[name + "_grad_logstd"], out[name + "_grad_pred"] = grads
    save("g14_ppo_loss", **out)


def gen_normalizer():
    """G15: Normalizer.record / update / normalize / unnormalize (learning/normalizer.py:18-86) over three update rounds, with
    the tracker's non-normalised index set (contacts + heightmap columns) and clip 10."""
    _, _, _, normalizer, _, _ = _import_learning()
    rng = np.random.default_rng(15)
    D = 1312
    nn_idx = torch.arange(766, D)                      # tar_contacts, char_contacts, hf (dm_ppo_agent.py:78-117)
    nz = normalizer.Normalizer((D,), "cpu", clip=10.0, non_norm_indices=nn_idx)
    out = {"non_norm_indices": nn_idx, "clip": np.float64(10.0)}
    scale = (rng.random(D) * 3 + 0.01).astype(np.float32)
    shift = rng.standard_normal(D).astype(np.float32) * 2
    scale[5] = 0.0                                     # a constant column: variance floor (min_std 1e-4)
    for r in range(3):
        for k in range(2):                             # two record calls per round, one [T, N, D] and one [N, D]
            x = (rng.standard_normal((3, 8, D) if k == 0 else (12, D)).astype(np.float32) * scale + shift * (1 + 0.2 * r))
            out["x_%d_%d" % (r, k)] = x
            nz.record(t(x))
            if r == 0 and k == 1:
                out["new_count_r0"], out["new_sum_r0"], out["new_sum_sq_r0"] = np.int64(nz._new_count), nz._new_sum.clone(), nz._new_sum_sq.clone()
        nz.update()
        out["count_%d" % r], out["mean_%d" % r], out["std_%d" % r] = nz._count.clone(), nz._mean.clone(), nz._std.clone()
    q = (rng.standard_normal((16, D)).astype(np.float32) * 4 * np.maximum(scale, 0.05) + shift)
    q[0, :10] = 1e6                                    # clamps at +-clip
    out["query"], out["normalized"] = q, nz.normalize(t(q))
    out["unnormalized"] = nz.unnormalize(t(q))
    # the action normaliser built from the action bounds (base_agent.py:195-203)
    lo, hi = (-rng.random(28) * 2 - 0.1).astype(np.float32), (rng.random(28) * 2 + 0.1).astype(np.float32)
    an = normalizer.Normalizer((28,), "cpu", init_mean=t(0.5 * (hi + lo)), init_std=t(0.5 * (hi - lo)))
    na = rng.standard_normal((32, 28)).astype(np.float32)
    out.update(a_low=lo, a_high=hi, a_mean=an._mean.clone(), a_std=an._std.clone(), norm_action=na, action=an.unnormalize(t(na)),
               action_renormalized=an.normalize(an.unnormalize(t(na))))
    save("g15_normalizer", **out)


def gen_trackers():
    """G16: DMPPOReturnTracker.update (learning/dm_ppo_return_tracker.py:66-99) and TrackingErrorTracker.update
    (learning/tracking_error_tracker.py:73-122) over a seeded 40-step episode stream of 24 envs."""
    _, _, rt, _, _, tet = _import_learning()
    rng = np.random.default_rng(16)
    S, N = 40, 24
    keys = ["total_r", "pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty"]
    rewards = rng.random((S, len(keys), N)).astype(np.float32)
    rewards[:, 6] *= -0.1
    done = rng.choice([0, 0, 0, 0, 0, 0, 0, 0, 1, 3], size=(S, N)).astype(np.int32)
    done[3] = 0
    done[4] = 0                                        # steps without any finished env
    done[7, :] = 1                                     # every env at once
    terr = rng.random((S, N, 7)).astype(np.float32)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        tr = rt.DMPPOReturnTracker(N, "cpu")
    te = tet.TrackingErrorTracker(N, "cpu")
    means, ep_len, episodes, te_means = [], [], [], []
    for s in range(S):
        info = {"rewards": {k: t(rewards[s, i]) for i, k in enumerate(keys)}}
        tr.update(info, t(done[s], torch.int))
        te.update(t(terr[s]), t(done[s], torch.int))
        means.append(np.array([float(tr.get_specific_mean_return(k)) for k in keys], np.float32))
        ep_len.append(float(tr.get_mean_ep_len()))
        episodes.append(tr.get_episodes())
        te_means.append(np.array([float(x) for x in (te.get_mean_root_pos_err(), te.get_mean_root_rot_err(), te.get_mean_body_pos_err(),
                                                     te.get_mean_body_rot_err(), te.get_mean_dof_vel_err(), te.get_mean_root_vel_err(),
                                                     te.get_mean_root_ang_vel_err())], np.float32))
    save("g16_trackers", keys=np.array(keys), rewards=rewards, done=done, tracking_error=terr, mean_returns=np.stack(means),
         mean_ep_len=np.array(ep_len, np.float32), episodes=np.array(episodes, np.int64), eps_per_env=tr.get_eps_per_env().clone(),
         return_bufs=torch.stack([tr._return_bufs[k] for k in keys]), ep_len_buf=tr._ep_len_buf.clone(), te_means=np.stack(te_means),
         te_names=np.array(["root_pos", "root_rot", "body_pos", "body_rot", "dof_vel", "root_vel", "root_ang_vel"]))


def gen_action_head():
    """G17: PPOAgent._decide_action (learning/ppo_agent.py:87-119) in TRAIN and TEST mode with DistributionGaussianDiag
    sample / mode / log_prob / entropy / param_reg (learning/distribution_gaussian_diag.py:39-102), called on a stub agent.  The two
    random draws (torch.normal, torch.bernoulli) are replaced by stored arrays for the duration of the call, so the result is a
    function of the fixture's inputs."""
    base_agent, dgd, _, normalizer, ppo_agent, _ = _import_learning()
    rng = np.random.default_rng(17)
    n, A = 200, 28
    mean = (rng.standard_normal((n, A)) * 0.8).astype(np.float32)
    logstd = (rng.standard_normal(A) * 0.3 - 2.0).astype(np.float32)
    noise = rng.standard_normal((n, A)).astype(np.float32)
    bern = (rng.random((n, 1)) < 0.7).astype(np.float32)
    lo, hi = (-rng.random(A) * 2 - 0.1).astype(np.float32), (rng.random(A) * 2 + 0.1).astype(np.float32)
    ag = _Stub()
    ag._device = "cpu"
    ident = _Stub()
    ident.normalize = lambda x: x
    ag._obs_norm = ident
    ag._a_norm = normalizer.Normalizer((A,), "cpu", init_mean=t(0.5 * (hi + lo)), init_std=t(0.5 * (hi - lo)))
    dist = dgd.DistributionGaussianDiag(t(mean), torch.broadcast_to(t(logstd), (n, A)))
    model = _Stub()
    model.eval_actor = lambda o: dist
    ag._model = model
    ag._get_exp_prob = lambda: 0.7
    out = dict(mean=mean, logstd=logstd, noise=noise, bernoulli=bern[:, 0], a_low=lo, a_high=hi, entropy=dist.entropy(), param_reg=dist.param_reg(),
               logp_of_noise=dist.log_prob(t(mean + np.exp(logstd) * noise)))
    real_normal, real_bern = torch.normal, torch.bernoulli
    torch.normal = lambda m, s: t(noise)
    torch.bernoulli = lambda p: t(bern)
    try:
        for mode in ("TRAIN", "TEST"):
            ag._mode = base_agent.AgentMode[mode]
            a, info = ppo_agent.PPOAgent._decide_action(ag, torch.zeros(n, 4), None)
            out["action_" + mode], out["a_logp_" + mode], out["mask_" + mode] = a, info["a_logp"], info["rand_action_mask"]
    finally:
        torch.normal, torch.bernoulli = real_normal, real_bern
    # action-bound penalty of the same distribution (base_agent.py:456-475)
    ag._env = _Stub()
    ag._env.get_action_space = lambda: _Box(lo, hi)
    out["action_bound_loss"] = base_agent.BaseAgent._compute_action_bound_loss(ag, dist)
    save("g17_action_head", **out)


def gen_recorded_files():
    """G19: the files THIS package wrote on the GPU (tests/golden/recorded/: a clip recorded by record mode and the terrain cache,
    produced by tools/make_recorded_fixture.py) opened by the REFERENCE's readers: MotionLib._load_motions (anim/motion_lib.py:204-380,
    plain pickle.load) and DeepMimicEnv.load_terrain (envs/ig_parkour/dm_env.py:493-507)."""
    rec_dir = os.path.join(HERE, "recorded")
    km = load_char()
    clip = os.path.join(rec_dir, "recorded_clip_dm.pkl")
    ml = motion_lib.MotionLib(clip, km, "cpu", contact_info=True)
    with open(clip, "rb") as f:
        raw = pickle.load(f)                       # (a file of our own making; SubTerrain resolves to the reference's class here)
    assert type(raw["terrain"]).__module__ == "util.terrain_util" and raw["terrain"].__class__ is terrain_util.SubTerrain
    ter = ml._terrains[0]
    import envs.ig_parkour.dm_env as ref_dm_env
    stub = _Stub()
    stub._device = "cpu"
    verts, tris = ref_dm_env.DeepMimicEnv.load_terrain(stub, os.path.join(rec_dir, "terrain.pkl"))
    rp, rr, rv, rav, jr, dv, cont = ml.calc_motion_frame(torch.zeros(3, dtype=torch.int64), t([0.0, 0.21, 10.0]))
    save("g19_recorded_files", num_frames=ml._motion_num_frames, length=ml._motion_lengths, fps=ml._motion_fps, loop_mode=ml._motion_loop_modes,
         frame_root_pos=ml._frame_root_pos, frame_root_rot=ml._frame_root_rot, frame_joint_rot=ml._frame_joint_rot,
         frame_root_vel=ml._frame_root_vel, frame_dof_vel=ml._frame_dof_vel, frame_contacts=ml._frame_contacts,
         obs=raw["obs"], obs_shape_names=np.array(list(raw["obs_shapes"].keys())),
         ter_hf=ter.hf, ter_min_point=ter.min_point, ter_dxdy=ter.dxdy, ter_dims=ter.dims, ter_hf_mask=ter.hf_mask, ter_hf_maxmin=ter.hf_maxmin,
         q_root_pos=rp, q_root_rot=rr, q_joint_rot=jr, q_contacts=cont,
         cache_hf=stub._terrain.hf, cache_min_point=stub._terrain.min_point, cache_dxdy=stub._terrain.dxdy, cache_dims=stub._terrain.dims,
         cache_motion_offsets=stub._dm_motion_offsets, cache_terrains_per_motion=np.int64(stub._terrains_per_motion),
         cache_num_vert_lists=np.int64(len(verts)), cache_verts_0=np.asarray(verts[0][0]), cache_tris_0=np.asarray(tris[0][0]))


STAGES = ["core", "voxel-mesh", "dataset-yaml", "procgen", "terrain-geometry", "done-branches", "ppo-loss", "normalizer", "trackers",
          "action-head", "recorded-files", "motion-opt", "mgdm", "motion-edit", "sim-config", "stage-scripts", "experience-buffer",
          "obs-variants"]


def _icosa_points(radius):
    """12 icosahedron vertices scaled to `radius`: the generator's own sample points for sphere geoms (trimesh, which the reference
    would ask for them, is absent); the loss functions take the point lists as an argument, so these are inputs of the fixture."""
    g = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, g, 0], [1, g, 0], [-1, -g, 0], [1, -g, 0], [0, -1, g], [0, 1, g], [0, -1, -g], [0, 1, -g], [g, 0, -1], [g, 0, 1],
                  [-g, 0, -1], [-g, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return (v * radius).astype(np.float32)


def gen_motion_opt():
    """G20: stage 2's motion optimiser (tools/motion_opt/motion_optimization.py): compute_approx_body_constraints (:34-181),
    motion_terrain_contact_loss (:183-395: the nine terms, the total, and its autograd gradient with respect to the three optimised
    tensors) and motion_contact_optimization (:404-500: 40 Adam iterations, loss per iteration and final frames), on 40 frames of
    the civilization clip over the slice of its terrain around them.  The sample points are an input: capsule / box points from the
    reference's geom_util, sphere geoms from _icosa_points above (get_char_point_samples would need trimesh for them)."""
    import tools.motion_opt.motion_optimization as mo
    rng = np.random.default_rng(20)
    torch.manual_seed(20)
    km = load_char()
    # body sample points: reference sampler on a model copy without spheres, own points for the sphere geoms
    km_ns = load_char()
    sph = {}
    for b in range(km_ns.get_num_joints()):
        for g in km_ns._geoms[b]:
            if g._shape_type == kin_char_model.GeomType.SPHERE:
                sph.setdefault(b, []).append(t(_icosa_points(float(g._dims))) + g._offset.reshape(1, 3))
        km_ns._geoms[b] = [g for g in km_ns._geoms[b] if g._shape_type != kin_char_model.GeomType.SPHERE]
    pts = geom_util.get_char_point_samples(km_ns)
    pts = [torch.cat([p.reshape(-1, 3)] + sph.get(b, []), dim=0) for b, p in enumerate(pts)]
    assert all(p.shape[0] > 0 for p in pts)
    civ = load_motion_file_safe(os.path.join(REF, "data/terrains/civilization.pkl"))
    frames = np.asarray(civ["frames"], np.float32)[150:190].copy()
    contacts = np.asarray(civ["contacts"], np.float32)[150:190].copy()
    lh = km.get_body_id("left_hand")
    contacts[8:15, lh] = 1.0                        # a hand contact so that the sphere-constraint branch runs
    contacts[30, lh] = 1.0                          # ... and a one-frame group
    ter_full = ref_terrain_from_dict(civ["terrain"])
    ter, src = terrain_util.slice_terrain_around_motion(t(frames), ter_full, padding=0.8)
    src = src.clone()
    src[:, 2] -= 0.04                               # slightly into the ground: penetration term active
    con = t(contacts)
    out = dict(pts_count=np.array([p.shape[0] for p in pts]), pts=torch.cat(pts, dim=0), src_frames=src, contacts=con, hf=ter.hf,
               min_point=ter.min_point, dxdy=ter.dxdy)
    # body constraints (the reference calls its point sampler there and drops the result: bypass that one call)
    keep = geom_util.get_char_point_samples
    geom_util.get_char_point_samples = lambda *a, **k: []
    try:
        bc = mo.compute_approx_body_constraints(src[:, 0:3], torch_util.exp_map_to_quat(src[:, 3:6]), km.dof_to_rot(src[:, 6:]), con, km, ter)
    finally:
        geom_util.get_char_point_samples = keep
    rows = [[b, c.start_frame_idx, c.end_frame_idx] + c.constraint_point.tolist() for b, lst in enumerate(bc) for c in lst]
    out["body_constraints"] = np.array(rows, dtype=np.float64)
    print("body constraints:", len(rows))
    w = dict(w_root_pos=1.0, w_root_rot=10.0, w_joint_rot=1.0, w_smoothness=10.0, w_penetration=1000.0, w_contact=1000.0, w_sliding=10.0,
             w_body_constraints=1000.0, w_jerk=1000.0)
    max_jerk = 1000.0
    out["weights"] = np.array([w[k] for k in ("w_root_pos", "w_root_rot", "w_joint_rot", "w_smoothness", "w_penetration", "w_contact", "w_sliding",
                                              "w_body_constraints", "w_jerk")] + [max_jerk])
    # the source-side quantities exactly as motion_contact_optimization derives them (:428-436)
    s_rp, s_rq = src[:, 0:3], torch_util.exp_map_to_quat(src[:, 3:6])
    s_jr = km.dof_to_rot(src[:, 6:34])
    s_bp, s_br = km.forward_kinematics(s_rp, s_rq, s_jr)
    s_bv = s_bp[1:] - s_bp[:-1]
    s_brv = torch_util.quat_diff_angle(s_br[1:], s_br[:-1])
    tgt = src + t(rng.normal(0.0, 0.02, size=src.shape))          # a perturbed target: every term non-zero
    out["tgt_frames"] = tgt

    def evaluate(tag, body_constraints, **override):
        ww = dict(w)
        ww.update(override)
        a, b_, c = tgt[:, 0:3].clone().requires_grad_(True), tgt[:, 3:6].clone().requires_grad_(True), tgt[:, 6:].clone().requires_grad_(True)
        loss, ld = mo.motion_terrain_contact_loss(a, b_, c, s_rp, s_rq, s_jr, s_bv, s_brv, con, ter, pts, km, body_constraints=body_constraints,
                                                  max_jerk=max_jerk, **ww)
        loss.backward()
        out[tag + "_loss"] = loss.detach()
        out[tag + "_terms"] = np.array([float(ld[k]) for k in mo.LossType if k in ld], dtype=np.float64)
        out[tag + "_term_ids"] = np.array([k.value for k in mo.LossType if k in ld])
        out[tag + "_grad"] = torch.cat([a.grad, b_.grad, c.grad], dim=-1)
    evaluate("full", bc)
    evaluate("nocon", None, w_contact=0.0, w_sliding=0.0)
    # 40 iterations of the optimiser itself; the loss of every iteration is recorded through a wrapper around the loss function
    trace = []
    inner = mo.motion_terrain_contact_loss

    def recording(**kw):
        loss, ld = inner(**kw)
        trace.append(loss.item())
        return loss, ld
    mo.motion_terrain_contact_loss = recording
    sys.modules["wandb"].run = None                 # the logger asks the (stubbed, never connected) wandb module for its active run
    try:
        opt = mo.motion_contact_optimization(src_frames=src, contacts=con, body_points=pts, terrain=ter, char_model=km, num_iters=40, step_size=0.001,
                                             body_constraints=bc, max_jerk=max_jerk, exp_name="g20", use_wandb=False, log_file=None, **w)
    finally:
        mo.motion_terrain_contact_loss = inner
    out["opt_frames"] = opt
    out["opt_loss_trace"] = np.array(trace, dtype=np.float64)
    save("g20_motion_opt", **out)


def gen_mgdm():
    """G21: the motion-generator sub-env (envs/ig_parkour/mgdm_env.py MotionGenDeepMimicEnv) driven on CPU the way
    IGParkourEnv drives it (reset -> pre_physics_step -> [physics] -> update_time -> update_misc -> _update_ref_motion -> compute_tar_obs ->
    refresh_obs_hfs -> update_done -> reset(done ids)), 8 envs, 27 steps, plan length 0.3 s.  The trained diffusion model is not in the
    tree (external download): `load_mdm` / `gen_util.gen_mdm_motion` are replaced by a deterministic stand-in whose INPUTS and OUTPUTS are
    stored per call, so a test can replay the outputs and check that the env under test hands the generator the same inputs.  "Physics" is
    the script's own: the character is put near the reference pose with seeded noise (some envs are pushed out of pose / out of bounds /
    too high to reach every termination branch); these states are stored as inputs.  Every torch.rand draw of the sub-env is stored in
    call order (`rand_k`), which is what lets a GPU implementation reproduce the sampling."""
    import random
    import envs.ig_parkour.mgdm_env as me
    import util.motion_util as motion_util
    import envs.base_env as base_env
    rng = np.random.default_rng(21)
    torch.manual_seed(21)
    random.seed(21)
    km = load_char()
    N, F, dt = 8, 16, 1.0 / 30.0
    B, D = km.get_num_joints(), km.get_dof_size()
    out = {}
    calls = []
    real_rand = torch.rand

    class StubGen:
        _num_prev_states, _sequence_fps, _device = 2, 30, "cpu"
        _dx = _dy = 0.4
        _num_x_neg, _num_x_pos, _num_y_neg, _num_y_pos = 2, 5, 3, 3
        _target_type = me.mdm.TargetType.XY_DIR

    def stub_generate(target_world_pos, prev_frames, terrain, mdm_model, char_model, mdm_settings, verbose=True):
        k = len(calls)
        n = prev_frames.root_pos.shape[0]
        last_p, last_q, last_j = prev_frames.root_pos[:, -1], prev_frames.root_rot[:, -1], prev_frames.joint_rot[:, -1]
        d = target_world_pos[:, 0:2] - last_p[:, 0:2]
        d = d / torch.linalg.norm(d, dim=-1, keepdim=True).clamp(min=1e-3)
        tt = torch.arange(F, dtype=torch.float32).reshape(1, F, 1) / 30.0
        rp = last_p.unsqueeze(1).repeat(1, F, 1)
        rp[..., 0:2] += 1.1 * tt * d.unsqueeze(1)
        rp[..., 2] += 0.02 * torch.sin(6.0 * tt[..., 0])
        yaw = 0.4 * tt[..., 0] * (0.5 - (torch.arange(n) % 2).float()).unsqueeze(-1)
        dq = torch.stack([torch.zeros_like(yaw), torch.zeros_like(yaw), torch.sin(yaw / 2), torch.cos(yaw / 2)], dim=-1)
        rq = torch_util.quat_mul(dq, last_q.unsqueeze(1).repeat(1, F, 1))
        g = torch.Generator().manual_seed(1000 + k)
        wob = torch_util.exp_map_to_quat(0.08 * torch.randn((n, 1, B - 1, 3), generator=g) * torch.sin(4.0 * tt).unsqueeze(-1))
        jr = torch_util.quat_mul(last_j.unsqueeze(1).repeat(1, F, 1, 1), wob)
        con = (real_rand((n, F, B), generator=g) > 0.6).float()
        calls.append(True)
        out.update({"gen%d_target" % k: target_world_pos.clone(), "gen%d_prev_root_pos" % k: prev_frames.root_pos.clone(),
                    "gen%d_prev_root_rot" % k: prev_frames.root_rot.clone(), "gen%d_prev_joint_rot" % k: prev_frames.joint_rot.clone(),
                    "gen%d_prev_contacts" % k: prev_frames.contacts.clone(),
                    "gen%d_use_prev_state" % k: mdm_settings.use_prev_state.clone(), "gen%d_prev_state_ind_key" % k: mdm_settings.prev_state_ind_key.clone(),
                    "gen%d_out_root_pos" % k: rp, "gen%d_out_root_rot" % k: rq, "gen%d_out_joint_rot" % k: jr, "gen%d_out_contacts" % k: con})
        return motion_util.MotionFrames(root_pos=rp, root_rot=rq, joint_rot=jr, contacts=con)

    me.load_mdm = lambda path: StubGen()
    me.gen_util.gen_mdm_motion = stub_generate
    rands = []

    def recording_rand(*a, **k):
        r = real_rand(*a, **k)
        rands.append(r.clone())
        return r
    cfg = {"env": {"control_freq": 30, "rand_root_pos_offset_scale": 0.0, "max_obs_h": 3.0, "min_obs_h": -3.0, "demo_mode": False, "target_radius": 0.5,
                   "mgdm": {"plan_length": 0.3, "ddim_stride": 50, "max_replans": 2, "cfg_scale": 0.7, "target_dist_max": 4.0, "target_dist_min": 1.0,
                            "target_dur_max": 0.5, "target_dur_min": 0.2, "target_heading_scale": 0.5, "model_path": "unused",
                            "heightmap": {"horizontal_scale": 0.4, "sq_m_per_env": 0.3, "safety_region": 2.0, "num_segments": 5,
                                          "platform_heights": [0.4, 0.8]}}}}
    out["config_json"] = np.frombuffer(__import__("json").dumps(cfg).encode(), dtype=np.uint8)
    env = me.MotionGenDeepMimicEnv(cfg, N, "cpu", False, km)
    z = torch.zeros
    ref = dict(ref_root_pos=z(N, 3), ref_root_rot=z(N, 4), ref_root_vel=z(N, 3), ref_root_ang_vel=z(N, 3), ref_body_pos=z(N, B, 3),
               ref_joint_rot=z(N, B - 1, 4), ref_dof_pos=z(N, D), ref_dof_vel=z(N, D), ref_contacts=z(N, B))
    ref["ref_root_rot"][:, 3] = 1
    ref["ref_joint_rot"][..., 3] = 1
    ch = dict(char_root_pos=z(N, 3), char_root_rot=z(N, 4), char_root_vel=z(N, 3), char_root_ang_vel=z(N, 3), char_dof_pos=z(N, D), char_dof_vel=z(N, D),
              char_contact_forces=z(N, B, 3), char_rigid_body_pos=z(N, B, 3), char_rigid_body_vel=z(N, B, 3), char_rigid_body_ang_vel=z(N, B, 3))
    ch["char_root_rot"][:, 3] = 1
    env.get_sim_tensor_views(**ref, **ch)
    env_offsets = z(N, 3)
    env_offsets[:, 0] = 0.3 * (torch.arange(N) % 4)
    env_offsets[:, 1] = 0.3 * (torch.arange(N) // 4)
    key_ids = torch.tensor([km.get_body_id(n) for n in ("right_hand", "left_hand", "right_foot", "left_foot")])
    rays = geom_util.get_xy_points_cone(center=torch.zeros(2), dx=0.2, num_neg=1, num_pos=5, num_rays_neg=1, num_rays_pos=1, angle_between_rays=0.3)
    bufs = dict(reward_buf=z(N), done_buf=z(N, dtype=torch.int), time_buf=z(N), timestep_buf=z(N, dtype=torch.int),
                actors_need_reset=z(N, 1, dtype=torch.bool), target_xy=z(N, 2), next_target_xy_time=z(N), env_offsets=env_offsets,
                key_body_ids=key_ids, ray_xy_points=rays, ray_hfs=z(N, rays.shape[0]))
    env.get_data_buffer_views(**bufs)
    out.update(env_offsets=env_offsets, key_body_ids=key_ids, ray_xy_points=rays)
    tmpd = tempfile.mkdtemp(prefix="parc_golden_mgdm_")
    verts, tris, min_pt = env.build_terrain(cfg["env"], os.path.join(tmpd, "mgdm_terrain.pkl"))
    out.update(terrain_hf=env._terrain.hf, terrain_min_point=env._terrain.min_point, terrain_dxdy=env._terrain.dxdy,
               spawn=np.array([env._spawn_min_x, env._spawn_max_x, env._spawn_min_y, env._spawn_max_y, env._oob_region]),
               mesh_counts=np.array([verts.shape[0], tris.shape[0]]), local_grid=env._mgdm_local_xy_points)
    tar_steps = torch.tensor([1, 2, 5])
    out["tar_obs_steps"] = tar_steps
    done_args = dict(termination_height=0.15, episode_length=0.8, contact_body_ids=torch.tensor([km.get_body_id("right_foot"), km.get_body_id("left_foot")]),
                     pose_termination=True, pose_termination_dist=torch.tensor([0.7, 1.0, 0.7, 0.7, 0.7, 0.7, 0.7, 0.7, 1.0, 1.2, 10.0, 1.0, 1.2, 10.0]),
                     global_obs=False, enable_early_termination=True, track_root=True, root_pos_termination_dist=0.6,
                     root_rot_termination_angle=1.309)
    out["pose_termination_dist"] = done_args["pose_termination_dist"]

    def snap(tag):
        out.update({tag + "_" + k: v.clone() for k, v in ref.items()})
        out.update({tag + "_" + k: ch[k].clone() for k in ("char_root_pos", "char_root_rot", "char_root_vel", "char_root_ang_vel", "char_dof_pos", "char_dof_vel")})
        out.update({tag + "_done": bufs["done_buf"].clone(), tag + "_time": bufs["time_buf"].clone(), tag + "_timestep": bufs["timestep_buf"].clone(),
                    tag + "_target_xy": bufs["target_xy"].clone(), tag + "_next_target_time": bufs["next_target_xy_time"].clone(),
                    tag + "_replan_buf": env._replan_buf.clone(), tag + "_replan_counter": env._replan_counter.clone(),
                    tag + "_plan_time": env._mgdm_time_buf.clone(), tag + "_replan_flag": np.array([int(env._replan_flag)]),
                    tag + "_need_reset": bufs["actors_need_reset"].clone(), tag + "_num_rand": np.array([len(rands)]),
                    tag + "_num_gen": np.array([len(calls)]),
                    tag + "_agent_hist_root_pos": env._agent_state_hist.root_pos.clone(), tag + "_ref_hist_root_pos": env._ref_state_hist.root_pos.clone(),
                    tag + "_agent_hist_joint_rot": env._agent_state_hist.joint_rot.clone()})
    me.torch.rand = recording_rand
    try:
        env.replan()                          # IGParkourEnv._build_data_buffers :797-798
        snap("init")
        bufs["actors_need_reset"][:] = False
        env.reset(torch.arange(N))            # the agent's first reset(None): every env, no replan pending -> soft reset
        snap("reset0")
        bufs["actors_need_reset"][:] = False
        K = 27
        for k in range(K):
            env.pre_physics_step()
            # the script's "physics": the character lands near the reference pose one step ahead
            nxt = env._motion_lib.calc_motion_frame(env._motion_ids, env._mgdm_time_buf.expand(N) + dt)
            ch["char_root_pos"][:] = nxt[0] + t(rng.normal(0, 0.02, (N, 3)))
            ch["char_root_rot"][:] = torch_util.quat_mul(torch_util.exp_map_to_quat(t(rng.normal(0, 0.03, (N, 3)))), nxt[1])
            ch["char_root_vel"][:] = nxt[2]
            ch["char_root_ang_vel"][:] = nxt[3]
            ch["char_dof_pos"][:] = km.rot_to_dof(nxt[4]) + t(rng.normal(0, 0.03, (N, D)))
            ch["char_dof_vel"][:] = nxt[5]
            if k == 4:
                ch["char_dof_pos"][1, 0:3] += 1.5           # out of pose
            if k == 6:
                ch["char_root_pos"][2, 2] = 3.4             # too high
            if k == 12:
                ch["char_root_pos"][3, 0] = env._terrain.min_point[0] + env._oob_region * 0.5 - env_offsets[3, 0]      # out of bounds
            if k == 15:
                ch["char_root_pos"][4, 1] += 0.9            # root position
            bp, _ = km.forward_kinematics(ch["char_root_pos"], ch["char_root_rot"], km.dof_to_rot(ch["char_dof_pos"]))
            ch["char_rigid_body_pos"][:] = bp
            ch["char_contact_forces"][:] = t((rng.random((N, B, 1)) > 0.8) * rng.normal(0, 30.0, (N, B, 3)))
            tag = "s%d" % k
            out.update({tag + "_in_" + kk: ch[kk].clone() for kk in ("char_root_pos", "char_root_rot", "char_root_vel", "char_root_ang_vel", "char_dof_pos",
                                                                     "char_dof_vel", "char_rigid_body_pos", "char_contact_forces")})
            bufs["timestep_buf"] += 1
            bufs["time_buf"][:] = dt * bufs["timestep_buf"]
            env.update_time(dt)
            env.update_misc()
            env._update_ref_motion()
            tar = env.compute_tar_obs(tar_steps)
            out.update({tag + "_tar_root_pos": tar[0], tag + "_tar_root_rot": tar[1], tag + "_tar_joint_rot": tar[2], tag + "_tar_key_pos": tar[3],
                        tag + "_tar_contacts": tar[4]})
            gpos = ch["char_root_pos"] + env_offsets
            env.refresh_obs_hfs(gpos, torch_util.calc_heading(ch["char_root_rot"]))
            out.update({tag + "_ray_hfs": bufs["ray_hfs"].clone(), tag + "_mgdm_hfs": env._mgdm_hfs.clone(), tag + "_floor": env._mgdm_floor_heights.clone()})
            env.update_done(**done_args)
            out.update({tag + "_done_after_update": bufs["done_buf"].clone(), tag + "_replan_buf_after_update": env._replan_buf.clone(),
                        tag + "_replan_flag_after_update": np.array([int(env._replan_flag)])})
            ids = (bufs["done_buf"] != base_env.DoneFlags.NULL.value).nonzero().flatten()
            env.reset(ids)
            snap(tag)
            bufs["actors_need_reset"][:] = False
        out["num_steps"] = np.array([K])
    finally:
        me.torch.rand = real_rand
    for i, r in enumerate(rands):
        out["rand_%d" % i] = r
    out["num_rand"] = np.array([len(rands)])
    out["num_gen"] = np.array([len(calls)])
    save("g21_mgdm", **out)


def gen_motion_edit():
    """G22: the motion-file helpers between stage 2 and the tracker (zmotion_editing_tools/motion_edit_lib.py): flip_motion_about_XZ_plane
    (:514-610) with SubTerrain.flip_by_XZ_axis (util/terrain_util.py:169-178) - the mirrored copy parc_2_kin_gen.py:493-511 adds for
    every optimised clip - and remove_hesitation_frames (:1242-1319), on 60 frames of the civilization clip with two stretches of
    near-repeated poses spliced in; MotionData / save_motion_data / load_motion_file round trip of the written file."""
    import zmotion_editing_tools.motion_edit_lib as medit
    rng = np.random.default_rng(22)
    km = load_char()
    civ = load_motion_file_safe(os.path.join(REF, "data/terrains/civilization.pkl"))
    frames = np.asarray(civ["frames"], np.float32)[40:100].copy()
    contacts = np.asarray(civ["contacts"], np.float32)[40:100].copy()
    out = dict(frames=frames, contacts=contacts)
    flipped, fcon = medit.flip_motion_about_XZ_plane(t(frames), km, contact_frames=t(contacts))
    out.update(flip_frames=flipped, flip_contacts=fcon)
    ter = ref_terrain_from_dict(civ["terrain"])
    sl, _ = terrain_util.slice_terrain_around_motion(t(frames), ter, padding=0.8)
    out.update(ter_hf=sl.hf.clone(), ter_min_point=sl.min_point.clone(), ter_dxdy=sl.dxdy.clone(), ter_mask=sl.hf_mask.clone(),
               ter_maxmin=sl.hf_maxmin.clone())
    sl.flip_by_XZ_axis()
    out.update(ter_flip_hf=sl.hf, ter_flip_min_point=sl.min_point, ter_flip_mask=sl.hf_mask, ter_flip_maxmin=sl.hf_maxmin)
    # hesitation: frames 20..27 hover around frame 19 and 45..47 around frame 44 (a run shorter than the minimum length: kept)
    hes = frames.copy()
    hcon = contacts.copy()
    for k in range(20, 28):
        hes[k] = hes[19] + rng.normal(0, 0.004, size=hes.shape[1]).astype(np.float32)
    for k in range(45, 48):
        hes[k] = hes[44] + rng.normal(0, 0.004, size=hes.shape[1]).astype(np.float32)
    nf, nc = medit.remove_hesitation_frames(t(hes), t(hcon), km)
    out.update(hes_frames=hes, hes_contacts=hcon, hes_out_frames=nf, hes_out_contacts=nc)
    nf2, _ = medit.remove_hesitation_frames(t(hes), t(hcon), km, hesitation_val=0.3, hesitation_min_seq_len=2)
    out["hes_out2_count"] = np.array([nf2.shape[0]])
    # file round trip
    tmp = tempfile.mkdtemp(prefix="parc_golden_medit_")
    path = os.path.join(tmp, "clip.pkl")
    ter2 = ref_terrain_from_dict(civ["terrain"])
    medit.save_motion_data(path, t(frames), t(contacts), ter2, 30, "CLAMP", loss=1.5, min_point_offset=t([0.25, -0.5]))
    md = medit.load_motion_file(path)
    out.update(rt_fps=np.array([md.get_fps()]), rt_frames=md.get_frames(), rt_contacts=md.get_contacts(), rt_hf=md.get_terrain().hf,
               rt_keys=np.array(sorted(md._data.keys())))
    save("g22_motion_edit", **out)


class _Rec:
    """Recording stand-in for an Isaac Gym parameter object (gymapi.SimParams / AssetOptions / TriangleMeshParams ...): attribute
    writes are kept, attribute reads create nested recorders - so running the reference's own set-up code on it yields exactly the
    values the reference hands to Isaac Gym."""

    def __init__(self):
        object.__setattr__(self, "_d", {})

    def __getattr__(self, k):
        d = object.__getattribute__(self, "_d")
        if k not in d:
            d[k] = _Rec()
        return d[k]

    def __setattr__(self, k, v):
        object.__getattribute__(self, "_d")[k] = v

    def tree(self):
        out = {}
        for k, v in object.__getattribute__(self, "_d").items():
            out[k] = v.tree() if isinstance(v, _Rec) else (v if isinstance(v, (int, float, str, bool)) or v is None else repr(v))
        return out


def gen_sim_config():
    """G23: the simulator's CONFIGURATION - the only part of row a1 that can be pinned (Isaac Gym's arithmetic is an absent binary).
    (1) data/assets/humanoid.xml parsed with ElementTree: every body / joint / geom / motor attribute, class defaults resolved;
    (2) what the reference's own set-up code hands to Isaac Gym, recorded by running it on recording stand-ins for the gymapi objects:
        IGEnv._parse_sim_params (envs/ig_env.py:131-164) on PARC/tracker_config/dm_env_default.yaml, IGCharEnv._build_char_asset_options
        and _build_character (envs/ig_char_env.py:105-146), util/ig_util.add_trimesh_to_gym (:6-22);
        gymutil.parse_sim_config is Isaac Gym's helper (absent): its stand-in copies the YAML's `sim:` keys onto the parameter object,
        which is its documented behaviour (scalars onto SimParams, the `physx:` block onto SimParams.physx);
    (3) the two default YAML trees (env + agent) as JSON."""
    import json
    import xml.etree.ElementTree as ET
    import yaml
    import envs.ig_env as ig_env
    import util.ig_util as ig_util
    root = ET.parse(os.path.join(REF, CHAR_FILE)).getroot()
    dflt = {"motor": dict(root.find("default").find("motor").attrib)}
    for d in root.find("default").findall("default"):
        dflt[d.attrib["class"]] = {c.tag: dict(c.attrib) for c in d}
    nums = lambda s_: [float(x) for x in s_.split()]
    bodies = []

    def walk(el, parent, cls):
        cls = el.attrib.get("childclass", cls)
        b = {"name": el.attrib["name"], "parent": parent, "pos": nums(el.attrib["pos"]), "class": cls, "freejoint": el.find("freejoint") is not None,
             "joints": [], "geoms": []}
        for j in el.findall("joint"):
            a = dict(dflt[cls]["joint"])
            a.update(j.attrib)
            b["joints"].append({"name": a["name"], "type": a["type"], "axis": nums(a["axis"]), "range_deg": nums(a["range"]),
                                "stiffness": float(a["stiffness"]), "damping": float(a["damping"]), "armature": float(a["armature"]),
                                "limited": a["limited"]})
        for g in el.findall("geom"):
            a = dict(dflt[cls]["geom"])
            a.update(g.attrib)
            e = {"name": a["name"], "type": a["type"], "size": nums(a["size"]), "density": float(a["density"]), "friction": nums(a["friction"]),
                 "condim": int(a["condim"])}
            if "fromto" in a:
                e["fromto"] = nums(a["fromto"])
            if "pos" in a:
                e["pos"] = nums(a["pos"])
            b["geoms"].append(e)
        bodies.append(b)
        for c in el.findall("body"):
            walk(c, b["name"], cls)
    for top in root.find("worldbody").findall("body"):
        walk(top, None, None)
    motors = [{"name": m.attrib["name"], "joint": m.attrib["joint"], "gear": float(m.attrib["gear"])} for m in root.find("actuator").findall("motor")]
    with open(os.path.join(REF, "PARC/tracker_config/dm_env_default.yaml")) as f:
        env_yaml = yaml.safe_load(f)
    with open(os.path.join(REF, "PARC/tracker_config/dm_agent_default.yaml")) as f:
        agent_yaml = yaml.safe_load(f)
    # ---- what the reference's set-up code hands to Isaac Gym
    gymapi, gymutil = sys.modules["isaacgym.gymapi"], sys.modules["isaacgym.gymutil"]
    saved = {k: getattr(gymapi, k, None) for k in ("SimParams", "AssetOptions", "TriangleMeshParams", "Transform", "Vec3", "Quat", "UP_AXIS_Z",
                                                    "DOF_MODE_POS", "DOF_MODE_VEL", "DOF_MODE_EFFORT", "DOF_MODE_NONE")}
    gymapi.SimParams = gymapi.AssetOptions = gymapi.TriangleMeshParams = gymapi.Transform = _Rec
    gymapi.Vec3 = lambda *a: list(a)
    gymapi.Quat = lambda *a: list(a)
    gymapi.UP_AXIS_Z = "UP_AXIS_Z"
    for m_ in ("DOF_MODE_POS", "DOF_MODE_VEL", "DOF_MODE_EFFORT", "DOF_MODE_NONE"):
        setattr(gymapi, m_, m_)

    def parse_sim_config(sim, params):
        for k, v in sim.items():
            if k == "physx":
                for kk, vv in v.items():
                    setattr(params.physx, kk, vv)
            else:
                setattr(params, k, v)
    gymutil.parse_sim_config = parse_sim_config
    try:
        env_cfg = env_yaml["env"]
        sim_freq, control_freq = env_cfg.get("sim_freq", 60), env_cfg.get("control_freq", 10)
        stub = types.SimpleNamespace(_device="cuda:0")
        sim_params = ig_env.IGEnv._parse_sim_params(stub, env_yaml, 1.0 / sim_freq).tree()
        calls = []

        class Gym:
            def create_actor(self, env_ptr, asset, pose, name, col_group, col_filter, seg_id):
                calls.append({"create_actor": {"name": name, "collision_group": col_group, "collision_filter": col_filter,
                                               "segmentation_id": seg_id, "pose": pose.tree()}})
                return "actor"

            def get_asset_dof_properties(self, asset):
                return {"driveMode": None, "stiffness": "from_asset", "damping": "from_asset"}

            def set_actor_dof_properties(self, env_ptr, handle, prop):
                calls.append({"set_actor_dof_properties": dict(prop)})

            def add_triangle_mesh(self, sim, verts, tris, params):
                calls.append({"add_triangle_mesh": params.tree()})

            def enable_actor_dof_force_sensors(self, *a):
                calls.append({"enable_actor_dof_force_sensors": True})
        cstub = types.SimpleNamespace(_char_control_mode=ig_char_env.ControlMode[env_cfg["control_mode"]], _gym=Gym(), _char_asset="asset",
                                      _char_handles=[], _enable_dof_force_sensors=lambda: False)
        cstub._control_mode_to_drive_mode = lambda m: ig_char_env.IGCharEnv._control_mode_to_drive_mode(cstub, m)
        asset_options = ig_char_env.IGCharEnv._build_char_asset_options(cstub, env_yaml).tree()
        ig_char_env.IGCharEnv._build_character(cstub, 7, "env_ptr", env_yaml)          # env id 7: the collision group must come back as 7
        ig_util.add_trimesh_to_gym(np.zeros((3, 3), np.float32), np.zeros((1, 3), np.int32), "sim", cstub._gym)
    finally:
        for k, v in saved.items():
            if v is None:
                if hasattr(gymapi, k):
                    delattr(gymapi, k)
            else:
                setattr(gymapi, k, v)
    out = {"mjcf": {"model": root.attrib.get("model"), "defaults": dflt, "bodies": bodies, "motors": motors},
           "isaac_gym": {"sim_params": sim_params, "sim_steps_per_control_step": int(sim_freq / control_freq), "sim_dt": 1.0 / sim_freq,
                         "control_dt": 1.0 / control_freq, "asset_options": asset_options, "calls": calls},
           "env_yaml": env_yaml, "agent_yaml": agent_yaml}
    path = os.path.join(OUT, "g23_sim_config.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)



def gen_stage_scripts():
    """G24: the reference's stage scripts parc_3_tracker.train_tracker (parc_3_tracker.py:8-78) and parc_4_phys_record.record_motions
    (parc_4_phys_record.py:8-65) run UNCHANGED on top of this package's module aliases (tests/golden/stage_scripts_child.py, a fresh
    process per script like `python parc_3_tracker.py`), on a small config derived from PARC/tracker_default.yaml /
    phys_record_default.yaml / create_dataset_config.yaml: same keys, the two default tracker YAMLs as env / agent config, a 4-file
    dataset of real motion, 32 envs.  env_builder.build_env / agent_builder.build_agent are recorders.  Stored: the files the scripts
    write (dm_env.yaml, agent_config.yaml, train_args.txt, record_env.yaml, record_args.txt, the dataset YAML of create_dataset), the
    argv handed to run.main and every call the reference's run.run makes into the package, with the scratch directory spelled <TMP>.
    Three runs: tracker with in_model_file + create_dataset_config, tracker without either, record."""
    import json
    import subprocess
    import yaml
    sys.path.insert(0, HERE)
    import dataset_tree
    from parc_amd.util import terrain_util as our_terrain_util
    tmp = tempfile.mkdtemp(prefix="parc_golden_stage_")

    def make_terrain(hf, min_point, dxdy):
        return our_terrain_util.SubTerrain.from_arrays(hf, min_point, dxdy, device="cpu").numpy_copy()
    # files in the reference's format (class path util.terrain_util.SubTerrain), written by the package's own writer
    folders = dataset_tree.build_clip_tree(os.path.join(tmp, "tree"), make_terrain, dump=our_terrain_util.dump_reference_pickle)
    ds_cfg = {"save_path": os.path.join(tmp, "dataset", "motions.yaml"), "folder_paths": folders, "char_filepath": "data/assets/humanoid.xml",
              "compute_preprocessing_data": False, "cut_some_classes_in_half": True, "motion_classes_to_cut_in_half": ["running"],
              "max_terrain_dim_x": 120, "max_terrain_dim_y": 120}
    os.makedirs(os.path.join(tmp, "dataset"))
    ds_cfg_path = os.path.join(tmp, "create_dataset_config.yaml")
    with open(ds_cfg_path, "w") as f:
        yaml.safe_dump(ds_cfg, f)
    base = {"env_config": "PARC/tracker_config/dm_env_default.yaml", "agent_config": "PARC/tracker_config/dm_agent_default.yaml",
            "num_envs": 32, "max_samples": 2048, "device": "cuda:0", "dataset_file": ds_cfg["save_path"]}
    runs = {
        "tracker_resume": ("tracker", dict(base, output_dir=os.path.join(tmp, "tracker_resume") + "/", in_model_file=os.path.join(tmp, "tracker_fresh", "model.pt"),
                                           create_dataset_config=ds_cfg_path)),
        "tracker_fresh": ("tracker", dict(base, output_dir=os.path.join(tmp, "tracker_fresh") + "/")),
        "record": ("record", {"output_dir": os.path.join(tmp, "record") + "/", "device": "cuda:0", "create_dataset_config": ds_cfg_path,
                              "model_file": os.path.join(tmp, "tracker_fresh", "model.pt"), "agent_file": os.path.join(tmp, "tracker_fresh", "agent_config.yaml"),
                              "env_file": os.path.join(tmp, "tracker_fresh", "dm_env.yaml")}),
    }
    os.makedirs(os.path.join(tmp, "record"))          # parc_4_phys_record writes into output_dir without creating it
    sub = lambda x: json.loads(json.dumps(x).replace(tmp, "<TMP>"))
    out = {"create_dataset_config": sub(ds_cfg), "runs": {}}
    for name in ("tracker_resume", "tracker_fresh", "record"):
        which, cfg = runs[name]
        cfg_path = os.path.join(tmp, name + "_config.yaml")
        with open(cfg_path, "w") as f:
            yaml.safe_dump(cfg, f)
        env = dict(os.environ, PYTHONHASHSEED="0")
        r = subprocess.run([sys.executable, os.path.join(HERE, "stage_scripts_child.py"), which, cfg_path, REF], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("G24JSON ")][-1]
        res = json.loads(line[len("G24JSON "):])
        assert res["modules_loaded_from_the_reference"] == ["parc_3_tracker" if which == "tracker" else "parc_4_phys_record", "run"], res["modules_loaded_from_the_reference"]
        files = {}
        od = cfg["output_dir"]
        for fn in sorted(os.listdir(od)):
            if os.path.isfile(os.path.join(od, fn)):
                files[fn] = open(os.path.join(od, fn)).read().replace(tmp, "<TMP>")
        out["runs"][name] = {"script": {"tracker": "parc_3_tracker.train_tracker", "record": "parc_4_phys_record.record_motions"}[which],
                             "config": sub(cfg), "files_written_to_output_dir": files, "argv": sub(res["argv"]), "calls": sub(res["calls"])}
    out["dataset_yaml"] = open(ds_cfg["save_path"]).read().replace(tmp, "<TMP>")
    # random master port and time-derived seed differ per run: keep their kind, not their value
    for rn in out["runs"].values():
        for c in rn["calls"]:
            if c["call"] == "mp_util.init":
                assert 6000 <= c["args"][3] < 7000
                c["args"][3] = "<port in [6000, 7000)>"
            if c["call"] == "util.set_rand_seed":
                assert c["args"][0]["numpy"] == "uint64"
                c["args"][0] = {"numpy": "uint64", "int": "<time-derived>"}
    path = os.path.join(OUT, "g24_stage_scripts.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)



def gen_experience_buffer():
    """G25: the experience buffer as the reference's agent builds and samples it.  (1) DMPPOAgent._build_exp_buffer -> PPOAgent ->
    BaseAgent (learning/dm_ppo_agent.py:281-313, ppo_agent.py:62-80, base_agent.py:224-253) run on an agent object that skipped
    __init__ (env = obs / action spaces of the tracker, `ig_parkour` => the two replan buffers): names in insertion order, dtypes,
    shapes.  (2) ExperienceBuffer (learning/experience_buffer.py:3-115): record / inc / sample over a partly filled and a full
    buffer, with minibatch sizes that do NOT divide the buffer so that _sample_rand_idx wraps in mid-call; every torch.randperm the
    class draws is recorded, and so is the index list of every sample() call."""
    import json
    import learning.dm_ppo_agent as dm_ppo_agent
    import learning.experience_buffer as experience_buffer
    T_, N_ = 5, 3
    ag = object.__new__(dm_ppo_agent.DMPPOAgent)
    torch.nn.Module.__init__(ag)
    ag._device = "cpu"
    ag._steps_per_iter = T_
    ag._is_terrain_runner = True
    env = _Stub()
    env.get_obs_space = lambda: _Box(-np.ones(1312, np.float32), np.ones(1312, np.float32))
    env.get_action_space = lambda: _Box(-np.ones(28, np.float32), np.ones(28, np.float32))
    env.get_num_envs = lambda: N_
    ag._env = env
    dm_ppo_agent.DMPPOAgent._build_exp_buffer(ag, {})
    table = [[k, str(v.dtype).replace("torch.", ""), list(v.shape)] for k, v in ag._exp_buffer._buffers.items()]
    flat = [[k, list(v.shape)] for k, v in ag._exp_buffer._flat_buffers.items()]
    # ---- sampler walk
    perms = []
    real = torch.randperm

    def randperm(*a, **k):
        p_ = real(*a, **k)
        perms.append(p_.tolist())
        return p_
    experience_buffer.torch.randperm = randperm
    try:
        torch.manual_seed(25)
        buf = experience_buffer.ExperienceBuffer(buffer_length=T_, batch_size=N_, device="cpu")
        buf.add_buffer("x", torch.zeros([T_, N_, 2]))
        walk = []
        for t_ in range(2):                                     # two of five rows written: 6 of 15 samples valid
            buf.record("x", torch.full([N_, 2], float(t_ + 1)))
            buf.inc()
        for n_ in (4, 4):
            idx = buf._sample_rand_idx(n_)
            walk.append({"phase": "partly filled", "n": n_, "sample_count": buf.get_sample_count(), "idx": idx.tolist(), "head_after": buf._sample_buf_head})
        for t_ in range(2, 7):                                  # fill it and go round once more (head wraps: 7 % 5 = 2)
            buf.record("x", torch.full([N_, 2], float(t_ + 1)))
            buf.inc()
        head_row, total = buf._buffer_head, buf.get_total_samples()
        buf.reset()                                             # _init_iter of every training iteration (ppo_agent.py:82-85)
        for n_ in (4, 4, 4, 4, 4, 7, 15, 3):                    # 4th call wraps in mid-call (12 + 4 > 15); 15 = the whole buffer at once
            idx = buf._sample_rand_idx(n_)
            walk.append({"phase": "full", "n": n_, "sample_count": buf.get_sample_count(), "idx": idx.tolist(), "head_after": buf._sample_buf_head})
        out_s = buf.sample(4)
        sample_keys = list(out_s.keys())
        x_rows = buf.get_data("x")[:, 0, 0].tolist()
    finally:
        experience_buffer.torch.randperm = real
    out = {"T": T_, "N": N_, "buffers": table, "flat_views": flat, "randperm_draws": perms, "index_walk": walk,
           "buffer_head_after_7_incs": head_row, "total_samples_after_7_incs": total, "rows_after_7_records": x_rows, "sample_keys": sample_keys}
    path = os.path.join(OUT, "g25_experience_buffer.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)



def gen_obs_variants():
    """G26: the non-default variants of IGParkourEnv._compute_obs / _update_reward (ig_parkour_env.py:1054-1244,1275-1404) - the
    reference's OWN methods, called unbound on an object that carries the G6 state under the reference's attribute names (its
    constructor needs Isaac Gym; these two methods do not) - and the two env configurations the reference ships
    (data/envs/ig_parkour_env.yaml: the motion-generator env with has_target_xy_obs; data/terrains/dm_env_civilization.yaml), parsed,
    with the observation segment table each one produces."""
    import json
    import yaml
    import envs.ig_parkour.ig_parkour_env as ipe
    z = np.load(os.path.join(HERE, "g6_step.npz"))
    km = load_char()
    n = z["obs"].shape[0]
    rng = np.random.default_rng(26)
    target_xy = t(z["char_root_pos"][:, 0:2] + rng.standard_normal((n, 2)).astype(np.float32) * 3.0)
    target_xy[0] = t(z["char_root_pos"][0, 0:2]) + 0.3                 # inside the target radius: task reward saturates
    target_xy[1] = t(z["char_root_pos"][1, 0:2])                       # on the target: direction undefined -> zeros
    plan_clock = torch.tensor(0.2333, dtype=torch.float32)

    class DM:
        def compute_tar_obs(self, steps, env_ids):
            assert env_ids is None
            return t(z["tar_root_pos"]), t(z["tar_root_rot"]), t(z["tar_joint_rot"]), t(z["tar_key_pos"]), t(z["tar_contacts"])

        def get_mgdm_time_buf(self):
            return plan_clock.reshape(1)

    def make(cfg):
        e = object.__new__(ipe.IGParkourEnv)
        e._device, e._num_envs, e._visualize, e._report_tracking_error = "cpu", n, False, False
        e._kin_char_model = km
        for k in ("char_root_pos", "char_root_rot", "char_root_vel", "char_root_ang_vel", "char_dof_pos", "char_dof_vel", "char_rigid_body_pos"):
            setattr(e, "_" + k, t(z[k]).clone())
        e._char_contact_forces = t(z["contact_forces"])
        for k in ("ref_root_pos", "ref_root_rot", "ref_root_vel", "ref_root_ang_vel", "ref_joint_rot", "ref_dof_vel", "ref_contacts", "ref_body_pos"):
            setattr(e, "_" + k, t(z[k]).clone())
        e._ray_hfs = t(z["ray_hfs"])
        e._target_xy = target_xy.clone()
        e._key_body_ids = t(z["key_body_ids"], torch.int64)
        e._tar_obs_steps = t(z["tar_obs_steps"], torch.int)
        e._joint_err_w, e._dof_err_w = t(z["joint_err_w"]), t(z["dof_err_w"])
        w = np.array([cfg["pose_w"], cfg["vel_w"], cfg["root_pos_w"], cfg["root_vel_w"], cfg["key_pos_w"]])
        tot = w.sum()
        e._pose_w, e._vel_w, e._root_pos_w, e._root_vel_w, e._key_pos_w = [float(v / tot) for v in w]
        e._contact_weights = t(cfg["contact_weights"])
        e._global_obs, e._use_heightmap = bool(cfg["global_obs"]), True
        e._global_root_height_obs, e._enable_tar_obs = bool(cfg["global_root_height_obs"]), bool(cfg.get("enable_tar_obs", True))
        e._use_contact_info, e._has_target_xy_obs = bool(cfg["use_contact_info"]), bool(cfg["has_target_xy_obs"])
        e._track_root, e._track_root_h = bool(cfg["track_root"]), bool(cfg["track_root_h"])
        e._task1_w, e._task2_w, e._target_radius = cfg["task1_w"], cfg["task2_w"], cfg["target_radius"]
        e._rel_deepmimic_w, e._rel_task_w = cfg["rel_deepmimic_w"], cfg["rel_task_w"]
        mg = bool(cfg.get("_mgdm", False))
        e._num_dm_envs, e._num_mgdm_envs = (0, n) if mg else (n, 0)
        e._enable_replan_timer_obs = bool(cfg.get("enable_replan_timer_obs", False)) and mg
        e._dm_env = e._mgdm_env = DM()
        e._reward_buf = torch.zeros(n)
        e._info = dict()
        return e

    base = dict(pose_w=0.5, vel_w=0.1, root_pos_w=0.15, root_vel_w=0.1, key_pos_w=0.15, contact_weights=[5.0] * 15, global_obs=False,
                global_root_height_obs=False, enable_tar_obs=True, use_contact_info=True, has_target_xy_obs=False, track_root=True,
                track_root_h=True, task1_w=0.7, task2_w=0.3, target_radius=1.0, rel_deepmimic_w=1.0, rel_task_w=0.0)
    variants = {"default": {}, "target_xy": {"has_target_xy_obs": True}, "root_height": {"global_root_height_obs": True},
                "no_tar_obs": {"enable_tar_obs": False}, "no_contact_info": {"use_contact_info": False},
                "no_root_h_tracking": {"track_root_h": False}, "task_product": {"rel_task_w": 0.5, "rel_deepmimic_w": 0.7},
                "everything": {"has_target_xy_obs": True, "global_root_height_obs": True, "track_root_h": False, "rel_task_w": 1.0},
                "mgdm_shipped": {"has_target_xy_obs": True, "enable_replan_timer_obs": True, "_mgdm": True},
                "global_obs": {"global_obs": True, "has_target_xy_obs": True}, "no_root_tracking": {"track_root": False},
                "no_root_tracking_at_all": {"track_root": False, "track_root_h": False, "global_obs": True}}
    arrs = {"target_xy": target_xy, "plan_clock": plan_clock}
    tables = {}
    import contextlib
    import io
    for tag, over in variants.items():
        cfg = dict(base, **over)
        e = make(cfg)
        with contextlib.redirect_stdout(io.StringIO()):                # (_compute_obs prints its segment table)
            shapes = ipe.IGParkourEnv._compute_obs(e, ret_obs_shapes=True)
        obs = ipe.IGParkourEnv._compute_obs(e)
        ipe.IGParkourEnv._update_reward(e)
        arrs[tag + "_obs"] = obs
        arrs[tag + "_reward"] = e._reward_buf.clone()
        if not cfg["track_root"]:
            # compute_done without the root checks (mgdm_dm_util.py:392-460: `if (track_root)` guards root position / rotation failure)
            arrs[tag + "_done"] = dmu.compute_done(
                done_buf=torch.zeros(n, dtype=torch.int), time=t(z["time_buf"]), ep_len=10.0, root_rot=e._char_root_rot,
                body_pos=e._char_rigid_body_pos, char_root_pos=e._char_root_pos, tar_root_rot=e._ref_root_rot, tar_body_pos=e._ref_body_pos,
                contact_force=e._char_contact_forces, contact_body_ids=torch.zeros(0, dtype=torch.int64),
                termination_heights=t(z["termination_heights"]), pose_termination=True, pose_termination_dist=t(z["pose_termination_dist"]),
                global_obs=bool(cfg["global_obs"]), enable_early_termination=True, track_root=False, root_pos_termination_dist=0.6,
                root_rot_termination_angle=1.309)
        for k, v in e._info["rewards"].items():
            arrs[tag + "_r_" + k] = v
        tables[tag] = {"config": {k: v for k, v in over.items()},
                       "obs_shapes": [[k, bool(v["use_normalizer"]), [int(d) for d in v["shape"]]] for k, v in shapes.items()],
                       "obs_dim": int(obs.shape[1])}
    assert np.array_equal(npy(arrs["default_obs"]), z["obs"]) and np.allclose(npy(arrs["default_reward"]), z["reward"], atol=1e-7)
    save("g26_obs_variants", **arrs)
    shipped = {}
    for name in ("data/envs/ig_parkour_env.yaml", "data/terrains/dm_env_civilization.yaml"):
        with open(os.path.join(REF, name)) as f:
            shipped[name] = yaml.safe_load(f)
    with open(os.path.join(OUT, "g26_obs_variants.json"), "w") as f:
        json.dump({"variants": tables, "shipped_configs": shipped}, f, indent=1, sort_keys=True)
    print("wrote g26_obs_variants.json", {k: v["obs_dim"] for k, v in tables.items()})


def gen_core():
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    civ = load_motion_file_safe(os.path.join(REF, "data/terrains/civilization.pkl"))
    teaser = load_motion_file_safe(os.path.join(REF, "data/terrains/TEASER_TERRAIN.pkl"))
    km = load_char()
    gen_quat(rng)
    gen_char(km)
    gen_kin(rng, km, civ["frames"])
    mlib, _ = gen_motion(rng, km, civ, teaser)
    rays = gen_rays()
    gen_heightmap(rng, civ, teaser, rays)
    gen_obs_reward_done(rng, km, mlib, civ, rays)
    gen_td_lambda(rng)


def main():
    """No flag: every stage, in order (reproduces all committed fixtures).  --only-<stage>: that stage alone."""
    run = {"core": gen_core, "voxel-mesh": lambda: gen_voxel_mesh(np.random.default_rng(10)), "dataset-yaml": gen_dataset_yaml,
           "procgen": gen_procgen, "terrain-geometry": gen_terrain_geometry, "done-branches": gen_done_branches, "ppo-loss": gen_ppo_loss,
           "normalizer": gen_normalizer, "trackers": gen_trackers, "action-head": gen_action_head, "recorded-files": gen_recorded_files,
           "motion-opt": gen_motion_opt, "mgdm": gen_mgdm, "motion-edit": gen_motion_edit, "sim-config": gen_sim_config, "stage-scripts": gen_stage_scripts, "experience-buffer": gen_experience_buffer,
           "obs-variants": gen_obs_variants}
    picked = [s_ for s_ in STAGES if "--only-" + s_ in sys.argv]
    if "--check" in sys.argv:
        # regenerate everything into a scratch directory and compare with the committed fixtures array by array
        global OUT
        OUT = tempfile.mkdtemp(prefix="parc_golden_check_")
    for s_ in picked or STAGES:
        print("== stage", s_)
        run[s_]()
    if "--check" in sys.argv:
        bad = 0
        for f in sorted(os.listdir(OUT)):
            if f.endswith(".json"):
                same = open(os.path.join(OUT, f)).read() == open(os.path.join(HERE, f)).read()
                bad += not same
                continue
            a, b = np.load(os.path.join(OUT, f)), np.load(os.path.join(HERE, f))
            for k in sorted(set(a.files) | set(b.files)):
                same = k in a.files and k in b.files and a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f")
                if not same:
                    bad += 1
                    print("DIFFERS", f, k)
        print("check: {} arrays differ from the committed fixtures".format(bad))
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
