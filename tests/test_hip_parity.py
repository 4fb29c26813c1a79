"""GPU parity: the HIP path (through the C-ABI of libparc_hip.so) against the CPU oracle on the same seeded
inputs and against the golden vectors the reference's Python produced.  fp32; tolerances as in
test_oracle_golden.py (device libm vs glibc/torch differ by a few ulp in sin/cos/atan2/acos)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, atol=2e-5, rtol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    bad = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)) > atol + rtol * np.abs(np.asarray(b, np.float64))
    if bad.any():
        idx = np.argwhere(bad)
        last = np.unique(idx[:, -1])
        raise AssertionError("{} of {} differ (atol {}, rtol {}); max |d| {:.3e}; trailing indices {}; first {}".format(
            int(bad.sum()), bad.size, atol, rtol, float(np.abs(np.asarray(a, np.float64) - b)[bad].max()), last[:40].tolist(),
            [(tuple(i), float(np.asarray(a)[tuple(i)]), float(np.asarray(b)[tuple(i)])) for i in idx[:6]]))


def T(x, dtype=torch.float32):
    return torch.tensor(np.asarray(x), dtype=dtype, device=DEV)


@pytest.fixture(scope="module")
def km():
    from parc_amd.anim.kin_char_model import KinCharModel
    from parc_amd.assets import humanoid_spec
    m = KinCharModel(DEV)
    m.load_char_file(humanoid_spec.write_mjcf())
    return m


@pytest.fixture(scope="module")
def mlib(km):
    from parc_amd.anim.motion_lib import MotionLib, LoopMode
    import pickle, tempfile, os, yaml
    z = golden("g3_motion")
    tmp = tempfile.mkdtemp(prefix="parc_test_")
    entries = []
    for i in range(4):
        p = os.path.join(tmp, str(z["clip_names"][i]) + ".pkl")
        with open(p, "wb") as f:
            pickle.dump({"fps": float(z["clip_fps"][i]), "loop_mode": LoopMode(int(z["clip_loop"][i])).name,
                         "frames": z["frames_%d" % i], "contacts": z["contacts_%d" % i]}, f)
        entries.append({"file": p, "weight": float(z["clip_weights_in"][i])})
    yp = os.path.join(tmp, "motions.yaml")
    with open(yp, "w") as f:
        yaml.safe_dump({"motions": entries}, f)
    return MotionLib(yp, km, DEV, contact_info=True)


def test_extension_loaded():
    from parc_amd import _hip
    assert _hip.lib().parc_abi_version() == 1


def test_g2_dof_rot_fk(km, oracle, ref_char):
    z = golden("g2_kin")
    jr = km.dof_to_rot(T(z["dof"]))
    close(jr, z["joint_rot"])
    close(jr, oracle.dof_to_rot(ref_char, z["dof"]), atol=2e-6)
    close(km.rot_to_dof(T(z["joint_rot"])), z["dof_back"])
    close(km.rot_to_dof(T(z["rand_joint_rot"])), z["dof_from_rand"])
    bp, br = km.forward_kinematics(T(z["root_pos"]), T(z["root_rot"]), T(z["joint_rot"]))
    close(bp, z["body_pos"])
    close(br, z["body_rot"])
    # ragged / empty batch
    e = km.dof_to_rot(torch.zeros((0, 28), device=DEV))
    assert e.shape == (0, 14, 4)
    one = km.dof_to_rot(T(z["dof"][:1]))
    close(one, z["joint_rot"][:1])


def test_g3_motion_lib_build_and_sample(mlib, ref_mlib):
    z = golden("g3_motion")
    close(mlib._motion_lengths, z["motion_lengths"], atol=1e-6)
    close(mlib._motion_weights, z["motion_weights"], atol=1e-6)
    close(mlib._motion_root_pos_delta, z["motion_root_pos_delta"], atol=1e-6)
    close(mlib._frame_root_pos, z["frame_root_pos"], atol=1e-6)
    close(mlib._frame_root_rot, z["frame_root_rot"], atol=2e-6)
    close(mlib._frame_joint_rot, z["frame_joint_rot"], atol=2e-6)
    close(mlib._frame_root_vel, z["frame_root_vel"], atol=1e-4)
    close(mlib._frame_root_ang_vel, z["frame_root_ang_vel"], atol=2e-4)
    close(mlib._frame_dof_vel, z["frame_dof_vel"], atol=2e-4)
    close(mlib._frame_contacts, z["frame_contacts"], atol=0)
    out = mlib.calc_motion_frame(T(z["q_ids"], torch.int64), T(z["q_times"]))
    names = ["q_root_pos", "q_root_rot", "q_root_vel", "q_root_ang_vel", "q_joint_rot", "q_dof_vel", "q_contacts"]
    tol = dict(q_root_vel=1e-4, q_root_ang_vel=2e-4, q_dof_vel=2e-4)
    o = ref_mlib.calc_motion_frame(z["q_ids"], z["q_times"])
    okeys = ["root_pos", "root_rot", "root_vel", "root_ang_vel", "joint_rot", "dof_vel", "contacts"]
    for t, n, ok in zip(out, names, okeys):
        close(t, z[n], atol=tol.get(n, 2e-5))
        close(t, o[ok], atol=tol.get(n, 2e-5))
    # empty query
    e = mlib.calc_motion_frame(torch.zeros(0, dtype=torch.int64, device=DEV), torch.zeros(0, device=DEV))
    assert e[0].shape == (0, 3)


def _hf_check(out, ref, boundary):
    out = out.detach().cpu().numpy()
    bad = out != ref
    assert not np.any(bad & ~(boundary < 1e-4)), "mismatch away from cell boundaries: %d" % int(np.sum(bad & ~(boundary < 1e-4)))
    assert np.mean(bad) < 2e-3


@pytest.mark.parametrize("name", ["g5_hf_civ", "g5_hf_teaser"])
def test_g5_heightmap_gather(name, oracle):
    from parc_amd import _hip
    z = golden(name)
    rays = T(golden("g4_rays")["ray_xy_points"])
    hf = T(z["hf"])
    ter = _hip.terrain_struct(hf, z["min_point"].tolist(), z["dxdy"].tolist())
    root, heading = T(z["root_pos"]), T(z["heading"])
    n, P = root.shape[0], rays.shape[0]
    ref_o = oracle.refresh_ray_obs_hfs(z["ray_xy_points"] if "ray_xy_points" in z else golden("g4_rays")["ray_xy_points"],
                                       z["root_pos"], z["heading"], z["hf"], z["min_point"], z["dxdy"])
    # (a) dense [N,P] output: rows are not 16-byte aligned -> scalar kernel
    out = torch.full((n, P), -99.0, device=DEV)
    _hip.check(_hip.lib().parc_refresh_ray_obs_hfs(_hip.stream(), n, _hip.ptr(rays), P, _hip.ptr(root), _hip.ptr(heading), ter,
                                                  -3.0, 3.0, _hip.ptr(out), P), "hf")
    _hf_check(out, z["ray_hfs"], z["boundary_dist"])
    _hf_check(out, ref_o, z["boundary_dist"])
    # (b) written into the observation row layout [N,1312] at column 871 -> vectorised float4 kernel
    obs = torch.full((n, 1312), -99.0, device=DEV)
    dst = _hip.c_vp(obs.data_ptr() + 4 * 871)
    _hip.check(_hip.lib().parc_refresh_ray_obs_hfs(_hip.stream(), n, _hip.ptr(rays), P, _hip.ptr(root), _hip.ptr(heading), ter,
                                                  -3.0, 3.0, dst, 1312), "hf")
    _hf_check(obs[:, 871:], z["ray_hfs"], z["boundary_dist"])
    assert torch.all(obs[:, :871] == -99.0)          # nothing outside the heightmap columns is touched
    assert torch.equal(obs[:, 871:], out) or np.mean((obs[:, 871:] != out).cpu().numpy()) < 1e-3
    # (c) n = 0 and a ragged n (not a multiple of the envs-per-block)
    assert _hip.lib().parc_refresh_ray_obs_hfs(_hip.stream(), 0, _hip.ptr(rays), P, _hip.ptr(root), _hip.ptr(heading), ter,
                                               -3.0, 3.0, dst, 1312) == 0
    obs2 = torch.full((n, 1312), -99.0, device=DEV)
    dst2 = _hip.c_vp(obs2.data_ptr() + 4 * 871)
    _hip.check(_hip.lib().parc_refresh_ray_obs_hfs(_hip.stream(), 7, _hip.ptr(rays), P, _hip.ptr(root), _hip.ptr(heading), ter,
                                                  -3.0, 3.0, dst2, 1312), "hf")
    assert torch.equal(obs2[:7, 871:], obs[:7, 871:]) and torch.all(obs2[7:] == -99.0)


def _core_from_golden(km, mlib, name="g6_step", **env_over):
    from parc_amd.tracker_core import TrackerConfig, TrackerCore
    from parc_amd.util.terrain_util import SubTerrain
    from parc_amd.envs.ig_parkour.default_config import default_env_config
    z = golden(name)
    cfg = TrackerConfig(dict(default_env_config()["env"], **env_over), km, 441)
    n = z["motion_ids"].shape[0]
    core = TrackerCore(n, DEV, km, mlib, cfg, T(z["rays"]))
    core.set_terrain(SubTerrain.from_arrays(z["hf"], z["min_point"], z["dxdy"], device=DEV))
    core.root_state[:, 0:3] = T(z["char_root_pos"])
    core.root_state[:, 3:7] = T(z["char_root_rot"])
    core.root_state[:, 7:10] = T(z["char_root_vel"])
    core.root_state[:, 10:13] = T(z["char_root_ang_vel"])
    ds = core.dof_state.view(n, 28, 2)
    ds[..., 0] = T(z["char_dof_pos"])
    ds[..., 1] = T(z["char_dof_vel"])
    core.rigid_body_state.view(n, 15, 13)[..., 0:3] = T(z["char_rigid_body_pos"])
    core.contact_forces.view(n, 15, 3)[:] = T(z["contact_forces"])
    core.env_offsets[:] = T(z["env_offsets"])
    core.motion_ids[:] = T(z["motion_ids"], torch.int64)
    core.motion_time_offsets[:] = T(z["motion_time_offsets"])
    core.motion_xy_offset[:] = T(z["motion_offsets"][z["motion_ids"], 0])
    core.time_buf[:] = T(z["time_buf"])
    return core, z


def test_g6_fused_post_step(km, mlib, oracle, ref_char, ref_mlib):
    from parc_amd import _hip
    core, z = _core_from_golden(km, mlib)
    core.refresh_obs_hfs()
    core.post_step(_hip.POST_REF | _hip.POST_OBS | _hip.POST_REWARD_DONE)
    torch.cuda.synchronize()
    _hf_check(core.ray_hfs, z["ray_hfs"], z["hf_boundary"])
    for k in ("ref_root_pos", "ref_root_rot", "ref_joint_rot", "ref_contacts", "ref_body_pos", "ref_dof_pos"):
        close(getattr(core, k), z[k])
    for k in ("ref_root_vel", "ref_root_ang_vel", "ref_dof_vel"):
        close(getattr(core, k), z[k], atol=2e-4)
    obs = core.obs.cpu().numpy()
    close(obs[:, 0:136], z["char_obs"])
    close(obs[:, 136:766], z["tar_obs"], atol=3e-5)
    close(obs[:, 766:856], z["tar_contacts"].reshape(64, -1))
    close(obs[:, 856:871], (np.linalg.norm(z["contact_forces"], axis=-1) > 1e-5).astype(np.float32), atol=0)
    close(core.reward, z["reward"])
    close(core.reward_terms[0:5].t(), z["reward_terms"])
    close(core.reward_terms[5], z["contact_penalty"])
    np.testing.assert_array_equal(core.done.cpu().numpy(), z["done_final"])
    # fail-rate EMA in env order
    fr = torch.ones(4, device=DEV)
    core.update_fail_rates(fr, 0.01)
    close(fr, z["fail_rates"], atol=1e-6)
    # oracle on the same inputs (full obs row incl. heightmap columns where they agree bit-exactly)
    off = z["motion_offsets"][z["motion_ids"], 0] - z["env_offsets"][:, 0:2]
    times = z["time_buf"] + z["motion_time_offsets"]
    tar_dt = (z["tar_obs_steps"].astype(np.float32) * np.float32(1.0 / 30.0)).astype(np.float32)
    o_obs = oracle.compute_obs(ref_char, ref_mlib, tar_dt, z["key_body_ids"], z["motion_ids"], times, off, z["char_root_pos"],
                               z["char_root_rot"], z["char_root_vel"], z["char_root_ang_vel"], z["char_dof_pos"], z["char_dof_vel"],
                               z["contact_forces"], z["ray_hfs"])
    close(obs[:, :871], o_obs[:, :871], atol=3e-5)


@pytest.mark.parametrize("tag", ["target_xy", "root_height", "no_tar_obs", "no_contact_info", "no_root_h_tracking", "task_product",
                                 "everything", "mgdm_shipped", "global_obs", "no_root_tracking", "no_root_tracking_at_all"])
def test_g26_observation_and_reward_variants(km, mlib, tag):
    """Fixture G26 = the reference's own IGParkourEnv._compute_obs / _update_reward (ig_parkour_env.py:1054-1244,1275-1404) run on the
    G6 state under the non-default switches: has_target_xy_obs (the configuration data/envs/ig_parkour_env.yaml ships),
    global_root_height_obs, enable_tar_obs / use_contact_info off, track_root_h off, rel_task_w > 0 (multiplicative task reward), the
    shipped motion-generator layout with the replan timer behind the target columns, global_obs (the row in world axes) and
    track_root off (root error / velocities / key bodies in each character's own heading frame; termination without the root checks:
    compute_done, mgdm_dm_util.py:392-460).  Device: the fused launch + obs_aux, the row
    gather parc_assemble_obs, the reward flags of parc_track_cfg_t, the one multiply of rel_task_w."""
    import json
    from parc_amd import _hip
    g = golden("g26_obs_variants")
    table = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g26_obs_variants.json")))["variants"][tag]
    over = {k: v for k, v in table["config"].items() if not k.startswith("_") and k != "enable_replan_timer_obs"}
    if not over.get("track_root", True):
        over.update(pose_termination=True, enable_early_termination=True, episode_length=10.0)       # what the fixture's compute_done was given
    core, z = _core_from_golden(km, mlib, **over)
    replan = tag == "mgdm_shipped"
    core.target_xy[:] = T(g["target_xy"])
    core.next_target_xy_time[:] = 1e9                      # no resample: the fixture's targets stay
    shapes, cols = core.cfg.obs_layout(replan)
    assert [[k, v["use_normalizer"], list(v["shape"])] for k, v in shapes.items()] == table["obs_shapes"]
    core.post_step(_hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF | _hip.POST_TARGETS)
    if core.cfg.rel_task_w > 0:
        torch.mul(core.reward, core.reward_terms[8], out=core.reward)       # IGParkourEnv._finish_reward
    want = g[tag + "_obs"]
    if cols is None:
        out = core.obs
    else:
        assert len(cols) == table["obs_dim"] == want.shape[1]
        out = torch.full((core.N, len(cols)), -7.0, device=DEV)
        clock = T(g["plan_clock"]).reshape(1)
        core.assemble_obs(T(cols, torch.int32), out, scalar=clock if replan else None)
        # a subset call rewrites exactly those rows
        out2 = out.clone()
        ids = torch.tensor([3, 17, 40], device=DEV)
        out2[ids] = 0.0
        core.assemble_obs(T(cols, torch.int32), out2, scalar=clock if replan else None, env_ids=ids)
        assert torch.equal(out, out2)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    P = 441
    names = [r[0] for r in table["obs_shapes"]]
    widths = [int(np.prod(r[2])) for r in table["obs_shapes"]]
    o = 0
    for name, w in zip(names, widths):
        if name == "hf":
            _hf_check(torch.tensor(got[:, o:o + w]), want[:, o:o + w], z["hf_boundary"])
        else:
            close(got[:, o:o + w], want[:, o:o + w], atol=3e-5)
        o += w
    assert o == got.shape[1]
    close(core.reward, g[tag + "_reward"], atol=2e-6)
    if tag + "_done" in g.files:           # RefCharEnv's flags before the motion-end override (dm_env.py:746-783 turns a clip's end into FAIL)
        end = z["motion_end"].astype(bool)
        np.testing.assert_array_equal(core.done.cpu().numpy()[~end], g[tag + "_done"][~end])
        assert int((g[tag + "_done"] != z["done_nocontact"]).sum()) >= 1                              # the root checks did matter in G6
    for i, k in enumerate(("pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty", "task_r1", "task_r2", "total_task_r")):
        if tag + "_r_" + k in g.files:
            close(core.reward_terms[i], g[tag + "_r_" + k], atol=2e-6)


def test_g6_fused_heightmap_equals_standalone(km, mlib):
    """PARC_POST_HF: the fused kernel writes the whole 1312-float row; same values as K5 + post-step."""
    from parc_amd import _hip
    core, z = _core_from_golden(km, mlib)
    core.refresh_obs_hfs()
    core.post_step(_hip.POST_REF | _hip.POST_OBS | _hip.POST_REWARD_DONE)
    two = core.obs.clone()
    core.obs[:] = -5.0
    core.post_step(_hip.POST_REF | _hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF)
    assert torch.equal(core.obs, two)
    _hf_check(core.ray_hfs, z["ray_hfs"], z["hf_boundary"])
    close(core.obs[:, :871], z["obs"][:, :871], atol=3e-5)
    # subset ids: only those rows change, heightmap columns included
    core.obs[:] = -5.0
    ids = torch.tensor([0, 63], dtype=torch.int64, device=DEV)
    core.post_step(_hip.POST_OBS | _hip.POST_HF, ids)
    assert torch.equal(core.obs[[0, 63]], two[[0, 63]]) and torch.all(core.obs[1:63] == -5.0)


def test_g6_post_step_subset_and_contact_bodies(km, mlib):
    """reset path: observations only, for a subset of env ids; and the fall-contact termination branch."""
    from parc_amd import _hip
    core, z = _core_from_golden(km, mlib)
    core.obs[:] = -7.0
    ids = torch.tensor([3, 9, 40], dtype=torch.int64, device=DEV)
    core.post_step(_hip.POST_OBS, ids)
    obs = core.obs.cpu().numpy()
    close(obs[[3, 9, 40], :871], z["obs"][[3, 9, 40], :871], atol=3e-5)
    mask = np.ones(64, bool)
    mask[[3, 9, 40]] = False
    assert np.all(obs[mask] == -7.0)
    core.post_step(_hip.POST_OBS, torch.zeros(0, dtype=torch.int64, device=DEV))   # empty id list is a no-op
    # contact bodies = feet -> done_feet golden (before the motion-end override; compare where no motion end)
    core.cfg.struct.num_contact_bodies = 2
    core.cfg.struct.contact_body_mask[11] = 1
    core.cfg.struct.contact_body_mask[14] = 1
    core.post_step(_hip.POST_REF | _hip.POST_REWARD_DONE)
    done = core.done.cpu().numpy()
    me = z["motion_end"]
    np.testing.assert_array_equal(done[~me], z["done_feet"][~me])
    assert np.all(done[me] == 1)
    core.cfg.struct.num_contact_bodies = 0


def test_g8b_done_every_branch_on_the_device(km, mlib):
    """The fused post-step kernel's termination logic on fixture G8b: every branch of compute_done / update_done tripped on its own
    (per-body pose distance for each of the 14 bodies, root position, root rotation, fall = height AND force, first-step
    exemption, timeout, motion end on CLAMP vs WRAP), flags as returned by the reference (mgdm_dm_util.py:392-460, dm_env.py:746-783)."""
    from parc_amd import _hip
    core, z = _core_from_golden(km, mlib, "g8b_done_branches")
    case = [str(c) for c in z["case"]]
    for tag, feet in (("nocontact", []), ("feet", [int(b) for b in z["feet"]])):
        core.cfg.struct.num_contact_bodies = len(feet)
        for b in range(15):
            core.cfg.struct.contact_body_mask[b] = 1 if b in feet else 0
        core.post_step(_hip.POST_REF | _hip.POST_REWARD_DONE)
        torch.cuda.synchronize()
        close(core.ref_body_pos, z["ref_body_pos"])
        done = core.done.cpu().numpy()
        bad = np.nonzero(done != z["done_final_" + tag])[0]
        assert bad.size == 0, [(case[i], int(done[i]), int(z["done_final_" + tag][i])) for i in bad]
        # before the motion-end override the flag is the reference's compute_done output
        me = z["motion_end"]
        np.testing.assert_array_equal(done[~me], z["done_" + tag][~me])
        fr = torch.ones(4, device=DEV)
        core.update_fail_rates(fr, 0.01)
        close(fr, z["fail_rates_" + tag], atol=1e-6)
    core.cfg.struct.num_contact_bodies = 0


def test_g9_td_lambda_and_advantage(oracle):
    from parc_amd.learning import rl_util
    z = golden("g9_td_lambda")
    g, lam, clip = [float(x) for x in z["params"]]
    ret = rl_util.compute_td_lambda_return(T(z["r"]), T(z["next_vals"]), T(z["done"], torch.int32), g, lam)
    close(ret, z["ret"], atol=1e-4, rtol=1e-6)
    close(ret, oracle.td_lambda_return(z["r"], z["next_vals"], z["done"], g, lam), atol=1e-4, rtol=1e-6)
    ret1 = rl_util.compute_td_lambda_return(T(z["r"][:1]), T(z["next_vals"][:1]), T(z["done"][:1], torch.int32), g, lam)
    close(ret1, z["ret_T1"])
    adv, ms = rl_util.normalize_advantage(T(z["ret"]), T(z["vals"]), T(z["rand_action_mask"]), clip)
    assert abs(ms[0].item() - float(z["adv_mean"])) < 1e-3 and abs(ms[1].item() - float(z["adv_std"])) < 1e-3
    close(adv, z["norm_adv"], atol=1e-4)


def test_full_size_properties(km, mlib):
    """BASELINE sizes (4096 envs): size-independent properties instead of an oracle run.
    heightmap: translation by whole cells shifts the lookup; flat terrain gives -z; TD(lambda) with lambda=0 is
    one-step TD; with no resets and constant r,v it is the closed form."""
    from parc_amd import _hip
    from parc_amd.learning import rl_util
    n, P = 4096, 441
    g = torch.Generator(device="cpu").manual_seed(1)
    rays = T(golden("g4_rays")["ray_xy_points"])
    hf = torch.rand((144, 144), generator=g).to(DEV) * 2.0
    ter = _hip.terrain_struct(hf, [-28.8, -28.8], [0.4, 0.4])
    root = torch.zeros((n, 3))
    root[:, 0:2] = (torch.rand((n, 2), generator=g) - 0.5) * 40.0
    root[:, 2] = torch.rand(n, generator=g)
    heading = (torch.rand(n, generator=g) - 0.5) * 6.28
    root, heading = root.to(DEV), heading.to(DEV)
    obs = torch.zeros((n, 1312), device=DEV)
    dst = _hip.c_vp(obs.data_ptr() + 4 * 871)
    L = _hip.lib()
    _hip.check(L.parc_refresh_ray_obs_hfs(_hip.stream(), n, _hip.ptr(rays), P, _hip.ptr(root), _hip.ptr(heading), ter, -3.0, 3.0, dst, 1312), "hf")
    a = obs[:, 871:].clone()
    assert a.min() >= -3.0 and a.max() <= 3.0
    # shift terrain origin and roots by 3 cells in x: identical lookups
    ter2 = _hip.terrain_struct(hf, [-28.8 + 1.2, -28.8], [0.4, 0.4])
    root2 = root.clone()
    root2[:, 0] += 1.2
    _hip.check(L.parc_refresh_ray_obs_hfs(_hip.stream(), n, _hip.ptr(rays), P, _hip.ptr(root2), _hip.ptr(heading), ter2, -3.0, 3.0, dst, 1312), "hf")
    assert (obs[:, 871:] != a).float().mean().item() < 2e-3     # only fp32 boundary flips
    # flat terrain: exactly clamp(h0 - z)
    flat = torch.full((144, 144), 0.25, device=DEV)
    ter3 = _hip.terrain_struct(flat, [-28.8, -28.8], [0.4, 0.4])
    _hip.check(L.parc_refresh_ray_obs_hfs(_hip.stream(), n, _hip.ptr(rays), P, _hip.ptr(root), _hip.ptr(heading), ter3, -3.0, 3.0, dst, 1312), "hf")
    assert torch.equal(obs[:, 871:], (0.25 - root[:, 2:3]).clamp(-3, 3).expand(-1, P))
    # TD(lambda)
    Tn = 32
    r = torch.rand((Tn, n), device=DEV)
    v = torch.rand((Tn, n), device=DEV) * 50
    done = torch.zeros((Tn, n), dtype=torch.int32, device=DEV)
    ret0 = rl_util.compute_td_lambda_return(r, v, done, 0.99, 0.0)
    assert torch.allclose(ret0, r + 0.99 * v, atol=1e-5)
    rc = torch.full((Tn, n), 0.5, device=DEV)
    vc = torch.full((Tn, n), 50.0, device=DEV)          # fixed point of v = r + g v
    retc = rl_util.compute_td_lambda_return(rc, vc, done, 0.99, 0.95)
    assert torch.allclose(retc, vc, atol=1e-3)
    done_all = torch.ones((Tn, n), dtype=torch.int32, device=DEV)
    ret1 = rl_util.compute_td_lambda_return(r, v, done_all, 0.99, 0.95)   # reset everywhere -> one-step TD
    assert torch.allclose(ret1, r + 0.99 * v, atol=1e-5)


@pytest.mark.parametrize("variant,n", [(1, 64), (1, 61), (0, 64)])
def test_sim_device_matches_host_build(km, variant, n):
    """Both simulator kernels -- body per lane (variant 1, the product's: parc_sim_bpl.h, through parc_sim_step) and one env per lane
    (variant 0, the single-source core; diagnostics library only, tools/parc_diag.py) -- against the HOST build of that core
    (oracle/sim_host.cpp): a few env steps of random actions
    on a bumpy terrain; n = 61 leaves a partially filled last workgroup.  fp32 with different libm / contraction /
    summation order: tolerance 1e-3 after 3 steps (contacts make the dynamics locally stiff)."""
    from parc_amd import _hip
    from parc_amd.sim_model import SimModel
    from oracle.sim_host import HostSim
    rng = np.random.default_rng(2)
    sm = SimModel(km)
    hf = (rng.random((40, 40)) * 0.3).astype(np.float32)
    host = HostSim(sm.struct, n, hf, [-4.0, -4.0], [0.4, 0.4])
    host.root_state[:, 0:2] = rng.random((n, 2)) * 6.0
    host.root_state[:, 2] = 1.15 + rng.random(n) * 0.2
    host.root_state[:, 7:13] = rng.standard_normal((n, 6)) * 0.3
    host.dof_state[:, :, 0] = rng.standard_normal((n, 28)) * 0.2
    host.dof_state[:, :, 1] = rng.standard_normal((n, 28)) * 0.5
    host.env_offsets[:, 0] = rng.random(n) * 2.0
    rs, ds = T(host.root_state.copy()), T(host.dof_state.copy())
    rb, cf = torch.zeros((n, 15, 13), device=DEV), torch.zeros((n, 15, 3), device=DEV)
    eo, lo, hi = T(host.env_offsets), T(host.act_lo), T(host.act_hi)
    d_hf = T(hf)
    ter = _hip.terrain_struct(d_hf, [-4.0, -4.0], [0.4, 0.4])
    L = _hip.lib()
    if variant == 0:
        import parc_diag                  # tools/parc_diag.py (tests/conftest.py puts tools/ on the path)
        step_fn, extra = parc_diag.lib().parc_diag_sim_step_env_per_lane, (64,)
    else:
        step_fn, extra = L.parc_sim_step, ()
    for step in range(3):
        act = (rng.standard_normal((n, 28)) * 0.5).astype(np.float32)
        host.step(act, n_sub=4, h=1.0 / 120.0)
        a = T(act)
        _hip.check(step_fn(_hip.stream(), sm.device_ptr(DEV), ter, n, _hip.ptr(rs), _hip.ptr(ds), _hip.ptr(rb), _hip.ptr(cf),
                           _hip.ptr(eo), _hip.ptr(a), _hip.ptr(lo), _hip.ptr(hi), 4, 1.0 / 120.0, *extra), "parc_sim_step")
        torch.cuda.synchronize()
    assert torch.isfinite(rs).all() and torch.isfinite(ds).all()
    assert float(np.abs(host.contact_forces).max()) > 10.0           # the scene does have contacts
    close(cf, host.contact_forces, atol=2.0, rtol=2e-2)                # N; forces are stiff in the penetration depth
    close(rb[..., 3:7], host.rigid_body_state[..., 3:7], atol=2e-3)
    close(rs[:, 0:7], host.root_state[:, 0:7], atol=1e-3)
    close(rs[:, 7:13], host.root_state[:, 7:13], atol=2e-2)
    close(ds[..., 0], host.dof_state[..., 0], atol=2e-3)
    close(rb[..., 0:3], host.rigid_body_state[..., 0:3], atol=2e-3)
    # refresh kernel == host refresh
    rb2, cf2 = torch.ones((n, 15, 13), device=DEV), torch.ones((n, 15, 3), device=DEV)
    ids = torch.arange(0, n, 2, device=DEV)
    _hip.check(L.parc_sim_refresh_bodies(_hip.stream(), sm.device_ptr(DEV), n, _hip.ptr(ids), int(ids.numel()), _hip.ptr(rs), _hip.ptr(ds),
                                         _hip.ptr(rb2), _hip.ptr(cf2)), "refresh")
    close(rb2[0::2, :, 0:7], rb[0::2, :, 0:7].cpu().numpy(), atol=1e-4)
    assert torch.all(rb2[1::2] == 1.0) and torch.all(cf2[0::2] == 0.0)


@pytest.mark.parametrize("case", ["default", "all_terms_l1", "critic_off_gate"])
def test_fused_ppo_loss_matches_autograd(case):
    """parc_ppo_loss (value + gradient in one pass) against a composite torch expression of PPOAgent._compute_loss
    (learning/ppo_agent.py:186-330), gradients through torch.autograd, at a batch size that is not a multiple of the block.
    The REFERENCE-generated check of the same kernel is tests/test_learner_gpu.py::test_g14_* (fixture G14, which also holds
    the batch without any random action: NaN in the reference and here)."""
    from parc_amd.learning import rl_util
    g = torch.Generator().manual_seed(11)
    B, A = 3000, 28            # not a multiple of the 256-thread block
    mean = (torch.randn(B, A, generator=g) * 0.7).to(DEV).requires_grad_(True)
    logstd = (torch.randn(A, generator=g) * 0.2 - 1.5).to(DEV).requires_grad_(True)
    pred = torch.randn(B, generator=g).to(DEV).requires_grad_(True)
    norm_a = (mean.detach() + torch.exp(logstd.detach()) * torch.randn(B, A, generator=g).to(DEV)).contiguous()
    adv = torch.randn(B, generator=g).to(DEV)
    mask = (torch.rand(B, generator=g) < 0.8).float().to(DEV)
    tar = torch.randn(B, generator=g).to(DEV)
    clip, bw, ew, rw, cw, l1 = 0.2, 10.0, 0.0, 0.0, 0.5, False
    if case == "all_terms_l1":
        ew, rw, l1 = 0.01, 0.003, True
    if case == "critic_off_gate":
        tar = tar + 9.0                      # critic loss > 20: the actor term must give no gradient

    def logp_of(mu, ls):
        z = (norm_a - mu) / torch.exp(ls)
        return -0.5 * torch.sum(z * z, dim=-1) + (-0.5 * A * np.log(2.0 * np.pi) - torch.sum(ls))
    old_logp = (logp_of(mean.detach(), logstd.detach()) + 0.3 * torch.randn(B, generator=g).to(DEV)).contiguous()

    # composite (reference formulation)
    diff = tar - pred
    critic_loss = diff.abs().mean() if l1 else diff.square().mean()
    m = (mask == 1.0).float()
    cnt = m.sum().clamp_min(1.0)
    ratio = torch.exp(logp_of(mean, logstd) - old_logp)
    l0, l1_ = adv * ratio, adv * torch.clamp(ratio, 1 - clip, 1 + clip)
    actor = -(torch.minimum(l0, l1_) * m).sum() / cnt
    viol = torch.clamp_max(mean + 1.0, 0.0).square().sum(-1) + torch.clamp_min(mean - 1.0, 0.0).square().sum(-1)
    actor = actor + bw * (viol * m).sum() / cnt
    ent = ((logstd.sum() + 0.5 * A * np.log(2.0 * np.pi * np.e)) * m).sum() / cnt
    actor = actor - ew * ent
    actor = actor + rw * (mean.square().sum(-1) * m).sum() / cnt
    term = torch.where(critic_loss.detach() > 20.0, actor.detach(), actor)
    loss_ref = term + cw * critic_loss
    g_ref = torch.autograd.grad(loss_ref, [mean, logstd, pred], allow_unused=True)

    loss, out = rl_util.ppo_loss(mean, logstd, pred, norm_a, old_logp, adv, mask, tar, clip, bw, ew, rw, cw, 20.0, l1)
    g_f = torch.autograd.grad(loss, [mean, logstd, pred])
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    ref_info = [loss_ref, critic_loss, actor, ((torch.abs(ratio - 1.0) > clip).float() * m).sum() / cnt, (ratio * m).sum() / cnt,
                (viol * m).sum() / cnt, ent, (mean.square().sum(-1) * m).sum() / cnt, cnt]
    for k, r in enumerate(ref_info):
        assert abs(out[k].item() - r.item()) <= 2e-5 * max(1.0, abs(r.item())), (k, out[k].item(), r.item())
    for a, b, name in zip(g_f, g_ref, ("mean", "logstd", "pred")):
        b = torch.zeros_like(a) if b is None else b
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-9, (name, float((a - b).abs().max()), scale)
    if case == "critic_off_gate":
        assert float(g_f[0].abs().max()) == 0.0 and float(g_f[1].abs().max()) == 0.0 and float(g_f[2].abs().max()) > 0.0


def test_post_step_and_sim_are_per_env_functions_at_8192_envs(km):
    """BASELINE configs[4] size (8192 envs/GPU): size-independent property instead of an oracle run - every output row
    depends on that env's inputs only, so permuting the envs permutes the outputs bit for bit (fused post-step kernel,
    simulator step), and an env_ids subset launch reproduces the same rows."""
    from parc_amd import _hip, synthetic
    from parc_amd.anim.motion_lib import MotionLib
    from parc_amd.envs.ig_parkour.default_config import default_env_config
    from parc_amd.sim_model import SimModel
    from parc_amd.tracker_core import TrackerConfig, TrackerCore
    from parc_amd.util import geom_util
    from parc_amd.util.terrain_util import SubTerrain
    n = 8192
    clips = synthetic.make_dataset(16, seed=3)
    ml = MotionLib(clips, km, DEV, init_type="clips", contact_info=True)
    hf, mn, dxdy, offs = synthetic.tile_square(clips)
    rays = geom_util.get_xy_points_cone(torch.zeros(2), 0.05, 2, 60, 3, 3, 0.26179938779)
    cfg = TrackerConfig(default_env_config()["env"], km, rays.shape[0])
    g = torch.Generator().manual_seed(9)
    perm = torch.randperm(n, generator=g).to(DEV)
    mids = torch.randint(0, 16, (n,), generator=g).to(DEV)
    toff = (torch.rand(n, generator=g) * 3.0).to(DEV)
    tbuf = (torch.randint(1, 60, (n,), generator=g).float() / 30.0).to(DEV)
    jitter = (torch.randn((n, 3), generator=g) * 0.05).to(DEV)
    djit = (torch.randn((n, 28), generator=g) * 0.1).to(DEV)
    act = (torch.randn((n, 28), generator=g) * 0.3).to(DEV)
    sm = SimModel(km)
    lo = torch.full((28,), -3.0, device=DEV)
    hi = torch.full((28,), 3.0, device=DEV)

    def run(order):
        core = TrackerCore(n, DEV, km, ml, cfg, rays)
        core.set_terrain(SubTerrain.from_arrays(hf, mn, dxdy, device=DEV))
        core.motion_ids[:] = mids[order]
        core.motion_xy_offset[:] = torch.tensor(offs[:, 0]).to(DEV)[core.motion_ids]
        core.motion_time_offsets[:] = toff[order]
        core.time_buf[:] = tbuf[order]
        core.post_step(_hip.POST_REF)
        core.root_state[:, 0:3] = core.ref_root_pos + jitter[order]
        core.root_state[:, 3:7] = core.ref_root_rot
        core.dof_state.view(n, 28, 2)[..., 0] = core.ref_dof_pos + djit[order]
        L = _hip.lib()
        _hip.check(L.parc_sim_step(_hip.stream(), sm.device_ptr(DEV), core._terrain_struct, n, _hip.ptr(core.root_state), _hip.ptr(core.dof_state),
                                   _hip.ptr(core.rigid_body_state), _hip.ptr(core.contact_forces), _hip.ptr(core.env_offsets),
                                   _hip.ptr(act[order].contiguous()), _hip.ptr(lo), _hip.ptr(hi), 4, 1.0 / 120.0), "sim")
        core.post_step(_hip.POST_REF | _hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF)
        torch.cuda.synchronize()
        return core
    ident = torch.arange(n, device=DEV)
    a, b = run(ident), run(perm)
    for name in ("root_state", "obs", "reward", "done", "ref_body_pos", "ref_dof_pos"):
        x, y = getattr(a, name), getattr(b, name)
        assert torch.equal(x.reshape(n, -1)[perm], y.reshape(n, -1)), name
    assert torch.equal(a.rigid_body_state.reshape(n, -1)[perm], b.rigid_body_state.reshape(n, -1))
    assert torch.equal(a.reward_terms[:, perm], b.reward_terms)
    assert torch.isfinite(a.obs).all() and torch.isfinite(a.root_state).all()
    # subset launch: the listed rows equal the all-env launch, the rest is untouched
    ids = perm[:777].sort().values
    snap = a.obs.clone()
    a.obs[:] = -9.0
    a.post_step(_hip.POST_OBS | _hip.POST_HF, ids)
    torch.cuda.synchronize()
    assert torch.equal(a.obs[ids], snap[ids])
    rest = torch.ones(n, dtype=torch.bool, device=DEV)
    rest[ids] = False
    assert torch.all(a.obs[rest] == -9.0)


def test_normalize_and_action_head_match_torch_expressions():
    """K12 (parc_normalize_clamp) bit-identical to clamp((x - mean) / std); K14 (parc_action_head) equal to the reference's
    sample / log_prob / unnormalize chain on the same noise."""
    from parc_amd import _hip
    from parc_amd.learning.normalizer import Normalizer
    g = torch.Generator().manual_seed(21)
    nrm = Normalizer((1312,), device=DEV, clip=10.0)
    nrm._mean[:] = torch.randn(1312, generator=g).to(DEV)
    nrm._std[:] = (torch.rand(1312, generator=g) * 2 + 0.01).to(DEV)
    x = (torch.randn((3, 37, 1312), generator=g) * 8).to(DEV)
    want = torch.clamp((x - nrm._mean) / nrm._std, -10.0, 10.0)
    got = nrm.normalize(x)
    assert torch.equal(got, want) and float(want.abs().max()) == 10.0
    buf = torch.empty_like(x)
    assert nrm.normalize(x, out=buf) is buf and torch.equal(buf, want)
    # action head
    n, A = 1000, 28
    mean = torch.randn((n, A), generator=g).to(DEV)
    logstd = (torch.randn(A, generator=g) * 0.3 - 2.0).to(DEV)
    noise = torch.randn((n, A), generator=g).to(DEV)
    mask = (torch.rand(n, generator=g) < 0.7).float().to(DEV)
    a_mean, a_std = torch.randn(A, generator=g).to(DEV), (torch.rand(A, generator=g) + 0.5).to(DEV)
    a, logp = torch.empty_like(mean), torch.empty(n, device=DEV)
    p = _hip.ptr
    _hip.check(_hip.lib().parc_action_head(_hip.stream(), n, A, p(mean), p(logstd), p(noise), p(mask), p(a_mean), p(a_std), p(a), p(logp)), "head")
    std = torch.exp(logstd)
    na = torch.where(mask.unsqueeze(-1) == 1.0, mean + std * noise, mean)
    lp = -0.5 * torch.sum(torch.square((na - mean) / std), dim=-1) + (-0.5 * A * np.log(2.0 * np.pi) - torch.sum(logstd))
    torch.cuda.synchronize()
    assert torch.allclose(a, na * a_std + a_mean, rtol=1e-6, atol=1e-6)
    assert torch.allclose(logp, lp, rtol=1e-5, atol=1e-4)
    assert torch.all(logp[mask == 0] == logp[mask == 0][0])            # mode actions: z = 0, log-prob is the normalising constant


def test_g13_points_hf_sdf_kernel(oracle):
    from parc_amd.util import terrain_util
    g = golden("g13_terrain_geometry")
    a = (T(g["sdf_points"]), T(g["sdf_hf"]), T(g["sdf_mbc"]), T(g["sdf_dxdy"]))
    n = (g["sdf_points"], g["sdf_hf"], g["sdf_mbc"], g["sdf_dxdy"])
    for kw, key in (({}, "sdf_inverted"), (dict(base_z=-5.0, inverted=False), "sdf_plain"), (dict(inverted=False, radius=0.07), "sdf_round")):
        out = terrain_util.points_hf_sdf(*a, **kw)
        close(out, g[key], atol=5e-7, rtol=0)                       # the reference's output (2 ulp: its 3-term norm)
        close(out, oracle.points_hf_sdf(*n, **kw), atol=5e-7, rtol=0)
    # ragged sizes: more cells than one LDS tile, point count not a multiple of the workgroup, one-cell and one-point fields
    rng = np.random.default_rng(5)
    for B, N, X, Y in ((2, 1, 1, 1), (1, 257, 50, 47), (3, 513, 3, 90)):
        p = rng.uniform(-2, 6, size=(B, N, 3)).astype(np.float32)
        hf = rng.uniform(-1, 1, size=(B, X, Y)).astype(np.float32)
        mbc = rng.uniform(-1, 1, size=(B, 2)).astype(np.float32)
        dxdy = np.array([0.4, 0.25], np.float32)
        close(terrain_util.points_hf_sdf(T(p), T(hf), T(mbc), T(dxdy)), oracle.points_hf_sdf(p, hf, mbc, dxdy), atol=1e-6, rtol=0)
    assert terrain_util.points_hf_sdf(torch.zeros((2, 0, 3), device=DEV), T(hf[:2]), T(mbc[:2]), T(dxdy)).shape == (2, 0)


def _capsule_box_points(km):
    from parc_amd.anim import kin_char_model as kcm
    from parc_amd.util import geom_util
    saved = [list(x) for x in km._geoms]
    for b in range(km.get_num_joints()):
        km._geoms[b] = [x for x in km._geoms[b] if x._shape_type != kcm.GeomType.SPHERE]
    pts = geom_util.get_char_point_samples(km)
    km._geoms = saved
    return pts


def test_g13_penetration_loss_and_hf_preprocessing(km):
    from parc_amd.util import terrain_util
    g = golden("g13_terrain_geometry")
    pts = _capsule_box_points(km)
    close(torch.cat(pts), g["pts"], atol=0, rtol=0)
    hf2 = T(np.stack([g["civ_hf"]] * 2)); mbc2 = T(np.stack([g["civ_min_point"]] * 2))
    loss, lp, lsdf = terrain_util.motion_frames_hf_sdf_loss(T(g["loss_frames"]), pts, hf2, mbc2, T(g["civ_dxdy"]), km, ret_vis_info=True)
    close(lp, g["loss_points"], atol=5e-6, rtol=0)
    close(lsdf, g["loss_sdf"], atol=1e-5, rtol=0)
    close(loss, g["loss"], atol=1e-5, rtol=1e-4)
    assert float(g["loss"][1]) > 10 * float(g["loss"][0]) > 0            # the lowered clip is the one that penetrates
    # dataset preprocessing: index / mask work is exact
    ter = terrain_util.SubTerrain.from_arrays(g["civ_hf"], g["civ_min_point"], g["civ_dxdy"], device=DEV)
    inds = terrain_util.compute_hf_extra_vals(T(g["extra_frames"]), ter, km, pts)
    assert [int(i.shape[0]) for i in inds] == g["extra_inds_count"].tolist()
    assert np.array_equal(torch.cat(inds).cpu().numpy(), g["extra_inds"])
    assert np.array_equal(ter.hf_mask.cpu().numpy(), g["extra_mask"])
    close(ter.hf_maxmin, g["extra_maxmin"], atol=5e-6, rtol=0)


def test_moments_kernel_matches_column_sums():
    """K22 (Normalizer.record): sum x and sum x^2 per column in one pass; fp32 with a fixed order, compared with float64 sums."""
    from parc_amd.learning.normalizer import Normalizer
    torch.manual_seed(3)
    for rows, dim in ((4096, 1312), (61, 28), (1, 4), (200, 1312)):
        x = torch.randn((rows, dim), device=DEV) * 3.0 + 0.5
        nz = Normalizer((dim,), DEV, clip=10.0)
        nz.record(x)
        nz.record(x[: max(rows // 2, 1)].contiguous())
        x2 = torch.cat([x, x[: max(rows // 2, 1)]]).double()
        close(nz._new_sum, x2.sum(0).cpu().numpy(), atol=1e-3, rtol=2e-6)
        close(nz._new_sum_sq, (x2 * x2).sum(0).cpu().numpy(), atol=1e-3, rtol=2e-6)
        assert nz._new_count == rows + max(rows // 2, 1)
        # and the update that follows reproduces the moments
        nz.update()
        close(nz._mean, x2.mean(0).cpu().numpy(), atol=1e-5, rtol=1e-5)
        close(nz._std, x2.std(0, unbiased=False).clamp_min(1e-4).cpu().numpy(), atol=1e-4, rtol=1e-4)


def test_fail_rate_kernel_is_the_sequential_loop_bit_for_bit():
    """dm_env.py:758-772: per finished env, in env order, fr[clip] = fr[clip] * (1 - w) (+ w if it failed).  The kernel walks the
    ballot masks in env order with the same two roundings, so it equals the loop exactly - also with many finishers per clip,
    env counts that are not multiples of the workgroup and more than one 16384-env segment."""
    from parc_amd import _hip
    rng = np.random.default_rng(11)
    for n, M, p_done in ((4096, 64, 0.02), (8192, 7, 0.5), (61, 3, 1.0), (20000, 5, 0.3)):
        mids = rng.integers(0, M, size=n)
        kind = np.where(rng.random(n) < p_done, rng.integers(1, 3, size=n), 0).astype(np.int32)
        fr0 = rng.random(M).astype(np.float32)
        w = np.float32(0.01)
        keep = np.float32(1.0 - 0.01)
        exp = fr0.copy()
        for e in range(n):
            if kind[e] != 0:
                v = np.float32(exp[mids[e]] * keep)
                exp[mids[e]] = np.float32(v + w) if kind[e] == 1 else v
        fr, d_mids, d_kind = T(fr0), T(mids, torch.int64), T(kind, torch.int32)
        _hip.check(_hip.lib().parc_update_fail_rates(_hip.stream(), n, M, _hip.ptr(d_mids), _hip.ptr(d_kind), 0.01, _hip.ptr(fr)),
                   "parc_update_fail_rates")
        assert np.array_equal(fr.cpu().numpy(), exp)


def test_g13_penetration_loss_gradients(km):
    """The terrain-penetration loss is what the motion optimiser differentiates: gradients with respect to the points and to the
    motion frames against the reference's autograd (fixture G13).  The kernel picks the column, torch re-evaluates that branch."""
    from parc_amd.util import terrain_util
    g = golden("g13_terrain_geometry")
    p = T(g["sdf_points"]).requires_grad_(True)
    sd = terrain_util.points_hf_sdf(p, T(g["sdf_hf"]), T(g["sdf_mbc"]), T(g["sdf_dxdy"]))
    close(sd.detach(), g["sdf_inverted"], atol=5e-7, rtol=0)
    sd.sum().backward()
    # the one-launch adjoint (terrain fixed) and the torch re-evaluation of the selected column (terrain differentiable) are the same map
    for kw in ({}, dict(inverted=False, base_z=-5.0), dict(inverted=False, radius=0.07)):
        pa, pb = T(g["sdf_points"]).requires_grad_(True), T(g["sdf_points"]).requires_grad_(True)
        hfb = T(g["sdf_hf"]).requires_grad_(True)
        wts = torch.randn(g["sdf_points"].shape[:2], device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        (terrain_util.points_hf_sdf(pa, T(g["sdf_hf"]), T(g["sdf_mbc"]), T(g["sdf_dxdy"]), **kw) * wts).sum().backward()
        (terrain_util.points_hf_sdf(pb, hfb, T(g["sdf_mbc"]), T(g["sdf_dxdy"]), **kw) * wts).sum().backward()
        assert float((pa.grad - pb.grad).abs().max()) < 1e-5 and float(hfb.grad.abs().sum()) > 0
    ref = g["sdf_inverted_grad"]
    got = p.grad.cpu().numpy()
    # the gradient is a unit vector per point; points equidistant from two columns (a tie inside fp32 rounding) may legitimately
    # pick the other column - allow a handful of those, everything else must agree
    bad = np.abs(got - ref).max(axis=-1) > 1e-4
    assert bad.mean() < 0.01, float(bad.mean())
    pts = _capsule_box_points(km)
    mf = T(g["loss_frames"]).requires_grad_(True)
    hf2 = T(np.stack([g["civ_hf"]] * 2)); mbc2 = T(np.stack([g["civ_min_point"]] * 2))
    loss = terrain_util.motion_frames_hf_sdf_loss(mf, pts, hf2, mbc2, T(g["civ_dxdy"]), km)
    close(loss.detach(), g["loss"], atol=1e-5, rtol=1e-4)
    loss.sum().backward()
    gr, ref = mf.grad.cpu().numpy(), g["loss_grad"]
    assert np.abs(ref).max() > 1.0
    assert np.abs(gr - ref).max() < 2e-3 * np.abs(ref).max(), float(np.abs(gr - ref).max())
    # without requires_grad the pose goes through the FK kernels and gives the same value
    loss_k = terrain_util.motion_frames_hf_sdf_loss(T(g["loss_frames"]), pts, hf2, mbc2, T(g["civ_dxdy"]), km)
    close(loss_k, loss.detach().cpu().numpy(), atol=1e-5, rtol=1e-4)


def test_clipped_norm_scale_matches_torch_expression():
    from parc_amd import _hip
    torch.manual_seed(2)
    for scale in (0.01, 100.0):
        x = torch.randn(100003, device=DEV) * scale
        ref = x * torch.clamp(1.0 / (torch.linalg.vector_norm(x) + 1e-6), max=1.0)
        nrm = torch.linalg.vector_norm(x).reshape(1)
        _hip.check(_hip.lib().parc_scale_by_clipped_norm(_hip.stream(), x.numel(), _hip.ptr(x), _hip.ptr(nrm), 1.0), "scale")
        assert torch.equal(x, ref)


def test_config0_motion_lib_and_fk_batch_at_its_full_size(km, oracle):
    """BASELINE.json configs[0] at the size SURVEY section 8(d) row 1 gives it: 64 clips = the authors' civilization clip (254 frames)
    under a per-clip yaw and translation, MotionLib + forward kinematics on 64 x 254 grid queries (every frame time of every clip)
    plus 28 672 random ones with times up to 1.1 clip lengths (past the end: clamped).  HIP calc_motion_frame + forward_kinematics
    against the C oracle on all 44 928 queries, at the parity tolerance; the oracle samples the clip rows the device stored, checked
    against its own first (see smoke_impl.adopt_device_frames for why)."""
    import smoke_impl
    from parc_amd.anim.motion_lib import MotionLib
    z = golden("g3_motion")
    fr0, con0 = z["frames_0"].astype(np.float32), z["contacts_0"].astype(np.float32)
    assert fr0.shape == (254, 34)
    rng = np.random.default_rng(80)

    def em_to_q(e):
        a = np.linalg.norm(e, axis=-1, keepdims=True)
        ax = np.where(a > 1e-8, e / np.maximum(a, 1e-8), np.array([0.0, 0.0, 1.0]))
        return np.concatenate([ax * np.sin(a / 2), np.cos(a / 2)], -1)

    def q_to_em(q):
        q = np.where(q[..., 3:4] < 0, -q, q)
        n = np.linalg.norm(q[..., :3], axis=-1, keepdims=True)
        ang = 2 * np.arctan2(n, q[..., 3:4])
        return np.where(n > 1e-8, q[..., :3] / np.maximum(n, 1e-8) * ang, 0.0)

    def q_mul(a, b):
        x1, y1, z1, w1 = [a[..., i] for i in range(4)]
        x2, y2, z2, w2 = [b[..., i] for i in range(4)]
        return np.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                         w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], -1)
    clips = []
    for k in range(64):
        yaw = rng.random() * 2 * np.pi
        c, s_ = np.cos(yaw), np.sin(yaw)
        f = fr0.astype(np.float64).copy()
        xy = f[:, 0:2].copy()
        f[:, 0] = c * xy[:, 0] - s_ * xy[:, 1] + rng.normal() * 3.0
        f[:, 1] = s_ * xy[:, 0] + c * xy[:, 1] + rng.normal() * 3.0
        qy = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
        f[:, 3:6] = q_to_em(q_mul(np.broadcast_to(qy, (254, 4)), em_to_q(f[:, 3:6])))
        clips.append(dict(frames=f.astype(np.float32), contacts=con0, fps=30.0, loop=0, weight=1.0, name="civ_%02d" % k))
    ml = MotionLib(clips, km, DEV, init_type="clips", contact_info=True)
    ids = np.concatenate([np.repeat(np.arange(64), 254), rng.integers(0, 64, 28672)]).astype(np.int64)
    length = 253.0 / 30.0
    times = np.concatenate([np.tile(np.arange(254) / 30.0, 64), rng.random(28672) * 1.1 * length]).astype(np.float32)
    assert ids.shape == (44928,)
    rp, rr, rv, rav, jr, dv, con = ml.calc_motion_frame(T(ids, torch.int64), T(times))
    bp, br = km.forward_kinematics(rp, rr, jr)
    torch.cuda.synchronize()
    # ---- oracle
    full = lambda t: t.detach().cpu().numpy()
    char = oracle.Char(full(km._parent_indices), full(km._local_translation), full(km._local_rotation), [j.joint_type.value for j in km._joints],
                       [full(j.axis) if j.axis is not None else np.zeros(3, np.float32) for j in km._joints], [j.dof_idx for j in km._joints])
    om = oracle.MotionLib(char, [c_["frames"] for c_ in clips], [30.0] * 64, [0] * 64, [1.0] * 64, [c_["contacts"] for c_ in clips])
    om.fps_max = 30.0
    st = {}
    smoke_impl.adopt_device_frames(om, ml, st)
    assert st["stored_frames_device_vs_oracle"]["root_rot"]["max_abs_diff"] <= 5e-7
    o = om.calc_motion_frame(ids, times)
    obp, obr = oracle.forward_kinematics(char, o["root_pos"], o["root_rot"], o["joint_rot"])
    close(rp, o["root_pos"], atol=2e-5)
    close(rr, o["root_rot"], atol=2e-5)
    close(jr, o["joint_rot"], atol=2e-5)
    close(rv, o["root_vel"], atol=0, rtol=0)
    close(rav, o["root_ang_vel"], atol=0, rtol=0)
    close(dv, o["dof_vel"], atol=0, rtol=0)
    close(con, o["contacts"], atol=2e-5)                    # the blend weight carries the rounding of time / length x (frames - 1)
    close(bp, obp, atol=5e-5)
    close(br, obr, atol=2e-5)
    # queries past the end of a clamped clip sit on its last frame
    past = times > length
    assert past.sum() > 1000
    last = np.stack([c_["frames"][-1, 0:3] for c_ in clips])[ids[past]]
    close(rp[T(past, torch.bool)], last, atol=1e-6)


def test_step_tail_equals_the_two_separate_launches():
    """parc_step_tail co-schedules the per-step publication of the reference state (ref_* buffers) with the fail-rate update in one launch
    of heterogeneous workgroups; the env's step uses it and leaves PARC_POST_REF out of the fused launch.  Same bits as the two
    separate paths (parc_track_post_step with PARC_POST_REF -> ref_state_kernel, parc_update_fail_rates), and the fused launch without
    PARC_POST_REF leaves the reference buffers alone while reward / termination (which sample the reference pose themselves) do not
    change."""
    from parc_amd import _hip, workloads
    n = 1000                                             # not a multiple of 16: a partly filled last state workgroup
    A, clips, _ = workloads.build_core("boxes_64clips", n, DEV, seed=6)
    Bc, _, _ = workloads.build_core("boxes_64clips", n, DEV, seed=6)
    g = torch.Generator().manual_seed(3)
    fused = _hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF | _hip.POST_TARGETS
    M = len(clips)
    frA = (torch.rand(M, generator=g) * 0.5 + 0.25).to(DEV)
    frB = frA.clone()
    names = ("ref_root_pos", "ref_root_rot", "ref_root_vel", "ref_root_ang_vel", "ref_joint_rot", "ref_dof_pos", "ref_dof_vel", "ref_contacts",
             "ref_body_pos")
    for step in range(3):
        for c in (A, Bc):
            c.time_buf += 1.0 / 30.0
            c.root_state[:, 0:3] += 0.01 * (step + 1)
            if step == 1:                                 # push some envs over a termination threshold so that fail rates move
                c.rigid_body_state.view(n, 15, 13)[::5, :, 2] += 1.0
            c.target_rand.copy_(A.target_rand)
        for nm in names:
            getattr(Bc, nm).fill_(-7.0)
        A.post_step(fused | _hip.POST_REF)
        A.update_fail_rates(frA, 0.01)
        Bc.post_step(fused)
        assert all(bool((getattr(Bc, nm) == -7.0).all()) for nm in names)          # the fused launch does not touch the reference state
        Bc.step_tail(frB, 0.01)
        torch.cuda.synchronize()
        for nm in names:
            assert torch.equal(getattr(A, nm), getattr(Bc, nm)), (step, nm)
        for nm in ("obs", "reward", "reward_terms", "done", "done_kind"):
            assert torch.equal(getattr(A, nm), getattr(Bc, nm)), (step, nm)
        assert torch.equal(frA, frB)
    fr0 = (torch.rand(M, generator=torch.Generator().manual_seed(3)) * 0.5 + 0.25).to(DEV)
    assert (A.done != 0).any() and not torch.equal(frA, fr0)                        # the scene did move the fail rates
