"""The CPU oracle (oracle/parc_oracle.c) against the golden vectors produced by the reference's own
Python (tests/golden/gen_golden.py).  fp32 throughout; the only differences allowed are libm-vs-torch
rounding of sin/cos/atan2/acos (a few ulp), hence 1e-5-scale tolerances, stated per test."""
import os

import numpy as np

from conftest import GOLDEN, golden

ATOL = 2e-6   # |x| <= ~1 quantities: a few fp32 ulp
RTOL = 1e-5


def close(a, b, atol=ATOL, rtol=RTOL):
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def test_g1_quaternion_ops(oracle):
    z = golden("g1_quat")
    close(oracle.quat_mul(z["a"], z["b"]), z["quat_mul"])
    close(oracle.quat_rotate(z["a"], z["v"]), z["quat_rotate"], atol=1e-5)
    close(oracle.exp_map_to_quat(z["exp_map"]), z["exp_map_to_quat"])
    close(oracle.quat_to_exp_map(z["a"]), z["quat_to_exp_map"], atol=1e-5)
    close(oracle.axis_angle_to_quat(z["axis"], z["angle"]), z["axis_angle_to_quat"])
    close(oracle.quat_to_tan_norm(z["a"]), z["quat_to_tan_norm"])
    close(oracle.slerp(z["a"], z["b"], z["blend"]), z["slerp"], atol=1e-5)
    close(oracle.calc_heading(z["a"]), z["calc_heading"], atol=1e-5)
    close(oracle.calc_heading_quat_inv(z["a"]), z["calc_heading_quat_inv"], atol=1e-5)
    close(oracle.quat_diff_angle(z["a"], z["b"]), z["quat_diff_angle"], atol=2e-5)


def test_g2_dof_rot_fk(oracle, ref_char):
    z = golden("g2_kin")
    jr = oracle.dof_to_rot(ref_char, z["dof"])
    close(jr, z["joint_rot"])
    close(oracle.rot_to_dof(ref_char, z["joint_rot"]), z["dof_back"], atol=1e-5)
    close(oracle.rot_to_dof(ref_char, z["rand_joint_rot"]), z["dof_from_rand"], atol=1e-5)
    bp, br = oracle.forward_kinematics(ref_char, z["root_pos"], z["root_rot"], z["joint_rot"])
    close(bp, z["body_pos"], atol=1e-5)
    close(br, z["body_rot"], atol=1e-5)


def test_g3_motion_lib_derived_arrays(ref_mlib):
    z = golden("g3_motion")
    m = ref_mlib
    np.testing.assert_array_equal(m.num_frames, z["motion_num_frames"])
    np.testing.assert_array_equal(m.start_idx, z["motion_start_idx"])
    close(m.length, z["motion_lengths"])
    close(m.weights, z["motion_weights"])
    close(m.pos_delta, z["motion_root_pos_delta"])
    close(m.root_pos, z["frame_root_pos"])
    close(m.root_rot, z["frame_root_rot"])
    close(m.joint_rot, z["frame_joint_rot"])
    # velocities are differences * fps: ulp-level rotation differences are amplified by fps (30x)
    close(m.root_vel, z["frame_root_vel"], atol=1e-4)
    close(m.root_ang_vel, z["frame_root_ang_vel"], atol=2e-4)
    close(m.dof_vel, z["frame_dof_vel"], atol=2e-4)
    close(m.contacts, z["frame_contacts"])


def test_g3_calc_motion_frame(ref_mlib):
    z = golden("g3_motion")
    o = ref_mlib.calc_motion_frame(z["q_ids"], z["q_times"])
    close(o["root_pos"], z["q_root_pos"], atol=1e-5)
    close(o["root_rot"], z["q_root_rot"], atol=1e-5)
    close(o["joint_rot"], z["q_joint_rot"], atol=1e-5)
    close(o["root_vel"], z["q_root_vel"], atol=1e-4)
    close(o["root_ang_vel"], z["q_root_ang_vel"], atol=2e-4)
    close(o["dof_vel"], z["q_dof_vel"], atol=2e-4)
    close(o["contacts"], z["q_contacts"], atol=1e-5)


def test_g4_ray_template(oracle):
    z = golden("g4_rays")
    p = z["params"]
    pts = oracle.xy_points_cone(float(p[0]), int(p[1]), int(p[2]), int(p[3]), int(p[4]), float(p[5]))
    assert pts.shape == (441, 2)
    close(pts, z["ray_xy_points"], atol=1e-6)


def _check_hf(out, z, key="ray_hfs", bkey="boundary_dist"):
    """Nearest-cell lookup is discontinuous: a query whose cell coordinate sits within 1e-4 of a .5
    rounding boundary may legitimately land in the neighbouring cell when sin/cos differ by an ulp.
    Everything else must be bit-exact."""
    ref = z[key]
    bad = out != ref
    near = z[bkey] < 1e-4
    assert not np.any(bad & ~near), "mismatch away from rounding boundaries: %d" % int(np.sum(bad & ~near))
    assert np.mean(bad) < 1e-3


def test_g5_local_heightmap(oracle):
    rays = golden("g4_rays")["ray_xy_points"]
    for name in ("g5_hf_civ", "g5_hf_teaser"):
        z = golden(name)
        out = oracle.refresh_ray_obs_hfs(rays, z["root_pos"], z["heading"], z["hf"], z["min_point"], z["dxdy"])
        _check_hf(out, z)
        assert out.min() >= -3.0 and out.max() <= 3.0


def _step_inputs(z):
    off = z["motion_offsets"][z["motion_ids"], 0] - z["env_offsets"][:, 0:2]
    times = z["time_buf"] + z["motion_time_offsets"]
    return off.astype(np.float32), times.astype(np.float32)


def test_g6_ref_motion_update(oracle, ref_char, ref_mlib):
    z = golden("g6_step")
    off, times = _step_inputs(z)
    o = oracle.update_ref_motion(ref_char, ref_mlib, z["motion_ids"], times, off)
    for k in ("ref_root_pos", "ref_root_rot", "ref_joint_rot", "ref_contacts"):
        close(o[k], z[k], atol=1e-5)
    for k in ("ref_root_vel", "ref_root_ang_vel", "ref_dof_vel"):
        close(o[k], z[k], atol=2e-4)
    close(o["ref_body_pos"], z["ref_body_pos"], atol=1e-5)
    close(o["ref_dof_pos"], z["ref_dof_pos"], atol=1e-5)


def test_g6_full_observation(oracle, ref_char, ref_mlib):
    z = golden("g6_step")
    off, times = _step_inputs(z)
    glob = z["char_root_pos"] + z["env_offsets"]
    heading = oracle.calc_heading(z["char_root_rot"])
    hfs = oracle.refresh_ray_obs_hfs(z["rays"], glob, heading, z["hf"], z["min_point"], z["dxdy"])
    _check_hf(hfs, z, "ray_hfs", "hf_boundary")
    tar_dt = (z["tar_obs_steps"].astype(np.float32) * np.float32(1.0 / 30.0)).astype(np.float32)
    obs = oracle.compute_obs(ref_char, ref_mlib, tar_dt, z["key_body_ids"], z["motion_ids"], times, off,
                             z["char_root_pos"], z["char_root_rot"], z["char_root_vel"], z["char_root_ang_vel"],
                             z["char_dof_pos"], z["char_dof_vel"], z["contact_forces"], z["ray_hfs"])
    assert obs.shape == (64, 1312)
    close(obs[:, 0:136], z["char_obs"], atol=1e-5)
    close(obs[:, 136:766], z["tar_obs"], atol=2e-5)
    close(obs, z["obs"], atol=2e-5)


def test_g7_reward(oracle, ref_char):
    z = golden("g6_step")
    contact_w = np.full(15, 5.0, np.float32)
    r, terms = oracle.compute_reward(ref_char, z["key_body_ids"], z, z, z["joint_err_w"], z["dof_err_w"], contact_w, z["reward_w"])
    close(terms[:, 0:5], z["reward_terms"], atol=1e-5)
    close(terms[:, 5], z["contact_penalty"], atol=1e-5)
    close(r, z["reward"], atol=1e-5)


def test_g26_observation_and_reward_switches(oracle, ref_char, ref_mlib):
    """every switch of IGParkourEnv._compute_obs / _update_reward (ig_parkour_env.py:1054-1244,1275-1404) that G26 holds the
    reference's own output for: the oracle's row, reward, every reward term and - without root tracking - the termination flags"""
    import json
    z = golden("g6_step")
    g = golden("g26_obs_variants")
    tables = json.load(open(os.path.join(GOLDEN, "g26_obs_variants.json")))["variants"]
    off, times = _step_inputs(z)
    tar_dt = (z["tar_obs_steps"].astype(np.float32) * np.float32(1.0 / 30.0)).astype(np.float32)
    base = dict(global_obs=False, global_root_height_obs=False, enable_tar_obs=True, use_contact_info=True, has_target_xy_obs=False,
                track_root=True, track_root_h=True, rel_deepmimic_w=1.0, rel_task_w=0.0, enable_replan_timer_obs=False)
    names = ("pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty", "task_r1", "task_r2", "total_task_r")
    assert len(tables) >= 12
    for tag, tab in tables.items():
        cfg = dict(base, **{k: v for k, v in tab["config"].items() if not k.startswith("_")})
        obs = oracle.compute_obs(ref_char, ref_mlib, tar_dt, z["key_body_ids"], z["motion_ids"], times, off, z["char_root_pos"],
                                 z["char_root_rot"], z["char_root_vel"], z["char_root_ang_vel"], z["char_dof_pos"], z["char_dof_vel"],
                                 z["contact_forces"], z["ray_hfs"], global_obs=cfg["global_obs"],
                                 global_root_height_obs=cfg["global_root_height_obs"], enable_tar_obs=cfg["enable_tar_obs"],
                                 use_contact_info=cfg["use_contact_info"], target_xy=g["target_xy"] if cfg["has_target_xy_obs"] else None,
                                 replan_t=float(g["plan_clock"]) if cfg["enable_replan_timer_obs"] else None)
        assert obs.shape == g[tag + "_obs"].shape == (64, tab["obs_dim"]), tag
        close(obs, g[tag + "_obs"], atol=2e-5)
        r, terms = oracle.compute_reward(ref_char, z["key_body_ids"], z, z, z["joint_err_w"], z["dof_err_w"], np.full(15, 5.0, np.float32),
                                         z["reward_w"], rel_dm_w=cfg["rel_deepmimic_w"], track_root=cfg["track_root"],
                                         track_root_h=cfg["track_root_h"], use_contact_info=cfg["use_contact_info"], target_xy=g["target_xy"],
                                         rel_task_w=cfg["rel_task_w"], all_terms=True)
        close(r, g[tag + "_reward"], atol=1e-5)
        for i, name in enumerate(names):
            if tag + "_r_" + name in g:
                close(terms[:, i], g[tag + "_r_" + name], atol=1e-5)
            else:
                assert name == "contact_penalty" and not cfg["use_contact_info"]
        if not cfg["track_root"]:
            pre, _, _ = oracle.update_done(
                time_buf=z["time_buf"], ep_len=10.0, char_root_rot=z["char_root_rot"], body_pos=z["char_rigid_body_pos"],
                ref_root_rot=z["ref_root_rot"], ref_body_pos=z["ref_body_pos"], contact_forces=z["contact_forces"], contact_body_ids=[],
                env_offsets=z["env_offsets"], hf=z["hf"], min_point=z["min_point"], dxdy=z["dxdy"], termination_height=0.15,
                pose_termination=True, pose_termination_dist=z["pose_termination_dist"], enable_early_termination=True, track_root=False,
                root_pos_term_dist=0.6, root_rot_term_angle=1.309, motion_ids=z["motion_ids"], motion_times=times, motion_len=ref_mlib.length,
                motion_loop_mode=ref_mlib.loop_mode, fail_rates=np.ones(4, np.float32))
            np.testing.assert_array_equal(pre, g[tag + "_done"])
            assert (pre != z["done_nocontact"]).any()                  # the root checks did decide some of the default flags


def test_g8_done_and_fail_rates(oracle, ref_mlib):
    z = golden("g6_step")
    _, times = _step_inputs(z)
    common = dict(time_buf=z["time_buf"], ep_len=10.0, char_root_rot=z["char_root_rot"], body_pos=z["char_rigid_body_pos"],
                  ref_root_rot=z["ref_root_rot"], ref_body_pos=z["ref_body_pos"], contact_forces=z["contact_forces"],
                  env_offsets=z["env_offsets"], hf=z["hf"], min_point=z["min_point"], dxdy=z["dxdy"], termination_height=0.15,
                  pose_termination=True, pose_termination_dist=z["pose_termination_dist"], enable_early_termination=True,
                  track_root=True, root_pos_term_dist=0.6, root_rot_term_angle=1.309, motion_ids=z["motion_ids"],
                  motion_times=times, motion_len=ref_mlib.length, motion_loop_mode=ref_mlib.loop_mode,
                  fail_rates=np.ones(4, np.float32))
    pre, fin, fr = oracle.update_done(contact_body_ids=[], **common)
    np.testing.assert_array_equal(pre, z["done_nocontact"])
    np.testing.assert_array_equal(fin, z["done_final"])
    close(fr, z["fail_rates"], atol=1e-6)
    pre2, _, _ = oracle.update_done(contact_body_ids=[11, 14], **common)
    np.testing.assert_array_equal(pre2, z["done_feet"])
    assert set(np.unique(fin)) <= {0, 1, 3}
    assert (fin == 1).sum() > 3 and (fin == 0).sum() > 3


def test_g8_tracking_error(oracle, ref_char):
    z = golden("g6_step")
    jr = oracle.dof_to_rot(ref_char, z["char_dof_pos"])
    bp, br = oracle.forward_kinematics(ref_char, z["char_root_pos"], z["char_root_rot"], jr)
    rbp, rbr = oracle.forward_kinematics(ref_char, z["ref_root_pos"], z["ref_root_rot"], z["ref_joint_rot"])
    te = oracle.tracking_error(z["char_root_pos"], z["char_root_rot"], br, bp, z["ref_root_pos"], z["ref_root_rot"], rbr, rbp,
                               z["char_root_vel"], z["char_root_ang_vel"], z["char_dof_vel"], z["ref_root_vel"],
                               z["ref_root_ang_vel"], z["ref_dof_vel"])
    close(te, z["tracking_error"], atol=2e-5)


def test_g9_td_lambda_and_advantage(oracle):
    z = golden("g9_td_lambda")
    g, lam, clip = [float(x) for x in z["params"]]
    ret = oracle.td_lambda_return(z["r"], z["next_vals"], z["done"], g, lam)
    close(ret, z["ret"], atol=1e-4, rtol=1e-6)
    close(oracle.td_lambda_return(z["r"][:1], z["next_vals"][:1], z["done"][:1], g, lam), z["ret_T1"], atol=1e-5)
    adv, mean, std = oracle.adv_normalize(z["ret"], z["vals"], z["rand_action_mask"], clip)
    assert abs(mean - float(z["adv_mean"])) < 1e-3 and abs(std - float(z["adv_std"])) < 1e-3
    close(adv, z["norm_adv"], atol=1e-4)


def test_td_lambda_brute_force_definition(oracle):
    """Independent O(T^2) definition of TD(lambda) (the reference keeps one at learning/rl_util.py:31-73)."""
    rng = np.random.default_rng(3)
    T, N, g, lam = 9, 7, 0.97, 0.9
    r = rng.random((T, N)).astype(np.float32)
    v = rng.random((T, N)).astype(np.float32) * 5
    done = rng.choice([0, 0, 0, 1, 3], size=(T, N)).astype(np.int32)
    ret = oracle.td_lambda_return(r, v, done, g, lam)
    brute = np.zeros((T, N))
    for i in range(N):
        for t0 in range(T):
            new_val, sum_r, cd, cl = 0.0, 0.0, 1.0, 1.0
            for t in range(t0, T):
                sum_r += cd * r[t, i]
                cur = sum_r + cd * g * v[t, i]
                if done[t, i] == 0 and t < T - 1:
                    new_val += (1 - lam) * cl * cur
                else:
                    new_val += cl * cur
                    break
                cd *= g
                cl *= lam
            brute[t0, i] = new_val
    np.testing.assert_allclose(ret, brute, atol=1e-4)


def test_g13_points_hf_sdf(oracle):
    # box SDF of point sets to heightfield columns (terrain_util.points_hf_sdf); exact ops except the 3-term norm, whose
    # summation inside torch's CPU kernel is not specified: 2 fp32 ulp of a ~1 m distance
    g = golden("g13_terrain_geometry")
    args = (g["sdf_points"], g["sdf_hf"], g["sdf_mbc"], g["sdf_dxdy"])
    close(oracle.points_hf_sdf(*args), g["sdf_inverted"], atol=5e-7, rtol=0)
    close(oracle.points_hf_sdf(*args, base_z=-5.0, inverted=False), g["sdf_plain"], atol=5e-7, rtol=0)
    close(oracle.points_hf_sdf(*args, inverted=False, radius=0.07), g["sdf_round"], atol=5e-7, rtol=0)
    # hand-checked cases on a 1 m block (cell 2,2 of a flat 0.4 m grid): a point inside the block is 0.2 m from the free air of the
    # neighbouring columns (nearer than the 0.25 m to its top); points in the air get the distance to the nearest face of their
    # own air column - the cell walls count, so it saturates at half a cell (0.2 m)
    hf = np.zeros((1, 5, 5), np.float32)
    hf[0, 2, 2] = 1.0
    pts = np.array([[[0.8, 0.8, 0.75], [0.8, 0.8, 1.5], [0.0, 0.0, 0.3], [0.0, 0.0, 0.05]]], np.float32)
    d = oracle.points_hf_sdf(pts, hf, np.zeros((1, 2), np.float32), np.array([0.4, 0.4], np.float32))
    close(d[0], [-0.2, 0.2, 0.2, 0.05], atol=1e-6)


def test_g8b_done_every_branch(oracle, ref_mlib):
    """Fixture G8b trips every branch of compute_done / update_done on its own (mgdm_dm_util.py:392-460, dm_env.py:746-783): each of
    the 14 per-body pose distances, root position, root rotation, the fall test (height AND force on non-contact bodies), the
    first-step exemption, timeout, motion end on CLAMP vs WRAP clips."""
    z = golden("g8b_done_branches")
    times = z["time_buf"] + z["motion_time_offsets"]
    common = dict(time_buf=z["time_buf"], ep_len=10.0, char_root_rot=z["char_root_rot"], body_pos=z["char_rigid_body_pos"],
                  ref_root_rot=z["ref_root_rot"], ref_body_pos=z["ref_body_pos"], contact_forces=z["contact_forces"],
                  env_offsets=z["env_offsets"], hf=z["hf"], min_point=z["min_point"], dxdy=z["dxdy"], termination_height=0.15,
                  pose_termination=True, pose_termination_dist=z["pose_termination_dist"], enable_early_termination=True,
                  track_root=True, root_pos_term_dist=0.6, root_rot_term_angle=1.309, motion_ids=z["motion_ids"],
                  motion_times=times, motion_len=ref_mlib.length, motion_loop_mode=ref_mlib.loop_mode,
                  fail_rates=np.ones(4, np.float32))
    for tag, cb in (("nocontact", []), ("feet", [int(b) for b in z["feet"]])):
        pre, fin, fr = oracle.update_done(contact_body_ids=cb, **common)
        bad = np.nonzero(pre != z["done_" + tag])[0]
        assert bad.size == 0, [(str(z["case"][i]), int(pre[i]), int(z["done_" + tag][i])) for i in bad]
        np.testing.assert_array_equal(fin, z["done_final_" + tag])
        close(fr, z["fail_rates_" + tag], atol=1e-6)
    # the fixture does exercise what it names (flags produced by the reference itself)
    case = [str(c) for c in z["case"]]
    exp = z["expect_fail"]
    for i, c in enumerate(case):
        tag = "done_feet" if c.startswith("fall") else "done_nocontact"
        if exp[i] >= 0:
            assert (z[tag][i] == 1) == bool(exp[i]), c
    assert sum(c.endswith("_over") and c.startswith("body") for c in case) == 14 * 3
    assert z["done_nocontact"][case.index("timeout")] == 3 and z["done_nocontact"][case.index("timeout_exact")] == 3
    assert z["done_nocontact"][case.index("just_before_timeout")] == 0 and z["done_nocontact"][case.index("timeout_and_fail")] == 1
    assert z["motion_end"][case.index("motion_end_clamp_past")] and not z["motion_end"][case.index("motion_end_wrap_past")]
    assert z["done_final_nocontact"][case.index("motion_end_clamp_past")] == 1 and z["done_final_nocontact"][case.index("motion_end_wrap_far_past")] == 0
