"""The motion-generator sub-env on the GPU.

1. Against fixture G21: the REFERENCE's envs/ig_parkour/mgdm_env.py MotionGenDeepMimicEnv driven on CPU for 27 steps (tests/golden/
   gen_golden.py stage mgdm) with a recorded stand-in generator.  Here the generator replays the recorded plans and ASSERTS that it is
   handed the recorded inputs; uniform draws are replayed in call order; the "physics" states are the recorded ones.  Compared after
   every call: reference pose buffers (the fused launch on the sub-env's rows with its own clip library and terrain), target poses,
   ray-fan heights, the generator's height grid, termination codes per branch, replan flags / counters / plan clock, targets, the state
   histories, the re-spawned character states.
2. IGParkourEnv with fraction_dm_envs < 1: both sub-envs in one env, the agent-style loop step -> reset(done ids)."""
import json

import numpy as np
import pytest
import torch

from conftest import golden
from test_hip_parity import DEV, T, close, km  # noqa: F401  (km is a fixture)

pytestmark = pytest.mark.gpu


class ReplayRand:
    def __init__(self, g):
        self.g, self.k = g, 0

    def __call__(self, n):
        r = self.g["rand_%d" % self.k]
        assert r.shape == (n,), ("draw %d" % self.k, r.shape, n)
        self.k += 1
        return T(r)


class ReplayGenerator:
    """returns the recorded plans; checks that the env hands over the recorded inputs"""
    _num_prev_states, _sequence_fps = 2, 30
    _dx = _dy = 0.4
    _num_x_neg, _num_x_pos, _num_y_neg, _num_y_pos = 2, 5, 3, 3

    def __init__(self, g):
        self.g, self.k = g, 0

    def __call__(self, target_xy, prev_frames, terrain, char_model, settings):
        from parc_amd.util.motion_util import MotionFrames
        g, p = self.g, "gen%d_" % self.k
        close(target_xy, g[p + "target"], atol=2e-5, rtol=0)
        close(prev_frames.root_pos, g[p + "prev_root_pos"], atol=2e-5, rtol=0)
        close(prev_frames.root_rot, g[p + "prev_root_rot"], atol=2e-5, rtol=0)
        close(prev_frames.joint_rot, g[p + "prev_joint_rot"], atol=2e-5, rtol=0)
        close(prev_frames.contacts, g[p + "prev_contacts"], atol=0, rtol=0)
        assert np.array_equal(settings.use_prev_state.cpu().numpy(), g[p + "use_prev_state"])
        assert np.array_equal(settings.prev_state_ind_key.cpu().numpy(), g[p + "prev_state_ind_key"])
        assert terrain.hf.shape == g["terrain_hf"].shape
        self.k += 1
        return MotionFrames(root_pos=T(g[p + "out_root_pos"]), root_rot=T(g[p + "out_root_rot"]), joint_rot=T(g[p + "out_joint_rot"]),
                            contacts=T(g[p + "out_contacts"]))


def _g21_env_config(g):
    from parc_amd.envs.ig_parkour.default_config import default_env_config
    cfg = default_env_config()
    e = cfg["env"]
    e.update(json.loads(bytes(g["config_json"]).decode())["env"])
    e.update(tar_obs_steps=[int(v) for v in g["tar_obs_steps"]], termination_height=0.15, episode_length=0.8, pose_termination=True,
             pose_termination_dist=[float(v) for v in g["pose_termination_dist"]], enable_early_termination=True, track_root=True,
             root_pos_termination_dist=0.6, root_rot_termination_angle=1.309, contact_bodies=["right_foot", "left_foot"],
             key_bodies=["right_hand", "left_hand", "right_foot", "left_foot"], fraction_dm_envs=0.0)
    return cfg


def _make_sub_env(g, km):
    from parc_amd.envs.ig_parkour import mgdm_env
    from parc_amd.tracker_core import TrackerConfig, TrackerCore
    from parc_amd.util import terrain_util
    cfg = _g21_env_config(g)
    N = int(g["env_offsets"].shape[0])
    rays = T(g["ray_xy_points"])
    tcfg = TrackerConfig(cfg["env"], km, int(rays.shape[0]))
    assert tcfg.key_body_ids == g["key_body_ids"].tolist()
    core = TrackerCore(N, DEV, km, None, tcfg, rays)
    core.env_offsets[:] = T(g["env_offsets"])
    gen, rnd = ReplayGenerator(g), ReplayRand(g)
    mg = mgdm_env.MotionGenDeepMimicEnv(cfg, N, DEV, False, km, generator=gen, rand_fn=rnd)
    close(mg._mgdm_local_xy_points, g["local_grid"], atol=1e-6, rtol=0)
    mg._terrain = terrain_util.SubTerrain.from_arrays(g["terrain_hf"], g["terrain_min_point"], g["terrain_dxdy"], device=DEV)
    mg._spawn_min_x, mg._spawn_max_x, mg._spawn_min_y, mg._spawn_max_y, mg._oob_region = [float(v) for v in g["spawn"]]
    mg.attach(core, 0)
    return cfg, core, mg, gen, rnd


def _check_snapshot(g, tag, core, mg, gen, rnd, need_refresh):
    for name in ("ref_root_pos", "ref_root_rot", "ref_root_vel", "ref_root_ang_vel", "ref_body_pos", "ref_joint_rot", "ref_dof_pos", "ref_dof_vel",
                 "ref_contacts"):
        tol = 3e-4 if name in ("ref_root_vel", "ref_root_ang_vel", "ref_dof_vel", "ref_joint_rot", "ref_dof_pos", "ref_body_pos") else 3e-5
        close(getattr(core, name), g[tag + "_" + name], atol=tol, rtol=1e-4)
    close(mg._char_root_pos, g[tag + "_char_root_pos"], atol=3e-5, rtol=0)
    close(mg._char_root_rot, g[tag + "_char_root_rot"], atol=3e-5, rtol=0)
    close(mg._char_root_vel, g[tag + "_char_root_vel"], atol=3e-4, rtol=1e-4)
    close(mg._char_dof_pos, g[tag + "_char_dof_pos"], atol=3e-5, rtol=0)
    close(mg._char_dof_vel, g[tag + "_char_dof_vel"], atol=3e-4, rtol=1e-4)
    assert np.array_equal(core.done.cpu().numpy(), g[tag + "_done"]), tag
    assert np.array_equal(core.timestep_buf.cpu().numpy(), g[tag + "_timestep"]), tag
    close(core.time_buf, g[tag + "_time"], atol=1e-6, rtol=0)
    close(core.target_xy, g[tag + "_target_xy"], atol=3e-5, rtol=0)
    close(core.next_target_xy_time, g[tag + "_next_target_time"], atol=1e-5, rtol=0)
    assert np.array_equal(mg._replan_buf.cpu().numpy(), g[tag + "_replan_buf"]), tag
    assert np.array_equal(mg._replan_counter.cpu().numpy(), g[tag + "_replan_counter"]), tag
    close(mg._mgdm_time_buf, g[tag + "_plan_time"], atol=0, rtol=0)
    assert float(mg._plan_time_host) == float(g[tag + "_plan_time"][0])            # the host image of the plan clock is the same fp32 number
    assert int(mg._replan_flag) == int(g[tag + "_replan_flag"][0])
    assert np.array_equal(need_refresh.cpu().numpy(), g[tag + "_need_reset"][:, 0]), tag
    assert rnd.k == int(g[tag + "_num_rand"][0]) and gen.k == int(g[tag + "_num_gen"][0]), (tag, rnd.k, gen.k)
    close(mg._agent_state_hist.root_pos, g[tag + "_agent_hist_root_pos"], atol=3e-5, rtol=0)
    close(mg._agent_state_hist.joint_rot, g[tag + "_agent_hist_joint_rot"], atol=3e-5, rtol=0)
    close(mg._ref_state_hist.root_pos, g[tag + "_ref_hist_root_pos"], atol=3e-5, rtol=0)


def test_g21_sub_env_follows_the_reference_step_by_step(km):
    from parc_amd import _hip
    from parc_amd.envs import base_env
    from parc_amd.util import torch_util
    g = golden("g21_mgdm")
    cfg, core, mg, gen, rnd = _make_sub_env(g, km)
    N, B, D = core.N, km.get_num_joints(), km.get_dof_size()
    dt = 1.0 / 30.0
    tar_steps = torch.tensor(g["tar_obs_steps"], device=DEV)
    rb = core.rigid_body_state.view(N, B, 13)

    def take_need():
        need = mg._need_refresh.clone()
        mg._need_refresh[:] = False
        return need
    mg.replan()                                           # IGParkourEnv construction (ig_parkour_env.py:796-798)
    _check_snapshot(g, "init", core, mg, gen, rnd, take_need())
    mg.reset(torch.arange(N, device=DEV))                 # the agent's first reset: a soft reset of every env
    _check_snapshot(g, "reset0", core, mg, gen, rnd, take_need())
    seen = set()
    for k in range(int(g["num_steps"][0])):
        tag = "s%d" % k
        mg.pre_physics_step()
        # the recorded "physics"
        mg._char_root_pos[:] = T(g[tag + "_in_char_root_pos"])
        mg._char_root_rot[:] = T(g[tag + "_in_char_root_rot"])
        mg._char_root_vel[:] = T(g[tag + "_in_char_root_vel"])
        mg._char_root_ang_vel[:] = T(g[tag + "_in_char_root_ang_vel"])
        mg._char_dof_pos[:] = T(g[tag + "_in_char_dof_pos"])
        mg._char_dof_vel[:] = T(g[tag + "_in_char_dof_vel"])
        bp, br = km.forward_kinematics(mg._char_root_pos.contiguous(), mg._char_root_rot.contiguous(), km.dof_to_rot(mg._char_dof_pos.contiguous()))
        close(bp, g[tag + "_in_char_rigid_body_pos"], atol=3e-5, rtol=0)
        rb[..., 0:3] = T(g[tag + "_in_char_rigid_body_pos"])
        rb[..., 3:7] = br
        mg._char_contact_forces[:] = T(g[tag + "_in_char_contact_forces"])
        core.timestep_buf += 1
        torch.mul(core.timestep_buf, dt, out=core.time_buf)
        mg.update_time(dt)
        mg.update_misc()
        mg._update_ref_motion()
        tar = mg.compute_tar_obs(tar_steps)
        for got, name in zip(tar, ("tar_root_pos", "tar_root_rot", "tar_joint_rot", "tar_key_pos", "tar_contacts")):
            # (joint rotations: slerp between nearly equal frames switches formula at sin(half angle) < 1e-3, a few 1e-4 either side)
            close(got, g[tag + "_" + name], atol=3e-4 if name in ("tar_joint_rot", "tar_key_pos") else 5e-5, rtol=0)
        mg.refresh_obs_hfs(mg._char_root_pos + mg._env_offsets, torch_util.calc_heading(mg._char_root_rot))
        close(mg._mgdm_hfs, g[tag + "_mgdm_hfs"], atol=3e-5, rtol=0)
        close(mg._mgdm_floor_heights, g[tag + "_floor"], atol=0, rtol=0)
        mg._post(_hip.POST_OBS | _hip.POST_HF)                                 # the ray fan is part of the fused launch
        # heights are piecewise constant in the query point: a ray point within rounding of a cell border may read the neighbour
        ray_err = (core.ray_hfs.cpu().numpy() - g[tag + "_ray_hfs"])
        assert (np.abs(ray_err) > 3e-5).mean() < 0.02, (tag, (np.abs(ray_err) > 3e-5).mean())
        mg.update_done()
        want = g[tag + "_done_after_update"]
        assert np.array_equal(core.done.cpu().numpy(), want), (tag, core.done.tolist(), want.tolist())
        assert np.array_equal(mg._replan_buf.cpu().numpy(), g[tag + "_replan_buf_after_update"]), tag
        assert int(mg._replan_flag) == int(g[tag + "_replan_flag_after_update"][0]), tag
        seen |= {(int(d), int(f)) for d, f in zip(want, [int(g[tag + "_replan_flag_after_update"][0])] * N)}
        ids = (core.done != base_env.DoneFlags.NULL.value).nonzero().flatten()
        mg.reset(ids)
        _check_snapshot(g, tag, core, mg, gen, rnd, take_need())
    # the fixture reaches failure, walking off the terrain (TIME without a replan pending), and the replan-time TIME
    assert {(1, 0), (3, 0), (3, 1), (0, 1)} <= seen, seen
    assert rnd.k == int(g["num_rand"][0]) and gen.k == int(g["num_gen"][0]) == 4


class WalkGenerator:
    """A deterministic stand-in planner for end-to-end runs: from the last previous frame, walk towards the target at 1 m/s keeping the
    pose, root height = floor + 0.9."""
    _num_prev_states, _sequence_fps = 2, 30
    _dx = _dy = 0.4
    _num_x_neg, _num_x_pos, _num_y_neg, _num_y_pos = 2, 5, 3, 3
    F = 45
    calls = 0

    def __call__(self, target_xy, prev_frames, terrain, char_model, settings):
        from parc_amd.util import terrain_util
        from parc_amd.util.motion_util import MotionFrames
        WalkGenerator.calls += 1
        F = self.F
        p, q, j = prev_frames.root_pos[:, -1], prev_frames.root_rot[:, -1], prev_frames.joint_rot[:, -1]
        n = p.shape[0]
        d = target_xy[:, 0:2] - p[:, 0:2]
        d = d / torch.linalg.vector_norm(d, dim=-1, keepdim=True).clamp(min=1e-3)
        tt = torch.arange(F, dtype=torch.float32, device=p.device).reshape(1, F, 1) / 30.0
        rp = p.unsqueeze(1).repeat(1, F, 1)
        rp[..., 0:2] += tt * d.unsqueeze(1)
        return MotionFrames(root_pos=rp, root_rot=q.unsqueeze(1).repeat(1, F, 1), joint_rot=j.unsqueeze(1).repeat(1, F, 1, 1),
                            contacts=torch.zeros((n, F, char_model.get_num_joints()), device=p.device))


@pytest.mark.parametrize("fraction,timer", [(0.5, False), (0.0, False), (0.5, True), (0.3, False)])   # 0.3: 19 + 45 rows, nothing aligned
def test_env_with_both_sub_envs(fraction, timer):
    """IGParkourEnv with fraction_dm_envs < 1 through the agent-style loop: dataset rows keep their clips and tile offsets, generator rows
    follow their plans (clip id = row, clip time = plan clock), replans happen on schedule, counters and episode bookkeeping move."""
    from parc_amd import workloads
    from parc_amd.envs import base_env
    N = 64
    WalkGenerator.calls = 0
    mg_cfg = {"plan_length": 0.5, "ddim_stride": 50, "max_replans": 3, "cfg_scale": 0.7, "target_dist_max": 4.0, "target_dist_min": 1.0,
              "target_dur_max": 2.0, "target_dur_min": 1.0, "target_heading_scale": 0.5, "generator": WalkGenerator(),
              "heightmap": {"horizontal_scale": 0.4, "sq_m_per_env": 0.5, "safety_region": 3.0, "num_segments": 6, "platform_heights": [0.0]}}
    env, _, _ = workloads.build_env("boxes_64clips", N, DEV, seed=3, env_overrides={"fraction_dm_envs": fraction, "mgdm": mg_cfg, "enable_replan_timer_obs": timer})
    n_dm = env._num_dm_envs
    assert n_dm == int(fraction * N) and env.has_mgdm_envs() and env.has_dm_envs() == (n_dm > 0)
    assert env.supports_graph_step()                                   # every launch of a step is fixed-shape, generator rows included
    mg = env.get_mgdm_env()
    assert WalkGenerator.calls == 1                                    # plans exist before the first observation
    obs, info = env.reset()
    assert obs.shape == (N, env._cfg.obs_dim + int(timer)) == (N, env.get_obs_space().shape[0]) and torch.isfinite(obs).all()
    assert ("replan_t" in env._compute_obs(ret_obs_shapes=True)) == timer
    low, high = env._action_bound_low, env._action_bound_high
    torch.manual_seed(0)
    replans, total_done = 0, 0
    for it in range(40):
        plan_before = float(mg._plan_time_host)
        a = env._ref_dof_pos.clone()                                   # PD target = the reference pose: a sensible controller
        obs, r, done, info = env.step(torch.minimum(torch.maximum(a, low), high))
        assert torch.isfinite(obs).all() and torch.isfinite(r).all()
        # generator rows: reference pose = their plan at the plan clock
        rp = mg._motion_lib.calc_motion_frame(mg._motion_ids, mg._mgdm_time_buf.expand(N - n_dm))[0]
        assert (env._ref_root_pos[n_dm:] - rp).abs().max() < 1e-4
        assert abs(float(mg._plan_time_host) - (plan_before + 1.0 / 30.0)) < 1e-5
        if timer:                                                       # the plan clock is the last observation column of EVERY row
            assert (obs[:, -1] == mg._mgdm_time_buf[0]).all() and torch.equal(obs[:, :-1], env._core.obs)
        ids = (done != base_env.DoneFlags.NULL.value).nonzero().flatten()
        total_done += int(ids.numel())
        calls = WalkGenerator.calls
        env.reset(ids)
        if WalkGenerator.calls > calls:
            replans += 1
            assert float(mg._plan_time_host) == np.float32(1.0 / 30.0) and not mg._replan_flag
            assert (mg._replan_counter >= 1).all() and (mg._replan_counter <= mg_cfg["max_replans"]).all()
        assert (env._done_buf[ids] == 0).all() if len(ids) else True
    assert replans == 2, replans                                        # plan_length 0.5 s at 30 Hz: after steps 15 and 30
    assert env.get_replan_counter().shape == (N,) and (env.get_replan_counter()[:n_dm] == 0).all()
    assert float(env.get_replan_time_buf()[0]) == float(mg._plan_time_host)
    if n_dm > 0:
        dm = env.get_dm_env()
        assert (env._core.motion_ids[:n_dm] < dm._motion_lib.num_motions()).all()
        assert (env._core.motion_ids[n_dm:] == torch.arange(N - n_dm, device=DEV)).all()
        assert dm._motion_id_fail_rates.shape[0] == dm._motion_lib.num_motions() and torch.isfinite(dm._motion_id_fail_rates).all()


def test_agent_trains_on_an_env_with_generator_rows():
    """The PPO agent on a split env: the rollout step runs as a captured hipGraph like on a pure tracker env (round 3: the generator
    rows' target picks, termination rules and restarts are fixed-shape), it records the replan timer / counter columns the reference's
    agent records for this env (dm_ppo_agent.py:278), and a training iteration goes through."""
    from parc_amd import workloads
    N = 32
    WalkGenerator.calls = 0
    mg_cfg = {"plan_length": 0.2, "ddim_stride": 50, "max_replans": 3, "cfg_scale": 0.7, "target_dist_max": 4.0, "target_dist_min": 1.0,
              "target_dur_max": 2.0, "target_dur_min": 1.0, "target_heading_scale": 0.5, "generator": WalkGenerator(),
              "heightmap": {"horizontal_scale": 0.4, "sq_m_per_env": 0.5, "safety_region": 3.0, "num_segments": 6, "platform_heights": [0.0, 0.4]}}
    env, _, _ = workloads.build_env("boxes_64clips", N, DEV, seed=5, env_overrides={"fraction_dm_envs": 0.5, "mgdm": mg_cfg})
    assert env._enable_replan_timer_obs and env.get_obs_space().shape[0] == env._cfg.obs_dim + 1       # the tracker config's default
    agent = workloads.build_agent(env, DEV, steps_per_iter=16, update_epochs=1, batch_size=2)
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    assert agent._graph_ok()
    w0 = agent._model._actor_layers[0].weight.clone()
    info = agent._train_iter()
    assert len(agent._graphs) >= 1 and agent._use_hip_graph           # captured, not fallen back to eager launches
    assert np.isfinite(info["critic_loss"].item()) and np.isfinite(info["actor_loss"].item())
    assert not torch.equal(w0, agent._model._actor_layers[0].weight)
    assert WalkGenerator.calls >= 3                                     # construction + a replan every 0.2 s = 6 steps
    eb = agent._exp_buffer
    rc = eb.get_data("replan_counter")
    assert rc.shape[-1] == N and (rc[:, :16] == 0).all() and (rc[:, 16:] >= 1).all()
    rt = eb.get_data("replan_timer")
    assert float(rt.max()) <= 0.2 + 2.0 / 30.0 + 1e-6 and float(rt.min()) >= 1.0 / 30.0 - 1e-6


@pytest.mark.parametrize("fraction", [0.5, 0.0])
def test_generator_rows_inside_the_captured_rollout_step(fraction):
    """Round-2 review item: a split env used to drop to eager stepping because the generator sub-env took host decisions inside a step
    (`nonzero` of expired target timers, `is it time to replan`).  Now the agent's rollout step - policy, record, simulator on both
    heightfields, both fused launches, target picks, termination rules, device-side restart of finished rows of BOTH kinds - is one
    captured hipGraph; the only host decision left is the replan, which is a reset, runs eagerly between two graph replays, and is
    predicted from the fp32 mirror of the plan clock (no read-back).  Checked over 3 iterations of 16 steps with a replan every 6 steps:
    the device clock and its host mirror never part, the generator is called exactly on schedule, generator rows follow their plans,
    rows that finish restart (episode counts move, clocks return to zero), counters stay within max_replans."""
    from parc_amd import workloads
    from parc_amd.envs import base_env
    N, T = 64, 16
    WalkGenerator.calls = 0
    mg_cfg = {"plan_length": 0.2, "ddim_stride": 50, "max_replans": 3, "cfg_scale": 0.7, "target_dist_max": 4.0, "target_dist_min": 1.0,
              "target_dur_max": 0.3, "target_dur_min": 0.1, "target_heading_scale": 0.5, "generator": WalkGenerator(),
              "heightmap": {"horizontal_scale": 0.4, "sq_m_per_env": 0.5, "safety_region": 3.0, "num_segments": 6, "platform_heights": [0.0, 0.4]}}
    env, _, _ = workloads.build_env("boxes_64clips", N, DEV, seed=7, env_overrides={"fraction_dm_envs": fraction, "mgdm": mg_cfg})
    n_dm = env._num_dm_envs
    mg = env.get_mgdm_env()
    agent = workloads.build_agent(env, DEV, steps_per_iter=T, update_epochs=1, batch_size=2)
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    calls0 = WalkGenerator.calls
    ep0 = env._ep_num_buf.clone()
    eager_resets = 0
    orig = agent._reset_done_envs

    def counting(done):
        nonlocal eager_resets
        eager_resets += 1
        return orig(done)
    agent._reset_done_envs = counting
    targets0 = env._target_xy[n_dm:].clone()
    for it in range(3):
        agent._train_iter()
        torch.cuda.synchronize()
        assert float(mg._mgdm_time_buf[0]) == float(mg._plan_time_host), (float(mg._mgdm_time_buf[0]), float(mg._plan_time_host))
    steps = 3 * T
    # plan_length 0.2 s at 30 Hz, plans start at one step: the clock passes 0.2 s on the 6th step after a replan -> a replan every 6 steps
    assert WalkGenerator.calls - calls0 == steps // 6, (WalkGenerator.calls - calls0, steps // 6)
    # only the steps that end in a replan (and the first two warm-up steps before the capture) take the eager reset path
    assert eager_resets <= steps // 6 + 2, eager_resets
    assert len(agent._graphs) == 2                                      # the step with and without the device-side restart
    assert torch.isfinite(env._obs_buf).all() and torch.isfinite(env._reward_buf).all()
    rp = mg._motion_lib.calc_motion_frame(mg._motion_ids, mg._mgdm_time_buf.expand(N - n_dm))[0]
    assert (env._ref_root_pos[n_dm:] - rp).abs().max() < 1e-4           # generator rows: reference pose = their plan at the plan clock
    assert not torch.equal(targets0, env._target_xy[n_dm:])              # targets were re-drawn under the mask (timers of 0.1 - 0.3 s)
    # the five exp(-err) reward terms of EVERY row are positive: a row-range launch writes them with the allocation's row length
    # (parc_env_buffers_t.reward_terms_stride), not with its own row count
    assert (env._core.reward_terms[0:5] > 0).all() and (env._core.reward_terms[0:5] <= 1.0).all()
    assert (mg._replan_counter >= 1).all() and (mg._replan_counter <= mg_cfg["max_replans"]).all()
    eb = agent._exp_buffer
    done = eb.get_data("done")
    ts = eb.get_data("timestep")
    epn = eb.get_data("ep_num")
    # a row that finished at step t starts the next step with its clock at zero and one more episode counted - for both kinds of rows,
    # whichever path restarted it (recorded timestep[t + 1] == 1 is the first step of the new episode)
    fin = done[:-1] != base_env.DoneFlags.NULL.value
    assert fin.any()
    assert (ts[1:][fin] == 1).all()
    assert (epn[1:][fin] == epn[:-1][fin] + 1).all() and (epn[1:][~fin] == epn[:-1][~fin]).all()
    assert (env._ep_num_buf >= ep0).all() and (env._ep_num_buf[n_dm:] > ep0[n_dm:]).any()
    rt = eb.get_data("replan_timer")
    assert float(rt.max()) <= 0.2 + 2.0 / 30.0 + 1e-6 and float(rt.min()) >= 1.0 / 30.0 - 1e-6


def test_the_reference_shipped_generator_env_config_builds_and_steps(tmp_path):
    """data/envs/ig_parkour_env.yaml - the one sample env configuration the reference ships for the motion-generator env
    (fraction_dm_envs 0.0, has_target_xy_obs, enable_replan_timer_obs) - as the reference's yaml parses it (fixture G26 holds the tree),
    handed to IGParkourEnv unchanged but for what does not exist here: the character file path (this package's generated copy of the
    same MJCF), the diffusion model (mgdm.model_path -> the stand-in planner as mgdm.generator) and the authors' output locations
    (../Data/terrains/..., output/_motions/... -> the test's scratch directory).  The observation row has
    the reference's segment table for that configuration (G26 'mgdm_shipped': 1315 columns), the two target columns are
    rotate_2d_vec(target_xy - root_xy, -heading) (ig_parkour_env.py:1215-1218) of the env's own state and the last column is the plan
    clock; the agent builds its normaliser index set from the table and trains one iteration on it."""
    import json
    import os
    from parc_amd import workloads
    from parc_amd.assets import humanoid_spec
    from parc_amd.envs.ig_parkour.ig_parkour_env import IGParkourEnv
    from parc_amd.util import torch_util
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g26_obs_variants.json")))
    tree = g["shipped_configs"]["data/envs/ig_parkour_env.yaml"]
    assert tree["env"]["has_target_xy_obs"] is True and tree["env"]["fraction_dm_envs"] == 0.0
    tree["env"]["char_file"] = humanoid_spec.write_mjcf()
    tree["env"]["mgdm"]["generator"] = WalkGenerator()
    tree["env"]["mgdm"]["terrain_save_path"] = str(tmp_path / "mgdm_terrain.pkl")
    tree["env"]["dm"]["terrain_save_path"] = str(tmp_path / "dm_terrain.pkl")
    tree["env"]["output_motion_dir"] = str(tmp_path / "recorded")
    N = 32
    torch.manual_seed(1)
    env = IGParkourEnv(tree, N, DEV, False)
    assert not env.has_dm_envs() and env.has_mgdm_envs()
    table = g["variants"]["mgdm_shipped"]
    shapes = env._compute_obs(ret_obs_shapes=True)
    assert [[k, v["use_normalizer"], list(v["shape"])] for k, v in shapes.items()] == table["obs_shapes"]
    obs, info = env.reset()
    assert obs.shape == (N, 1315) == (N, table["obs_dim"]) == (N, env.get_obs_space().shape[0])
    low, high = env._action_bound_low, env._action_bound_high
    for _ in range(5):
        obs, r, done, info = env.step(torch.minimum(torch.maximum(env._ref_dof_pos.clone(), low), high))
        assert torch.isfinite(obs).all() and torch.isfinite(r).all()
        assert torch.equal(obs[:, :1312], env._core.obs)
        heading = torch_util.calc_heading(env._char_root_rot)
        want = torch_util.rotate_2d_vec(env._target_xy - env._char_root_pos[:, 0:2], -heading)
        assert (obs[:, 1312:1314] - want).abs().max() < 3e-5 and want.abs().max() > 0.1
        assert (obs[:, 1314] == env.get_mgdm_env()._mgdm_time_buf[0]).all()
        env.reset((done != 0).nonzero().flatten())
    agent = workloads.build_agent(env, DEV, steps_per_iter=4, update_epochs=1, batch_size=2)
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    tinfo = agent._train_iter()
    assert np.isfinite(tinfo["critic_loss"].item())


def test_recording_with_generator_rows(tmp_path):
    """IGParkourEnv.write_agent_states on an env with both kinds of rows (ig_parkour_env.py:957-995 records whatever rows exist): a
    generator row records from the call until it FAILS - across time-outs, resets and replans - and is then written under the default
    name with a slice of the GENERATOR's terrain and its frames unshifted; a dataset row ends with its clip as before.  Checked against
    the reference's own scheme kept on the host (one list per env, appended every step), with the recorder's buffers forced to grow."""
    from parc_amd import workloads
    from parc_amd.envs import base_env
    from parc_amd.util import safe_pickle
    N = 32
    mg_cfg = {"plan_length": 0.5, "ddim_stride": 50, "max_replans": 3, "cfg_scale": 0.7, "target_dist_max": 4.0, "target_dist_min": 1.0,
              "target_dur_max": 2.0, "target_dur_min": 1.0, "target_heading_scale": 0.5, "generator": WalkGenerator(),
              "heightmap": {"horizontal_scale": 0.4, "sq_m_per_env": 0.5, "safety_region": 3.0, "num_segments": 6, "platform_heights": [0.0]}}
    env, _, _ = workloads.build_env("boxes_64clips", N, DEV, seed=5, env_overrides={"fraction_dm_envs": 0.5, "mgdm": mg_cfg,
                                                                                     "enable_replan_timer_obs": True, "has_target_xy_obs": True})
    n_dm = env._num_dm_envs
    env._output_motion_dir = str(tmp_path)
    env._bypass_record_fail = True
    obs, info = env.reset()
    assert obs.shape == (N, 1315)
    env.build_agent_states_dict("_rec", record_obs=True)
    env._rec_cap = 4                                   # the allocation has more rows; the next steps force _grow_recorder twice
    env.write_agent_states()                           # the frame at reset, like DMPPOAgent.record_motions (dm_ppo_agent.py:435-436)
    lists = [{"frames": [], "contacts": [], "obs": [], "on": True} for _ in range(N)]

    def host_append():
        fr, co = env._get_char_state_all()
        for e in range(N):
            if lists[e]["on"]:
                lists[e]["frames"].append(fr[e].cpu().numpy())
                lists[e]["contacts"].append(co[e].cpu().numpy())
                lists[e]["obs"].append(env._obs_buf[e].cpu().numpy())
    host_append()
    torch.manual_seed(3)
    low, high = env._action_bound_low, env._action_bound_high
    written = set()
    for it in range(60):
        a = low + (high - low) * torch.rand((N, 28), device=DEV)          # flailing: everybody falls within a second or two
        obs, r, done, info = env.step(a)                                   # (step() records, the flag being set)
        host_append()
        for e in (done == base_env.DoneFlags.FAIL.value).nonzero().flatten().tolist():
            if lists[e]["on"]:
                lists[e]["on"] = False
                written.add(e)
        env.reset((done != 0).nonzero().flatten())
        if not env.is_writing_agent_states():
            break
    assert env._rec_cap >= 16                                              # grew from 4
    gen_rows = sorted(e for e in written if e >= n_dm)
    assert len(gen_rows) >= 4, gen_rows
    for e in gen_rows:
        d = safe_pickle.load_motion_file_safe(str(tmp_path / ("dm_motion_" + str(e).zfill(3) + ".pkl")))
        T_ = len(lists[e]["frames"])
        assert d["frames"].shape == (T_, 34) and d["contacts"].shape == (T_, 15) and d["obs"].shape == (T_, 1315)
        assert list(d["obs_shapes"])[-2:] == ["target_xy", "replan_t"]
        np.testing.assert_array_equal(d["contacts"], np.stack(lists[e]["contacts"]))
        np.testing.assert_array_equal(d["obs"], np.stack(lists[e]["obs"]))
        want = np.stack(lists[e]["frames"])
        np.testing.assert_array_equal(d["frames"][:, 2:], want[:, 2:])      # everything but xy is untouched
        # xy: slice_terrain_around_motion re-bases the clip on the sliced terrain; differences between frames are preserved
        np.testing.assert_allclose(np.diff(d["frames"][:, 0:2], axis=0), np.diff(want[:, 0:2], axis=0), atol=2e-6)
        assert d["terrain"]["__class__"] == "util.terrain_util.SubTerrain"
    if any(e < n_dm for e in written):                                     # dataset rows still end up under their clip's name + suffix
        assert len(list(tmp_path.glob("*_rec.pkl"))) >= 1
