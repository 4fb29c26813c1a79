"""Drop-in surface of the environment and agent on the GPU: shapes, dtypes, aliasing conventions,
reset semantics, kinematic-replay upper bound of the reward, checkpoint key names, one training iteration."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def env():
    from parc_amd import workloads
    torch.manual_seed(0)
    e, clips, tiled = workloads.build_env("boxes_64clips", 96, DEV, seed=0)
    return e


def test_spaces_and_buffers(env):
    assert env.NAME == "ig_parkour" and env.get_num_envs() == 96
    o, a = env.get_obs_space(), env.get_action_space()
    assert tuple(o.shape) == (1312,) and tuple(a.shape) == (28,)
    assert np.all(a.high > a.low)
    # G13: PD action bounds derived from the MJCF ranges with the reference's formula (ig_char_env.py:308-348)
    assert abs(a.high[9] - (0.5 * np.deg2rad(160) + 0.7 * np.deg2rad(160))) < 1e-5      # right elbow hinge 0..160 deg
    assert abs(a.high[0] - 1.2 * np.deg2rad(90)) < 1e-5                                 # abdomen spherical: 1.2 * max |limit|
    assert env.get_reward_bounds() == (0.0, 1.0) and env.get_reward_fail() == 0.0
    shapes = env._compute_obs(ret_obs_shapes=True)
    assert list(shapes.keys()) == ["char_obs", "tar_obs", "tar_contacts", "char_contacts", "hf"]
    assert [v["use_normalizer"] for v in shapes.values()] == [True, True, False, False, False]


def test_reset_and_step_semantics(env):
    obs, info = env.reset()
    assert obs.data_ptr() == env._obs_buf.data_ptr() and obs.shape == (96, 1312) and obs.dtype == torch.float32
    assert torch.isfinite(obs).all()
    assert set(["timestep", "ep_num", "compute_time", "char_contact_forces"]) <= set(info.keys())
    ep0 = env._ep_num_buf.clone()
    # at reset the simulated character sits on the reference pose (+ xy noise <= 0.075 m)
    d = (env._char_root_pos - env._ref_root_pos).abs()
    assert d[:, 0:2].max() <= 0.0751 and d[:, 2].max() < 1e-6
    assert torch.allclose(env._char_dof_pos, env._ref_dof_pos)
    # empty id list: no-op (not the same as None)
    before = obs.clone()
    env.reset(torch.zeros(0, dtype=torch.long, device=DEV))
    assert torch.equal(env._obs_buf, before) and torch.equal(env._ep_num_buf, ep0)
    act = torch.zeros((96, 28), device=DEV)
    o2, r, done, info = env.step(act)
    assert o2.data_ptr() == env._obs_buf.data_ptr()
    assert r.shape == (96,) and r.dtype == torch.float32 and done.dtype == torch.int32
    assert torch.all(env._timestep_buf == 1) and torch.allclose(env._time_buf, torch.full((96,), 1.0 / 30.0, device=DEV))
    assert set(info["rewards"].keys()) == {"pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty", "task_r1",
                                           "task_r2", "total_task_r", "total_r"}
    assert info["char_contact_forces"].shape == (96, 15, 3) and info["char_contact_forces"].data_ptr() != env._char_contact_forces.data_ptr()
    assert torch.isfinite(r).all() and r.min() >= -1.7 and r.max() <= 1.7   # 5 exp kernels in [0,1] + contact term in [-5/3, 5/3]*mean
    assert set(done.unique().tolist()) <= {0, 1, 3}
    # subset reset touches only those rows
    ids = torch.tensor([5, 17], device=DEV)
    snap = env._obs_buf.clone()
    env.reset(ids)
    same = torch.ones(96, dtype=torch.bool, device=DEV)
    same[ids] = False
    assert torch.equal(env._obs_buf[same], snap[same]) and not torch.equal(env._obs_buf[ids], snap[ids])
    assert torch.all(env._timestep_buf[ids] == 0) and torch.all(env._ep_num_buf[ids] == ep0[ids] + 1)


def test_kinematic_replay_reward_upper_bound(env):
    """A character teleported onto the reference pose gets reward 1 (all five exp kernels at 1, zero contact term)."""
    from parc_amd import _hip
    env.reset()
    c = env._core
    c.time_buf += 0.5
    c.post_step(_hip.POST_REF)
    env._char_root_pos[:] = c.ref_root_pos
    env._char_root_rot[:] = c.ref_root_rot
    env._char_root_vel[:] = c.ref_root_vel
    env._char_root_ang_vel[:] = c.ref_root_ang_vel
    env._char_dof_pos[:] = c.ref_dof_pos
    env._char_dof_vel[:] = c.ref_dof_vel
    _hip.check(_hip.lib().parc_sim_refresh_bodies(_hip.stream(), env._sim_model.device_ptr(DEV), 96, _hip.c_vp(0), 0, _hip.ptr(c.root_state),
                                                  _hip.ptr(c.dof_state), _hip.ptr(c.rigid_body_state), _hip.ptr(c.contact_forces)), "refresh")
    c.post_step(_hip.POST_REF | _hip.POST_REWARD_DONE)
    assert torch.allclose(c.reward, torch.ones(96, device=DEV), atol=2e-3)
    assert torch.all(c.done[c.time_buf + c.motion_time_offsets < env._dm_env._motion_lib._motion_lengths[c.motion_ids]] == 0)
    c.time_buf -= 0.5


def test_long_rollout_stays_finite(env):
    env.reset()
    g = torch.Generator(device=DEV).manual_seed(3)
    lo, hi = env._action_bound_low, env._action_bound_high
    n_done = 0
    for _ in range(150):
        a = lo + (hi - lo) * torch.rand((96, 28), device=DEV, generator=g)
        obs, r, done, info = env.step(a)
        ids = (done != 0).nonzero().flatten()
        n_done += int(ids.numel())
        env.reset(ids)
    assert torch.isfinite(obs).all() and torch.isfinite(env._root_state).all() and torch.isfinite(env._dof_state).all()
    assert n_done > 0
    fr = env._dm_env._motion_id_fail_rates
    assert torch.all((fr > 0) & (fr <= 1.0))


def test_agent_checkpoint_keys_and_training_iteration(env, tmp_path):
    from parc_amd import workloads
    agent = workloads.build_agent(env, DEV, steps_per_iter=4, update_epochs=2, batch_size=2)
    keys = set(agent.state_dict().keys())
    expect = {"_model._actor_layers.0.weight", "_model._actor_layers.2.bias", "_model._actor_layers.4.weight",
              "_model._action_dist._mean_net.weight", "_model._action_dist._logstd_net", "_model._critic_layers.0.weight",
              "_model._critic_out.bias", "_obs_norm._count", "_obs_norm._mean", "_obs_norm._std", "_a_norm._mean", "_a_norm._std"}
    assert expect <= keys
    assert agent.calc_num_params() == 10638877             # the 42.6 MB flat fp32 gradient of SURVEY.md 2.2
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    w0 = agent._model._actor_layers[0].weight.clone()
    info = agent._train_iter()
    assert np.isfinite(info["critic_loss"].item()) and np.isfinite(info["actor_loss"].item())
    assert not torch.equal(w0, agent._model._actor_layers[0].weight)
    assert agent._obs_norm._count.item() == 4 * 96
    # non-normalised segments keep mean 0 / std 1 (contacts + heightmap = 546 dims)
    assert torch.all(agent._obs_norm._mean[766:] == 0) and torch.all(agent._obs_norm._std[766:] == 1)
    p = tmp_path / "model.pt"
    agent.save(str(p))
    agent.load(str(p))
    res = agent.test_model(2)
    assert np.isfinite(res["mean_return"]) and res["num_eps"] >= 96


def test_graph_rollout_matches_eager_bookkeeping(env):
    """The captured rollout step (hipGraph) must leave the same bookkeeping as the eager step: contiguous transitions in
    the experience buffer, timestep/episode counters, normaliser sample counts and return-tracker episode counts."""
    from parc_amd import workloads
    T = 12
    agent = workloads.build_agent(env, DEV, steps_per_iter=T, update_epochs=1, batch_size=2)
    assert agent._use_hip_graph
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    agent._rollout_train(T)
    torch.cuda.synchronize()
    assert len(agent._graphs) == 1                     # 2 eager warm-up steps, then capture + replays
    eb = agent._exp_buffer
    obs, nobs, done = eb.get_data("obs"), eb.get_data("next_obs"), eb.get_data("done")
    ts, epn, r = eb.get_data("timestep"), eb.get_data("ep_num"), eb.get_data("reward")
    assert torch.isfinite(obs).all() and torch.isfinite(nobs).all() and torch.isfinite(r).all()
    for t in range(T - 1):
        cont = done[t] == 0
        assert torch.equal(nobs[t][cont], obs[t + 1][cont])                 # graph steps included (t >= 2)
        assert torch.equal(ts[t + 1][cont], ts[t][cont] + 1)
        assert torch.all(ts[t + 1][~cont] == 1)                             # reset -> first step of the new episode
        assert torch.equal(epn[t + 1][~cont], epn[t][~cont] + 1)
    assert torch.all(eb.get_data("env_id") == torch.arange(96, device=DEV))
    assert torch.all(eb.get_data("compute_time")[T - 1] > 0)
    n_done = int((done != 0).sum().item())
    assert agent._train_return_tracker.get_episodes() == n_done
    assert agent._obs_norm._new_count == T * 96
    assert torch.allclose(agent._obs_norm._new_sum, obs.sum(dim=(0, 1)), rtol=1e-4, atol=1e-2)
    # prev/next contact forces chain like the observations
    pcf, ncf = eb.get_data("prev_char_contact_forces"), eb.get_data("next_char_contact_forces")
    for t in range(2, T - 1):
        cont = done[t] == 0
        assert torch.equal(ncf[t][cont], pcf[t + 1][cont])
    # V(next_obs) assembled from V(obs[t+1]) + a critic pass on finished envs / the last step == the direct second pass
    with torch.no_grad():
        vals, nv = agent._critic_values()
        direct = agent._model.eval_critic(agent._obs_norm.normalize(nobs)).squeeze(-1)
        direct_v = agent._model.eval_critic(agent._obs_norm.normalize(obs)).squeeze(-1)
    assert torch.allclose(nv, direct, rtol=1e-4, atol=1e-4) and torch.allclose(vals, direct_v, rtol=1e-4, atol=1e-4)
    assert torch.equal(eb.get_data("norm_obs"), agent._obs_norm.normalize(obs))
    # the captured graph survives a full training iteration (update changes the weights in place)
    info = agent._train_iter()
    assert np.isfinite(info["mean_return"]) and len(agent._graphs) == 1


def test_annealed_exploration_probability_reaches_the_captured_step(env):
    """ppo_agent.py:97-99: the exploration probability anneals with the sample count.  The captured rollout step must draw its
    Bernoulli mask with the CURRENT probability, not the one of the step it was captured on."""
    from parc_amd import workloads
    T = 16
    agent = workloads.build_agent(env, DEV, steps_per_iter=T, update_epochs=1, batch_size=2, exp_prob_beg=0.9, exp_prob_end=0.1,
                                  exp_anneal_samples=1000.0)
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    agent._sample_count = 0                     # exp_prob 0.9: two eager steps, capture, replays
    agent._rollout_train(T)
    agent._exp_buffer.reset()
    torch.cuda.synchronize()
    assert len(agent._graphs) == 1
    m_hi = agent._exp_buffer.get_data("rand_action_mask")[4:T].mean().item()
    agent._sample_count = 1000                  # annealed to 0.1: same graph key (< 1), replays only
    agent._rollout_train(T)
    torch.cuda.synchronize()
    assert len(agent._graphs) == 1
    m_lo = agent._exp_buffer.get_data("rand_action_mask").mean().item()
    n = 96 * (T - 4)
    assert abs(m_hi - 0.9) < 5 * (0.09 / n) ** 0.5 + 1e-3, m_hi
    assert abs(m_lo - 0.1) < 5 * (0.09 / (96 * T)) ** 0.5 + 1e-3, m_lo
    # masked-out samples carry the mode: their action equals the un-normalised mean, so the log-probability is the maximum
    lp = agent._exp_buffer.get_data("a_logp")
    mk = agent._exp_buffer.get_data("rand_action_mask")
    assert torch.all(lp[mk == 0] >= lp.max() - 1e-4)


def _snapshot(env):
    c = env._core
    names = ["root_state", "dof_state", "rigid_body_state", "contact_forces", "motion_ids", "motion_terrain_ids", "motion_time_offsets",
             "motion_xy_offset", "time_buf", "timestep_buf", "done", "obs", "ref_root_pos", "ref_root_rot", "ref_joint_rot", "ref_dof_pos",
             "ref_dof_vel", "ref_body_pos", "ref_contacts", "next_target_xy_time", "target_xy"]
    d = {n: getattr(c, n).clone() for n in names}
    d["ep_num"] = env._ep_num_buf.clone()
    return d


def _restore(env, snap):
    c = env._core
    for n, v in snap.items():
        (env._ep_num_buf if n == "ep_num" else getattr(c, n)).copy_(v)


def test_device_reset_equals_indexed_reset(env):
    """reset_done(done) (masked kernels, no host sync) must leave exactly the state reset(nonzero(done)) leaves when both
    draw the same clip / start time: demo mode + fixed start fraction + no root noise make the draw deterministic."""
    dm = env._dm_env
    env.reset()
    act = torch.zeros((96, 28), device=DEV)
    for _ in range(3):
        env.step(act)
    old = (dm._demo_mode, dm._rand_reset, dm._rand_root_pos_offset_scale)
    dm._demo_mode, dm._rand_reset, dm._rand_root_pos_offset_scale = True, False, 0.0
    dm._motion_start_time_fraction[:] = torch.linspace(0.05, 0.6, 96, device=DEV)
    try:
        done = torch.zeros(96, dtype=torch.int32, device=DEV)
        done[[0, 3, 4, 5, 6, 7, 31, 64, 95]] = torch.tensor([1, 1, 3, 1, 1, 2, 1, 3, 1], dtype=torch.int32, device=DEV)
        env._done_buf.copy_(done)
        snap = _snapshot(env)
        torch.manual_seed(5)
        env.reset(done.nonzero().flatten())
        a = _snapshot(env)
        _restore(env, snap)
        torch.manual_seed(5)
        assert env.supports_device_reset()
        env.reset_done(env._done_buf)
        b = _snapshot(env)
    finally:
        dm._demo_mode, dm._rand_reset, dm._rand_root_pos_offset_scale = old
        dm._motion_start_time_fraction.zero_()
    ids = done.nonzero().flatten()
    assert torch.all(b["timestep_buf"][ids] == 0) and torch.all(b["done"][ids] == 0) and torch.equal(b["ep_num"][ids], snap["ep_num"][ids] + 1)
    for n in a:
        if n in ("target_xy", "next_target_xy_time"):       # drawn from the RNG in a different order
            continue
        assert torch.equal(a[n], b[n]), n
    # both paths re-arm the xy target of the reset envs (drawn from different random streams, hence not compared above)
    for snap_ in (a, b):
        assert torch.all(snap_["next_target_xy_time"][ids] > 0.0) and torch.isfinite(snap_["target_xy"][ids]).all()
        assert torch.equal(snap_["next_target_xy_time"][done == 0], snap["next_target_xy_time"][done == 0])
    keep = done == 0
    for n in ("root_state", "dof_state", "obs", "motion_ids", "time_buf", "ref_root_pos"):
        assert torch.equal(b[n].reshape(96, -1)[keep], snap[n].reshape(96, -1)[keep]), n


def test_fused_linear_relu_matches_plain_layers():
    """FusedMLP (GEMM with bias + ReLU epilogue, explicit backward) against the plain Linear/ReLU stack it replaces."""
    from parc_amd.learning import dm_ppo_model as M
    torch.manual_seed(3)
    net, h = M.build_net("fc_3layers_2048units", 1312, torch.nn.ReLU)
    net = net.to(DEV)
    for p in net.parameters():
        if p.dim() == 1:
            torch.nn.init.normal_(p, std=0.1)
    x = torch.randn(512, 1312, device=DEV)
    y = net(x)
    y_ref = torch.nn.Sequential.forward(net, x)
    assert y.shape == (512, 512)
    assert torch.allclose(y, y_ref, rtol=1e-4, atol=1e-4)
    g = torch.autograd.grad(y.square().sum(), list(net.parameters()))
    g_ref = torch.autograd.grad(y_ref.square().sum(), list(net.parameters()))
    for a, b in zip(g, g_ref):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max())), float((a - b).abs().max())
    with torch.no_grad():
        assert torch.allclose(net(x), y_ref, rtol=1e-4, atol=1e-4)


def test_parkour_terrain_workload_rollout_and_record(tmp_path):
    """BASELINE configs[4] shape at test size: stairs / curvy paths / boxes tiles from the reference's generators, a short
    training rollout (graph path, device reset) and the deterministic record rollout of parc_4_phys_record."""
    from parc_amd import workloads
    from parc_amd.util import safe_pickle
    torch.manual_seed(1)
    e, clips, tiled = workloads.build_env("parkour_32clips", 64, DEV, seed=0)
    assert tiled[0].shape == (6 * 34, 6 * 34) and float(tiled[0].max()) > 1.0
    agent = workloads.build_agent(e, DEV, steps_per_iter=8, update_epochs=1, batch_size=2)
    agent._curr_obs, agent._curr_info = e.reset()
    agent._init_train()
    info = agent._train_iter()
    assert np.isfinite(info["mean_return"]) and torch.isfinite(e._obs_buf).all()
    hfcols = e._obs_buf[:, 871:]
    assert float(hfcols.min()) >= -3.0 and float(hfcols.max()) <= 3.0 and float(hfcols.std()) > 0.05    # the fans see relief
    # record mode: deterministic policy, one clip per env, files in the reference's motion format
    e._output_motion_dir = str(tmp_path / "recorded")
    ok = agent.record_motions(max_steps=40)
    files = sorted((tmp_path / "recorded").glob("*.pkl"))
    assert len(ok) == 64 or ok is not None
    if files:
        d = safe_pickle.load_motion_file_safe(str(files[0]))
        assert d["frames"].shape[1] == 34 and d["contacts"].shape[1] == 15 and d["fps"] == 30


def test_in_kernel_target_resample_matches_rule(env):
    """PARC_POST_TARGETS: envs whose timer expired take target = clip root xy at (clip time + U[min, max]) + N(0, 0.05) and a
    new timer; the others keep both.  Checked against the rule evaluated with torch on the same uniforms."""
    from parc_amd import _hip
    c, dm = env._core, env._dm_env
    env.reset()
    act = torch.zeros((96, 28), device=DEV)
    for _ in range(2):
        env.step(act)
    g = torch.Generator(device="cpu").manual_seed(3)
    u = torch.rand((96, 3), generator=g).to(DEV)
    c.target_rand.copy_(u)
    c.next_target_xy_time[:] = torch.where(torch.arange(96, device=DEV) % 3 == 0, torch.zeros(96, device=DEV), torch.full((96,), 1e9, device=DEV))
    old_xy, old_t = c.target_xy.clone(), c.next_target_xy_time.clone()
    c.post_step(_hip.POST_OBS | _hip.POST_TARGETS)
    torch.cuda.synchronize()
    due = old_t <= c.time_buf
    assert due.sum().item() == 32
    fut = u[:, 0] * (dm._target_xy_future_time_max - dm._target_xy_future_time_min) + dm._target_xy_future_time_min
    root = dm._motion_lib.calc_motion_frame(c.motion_ids, c.time_buf + c.motion_time_offsets + fut)[0]
    rr = 0.05 * torch.sqrt(-2.0 * torch.log(1.0 - u[:, 1]))
    noise = torch.stack([rr * torch.cos(2 * np.pi * u[:, 2]), rr * torch.sin(2 * np.pi * u[:, 2])], dim=-1)
    want = root[:, 0:2] + c.motion_xy_offset - c.env_offsets[:, 0:2] + noise
    assert torch.allclose(c.target_xy[due], want[due], atol=2e-5)
    assert torch.allclose(c.next_target_xy_time[due], (c.time_buf + fut)[due], atol=1e-6)
    assert torch.equal(c.target_xy[~due], old_xy[~due]) and torch.equal(c.next_target_xy_time[~due], old_t[~due])
    assert float(noise.std()) > 0.03 and float(noise.std()) < 0.07


def test_return_tracker_kernel_matches_torch_rule():
    """K21: parc_return_tracker_update against the torch statement of DMPPOReturnTracker.update over a random episode stream."""
    from parc_amd.learning.dm_ppo_return_tracker import DMPPOReturnTracker
    g = torch.Generator().manual_seed(5)
    N = 1000
    a, b = DMPPOReturnTracker(N, DEV, target_task=True), DMPPOReturnTracker(N, DEV, target_task=True)
    b._use_kernel = False
    names = ["total_r", "pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty", "task_r1", "task_r2", "total_task_r"]
    for step in range(40):
        blk = torch.rand((10, N), generator=g).to(DEV)
        done = (torch.rand(N, generator=g) < (0.0 if step in (0, 7) else 0.06)).to(torch.int32).to(DEV) * (1 + step % 3)
        info = {"rewards_all": (names, blk)}
        a.update(info, done)
        b.update(info, done)
    torch.cuda.synchronize()
    assert a.get_episodes() == b.get_episodes() and a.get_episodes() > 1000
    assert torch.equal(a._ep_len_buf, b._ep_len_buf) and torch.equal(a.get_eps_per_env(), b.get_eps_per_env())
    assert torch.allclose(a._return_buf, b._return_buf, rtol=1e-6, atol=1e-6)
    assert torch.allclose(a._mean_return, b._mean_return, rtol=2e-5) and torch.allclose(a.get_mean_ep_len(), b.get_mean_ep_len(), rtol=2e-5)
    assert abs(a.summary()["mean_return"] - b.summary()["mean_return"]) < 1e-4


def test_create_dataset_preprocessing_writes_mask_inds(tmp_path):
    """compute_preprocessing_data: every motion file gains hf_mask_inds (one cell list per frame) and the updated terrain, and a
    second run leaves the files alone (PARC/util/create_dataset.py:147-160)."""
    import pickle
    from parc_amd.util import create_dataset, safe_pickle, terrain_util
    g = golden("g13_terrain_geometry")
    cls_dir = tmp_path / "data" / "running"
    cls_dir.mkdir(parents=True)
    ter = terrain_util.SubTerrain.from_arrays(g["civ_hf"], g["civ_min_point"], g["civ_dxdy"])
    for k in range(2):
        terrain_util.dump_reference_pickle({"fps": 30, "loop_mode": "CLAMP", "frames": g["extra_frames"][10 * k:10 * k + 30], "terrain": ter.numpy_copy()},
                                           str(cls_dir / "clip_{}.pkl".format(k)))
    out = tmp_path / "dataset.yaml"
    create_dataset.create_dataset_yaml([tmp_path / "data"], out, compute_preprocessing_data=True, max_terrain_dim_x=64, max_terrain_dim_y=64)
    d = safe_pickle.load_motion_file_safe(str(cls_dir / "clip_0.pkl"))
    assert len(d["hf_mask_inds"]) == 30 and d["frames"].shape == (30, 34)
    assert np.asarray(d["terrain"]["hf_mask"]).sum() > 0
    before = (cls_dir / "clip_0.pkl").read_bytes()
    create_dataset.create_dataset_yaml([tmp_path / "data"], out, compute_preprocessing_data=True, max_terrain_dim_x=64, max_terrain_dim_y=64)
    assert (cls_dir / "clip_0.pkl").read_bytes() == before


def test_episode_length_attribute_reaches_the_kernels_and_rekeys_graphs():
    """env._episode_length is written by callers (record_motions): the kernels' config follows, and the agent's captured rollout
    graphs are keyed by the host-side step parameters so a stale graph is never replayed."""
    from parc_amd import workloads
    from parc_amd.envs.base_env import DoneFlags
    torch.manual_seed(0)
    e, _, _ = workloads.build_env("boxes_64clips", 64, DEV, seed=0)
    sig0 = e.host_step_signature()
    e.reset()
    e._episode_length = 0.01
    assert abs(e._cfg.struct.episode_length - 0.01) < 1e-9 and e.host_step_signature() != sig0
    a = torch.zeros((64, 28), device=DEV)
    _, _, done, _ = e.step(a)
    assert bool((done != DoneFlags.NULL.value).all())             # 1/30 s of env time is already past the limit
    e.set_rand_reset(False)
    assert e.host_step_signature()[1] is False


def test_fused_reset_sampler_distribution_and_bookkeeping():
    """parc_reset_sample_apply: only finished envs change; the clip draw follows weight * max(fail rate, floor), the tile draw is
    uniform, the start time lies inside the clip, the xy noise inside +-scale; counters are reset like reset(env_ids) does."""
    from parc_amd import workloads
    torch.manual_seed(1)
    e, _, _ = workloads.build_env("boxes_64clips", 4096, DEV, seed=0)
    e.reset()
    dm, c = e._dm_env, e._core
    M = dm._motion_lib.num_motions()
    fr = torch.rand(M, device=DEV)
    fr[:8] = 0.0                                   # floored at min_motion_weight
    dm._motion_id_fail_rates.copy_(fr)
    a = torch.zeros((4096, 28), device=DEV)
    e.step(a)
    done = torch.zeros(4096, dtype=torch.int32, device=DEV)
    done[::2] = 1
    before = {k: getattr(c, k).clone() for k in ("motion_ids", "motion_time_offsets", "root_state", "obs")}
    ep0 = e._ep_num_buf.clone()
    counts = torch.zeros(M, device=DEV)
    for it in range(30):
        e._done_buf.copy_(done)
        c.rand_pool_fresh = False
        e.reset_done(e._done_buf)
        counts += torch.bincount(c.motion_ids[::2], minlength=M).float()
        if it == 0:
            keep = done == 0
            for k, v in before.items():
                assert torch.equal(getattr(c, k)[keep], v[keep]), k
            assert torch.equal(e._ep_num_buf[keep], ep0[keep]) and torch.equal(e._ep_num_buf[~keep], ep0[~keep] + 1)
            assert torch.all(c.timestep_buf[~keep] == 0) and torch.all(c.done == 0) and torch.all(c.reset_mask == done)
            lens = dm._motion_lib._motion_lengths[c.motion_ids]
            assert torch.all(c.motion_time_offsets[~keep] >= 0) and torch.all(c.motion_time_offsets[~keep] < lens[~keep] + 1e-6)
            assert torch.all(c.init_noise_xy[~keep].abs() <= dm._rand_root_pos_offset_scale + 1e-7)
            assert torch.all((c.motion_terrain_ids >= 0) & (c.motion_terrain_ids < dm._terrains_per_motion))
            assert torch.isfinite(c.obs).all()
    w = torch.clamp(fr, min=dm._min_motion_weight) * dm._motion_lib._motion_weights
    p = (w / w.sum()).cpu().numpy()
    n = float(counts.sum())
    obs_p = counts.cpu().numpy() / n
    sigma = np.sqrt(p * (1 - p) / n)
    assert np.all(np.abs(obs_p - p) < 5 * sigma + 1e-4), float(np.max(np.abs(obs_p - p) / (sigma + 1e-9)))


def test_step_clock_rides_in_the_simulator_launch(env):
    """IGEnv._update_time inside parc_sim_step_tick: timestep += 1, time = timestep * dt (fp32)."""
    env.reset()
    a = torch.zeros((96, 28), device=DEV)
    ts0 = env._timestep_buf.clone()
    env.step(a)
    assert torch.equal(env._timestep_buf, ts0 + 1)
    assert torch.equal(env._time_buf, env._timestep_buf.to(torch.float32) * torch.tensor(env._timestep, dtype=torch.float32, device=DEV))


def test_device_recorder_equals_per_step_host_lists(env, tmp_path):
    """IGParkourEnv.write_agent_states keeps the recorded clips in device buffers with a per-env write row; the result must be
    what the reference's scheme produces: per env, the state of every step up to and including the one on which it failed."""
    from parc_amd.envs.base_env import DoneFlags
    env._output_motion_dir = str(tmp_path)
    old_bypass = env._bypass_record_fail
    env._bypass_record_fail = False
    try:
        torch.manual_seed(4)
        env.reset()
        env.build_agent_states_dict("_t", record_obs=True)
        ref = [{"frames": [], "contacts": [], "obs": [], "on": True} for _ in range(96)]

        def host_side():
            f, c = env._get_char_state_all()
            f, c, o, d = f.cpu().numpy(), c.cpu().numpy(), env._obs_buf.cpu().numpy(), env._done_buf.cpu().numpy()
            for e in range(96):
                if ref[e]["on"]:
                    ref[e]["frames"].append(f[e].copy()); ref[e]["contacts"].append(c[e].copy()); ref[e]["obs"].append(o[e].copy())
                    if d[e] == DoneFlags.FAIL.value:
                        ref[e]["on"] = False
        host_side()
        env.write_agent_states()
        act = torch.randn((96, 28), device=DEV) * 0.5
        for _ in range(30):
            if not env.is_writing_agent_states():
                break
            # step() records by itself while the flag is set; take the host-side copy of the same state right after
            _, _, done, _ = env.step(act)
            host_side()
            env.reset(done.nonzero().flatten())
        lens = env._rec_len.cpu().numpy()
        assert any(not r["on"] for r in ref)                                   # some envs did fail within the window
        for e in range(96):
            T = len(ref[e]["frames"])
            assert lens[e] == T, (e, lens[e], T)
            assert np.array_equal(env._rec_frames[:T, e].cpu().numpy(), np.stack(ref[e]["frames"]))
            assert np.array_equal(env._rec_contacts[:T, e].cpu().numpy(), np.stack(ref[e]["contacts"]))
            assert np.array_equal(env._rec_obs[:T, e].cpu().numpy(), np.stack(ref[e]["obs"]))
            assert env.is_writing_env_state(e) == ref[e]["on"]
    finally:
        env._bypass_record_fail = old_bypass
        env.set_write_agent_states_flag(False)


def test_config4_parkour_8192_envs_with_record_rollout(tmp_path):
    """BASELINE.json configs[4] at its per-GPU size: 8192 envs on the parkour terrains (stairs / curvy paths / boxes from the reference's
    generators), training rollout with oracle parity of every env, then the parc_4_phys_record loop (deterministic policy, device
    recorder on all 8192 envs, finished clips written in the reference's motion format)."""
    import smoke_impl
    from parc_amd import workloads
    from parc_amd.util import safe_pickle
    torch.manual_seed(0)
    n = 8192
    env, clips, tiled = workloads.build_env("parkour_32clips", n, DEV, seed=0)
    agent = workloads.build_agent(env, DEV, steps_per_iter=4, update_epochs=1, batch_size=2)
    obs, info = env.reset()
    for _ in range(3):
        a, _ = agent._decide_action(obs, info)
        obs, r, done, info = env.step(a)
    torch.cuda.synchronize()
    assert obs.shape == (n, 1312) and torch.isfinite(obs).all() and torch.isfinite(r).all()
    assert smoke_impl.oracle_compare(env, clips, tiled, obs, r) == n          # every one of the 8192 envs against the oracle
    hfcols = obs[:, 871:]
    assert float(hfcols.min()) >= -3.0 and float(hfcols.max()) <= 3.0 and float(hfcols.std()) > 0.05
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    tinfo = agent._train_iter()
    assert np.isfinite(tinfo["critic_loss"].item())
    # record mode on all 8192 envs (256 envs per clip in demo mode); an untrained policy falls early: bypass the tracked-to-the-end
    # filter so that the writer runs, and stop after 25 steps
    env._output_motion_dir = str(tmp_path / "recorded")
    env._bypass_record_fail = True
    ok = agent.record_motions(max_steps=25)
    assert len(ok) == n
    files = sorted((tmp_path / "recorded").glob("*.pkl"))
    assert len(files) >= 1
    d = safe_pickle.load_motion_file_safe(str(files[0]))
    T = d["frames"].shape[0]
    assert d["frames"].shape == (T, 34) and d["contacts"].shape == (T, 15) and d["obs"].shape == (T, 1312) and d["fps"] == 30
    assert d["terrain"]["__class__"] == "util.terrain_util.SubTerrain"


SWITCH_SETS = {
    "everything": {"has_target_xy_obs": True, "global_root_height_obs": True, "track_root_h": False, "rel_task_w": 1.0},
    "global_obs": {"global_obs": True, "has_target_xy_obs": True},
    "no_root_tracking": {"track_root": False},
    "no_root_tracking_at_all": {"track_root": False, "track_root_h": False, "global_obs": True},
    "bare_row": {"enable_tar_obs": False, "use_contact_info": False},
    "task_product": {"rel_task_w": 0.5, "rel_deepmimic_w": 0.7},
}


@pytest.mark.parametrize("tag", sorted(SWITCH_SETS))
def test_observation_and_reward_switches_at_full_size(tag):
    """The non-default settings of IGParkourEnv._compute_obs / _update_reward (ig_parkour_env.py:1054-1244,1275-1404) on the metric's
    configuration (4096 envs, 64 clips on box heightfields), three simulated steps in: EVERY env's handed-out observation row,
    reward, nine reward terms and termination flags against the oracle, whose own switches are pinned on the reference's output
    (G26, tests/test_oracle_golden.py).  The 64-env G26 state itself is checked through the C ABI in tests/test_hip_parity.py."""
    import smoke_impl
    from parc_amd import workloads
    torch.manual_seed(0)
    n = 4096
    over = SWITCH_SETS[tag]
    env, clips, tiled = workloads.build_env("boxes_64clips", n, DEV, seed=0, env_overrides=dict(over))
    obs, info = env.reset()
    lo, hi = env._action_bound_low, env._action_bound_high
    mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
    for _ in range(3):
        a = mid + 0.2 * half * torch.randn((n, 28), device=DEV)
        obs, r, done, info = env.step(a)
    torch.cuda.synchronize()
    width = 1312 + 2 * bool(over.get("has_target_xy_obs")) + bool(over.get("global_root_height_obs")) \
        - (6 * 105 + 6 * 15) * (not over.get("enable_tar_obs", True)) - 15 * (not over.get("use_contact_info", True))
    assert obs.shape == (n, width) and torch.isfinite(obs).all() and torch.isfinite(r).all()
    c = env._core
    if over.get("has_target_xy_obs") or over.get("rel_task_w", 0) > 0:
        # the xy targets were drawn (dm_env.py:617-654) and are not the characters' own positions
        assert float((c.target_xy - c.root_state[:, 0:2]).norm(dim=-1).mean()) > 0.2
    assert smoke_impl.oracle_compare(env, clips, tiled, obs, r) == n


@pytest.mark.parametrize("workload,num_envs", [("flat_1clip", 1024), ("boxes_64clips", 4096), ("iter0_1024clips", 4096)])
def test_baseline_config_workloads_at_full_size(workload, num_envs):
    """BASELINE.json configs[1] (1024 envs, flat terrain, one clip), configs[2] (4096 envs on procgen box heightfields, 64 clips: the
    configuration the metric is quoted on and bench.py times) and configs[3]'s single-GPU share (4096 envs on the iter-0 stand-in:
    1024 clips, 32 x 32 tiles, a 1504^2 heightfield and ~83 MB of clip rows, i.e. nothing fits in L2) at their full env counts:
    oracle parity of every env of the full launch, finiteness, per-env independence of the fused post-step kernel (a subset
    launch reproduces the rows of the full launch), and one PPO iteration end to end."""
    import smoke_impl
    from parc_amd import _hip, workloads
    torch.manual_seed(0)
    env, clips, tiled = workloads.build_env(workload, num_envs, DEV, seed=0)
    assert env.get_num_envs() == num_envs and len(clips) == {"flat_1clip": 1, "boxes_64clips": 64, "iter0_1024clips": 1024}[workload]
    if workload == "boxes_64clips":
        assert tiled[0].shape == (144, 144) and float(tiled[0].std()) > 0.3        # 8 x 8 tiles of 18^2 cells, boxes of U[-3, 3] m
    if workload == "iter0_1024clips":
        assert tiled[0].shape == (1504, 1504) and sum(c["frames"].shape[0] for c in clips) * 448 > 80e6
    obs, info = env.reset()
    lo, hi = env._action_bound_low, env._action_bound_high
    mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
    for _ in range(3):
        a = mid + 0.2 * half * torch.randn((num_envs, 28), device=DEV)
        obs, r, done, info = env.step(a)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(r).all() and torch.all((done >= 0) & (done <= 3))
    assert r.mean().item() > 0.1                                         # three steps after a reset on the reference pose
    c = env._core
    if workload == "iter0_1024clips":
        assert torch.unique(c.motion_ids).numel() > 900                  # the launch does touch (almost) the whole clip database
    # EVERY env of the launch against the oracle, at the tight tolerance with nothing excused (reference pose, observation, reward,
    # termination flags); the oracle samples the clip rows the device stored, after checking them against its own (smoke_impl)
    assert smoke_impl.oracle_compare(env, clips, tiled, obs, r) == num_envs
    ids = np.linspace(0, num_envs - 1, 64).astype(np.int64)
    # per-env function: relaunching a subset rewrites exactly those rows with the same values
    ref_obs = obs.clone()
    sub = torch.tensor(ids[::4], dtype=torch.int64, device=DEV)
    c.obs[sub] = -5.0
    c.post_step(_hip.POST_OBS | _hip.POST_HF, sub)
    assert torch.equal(c.obs, ref_obs)
    # one PPO iteration on it
    agent = workloads.build_agent(env, DEV, steps_per_iter=4, update_epochs=1, batch_size=2)
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    tinfo = agent._train_iter()
    assert np.isfinite(tinfo["critic_loss"].item()) and np.isfinite(tinfo["actor_loss"].item())


def test_reset_state_offsets(env):
    """RefCharEnv.apply_offsets_to_char_state (mgdm_dm_util.py:138-157): fixed per-env offsets on the character state at reset --
    position, heading quaternion (quat_multiply(offset, root_rot)), velocities and dofs -- as the GUI caller sets them
    (ig_parkour_env.py:414-427).  With offsets set, the device-side reset path steps aside."""
    from parc_amd.util import torch_util
    dm = env.get_dm_env()
    N = env.get_num_envs()
    scale = dm._rand_root_pos_offset_scale
    dm.set_rand_root_pos_offset_scale(0.0)
    try:
        pos = torch.tensor([0.5, -0.25, 0.1], device=DEV).expand(N, 3).contiguous()
        yaw = torch.linspace(-1.0, 1.0, N, device=DEV)
        rot = torch_util.heading_to_quat(yaw)
        dm.set_root_pos_offset(pos)
        dm.set_root_rot_offset(rot)
        dm.set_root_vel_offset(torch.full((N, 3), 0.2, device=DEV))
        dm.set_dof_pos_offset(torch.full((N, 28), 0.01, device=DEV))
        assert not env.supports_device_reset()
        env.reset()
        c = env._core
        torch.testing.assert_close(env._char_root_pos, c.ref_root_pos + pos, atol=1e-6, rtol=0)
        torch.testing.assert_close(env._char_root_rot, torch_util.quat_mul(rot, c.ref_root_rot), atol=1e-6, rtol=0)
        torch.testing.assert_close(env._char_root_vel, c.ref_root_vel + 0.2, atol=1e-6, rtol=0)
        torch.testing.assert_close(env._char_dof_pos, c.ref_dof_pos + 0.01, atol=1e-6, rtol=0)
        torch.testing.assert_close(env._char_dof_vel, c.ref_dof_vel, atol=0, rtol=0)
        # the published body poses follow the offset state (the root body is the root)
        torch.testing.assert_close(c.rigid_body_state.view(N, 15, 13)[:, 0, 0:3], env._char_root_pos, atol=1e-6, rtol=0)
        assert torch.isfinite(env._obs_buf).all()
        import pytest as _pt
        with _pt.raises(AssertionError):
            dm.set_root_rot_offset(torch.zeros((N, 3), device=DEV))
    finally:
        for setter in (dm.set_root_pos_offset, dm.set_root_rot_offset, dm.set_root_vel_offset, dm.set_root_ang_vel_offset, dm.set_dof_pos_offset,
                       dm.set_dof_vel_offset):
            setter(None)
        dm.set_rand_root_pos_offset_scale(scale)
    assert env.supports_device_reset()
    env.reset()


@pytest.mark.parametrize("steps", [[1, 2, 3, 4, 5], [2], []], ids=["five", "one", "none"])
def test_other_target_step_counts_against_the_oracle(steps):
    """The target waves of the fused post-step kernel carry two adjacent target steps per lane; an odd count leaves the last wave's second
    component empty, a single step uses one half-filled wave, no step launches no target wave at all (dm_env.py:686-718 takes any
    `tar_obs_steps`).  Every env's reference pose, observation row, reward and termination flag against the oracle, after a few steps and
    after a restart of half the envs."""
    import smoke_impl
    from parc_amd import workloads
    torch.manual_seed(0)
    n = 256
    env, clips, tiled = workloads.build_env("boxes_64clips", n, DEV, seed=1, env_overrides={"tar_obs_steps": steps})
    S = len(steps)
    assert int(env._cfg.struct.num_tar_steps) == S
    obs, info = env.reset()
    assert obs.shape[1] == 136 + S * 105 + S * 15 + 15 + 441
    lo, hi = env._action_bound_low, env._action_bound_high
    mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
    for _ in range(3):
        obs, r, done, info = env.step(mid + 0.2 * half * torch.randn((n, 28), device=DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(r).all()
    assert smoke_impl.oracle_compare(env, clips, tiled, obs, r) == n
    env.reset(torch.arange(0, n, 2, device=DEV))
    obs, r, done, info = env.step(mid + 0.2 * half * torch.randn((n, 28), device=DEV))
    torch.cuda.synchronize()
    assert smoke_impl.oracle_compare(env, clips, tiled, obs, r) == n


@pytest.mark.parametrize("n", [3, 250])
def test_env_counts_that_do_not_fill_the_last_workgroup(n):
    """The post-step kernel serves 4 envs per workgroup; a launch whose env count is not a multiple of 4 (or smaller than one workgroup)
    leaves lane groups of the last workgroup without an env.  Every env against the oracle, after steps and after a restart of some."""
    import smoke_impl
    from parc_amd import workloads
    torch.manual_seed(0)
    env, clips, tiled = workloads.build_env("boxes_64clips", n, DEV, seed=3)
    obs, info = env.reset()
    lo, hi = env._action_bound_low, env._action_bound_high
    mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
    for _ in range(3):
        obs, r, done, info = env.step(mid + 0.2 * half * torch.randn((n, 28), device=DEV))
    torch.cuda.synchronize()
    assert obs.shape == (n, 1312) and torch.isfinite(obs).all() and torch.isfinite(r).all()
    assert smoke_impl.oracle_compare(env, clips, tiled, obs, r) == n
    env.reset(torch.arange(0, n, 2, device=DEV))
    obs, r, done, info = env.step(mid + 0.2 * half * torch.randn((n, 28), device=DEV))
    torch.cuda.synchronize()
    assert smoke_impl.oracle_compare(env, clips, tiled, obs, r) == n
