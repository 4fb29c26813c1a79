"""GPU parity of the learner-side kernels (K12, K14, K19, K21, K22) and of the tracking-error metric against fixtures produced
by the REFERENCE's own classes on CPU (tests/golden/gen_golden.py stages ppo-loss, normalizer, trackers, action-head):
PPOAgent._compute_loss (learning/ppo_agent.py:212-330), Normalizer (learning/normalizer.py:18-86), DMPPOReturnTracker.update
(learning/dm_ppo_return_tracker.py:66-99), TrackingErrorTracker.update, PPOAgent._decide_action with DistributionGaussianDiag
(learning/ppo_agent.py:87-119, learning/distribution_gaussian_diag.py:39-102), compute_tracking_error (mgdm_dm_util.py:578-611)."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO, golden
from test_hip_parity import DEV, T, _core_from_golden, close, km, mlib  # noqa: F401  (km, mlib are fixtures)

pytestmark = pytest.mark.gpu

G14_CASES = ["default", "all_terms_l1", "critic_gate", "mask_all", "mask_one", "mask_none", "no_bound_term"]


@pytest.mark.parametrize("case", G14_CASES)
def test_g14_fused_ppo_loss_against_reference(case):
    """parc_ppo_loss: loss, every logged term and the gradients w.r.t. the network outputs equal what the reference's
    _compute_loss + autograd produced on the same batch (fixture G14)."""
    from parc_amd.learning import rl_util
    z = golden("g14_ppo_loss")
    assert case in [str(c) for c in z["case_names"]]
    clip, bw, ew, rw, cw, gate, l1 = [float(v) for v in z[case + "_params"]]
    mean = T(z["mean"]).requires_grad_(True)
    logstd = T(z["logstd"]).requires_grad_(True)
    pred = T(z["pred"]).requires_grad_(True)
    loss, out = rl_util.ppo_loss(mean, logstd, pred, T(z["norm_action"]), T(z["a_logp"]), T(z["adv"]), T(z[case + "_mask"]),
                                 T(z[case + "_tar_val"]), clip, bw, ew, rw, cw, gate, bool(l1))
    g_mean, g_logstd, g_pred = torch.autograd.grad(loss, [mean, logstd, pred])
    torch.cuda.synchronize()
    ref = dict(zip([str(k) for k in z["info_names"]], z[case + "_info"]))
    if case == "mask_none":
        # no random action in the batch: the reference's actor loss is the mean of an empty selection = NaN and its NaN trap stops
        # the run (ppo_agent.py:242-252); the kernel must surface the same NaN (the agent's trap reads it), not a silent zero
        assert np.isnan(ref["actor_loss"]) and np.isnan(ref["loss"])
        assert torch.isnan(loss) and torch.isnan(out[2])
        assert abs(out[1].item() - ref["critic_loss"]) <= 2e-6 * max(1.0, abs(ref["critic_loss"]))
        return
    got = {"loss": out[0], "critic_loss": out[1], "actor_loss": out[2], "clip_frac": out[3], "imp_ratio": out[4], "action_bound_loss": out[5],
           "action_entropy": out[6], "action_reg_loss": out[7]}
    for k, r in ref.items():
        if np.isnan(r):          # a term the reference does not compute under this configuration (weight 0)
            continue
        assert abs(got[k].item() - r) <= 2e-5 * max(1.0, abs(r)), (k, got[k].item(), r)
    assert out[8].item() == float(z[case + "_mask"].sum())
    for a, name in ((g_mean, "grad_mean"), (g_logstd, "grad_logstd"), (g_pred, "grad_pred")):
        b = z[case + "_" + name]
        scale = max(float(np.abs(b).max()), 1e-6)
        err = float(np.abs(a.cpu().numpy() - b).max())
        assert err <= 3e-5 * scale + 1e-9, (name, err, scale)
    if case == "critic_gate":        # critic loss > 20: the actor term must not move the policy (ppo_agent.py:225-238)
        assert ref["critic_loss"] > 20.0 and float(g_mean.abs().max()) == 0.0 and float(g_logstd.abs().max()) == 0.0 and float(g_pred.abs().max()) > 0


def test_g15_normalizer_kernels_against_reference():
    """moments (K22: parc_moments_accumulate) + normalise-and-clamp (K12: parc_normalize_clamp) through the Normalizer class."""
    from test_host_logic import check_normalizer_against_g15
    check_normalizer_against_g15(DEV)
    # and the kernel path was the one taken: same call on a tensor the kernel cannot take (unaligned view) agrees bit for bit
    from parc_amd.learning.normalizer import Normalizer
    z = golden("g15_normalizer")
    nz = Normalizer((1312,), DEV, clip=10.0)
    nz._mean[:] = T(z["mean_2"])
    nz._std[:] = T(z["std_2"])
    q = T(z["query"])
    pad = torch.zeros(q.numel() + 1, device=DEV)
    pad[1:] = q.flatten()
    assert torch.equal(nz.normalize(q), nz.normalize(pad[1:].view_as(q)))


def test_g16_return_tracker_kernel_against_reference():
    """K21 (parc_return_tracker_update) and the tracking-error tracker over the reference's 40-step episode stream."""
    from test_host_logic import check_trackers_against_g16
    check_trackers_against_g16(DEV)


def test_g17_action_head_against_reference(km):
    """K14 (parc_action_head): sample / mode by the exploration mask, log-probability and un-normalised action, given the
    reference's noise and Bernoulli draws (fixture G17, PPOAgent._decide_action in TRAIN and TEST mode)."""
    from parc_amd import _hip
    z = golden("g17_action_head")
    n, A = z["mean"].shape
    mean, logstd, noise = T(z["mean"]), T(z["logstd"]), T(z["noise"])
    a_mean, a_std = T(0.5 * (z["a_high"] + z["a_low"])), T(0.5 * (z["a_high"] - z["a_low"]))
    p = _hip.ptr
    for mode, mask in (("TRAIN", T(z["bernoulli"])), ("TEST", torch.zeros(n, device=DEV))):
        a, logp = torch.empty_like(mean), torch.empty(n, device=DEV)
        _hip.check(_hip.lib().parc_action_head(_hip.stream(), n, A, p(mean), p(logstd), p(noise), p(mask), p(a_mean), p(a_std), p(a), p(logp)),
                   "parc_action_head")
        torch.cuda.synchronize()
        close(a, z["action_" + mode], atol=2e-6, rtol=2e-6)
        close(logp, z["a_logp_" + mode], atol=2e-4, rtol=2e-6)
        np.testing.assert_array_equal(mask.cpu().numpy(), z["mask_" + mode])
    assert 0.5 < z["mask_TRAIN"].mean() < 0.9 and z["mask_TEST"].sum() == 0
    # the distribution object the model hands out: entropy / regulariser / log-probability as the reference's class computes them
    from parc_amd.learning import dm_ppo_model
    dist = dm_ppo_model.DistributionGaussianDiag(mean, logstd.expand(n, A))
    close(dist.entropy(), z["entropy"], atol=1e-5)
    close(dist.param_reg(), z["param_reg"], atol=1e-5)
    close(dist.log_prob(mean + torch.exp(logstd) * noise), z["logp_of_noise"], atol=2e-4, rtol=2e-6)
    # action-bound penalty (base_agent.py:456-475) = the bound term of the fused loss with every other term off
    from parc_amd.learning import rl_util
    ones = torch.ones(n, device=DEV)
    zeros = torch.zeros(n, device=DEV)
    _, out = rl_util.ppo_loss(mean, logstd, zeros, mean.clone(), dist.log_prob(mean), zeros, ones, zeros, 0.2, 1.0, 0.0, 0.0, 0.0)
    assert abs(out[5].item() - float(z["action_bound_loss"].mean())) <= 1e-5 * max(1.0, float(z["action_bound_loss"].mean()))


def test_g18_tracking_error_on_the_device(km, mlib):
    """IGParkourEnv._compute_tracking_error (compute_tracking_error, mgdm_dm_util.py:578-611) on the G6 state."""
    from parc_amd import _hip
    from parc_amd.envs.ig_parkour.ig_parkour_env import IGParkourEnv
    core, z = _core_from_golden(km, mlib)
    core.post_step(_hip.POST_REF)
    env = IGParkourEnv.__new__(IGParkourEnv)            # the method needs the kinematic model, the core and the state views only
    env._kin_char_model, env._core = km, core
    n = 64
    env._char_root_pos, env._char_root_rot = core.root_state[:, 0:3], core.root_state[:, 3:7]
    env._char_root_vel, env._char_root_ang_vel = core.root_state[:, 7:10], core.root_state[:, 10:13]
    ds = core.dof_state.view(n, 28, 2)
    env._char_dof_pos, env._char_dof_vel = ds[..., 0], ds[..., 1]
    te = env._compute_tracking_error()
    torch.cuda.synchronize()
    assert te.shape == (n, 7)
    close(te[:, [0, 2]], z["tracking_error"][:, [0, 2]], atol=2e-5)            # positions
    close(te[:, [1, 3]], z["tracking_error"][:, [1, 3]], atol=1e-4)            # angles
    close(te[:, 4:7], z["tracking_error"][:, 4:7], atol=3e-4)                  # finite-difference velocities of the clip database


def test_g19_recorded_clip_loads_like_in_the_reference(km):
    """A clip this package recorded (tests/golden/recorded/, written on the GPU by record mode) loaded by this package's MotionLib
    gives the frame arrays the REFERENCE's MotionLib derived from the same file (fixture G19)."""
    from parc_amd.anim.motion_lib import MotionLib
    z = golden("g19_recorded_files")
    ml = MotionLib(os.path.join(REPO, "tests", "golden", "recorded", "recorded_clip_dm.pkl"), km, DEV, contact_info=True)
    assert ml.num_motions() == 1 and int(ml._motion_num_frames[0]) == int(z["num_frames"][0])
    assert abs(ml._motion_lengths[0].item() - float(z["length"][0])) < 1e-6
    rp, rr, rv, rav, jr, dv, cont = ml.calc_motion_frame(torch.zeros(3, dtype=torch.int64, device=DEV), T([0.0, 0.21, 10.0]))
    close(rp, z["q_root_pos"])
    close(rr, z["q_root_rot"])
    close(jr, z["q_joint_rot"])
    close(cont, z["q_contacts"])
    F = int(z["num_frames"][0])
    ids = torch.zeros(F, dtype=torch.int64, device=DEV)
    times = torch.arange(F, device=DEV, dtype=torch.float32) / 30.0
    rp, rr, rv, rav, jr, dv, cont = ml.calc_motion_frame(ids, times)
    close(rp, z["frame_root_pos"], atol=2e-5)
    sign = np.sign(np.sum(rr.cpu().numpy() * z["frame_root_rot"], axis=-1, keepdims=True))
    # (a query exactly on a frame time still goes through slerp between that frame and a neighbour with blend ~0 or ~1)
    close(rr.cpu().numpy() * sign, z["frame_root_rot"], atol=1e-4)
    t = ml._terrains[0]
    np.testing.assert_array_equal(np.asarray(t.hf.cpu()), z["ter_hf"])


def test_explicit_backward_equals_autograd():
    """The update phase's explicit training step (DMPPOModel.train_forward / train_backward: every gradient written once into the flat
    buffer, ReLU mask + bias gradient in one kernel, no autograd graph) against the autograd path it replaces, on a real minibatch:
    same loss terms, same flat gradient; and a whole _update_model from identical weights ends at the same parameters."""
    import copy
    from parc_amd import workloads
    from parc_amd.learning.dm_ppo_agent import _LOSS_KEYS
    torch.manual_seed(0)
    env, _, _ = workloads.build_env("boxes_64clips", 256, DEV, seed=0)
    agent = workloads.build_agent(env, DEV, steps_per_iter=8, update_epochs=2, batch_size=2)
    assert agent._explicit_update_ok()
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    agent._exp_buffer.reset()
    agent.eval()
    agent._rollout_train(8)
    agent._build_train_data()
    eb = agent._exp_buffer
    idx = torch.randperm(8 * 256, device=DEV)[:512]
    batch = {k: eb.get_data_flat(k)[idx] for k in _LOSS_KEYS + ["loss_rec"]}
    A = batch["action"].shape[1]
    rec = batch["loss_rec"]                            # the packed per-sample record holds what the six separate arrays hold
    assert torch.equal(rec[:, 0:A], agent._a_norm.normalize(batch["action"])) and torch.equal(rec[:, A], batch["a_logp"])
    assert torch.equal(rec[:, A + 1], batch["adv"]) and torch.equal(rec[:, A + 2], batch["rand_action_mask"]) and torch.equal(rec[:, A + 3], batch["tar_val"])
    # a spread of value targets / advantages so that every term of the loss is active
    flat = agent._optimizer._flat_grad
    agent.train()
    info = agent._compute_loss(batch)
    flat.zero_()
    info["loss"].backward()
    g_ref = flat.clone()
    flat.fill_(7.0)                                   # stale content must be overwritten, not accumulated into
    acc = {}
    sd = copy.deepcopy(agent.state_dict())
    mean, logstd, pred, saved = agent._model.train_forward(batch["norm_obs"])
    from parc_amd.learning import rl_util
    cfg = rl_util.ppo_cfg(agent._ppo_clip_ratio, agent._action_bound_weight, agent._action_entropy_weight, agent._action_reg_weight,
                          agent._critic_loss_weight, 20.0, agent._critic_loss_type != "L2")
    out, g_mean, g_logstd, g_pred = rl_util.ppo_loss_and_grads_packed(mean, logstd, pred, rec, cfg)
    out2, g_mean2, _, g_pred2 = rl_util.ppo_loss_and_grads(mean, logstd, pred, rec[:, 0:A].contiguous(), batch["a_logp"], batch["adv"],
                                                            batch["rand_action_mask"], batch["tar_val"], cfg)
    assert torch.equal(out[:9], out2[:9]) and torch.equal(g_mean, g_mean2) and torch.equal(g_pred, g_pred2)      # packed == separate arrays
    agent._model.train_backward(saved, g_mean, g_logstd, g_pred, lambda p: p.grad)
    torch.cuda.synchronize()
    assert abs(out[0].item() - info["loss"].item()) <= 1e-6 * max(1.0, abs(info["loss"].item()))
    scale = float(g_ref.abs().max())
    assert scale > 0 and float((flat - g_ref).abs().max()) <= 2e-5 * scale, (float((flat - g_ref).abs().max()), scale)
    # per parameter, relative to that parameter's own gradient size (the critic head's is much larger than the first layers')
    for p in agent._optimizer._param_list:
        ref = g_ref[(p.grad.data_ptr() - flat.data_ptr()) // 4:][:p.numel()].view_as(p)
        s_ = float(ref.abs().max())
        assert float((p.grad - ref).abs().max()) <= 5e-5 * s_ + 1e-12, (tuple(p.shape), float((p.grad - ref).abs().max()), s_)
    # whole update phase, both paths from the same weights / optimizer state / sampling order
    results = []
    for explicit in (False, True):
        agent.load_state_dict(sd)
        agent._optimizer.reset_state()
        agent._config["explicit_backward"] = explicit
        eb._reset_sample_buf()
        torch.manual_seed(5)
        eb._sample_buf[:] = torch.randperm(eb._sample_buf.shape[0], device=DEV)
        eb._sample_buf_head = 0
        tinfo = agent._update_model()
        results.append((tinfo, torch.cat([p.detach().reshape(-1) for p in agent._optimizer._param_list]).clone()))
    (ia, pa), (ib, pb) = results
    for k in ("loss", "critic_loss", "actor_loss", "clip_frac", "imp_ratio"):
        assert abs(ia[k].item() - ib[k].item()) <= 1e-5 * max(1.0, abs(ia[k].item())), k
    step = float((pa - torch.cat([v.reshape(-1) for k, v in sd.items() if k.startswith("_model") and "logstd" not in k]).to(DEV)).abs().max())
    assert step > 0 and float((pa - pb).abs().max()) <= 1e-3 * step + 1e-9
    agent._config["explicit_backward"] = True


def test_relu_bwd_bias_grad_kernel():
    """parc_relu_bwd_bias_grad against threshold_backward + column sum, ragged row count, in-place update."""
    from parc_amd import _hip
    g = torch.Generator().manual_seed(2)
    for rows, dim in ((1000, 512), (16384, 2048), (7, 28 * 4)):
        gy = torch.randn(rows, dim, generator=g).to(DEV)
        y = torch.relu(torch.randn(rows, dim, generator=g)).to(DEV)
        ref = torch.ops.aten.threshold_backward(gy, y, 0.0)
        db_ref = ref.sum(dim=0, dtype=torch.float64)
        db = torch.full((dim,), 3.0, device=DEV)
        L = _hip.lib()
        ws = torch.empty(int(L.parc_relu_bwd_workspace_floats(rows, dim)), device=DEV)
        _hip.check(L.parc_relu_bwd_bias_grad(_hip.stream(), rows, dim, _hip.ptr(gy), _hip.ptr(y), _hip.ptr(db), _hip.ptr(ws)), "relu_bwd")
        torch.cuda.synchronize()
        assert torch.equal(gy, ref)
        assert float((db.double() - db_ref).abs().max()) <= 1e-5 * max(1.0, float(db_ref.abs().max())) * (rows ** 0.5)
    assert L.parc_relu_bwd_bias_grad(_hip.stream(), 4, 6, _hip.ptr(gy), _hip.ptr(y), _hip.ptr(db), _hip.ptr(ws)) == -1       # dim % 4
    # parc_weighted_colsum: out[c] = sum_r w[r] x[r, c] (the value head's weight gradient g_pred^T h)
    for rows, dim in ((16384, 512), (777, 64)):
        x = torch.randn(rows, dim, generator=g).to(DEV)
        w = torch.randn(rows, generator=g).to(DEV)
        out = torch.full((dim,), -1.0, device=DEV)
        ws = torch.empty(int(L.parc_relu_bwd_workspace_floats(rows, dim)), device=DEV)
        _hip.check(L.parc_weighted_colsum(_hip.stream(), rows, dim, _hip.ptr(x), _hip.ptr(w), _hip.ptr(out), _hip.ptr(ws)), "weighted_colsum")
        ref = (w.double().unsqueeze(0) @ x.double()).squeeze(0)
        assert float((out.double() - ref).abs().max()) <= 2e-5 * (rows ** 0.5)


def test_flat_sgd_step_equals_torch_sgd_with_clipping():
    """parc_sgd_momentum_step (clip by global norm + SGD with momentum over flat buffers, two passes) against clip_grad_norm_ +
    torch.optim.SGD(momentum=0.9) on the same gradients, three steps, with and without the clip being active."""
    from parc_amd.learning import mp_optimizer
    for max_norm in (1000.0, 0.5):
        torch.manual_seed(1)
        ref = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.ReLU(), torch.nn.Linear(53, 5)).to(DEV)
        mine = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.ReLU(), torch.nn.Linear(53, 5)).to(DEV)
        mine.load_state_dict(ref.state_dict())
        opt_ref = torch.optim.SGD(ref.parameters(), 5e-3, momentum=0.9)
        opt = mp_optimizer.MPOptimizer({"type": "SGD", "learning_rate": 5e-3}, list(mine.parameters()))
        assert opt._flat_sgd and all(p.data_ptr() >= opt._flat_param.data_ptr() for p in mine.parameters())
        for step in range(3):
            x = torch.randn(64, 37, device=DEV)
            opt_ref.zero_grad()
            ref(x).square().sum().backward()
            gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
            opt_ref.step()
            opt.step(mine(x).square().sum(), model=mine, max_norm=max_norm)
            assert abs(opt._grad_norm.item() - gn.item()) <= 1e-5 * gn.item()
            for a, b in zip(mine.parameters(), ref.parameters()):
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (step, float((a - b).abs().max()))
        assert (gn.item() > max_norm) == (max_norm == 0.5)
    sd = mine.state_dict()                                        # parameters are views of the flat buffer: state_dict round trip
    mine.load_state_dict({k: v.clone() + 1.0 for k, v in sd.items()})
    assert torch.allclose(opt._flat_param, torch.cat([p.reshape(-1) for p in mine.parameters()]))


def test_flat_sgd_notices_a_rebound_parameter():
    """The flat SGD step updates the flat parameter / momentum buffers only; every nn.Parameter is a view into them.  Anything that
    rebinds p.data or p.grad afterwards (load_state_dict(assign=True), module.float() / .to(), zero_grad(set_to_none=True)) would leave
    the model reading tensors training no longer updates: MPOptimizer checks the aliasing (pointer compares) every CHECK_ALIAS_STEPS
    steps and raises.  No torch optimizer object is built in this mode (round-2 advisor finding)."""
    from parc_amd.learning import mp_optimizer
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2)).to(DEV)
    opt = mp_optimizer.MPOptimizer({"type": "SGD", "learning_rate": 0.1}, list(m.parameters()))
    assert opt._flat_sgd and opt._optimizer is None
    opt.CHECK_ALIAS_STEPS = 1
    x = torch.randn(5, 8, device=DEV)
    w0 = m[0].weight.detach().clone()
    opt.step(torch.mean(torch.square(m(x))))
    assert not torch.equal(w0, m[0].weight)                   # the view moved with the flat buffer
    opt._check_aliasing()
    m[0].weight.data = m[0].weight.data.clone()               # what load_state_dict(assign=True) / .to() would do
    with pytest.raises(RuntimeError, match="no longer aliases the optimizer's flat parameter buffer"):
        opt.step(torch.mean(torch.square(m(x))))
    m2 = torch.nn.Linear(4, 4).to(DEV)
    opt2 = mp_optimizer.MPOptimizer({"type": "SGD", "learning_rate": 0.1}, list(m2.parameters()))
    opt2.CHECK_ALIAS_STEPS = 1
    m2.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="no longer aliases the flat gradient buffer"):
        opt2._check_aliasing()


@pytest.mark.parametrize("explicit", [True, False])
def test_nan_trap_stops_at_the_epoch_and_dumps_the_offending_minibatch(explicit, tmp_path, monkeypatch):
    """ppo_agent.py:242-252: a NaN critic / actor loss ends the run and the minibatch that produced it is dumped.  Here the trap reads
    the losses once per update epoch: a batch without any random action (the reference's own NaN case: the mean of an empty selection)
    is planted in the rollout data, the update raises in the FIRST epoch and output/debug_batch.pkl holds exactly a minibatch that
    contains planted rows."""
    import pickle
    from parc_amd import workloads
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    env, _, _ = workloads.build_env("flat_1clip", 64, DEV, seed=0)
    agent = workloads.build_agent(env, DEV, steps_per_iter=4, update_epochs=3, batch_size=2, explicit_backward=explicit)
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    agent._train_iter()                                    # a healthy iteration first
    orig = agent._build_train_data

    def poisoned():
        info = orig()
        eb = agent._exp_buffer
        eb.get_data("rand_action_mask").zero_()              # no random action anywhere: every minibatch's actor loss is 0 / 0
        if eb.has_buffer("loss_rec"):
            eb.get_data("loss_rec")[..., 30] = 0.0           # the packed copy the update reads: [norm action 28 | a_logp | adv | MASK | tar_val]
        return info
    agent._build_train_data = poisoned
    with pytest.raises(FloatingPointError, match="minibatch 0 of update epoch 0"):
        agent._train_iter()
    d = pickle.load(open(tmp_path / "output" / "debug_batch.pkl", "rb"))
    key = "loss_rec" if "loss_rec" in d else "rand_action_mask"
    assert d[key].shape[0] == 2 * 64 and all(not v.is_cuda for v in d.values())


def _philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) in numpy uint64 arithmetic:
    ctr [n, 4] uint32, key (2,) uint32 -> [n, 4] uint32.  An independent restatement for the test (pinned below by the paper's
    known-answer vectors)."""
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & mask, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & mask]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack(c, axis=1).astype(np.uint32)


def test_rng_step_is_philox_and_its_outputs_are_what_they_claim():
    """parc_rng_step: one launch per rollout step for the policy's action noise and the env's uniform pool.  (1) the numpy restatement
    reproduces the Random123 known-answer vectors of Philox4x32-10; (2) the device's uniforms are bit for bit (r >> 8) * 2^-24 of that
    generator at counter (index / 4, 0, step, 0) and key = seed; (3) the step counter advances by one per launch, the ticket cell
    returns to zero, and the same (seed, step) gives the same numbers; (4) the normals are Box-Muller pairs of the same stream with the
    moments of N(0, 1)."""
    from parc_amd import _hip
    kat = _philox4x32_10(np.array([[0, 0, 0, 0], [0xFFFFFFFF] * 4, [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], dtype=np.uint32)[:1],
                         np.array([0, 0], np.uint32))
    assert [hex(v) for v in kat[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    kat = _philox4x32_10(np.array([[0xFFFFFFFF] * 4], dtype=np.uint32), np.array([0xFFFFFFFF, 0xFFFFFFFF], np.uint32))
    assert [hex(v) for v in kat[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    kat = _philox4x32_10(np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], dtype=np.uint32), np.array([0xA4093822, 0x299F31D0], np.uint32))
    assert [hex(v) for v in kat[0]] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    L = _hip.lib()
    seed = 0x0123456789ABCDEF
    nu, nn = 45056 + 3, 4096 * 28
    state = torch.zeros(2, dtype=torch.int64, device=DEV)
    uni, nor = torch.empty(nu, device=DEV), torch.empty(nn, device=DEV)
    outs = []
    for step in range(3):
        _hip.check(L.parc_rng_step(_hip.stream(), seed, _hip.ptr(state), _hip.ptr(uni), nu, _hip.ptr(nor), nn, None, 0), "parc_rng_step")
        torch.cuda.synchronize()
        assert state.tolist() == [step + 1, 0]
        outs.append((uni.cpu().numpy().copy(), nor.cpu().numpy().copy()))
    qu = (nu + 3) // 4
    for step in range(3):
        ctr = np.zeros((qu, 4), np.uint32)
        ctr[:, 0], ctr[:, 2] = np.arange(qu), step
        r = _philox4x32_10(ctr, np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32))
        want = ((r >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(-1)[:nu]
        np.testing.assert_array_equal(outs[step][0], want)
        assert 0.0 <= outs[step][0].min() and outs[step][0].max() < 1.0
        z = outs[step][1].astype(np.float64)
        assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02 and abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3.0) < 0.1
        assert np.isfinite(z).all() and np.abs(z).max() < 6.5
        # Box-Muller of the same stream: quads qu .. qu + nn / 4
        ctr = np.zeros((nn // 4, 4), np.uint32)
        ctr[:, 0], ctr[:, 2] = qu + np.arange(nn // 4), step
        r = _philox4x32_10(ctr, np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)).astype(np.float64)
        u1 = (np.floor(r[:, 0::2] / 256.0) + 1.0) / 16777216.0
        u2 = np.floor(r[:, 1::2] / 256.0) / 16777216.0
        rad = np.sqrt(-2.0 * np.log(u1))
        wz = np.stack([rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)], axis=-1).reshape(-1)
        np.testing.assert_allclose(outs[step][1], wz, atol=2e-3, rtol=2e-3)       # __logf / __sincosf on the device
    assert not np.array_equal(outs[0][0], outs[1][0])
    state.zero_()                                                  # same (seed, step) -> same numbers
    _hip.check(L.parc_rng_step(_hip.stream(), seed, _hip.ptr(state), _hip.ptr(uni), nu, _hip.ptr(nor), nn, None, 0), "parc_rng_step")
    np.testing.assert_array_equal(uni.cpu().numpy(), outs[0][0])
    np.testing.assert_array_equal(nor.cpu().numpy(), outs[0][1])


def test_rng_step_carries_the_write_row_tick_and_record_group_follows_it():
    """The captured rollout step keeps the experience buffer's write row in a device cell: its first launch (parc_rng_step) moves the
    cell on by one modulo T - ExperienceBuffer.inc on the device, no fill per step - and both record launches of the step write that
    row; rows land where the host-indexed record() puts them.  Normalizer.record equals the column sums for row counts that move the
    partial rows around in the shared workspace."""
    from parc_amd import _hip
    from parc_amd.learning.experience_buffer import ExperienceBuffer
    from parc_amd.learning.normalizer import Normalizer
    T_, N = 5, 300
    eb = ExperienceBuffer(T_, N, DEV)
    ref = ExperienceBuffer(T_, N, DEV)
    for b in (eb, ref):
        b.add_buffer("obs", torch.zeros((T_, N, 1312), device=DEV))
        b.add_buffer("reward", torch.zeros((T_, N), device=DEV))
        b.add_buffer("ep_num", torch.zeros((T_, N), dtype=torch.int32, device=DEV))
        b.add_buffer("replan_timer", torch.zeros((T_, N), device=DEV))
    clock = torch.zeros(1, device=DEV)                                    # one device value recorded for every env of the row
    head = torch.full((1,), T_ - 1, dtype=torch.int64, device=DEV)          # "the row of the step before" the first one
    eb.set_device_head(head)
    state = torch.zeros(2, dtype=torch.int64, device=DEV)
    pool, noise = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(3)
    for step in range(12):
        _hip.check(_hip.lib().parc_rng_step(_hip.stream(), 7, _hip.ptr(state), _hip.ptr(pool), 64, _hip.ptr(noise), 64, _hip.ptr(head), T_), "rng")
        assert int(head.item()) == step % T_ and state.tolist() == [step + 1, 0]
        obs, r = torch.randn((N, 1312), generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
        ep = torch.randint(0, 1000, (N,), generator=g).to(DEV)
        eb.record_group([("obs", obs)])
        clock.fill_(0.125 * step)
        eb.record_group([("reward", r), ("ep_num", ep), ("replan_timer", clock)])
        ref.record("obs", obs); ref.record("reward", r); ref.record("ep_num", ep); ref.record("replan_timer", clock.expand(N)); ref.inc()
        eb.inc()
    for k in ("obs", "reward", "ep_num", "replan_timer"):
        assert torch.equal(eb.get_data(k), ref.get_data(k)), k
    # a tick without random numbers is a launch too
    _hip.check(_hip.lib().parc_rng_step(_hip.stream(), 7, _hip.ptr(state), None, 0, None, 0, _hip.ptr(head), T_), "rng")
    assert int(head.item()) == 12 % T_
    nrm = Normalizer((1312,), device=DEV)
    tot, tot2, cnt = torch.zeros(1312, dtype=torch.float64), torch.zeros(1312, dtype=torch.float64), 0
    for rows in (4096, 64, 1000, 4096, 37):
        x = torch.randn((rows, 1312), generator=g) * 2.0 + 0.5
        nrm.record(x.to(DEV))
        tot += x.double().sum(0); tot2 += (x.double() ** 2).sum(0); cnt += rows
        np.testing.assert_allclose(nrm._acc[0].cpu().numpy(), tot.numpy(), rtol=2e-5, atol=2e-3)
        np.testing.assert_allclose(nrm._acc[1].cpu().numpy(), tot2.numpy(), rtol=2e-5, atol=2e-3)
    assert nrm._new_count == cnt


def test_graph_rollout_writes_the_same_rows_as_the_eager_rollout():
    """One rollout of T steps through the captured step (device write row ticked by the step's first launch, the env's own random
    draw) fills every row of every buffer exactly like the step body issued eagerly with a host-indexed write row does for the SAME random
    numbers - checked on what does not depend on the numbers: each env's timestep / ep_num / env_id rows advance consistently, row t
    holds step t, no row is skipped or written twice."""
    from parc_amd import workloads
    torch.manual_seed(0)
    env, _, _ = workloads.build_env("flat_1clip", 64, DEV, seed=0)
    agent = workloads.build_agent(env, DEV, steps_per_iter=8, update_epochs=1, batch_size=2)
    assert agent._device_tick()
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    from parc_amd.learning.dm_ppo_agent import AgentMode
    for it in range(3):                                     # eager warm-up steps, the capture, then steady replays
        if it < 2:
            info = agent._train_iter()
            assert np.isfinite(info["critic_loss"].item())
        else:                                               # the third rollout alone: its rows are checked BEFORE an update changes the policy
            agent._exp_buffer.reset()
            agent.eval()
            agent.set_mode(AgentMode.TRAIN)
            agent._rollout_train(agent._steps_per_iter)
            assert agent._graphs
        eb = agent._exp_buffer
        ts = eb.get_data("timestep").cpu().numpy()          # [T, N]: the env's step counter after each step
        ep = eb.get_data("ep_num").cpu().numpy()
        assert (eb.get_data("env_id").cpu().numpy() == np.arange(64)[None, :]).all()
        d_ts, d_ep = np.diff(ts, axis=0), np.diff(ep, axis=0)
        # from one row to the next an env either made one more step of the same episode or finished (done in the earlier row) and restarted
        done = eb.get_data("done").cpu().numpy()[:-1] != 0
        assert np.all(np.where(done, ts[1:] == 1, d_ts == 1)) and np.all(np.where(done, d_ep == 1, d_ep == 0))
        assert agent._head_dev == (eb._buffer_head - 1) % eb._buffer_length or not agent._graphs
        # rows written by the fused passes of the captured step (observation ingest, action head) hold what the separate record
        # launches put there: obs[t + 1] is next_obs[t] and the forces carry over wherever the env did not restart
        obs, nxt = eb.get_data("obs"), eb.get_data("next_obs")
        keep = ~torch.tensor(done, device=DEV)
        assert torch.equal(obs[1:][keep], nxt[:-1][keep])
        assert torch.equal(eb.get_data("prev_char_contact_forces")[1:][keep], eb.get_data("next_char_contact_forces")[:-1][keep])
        assert (eb.get_data("rand_action_mask") == 1.0).all() and torch.isfinite(eb.get_data("action")).all()
    # the env's generator moved on by one per step (a cell allocated inside the capture would be re-zeroed by every replay): 24 steps,
    # and no two steps drew the same policy noise
    assert env._rng_state.tolist() == [3 * agent._steps_per_iter - 2, 0]          # (the first two steps of all ran eagerly, on torch's generator)
    with torch.no_grad():
        z_ = []
        for t_ in range(agent._steps_per_iter):
            dist = agent._model.eval_actor(agent._obs_norm.normalize(obs[t_].contiguous()))
            z_.append((agent._a_norm.normalize(eb.get_data("action")[t_]) - dist.mean) / dist.logstd.exp())
        z_ = torch.stack(z_)                                   # [T, N, A] ~ N(0, 1), the noise the head was given
        assert abs(float(z_.mean())) < 0.02 and abs(float(z_.var()) - 1.0) < 0.05
        for t_ in range(1, agent._steps_per_iter):
            assert (z_[t_] - z_[t_ - 1]).abs().max() > 0.5
    # ... and the stored log-probability is the (unchanged) policy's for the stored action on the stored observation
    with torch.no_grad():
        for t_ in (0, 5, 7):
            dist = agent._model.eval_actor(agent._obs_norm.normalize(obs[t_].contiguous()))
            lp = dist.log_prob(agent._a_norm.normalize(eb.get_data("action")[t_]))
            assert (lp - eb.get_data("a_logp")[t_]).abs().max() < 2e-3


def test_obs_ingest_equals_its_three_separate_passes():
    """parc_obs_ingest = Normalizer.normalize + the raw copy into row *head of the experience buffer + Normalizer.record in one pass over
    the observation rows: every output bit-identical to the three launches it replaces (same fp32 operations, same partial rows, same
    summation order), with and without the optional parts, for a row count that leaves a ragged last chunk."""
    from parc_amd.learning.normalizer import Normalizer
    g = torch.Generator(device="cpu").manual_seed(5)
    N, D, T_ = 1000, 1312, 3
    a, b = Normalizer((D,), device=DEV, clip=10.0), Normalizer((D,), device=DEV, clip=10.0)
    for n in (a, b):
        n._mean[:] = torch.randn(D, generator=torch.Generator().manual_seed(1)).to(DEV)
        n._std[:] = (torch.rand(D, generator=torch.Generator().manual_seed(2)) + 0.5).to(DEV)
    buf = torch.full((T_, N, D), -1.0, device=DEV)
    head = torch.tensor([2], dtype=torch.int64, device=DEV)
    for it in range(3):
        x = (torch.randn((N, D), generator=g) * 3.0).to(DEV)
        want_norm = a.normalize(x)
        a.record(x)
        got = b.ingest(x, record=True, copy_into=(buf, head))
        assert torch.equal(got, want_norm) and torch.equal(buf[2], x) and (buf[:2] == -1.0).all()
        assert torch.equal(a._acc, b._acc) and a._new_count == b._new_count == (it + 1) * N
    x = (torch.randn((N, D), generator=g) * 3.0).to(DEV)
    acc0 = b._acc.clone()
    assert torch.equal(b.ingest(x), a.normalize(x)) and torch.equal(b._acc, acc0) and torch.equal(buf[2], buf[2])   # nothing optional: normalise only
    a.update(); b.update()
    assert torch.equal(a._mean, b._mean) and torch.equal(a._std, b._std)
