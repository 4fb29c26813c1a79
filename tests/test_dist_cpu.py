"""Data-parallel path on CPU with gloo, world_size 2: the flat-gradient all-reduce of MPOptimizer, the normaliser merge
and the desync detector (reference: learning/mp_optimizer.py:20-90, learning/normalizer.py:28-58)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    from parc_amd.learning import mp_optimizer, normalizer
    from parc_amd.util import mp_util
    mp_util.init(rank, world, "cpu", port)
    torch.manual_seed(100 + rank)                       # different init per rank: sync() must broadcast rank 0's weights
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
    opt = mp_optimizer.MPOptimizer({"type": "SGD", "learning_rate": 0.1}, list(model.parameters()))
    w_init = [p.detach().clone() for p in model.parameters()]
    torch.manual_seed(7)
    x_all = torch.randn(8, 6)
    y_all = torch.randn(8, 1)
    x, y = x_all[rank * 4:(rank + 1) * 4], y_all[rank * 4:(rank + 1) * 4]      # each rank sees its own half
    loss = torch.mean(torch.square(model(x) - y))
    opt.step(loss, model=model, max_norm=1000.0)
    assert opt._check_synced()
    # sync() / _check_synced() are ONE broadcast of a flat buffer each, whatever the number of parameters (4 here, 16 in the agent)
    calls = []
    real_bc = torch.distributed.broadcast
    torch.distributed.broadcast = lambda t, src, *a, **k: (calls.append(int(t.numel())), real_bc(t, src, *a, **k))[1]
    try:
        n_all = sum(p.numel() for p in model.parameters())
        opt.sync()
        assert calls == [n_all], calls
        assert opt._check_synced() and calls == [n_all, n_all]
        if rank == 1:                               # a desynchronised rank is detected, and sync() repairs it
            with torch.no_grad():
                list(model.parameters())[2].add_(1e-3)
        assert not opt._check_synced()
        opt.sync()
        assert opt._check_synced()
    finally:
        torch.distributed.broadcast = real_bc
    # the reference's scaling rule at world 2 (base_agent.py:179-180, ppo_agent.py:27-29; bench.py --scaling reference): half the
    # rollout, half the minibatch per rank - the same number of optimizer steps per epoch, each one an exchange of the flat gradient
    from parc_amd.envs.ig_parkour.default_config import default_agent_config
    from parc_amd.learning.dm_ppo_agent import DMPPOAgent
    T_, B_ = DMPPOAgent.rollout_shape(default_agent_config(), world)
    assert (T_, B_) == (16, 2) and DMPPOAgent.rollout_shape(dict(default_agent_config(), mp_scale_rollout=False), world) == (32, 4)
    assert DMPPOAgent.rollout_shape(default_agent_config(), 8) == (4, 1) and DMPPOAgent.rollout_shape(default_agent_config(), 1) == (32, 4)
    torch.manual_seed(500)
    m5 = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    o5 = mp_optimizer.MPOptimizer({"type": "SGD", "learning_rate": 0.05, "allreduce_buckets": 3}, list(m5.parameters()))
    n_env = 3                                       # rows per env-step of this rank; a minibatch = B_ * n_env rows of its T_ * n_env samples
    torch.manual_seed(600 + rank)
    xs = torch.randn(T_ * n_env, 6)
    steps = 0
    for lo in range(0, T_ * n_env, B_ * n_env):
        o5.step(torch.mean(torch.square(m5(xs[lo:lo + B_ * n_env]) - 0.5)))
        steps += 1
    assert steps == 32 // 4 and o5.get_steps() == steps and o5._check_synced()      # as many optimizer steps per epoch as on one GPU
    nrm = normalizer.Normalizer((3,), device="cpu", non_norm_indices=torch.tensor([2]))
    nrm.record(torch.full((5, 3), float(rank + 1)))
    nrm.update()
    # plain numpy in the queue: torch tensors would be handed over through shared-memory handles that die with the child
    # bucketed exchange overlapped with backward == one all-reduce after backward (same data, same mean)
    res_ov = []
    for overlap in (True, False):
        torch.manual_seed(300)
        m3 = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
        o3 = mp_optimizer.MPOptimizer({"type": "SGD", "learning_rate": 0.1, "overlap_allreduce": overlap, "allreduce_buckets": 3},
                                      list(m3.parameters()))
        assert (len(o3._buckets) >= 2) == overlap
        for k in range(3):
            torch.manual_seed(400 + 10 * k + rank)
            xx = torch.randn(7, 6)
            o3.step(torch.mean(torch.square(m3(xx) - 1.0)))
        assert o3._check_synced()
        res_ov.append([p.detach().numpy().copy() for p in m3.parameters()])
    for a_, b_ in zip(*res_ov):
        assert (a_ == b_).all() or abs(a_ - b_).max() < 1e-7
    # explicit-gradient step (the caller writes every gradient slice and reports completion in backward order; the bucketed exchange is
    # started from those reports): same parameters as loss.backward() through step(), with and without buckets
    for overlap in (True, False):
        torch.manual_seed(300)
        m4 = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
        o4 = mp_optimizer.MPOptimizer({"type": "SGD", "learning_rate": 0.1, "overlap_allreduce": overlap, "allreduce_buckets": 3},
                                      list(m4.parameters()))
        for k in range(3):
            torch.manual_seed(400 + 10 * k + rank)
            xx = torch.randn(7, 6)
            grads = torch.autograd.grad(torch.mean(torch.square(m4(xx) - 1.0)), list(m4.parameters()))

            def write(grad_of, done, grads=grads, params=list(m4.parameters())):
                for p_, g_ in reversed(list(zip(params, grads))):          # last layer first, like a backward pass
                    grad_of(p_).copy_(g_)
                    done(p_)
            o4._flat_grad.fill_(123.0)                                     # stale content: every slice must be overwritten
            o4.step_explicit(write)
        assert o4._check_synced()
        for a_, b_ in zip(res_ov[0], [p.detach().numpy() for p in m4.parameters()]):
            assert abs(a_ - b_).max() < 1e-6
    # per-epoch cadence: local steps diverge, end_epoch() averages parameters and momentum buffers
    torch.manual_seed(200 + rank)
    m2 = torch.nn.Linear(4, 2)
    o2 = mp_optimizer.MPOptimizer({"type": "SGD", "learning_rate": 0.05, "grad_allreduce": "epoch"}, list(m2.parameters()))
    for k in range(3):
        xx = torch.randn(5, 4)
        o2.step(torch.mean(torch.square(m2(xx) - float(rank))))
    local = [p.detach().clone() for p in m2.parameters()]
    mom_local = [o2._optimizer.state[p]["momentum_buffer"].clone() for p in m2.parameters()]
    o2.end_epoch()
    epoch_res = ([p.numpy().copy() for p in local], [p.detach().numpy().copy() for p in m2.parameters()],
                 [b.numpy().copy() for b in mom_local], [o2._optimizer.state[p]["momentum_buffer"].numpy().copy() for p in m2.parameters()])
    out.put((rank, [p.detach().numpy().copy() for p in model.parameters()], [w.numpy().copy() for w in w_init],
             nrm._mean.detach().numpy().copy(), nrm._count.item(), mp_util.reduce_sum(rank + 1), epoch_res))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_dp_optimizer_and_normalizer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = torch.from_numpy
    epoch = [t[6] for t in res]
    res = [(r, [T(a) for a in w], [T(a) for a in i], T(m), c, s) for r, w, i, m, c, s, _e in res]
    # per-epoch cadence: after end_epoch both ranks hold the mean of the two local parameter / momentum sets
    import numpy as np
    for k in range(2):
        assert not np.allclose(epoch[0][0][k], epoch[1][0][k])                    # local steps really diverged
        mean_p = 0.5 * (epoch[0][0][k] + epoch[1][0][k])
        mean_m = 0.5 * (epoch[0][2][k] + epoch[1][2][k])
        for r in range(2):
            np.testing.assert_allclose(epoch[r][1][k], mean_p, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(epoch[r][3][k], mean_m, rtol=1e-6, atol=1e-7)
    (r0, w0, init0, m0, c0, s0), (r1, w1, init1, m1, c1, s1) = res
    for a, b in zip(init0, init1):
        assert torch.equal(a, b)                        # broadcast from rank 0 at construction
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)                        # identical after the all-reduced step
    # single-process reference: gradient of the mean over the two halves == mean of the per-rank gradients
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
    with torch.no_grad():
        for p, w in zip(model.parameters(), init0):
            p.copy_(w)
    torch.manual_seed(7)
    x_all, y_all = torch.randn(8, 6), torch.randn(8, 1)
    loss = 0.5 * (torch.mean(torch.square(model(x_all[:4]) - y_all[:4])) + torch.mean(torch.square(model(x_all[4:]) - y_all[4:])))
    loss.backward()
    for p, w in zip(model.parameters(), w0):
        assert torch.allclose(p.detach() - 0.1 * p.grad, w, atol=1e-6)
    assert c0 == 10 and c1 == 10 and torch.allclose(m0, torch.tensor([1.5, 1.5, 0.0])) and torch.equal(m0, m1)
    assert s0 == 3 and s1 == 3


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus N` (how the driver calls it) must run N ranks, not silently one: the launcher path is exercised
    without GPU work (--launch-check), and a request for more GPUs than are visible fails instead of reporting fewer."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PARC_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                         env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["scaling"] == "weak" and line["rollout_steps_per_rank"] == 32 and line["minibatch_envs_multiple"] == 4
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launch-check", "--scaling", "reference"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["rollout_steps_per_rank"] == 16 and line["minibatch_envs_multiple"] == 2
    env.pop("PARC_DIST_BACKEND")
    if not torch.cuda.is_available():
        bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=300)
        assert bad.returncode != 0 and "needs 8 visible GPUs" in bad.stderr and bad.stdout.strip() == ""


def test_rank_device_rule():
    """mp_util.rank_device: the reference's launcher hands the SAME --device string to every rank (run.py:98,150-162); a bare
    cuda / cuda:0 with more than one process therefore means "this rank's GPU"."""
    from parc_amd.util import mp_util
    rd = mp_util.rank_device
    assert [rd(r, 8, "cuda:0", num_devices=8, local_rank=r) for r in range(8)] == ["cuda:%d" % r for r in range(8)]
    assert rd(5, 8, "cuda", num_devices=8, local_rank=5) == "cuda:5"
    assert rd(1, 2, "cuda:0", num_devices=1, local_rank=1) == "cuda:0"        # two ranks rehearsing on a one-GPU box share it
    assert rd(3, 8, "cuda:6", num_devices=8, local_rank=3) == "cuda:6"        # an explicit device is taken as given
    assert rd(0, 1, "cuda:0", num_devices=8, local_rank=0) == "cuda:0" and rd(1, 2, "cpu", num_devices=0, local_rank=1) == "cpu"
    assert rd(9, 16, "cuda:0", num_devices=8, local_rank=1) == "cuda:1"       # LOCAL_RANK (torchrun), not the global rank
    # the count that needs no HIP call: a visibility variable wins over sysfs
    old = {k: os.environ.pop(k, None) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    try:
        os.environ["ROCR_VISIBLE_DEVICES"] = "2,3,5"
        assert mp_util.visible_device_count() == 3
        os.environ["HIP_VISIBLE_DEVICES"] = "0"
        assert mp_util.visible_device_count() == 1
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def _mapping_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["PARC_MP_BACKEND"] = "gloo"            # the mapping logic without GPUs: the group itself runs on gloo
    os.environ["HIP_VISIBLE_DEVICES"] = "0,1"         # "two GPUs visible" (nothing touches them: torch.cuda.is_available() is False here)
    os.environ.pop("LOCAL_RANK", None)
    torch.set_num_threads(1)
    from parc_amd.util import mp_util
    # exactly the calls the reference's run.run makes (run.py:95-117, fixture G24): the same string on every rank
    mp_util.init(rank, world, "cuda:0", port)
    t = torch.tensor([rank + 1.0])                     # (host tensor: the gloo group is real, the devices are only names here)
    torch.distributed.all_reduce(t)
    got = {"device": mp_util.get_device(), "env": mp_util.resolve_device("cuda:0"), "explicit": mp_util.resolve_device("cuda:5"),
           "sum": int(t.item())}
    out.put((rank, got))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_one_rank_per_gpu_under_the_reference_launcher_gloo_world2():
    """Two ranks started the way the reference's run.main starts them - both with --device cuda:0 - end up on cuda:0 and cuda:1, and
    that is the device env_builder / agent_builder build on (they resolve the launcher's string through mp_util.resolve_device)."""
    if torch.cuda.is_available():
        pytest.skip("mapping logic test for the CPU container (a GPU box has the real thing in test_dropin_gpu)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mapping_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert res[r] == {"device": "cuda:%d" % r, "env": "cuda:%d" % r, "explicit": "cuda:5", "sum": 3}, res
    import inspect
    from parc_amd.envs import env_builder
    from parc_amd.learning import agent_builder
    assert "mp_util.resolve_device(device)" in inspect.getsource(env_builder.build_env)
    assert "mp_util.resolve_device(device)" in inspect.getsource(agent_builder.build_agent)
