#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the tracker training loop (rollout + train data + PPO update).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One bench "step" = one PPO iteration of the reference's loop (learning/dm_ppo_agent.py:230-272, base_agent.py:290-309):
T = 32 env steps on every one of the 4096 envs of the rank (simulator + fused post-step kernel + policy forward +
experience record + resets), critic passes + TD(lambda) + advantage normalisation, then 5 epochs x 8 minibatches of
16384 samples with the gradient all-reduced over RCCL per minibatch (the reference's cadence).  Per-GPU work is fixed as
N grows ("weak").  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# fp32 throughout, like the reference (torch 1.13 default: no TF32; gfx950 has no TF32 path anyway)
torch.backends.cuda.matmul.allow_tf32 = False

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense fp32 matrix peak (spec; 155 measured)
# algorithmic bytes per env of the fused post-step kernel: K5 heightmap gather 3544 (16 in + 441*4 gathered +
# 441*4 stored) + K3 7 queries x 760 + simulator state read 456 + observation columns [0,871) 3484 + body states 780 + 8 out
POST_STEP_BYTES_PER_ENV = 3544 + 7 * 760 + 456 + 3484 + 780 + 8


def cpu_baseline(env, clips, tiled, budget_s=12.0):
    """The CPU oracle (C port of the reference's kinematic / observation / reward path, OpenMP) on the same 4096-env
    state, timed on this box's host cores.  Physics and the policy are NOT part of it (the reference's physics is the
    GPU-only Isaac Gym binary)."""
    from oracle import oracle as orc
    c = env._core
    km = env._kin_char_model
    z = lambda t: t.detach().cpu().numpy()
    char = orc.Char(z(km._parent_indices), z(km._local_translation), z(km._local_rotation), [j.joint_type.value for j in km._joints],
                    [z(j.axis) if j.axis is not None else np.zeros(3, np.float32) for j in km._joints], [j.dof_idx for j in km._joints])
    mlib = orc.MotionLib(char, [cl["frames"] for cl in clips], [cl["fps"] for cl in clips], [cl["loop"] for cl in clips],
                         [cl["weight"] for cl in clips], [cl["contacts"] for cl in clips])
    n = env.get_num_envs()
    mids, times = z(c.motion_ids), z(c.time_buf + c.motion_time_offsets)
    off = z(c.motion_xy_offset - c.env_offsets[:, 0:2])
    rs, ds = z(c.root_state), z(c.dof_state).reshape(n, 28, 2)
    cf = z(c.contact_forces).reshape(n, 15, 3)
    rb = z(c.rigid_body_state).reshape(n, 15, 13)[..., 0:3]
    rays = z(c.ray_xy_points)
    s = env._cfg.struct
    tar_dt = np.array(list(s.tar_dt), np.float32)
    st = dict(char_root_pos=rs[:, 0:3], char_root_rot=rs[:, 3:7], char_root_vel=rs[:, 7:10], char_root_ang_vel=rs[:, 10:13],
              char_dof_pos=np.ascontiguousarray(ds[..., 0]), char_dof_vel=np.ascontiguousarray(ds[..., 1]), char_rigid_body_pos=rb,
              contact_forces=cf)
    jw, dw, cw, w5 = list(s.joint_err_w)[:14], list(s.dof_err_w)[:28], list(s.contact_w)[:15], list(s.reward_w)

    def one_step():
        ref = orc.update_ref_motion(char, mlib, mids, times, off)
        glob = rs[:, 0:3] + z(c.env_offsets)
        hfs = orc.refresh_ray_obs_hfs(rays, glob, orc.calc_heading(rs[:, 3:7]), tiled[0], tiled[1], tiled[2])
        orc.compute_obs(char, mlib, tar_dt, env._cfg.key_body_ids, mids, times, off, st["char_root_pos"], st["char_root_rot"],
                        st["char_root_vel"], st["char_root_ang_vel"], st["char_dof_pos"], st["char_dof_vel"], cf, hfs)
        orc.compute_reward(char, env._cfg.key_body_ids, st, ref, jw, dw, cw, w5)
    one_step()
    t0 = time.time()
    k = 0
    while time.time() - t0 < budget_s:
        one_step()
        k += 1
    dt = time.time() - t0
    return {"value": n * k / dt, "unit": "env-steps/s", "cores": orc.num_threads(), "kind": "port",
            "sample": "{} env steps x {} envs of the kinematic/observation/reward path (K3 K2 K4 K5 K6-K9) in the C oracle with OpenMP; "
                      "no physics, no policy: the reference's physics is the GPU-only Isaac Gym binary".format(k, n)}


def post_step_in_rollout_by_the_profiler(workload, envs, steps=64, timeout_s=300):
    """The duration of the fused post-step launch where the rollout step issues it, by the DISPATCH's own time stamps: a child process runs
    the rollout's launch sequence eagerly (tools/rollout_only.py --eager: same kernels, same order, same device-side restarts as the
    captured step) under `rocprofv3 --kernel-trace`, and the trace is reduced by tools/rollout_trace_stats.py - the very numbers
    profiles/rNN_rollout_kernel_stats.csv holds.  HIP events cannot deliver this figure from inside the process: a pair recorded around
    the launch reads ~5 us more, a pair bound to the dispatch (hipExtLaunchKernel) ~1 us more (4.7 us more under the profiler) than the
    dispatch's begin -> end.  None when rocprofv3 is not installed or the child fails (the event figure is used then)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rollout_trace_stats
    out = tempfile.mkdtemp(prefix="parc_bench_prof_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        res = subprocess.run([exe, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
                              os.path.join(ROOT, "tools", "rollout_only.py"), str(steps), "--eager", "--workload=" + workload, "--envs=%d" % envs],
                             cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        if res.returncode != 0:
            return None
        for dirpath, _dirs, files in os.walk(out):
            for f in files:
                if f.endswith("kernel_trace.csv"):
                    return rollout_trace_stats.stats(os.path.join(dirpath, f))
        return None
    except Exception:                     # noqa: BLE001  (a context measurement: never let it take the metric down)
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) as a child torch.distributed.run job and
    relay rank 0's JSON line.  The parent makes no GPU call at all - the device count comes from the visibility variables or the KFD
    topology in sysfs (mp_util.visible_device_count), never from the HIP runtime - and it starts a child rather than replacing
    itself.  A count of 0 means "could not tell": the ranks then check for themselves (each refuses if the runtime shows fewer
    devices than ranks)."""
    import socket
    import subprocess
    backend = os.environ.get("PARC_DIST_BACKEND", "nccl")
    from parc_amd.util import mp_util
    n_dev = mp_util.visible_device_count()
    if 0 < n_dev < args.gpus:
        # sysfs says "fewer than asked for": before refusing, let a CHILD ask the runtime (this process stays free of HIP); the larger
        # answer counts - a wrong refusal would be worse than a late one (the ranks check again themselves)
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
            n_dev = max(n_dev, int(out.stdout.strip().splitlines()[-1]))
        except Exception:       # noqa: BLE001
            pass
    if backend == "nccl" and not args.launch_check and 0 < n_dev < args.gpus:
        sys.stderr.write("bench.py: --gpus {} needs {} visible GPUs, found {} (refusing to report a {}-GPU number from fewer)\n".format(
            args.gpus, args.gpus, n_dev, args.gpus))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"n_gpus"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0 or line is None:
        sys.stderr.write("bench.py: the {}-rank job failed (exit code {})\n".format(args.gpus, rc))
        return rc or 1
    print(line)
    return 0


def launch_check(rank, world, backend, scaling="weak"):
    """The launcher's plumbing without the workload: rendezvous, barrier, MAX over ranks of a timer, one line on rank 0."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend if torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
        torch.distributed.barrier()
    t0 = time.time()
    time.sleep(0.01 * (rank + 1))
    tt = torch.tensor([time.time() - t0], dtype=torch.float64)
    ranks = torch.ones(1, dtype=torch.float64)
    if world > 1:
        if torch.distributed.get_backend() == "nccl":
            tt, ranks = tt.cuda(), ranks.cuda()
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(ranks)
    if rank == 0:
        from parc_amd.envs.ig_parkour.default_config import default_agent_config
        from parc_amd.learning.dm_ppo_agent import DMPPOAgent
        T, B = DMPPOAgent.rollout_shape(dict(default_agent_config(), mp_scale_rollout=(scaling == "reference")), world)
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": int(ranks.item()), "max_rank_seconds": tt.item(),
                          "backend": torch.distributed.get_backend() if world > 1 else None,
                          "scaling": "weak" if scaling == "weak" else "strong", "rollout_steps_per_rank": T, "minibatch_envs_multiple": B}))
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--workload", default="boxes_64clips")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profiler-child", action="store_true",
                    help="do not start the rocprofv3 child that times the post-step launch inside the rollout (roofline.us_per_launch then "
                         "comes from HIP events bound to the dispatch)")
    ap.add_argument("--grad-allreduce", default="minibatch", choices=["minibatch", "epoch"],
                    help="minibatch = the reference's cadence (default); epoch = one parameter exchange per PPO epoch (north-star)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "reference"],
                    help="weak (default): every rank runs the one-GPU workload (T = 32 steps, minibatches of 4 x envs rows) - per-GPU work fixed; "
                         "reference: the reference's own rule (base_agent.py:179-180, ppo_agent.py:27-29): T = ceil(32 / P), minibatch = "
                         "ceil(4 / P) x envs rows per rank - what parc_3_tracker.py --num_workers P runs unchanged, total work fixed")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, rendezvous, barrier and MAX-reduce a timer (no GPU work, no metric): checks the launcher")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))          # `python bench.py --gpus N`: this process only starts the N ranks (no GPU call here)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus {} but WORLD_SIZE={}: launch with torch.distributed.run --nproc-per-node {}".format(
            args.gpus, world, args.gpus))
    backend = os.environ.get("PARC_DIST_BACKEND", "nccl")
    if args.launch_check:
        return launch_check(rank, world, backend, args.scaling)
    n_dev = torch.cuda.device_count()
    if n_dev < world and backend == "nccl":
        sys.exit("bench.py: --gpus {} needs {} visible GPUs, found {}".format(args.gpus, world, n_dev))
    dev = "cuda:{}".format(local_rank % max(n_dev, 1))     # (gloo rehearsal on a 1-GPU box: ranks share the device)
    torch.cuda.set_device(dev)
    from parc_amd import _hip, workloads
    from parc_amd.util import mp_util
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    mp_util.init(rank, world, dev)
    torch.manual_seed(0 + 41 * rank)       # run.py:90 of the reference
    np.random.seed(41 * rank)

    env, clips, tiled = workloads.build_env(args.workload, args.envs, dev, seed=0)
    agent = workloads.build_agent(env, dev, mp_scale_rollout=(args.scaling == "reference"))
    agent._optimizer._cadence = args.grad_allreduce
    N, T = env.get_num_envs(), agent._steps_per_iter
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()

    def barrier():
        if world > 1:
            if torch.distributed.get_backend() == "nccl":
                torch.distributed.barrier(device_ids=[torch.cuda.current_device()])
            else:
                torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        agent._train_iter()
    if args.warmup == 0 and agent._use_hip_graph:
        agent._use_hip_graph = False        # nothing warmed up: a capture inside the timed region would be timed
    rollout_s = [0.0]
    orig_rollout = agent._rollout_train

    def timed_rollout(n):
        torch.cuda.synchronize()
        t = time.time()
        orig_rollout(n)
        torch.cuda.synchronize()
        rollout_s[0] += time.time() - t
    agent._rollout_train = timed_rollout

    barrier()
    t0 = time.time()
    info = None
    for _ in range(args.steps):
        info = agent._train_iter()
    barrier()
    elapsed = time.time() - t0
    agent._rollout_train = orig_rollout
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()

    # dominant env kernel: the fused post-step launch (heightmap gather + reference pose + obs + reward + done).
    # Inside the timed region it runs as a node of the captured rollout graph, where a single node cannot be bracketed
    # with events; so (a) one extra EAGER rollout right after the region with an event pair around every launch, in the
    # loop's real context, and (b) 200 back-to-back launches (the figure the roofline uses).
    env._core.timing_events = []
    workloads.eager_rollout_like_the_graph(agent, T)
    torch.cuda.synchronize()
    pairs = env._core.timing_events
    env._core.timing_events = None
    # THE roofline figure: the launch where the product issues it - behind the simulator step of a rollout step, its clip rows and
    # destination lines evicted by the policy GEMMs in between -, by a pair of events bound to the dispatch itself (hipExtLaunchKernel:
    # the kernel's own begin / end time stamps, the ones rocprofv3 reports; an event pair recorded AROUND a launch reads ~5 us more)
    in_rollout = [pr.elapsed_us() for pr in pairs]
    kern_events_us = float(np.mean(in_rollout)) if in_rollout else float("nan")
    evs = pairs
    prof = None
    if rank == 0 and world == 1 and not args.no_profiler_child:
        prof = post_step_in_rollout_by_the_profiler(args.workload, N)
    kern_us = prof["full_us_mean"] if prof else kern_events_us
    # the product step's flags (the reference STATE is published by the step's tail launch, parc_step_tail, since round 3)
    full = _hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF | _hip.POST_TARGETS
    for _ in range(10):
        env._core.post_step(full)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200):
        env._core.post_step(full)
    e.record()
    torch.cuda.synchronize()
    kern_b2b_us = s.elapsed_time(e) * 1e3 / 200
    # standalone: 200 launches captured in one hipGraph, replayed, bracketed by events on the replay stream (warm caches, nothing
    # else on the chip; what profiles/rNN_post_step_*_kernel_stats.csv measures) - reported as frac_standalone, NOT the roofline figure
    kern_graph_us = kern_b2b_us
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(200):
                env._core.post_step(full)
        g.replay()
        torch.cuda.synchronize()
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        kern_graph_us = s.elapsed_time(e) * 1e3 / 200
        del g
    except Exception:
        pass
    alg_bytes = N * POST_STEP_BYTES_PER_ENV
    achieved = alg_bytes / (kern_us * 1e-6) / 1e9
    achieved_standalone = alg_bytes / (kern_graph_us * 1e-6) / 1e9
    # HBM traffic per launch and the VALU instruction count come from the PMC passes of tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE in
    # separate passes, gfx950-corrected; SQ_INSTS_VALU), committed under profiles/: counters cannot be read from inside this process, so
    # these two are PROFILE CONSTANTS valid for exactly this workload and env count (null otherwise), and are labelled as such.
    def committed_profile(workload, what):
        for rnd in ("r04", "r03", "r02", "r01"):
            for name in ("{}_post_step_{}_{}.json".format(rnd, workload, what), "{}_post_step_{}.json".format(rnd, what)):
                pth = os.path.join(ROOT, "profiles", name)
                if os.path.exists(pth) and (workload in name or workload == "boxes_64clips"):
                    with open(pth) as f:
                        return json.load(f), "profiles/" + name
        return None, None
    traffic = traffic_src = None
    tj, tsrc = committed_profile(args.workload, "pmc_traffic")
    if tj is not None and tj.get("envs") == N and tj.get("workload", "boxes_64clips") == args.workload:
        traffic, traffic_src = tj["traffic_bytes_corrected"], tsrc
    # the kernel's arithmetic intensity sits above the VALU/HBM ridge, so vector issue is its binding roofline: reported
    # next to the HBM figures (1024 SIMDs, 4 cycles per wave64 instruction, 2.4 GHz)
    valu = None
    sj, ssrc = committed_profile(args.workload, "sq_counters")
    if sj is not None and N == 4096:
        n_valu = (sj.get("per_dispatch_mean") or {k: v["mean"] for k, v in sj.items() if isinstance(v, dict) and "mean" in v}).get("SQ_INSTS_VALU")
        if n_valu:
            issue_us = n_valu * 4.0 / (1024 * 2.4e3)
            valu = {"wave_instructions_per_launch": n_valu, "source": ssrc + " (committed profile, not measured in this run)",
                    "lane_ops_per_algorithmic_byte": n_valu * 64.0 / alg_bytes,
                    "ridge_lane_ops_per_byte": 1024 * 16 * 2.4e9 / (HBM_PEAK_GBPS * 1e9), "issue_bound_us": issue_us,
                    "frac_of_valu_issue_peak": issue_us / kern_us, "frac_of_valu_issue_peak_standalone": issue_us / kern_graph_us}

    def time_launches(fn, iters):
        """us per launch: `iters` launches captured in one hipGraph and replayed between two events (kernel time without the host's
        dispatch gaps, like the roofline figure above); eager back-to-back launches if the capture is refused"""
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(iters):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            del g
        except Exception:
            torch.cuda.synchronize()
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / iters

    # what a plain device copy reaches on this box (the nominal 8 TB/s next to the achievable figure): 1 GiB read + 1 GiB written
    copy_gbps = None
    if rank == 0:
        try:
            src_ = torch.empty(256 << 20, dtype=torch.float32, device=dev)
            dst_ = torch.empty_like(src_)
            us_c = time_launches(lambda: dst_.copy_(src_), 10)
            copy_gbps = 2.0 * src_.numel() * 4 / us_c / 1e3
            del src_, dst_
        except Exception:           # noqa: BLE001
            copy_gbps = None
    extra = []
    if rank == 0:
        # the other hand-written kernels of the step, same HIP-event timing (context for the roofline object)
        c, L = env._core, _hip.lib()
        P = c.ray_xy_points.shape[0]
        for nn in (N, 65536):
            obs_big = c.obs if nn == N else torch.zeros((nn, c.cfg.obs_dim), device=dev)
            rs_big = c.root_state if nn == N else c.root_state.repeat((nn + N - 1) // N, 1)[:nn].contiguous()
            eo_big = c.env_offsets if nn == N else c.env_offsets.repeat((nn + N - 1) // N, 1)[:nn].contiguous()
            dst = _hip.c_vp(obs_big.data_ptr() + 4 * (c.cfg.obs_dim - P))
            us = time_launches(lambda: L.parc_refresh_obs_hfs(_hip.stream(), nn, _hip.ptr(c.ray_xy_points), P, _hip.ptr(rs_big), _hip.ptr(eo_big),
                                                               c._terrain_struct, c.cfg.struct.min_obs_h, c.cfg.struct.max_obs_h, dst,
                                                               c.cfg.obs_dim), 200 if nn == N else 50)
            byts = nn * 3544
            extra.append({"kernel": "hf_gather_kernel (K5 alone)", "envs": nn, "us_per_launch": us, "bound": "hbm", "algorithmic_bytes": byts,
                          "achieved_GBps": byts / us / 1e3, "frac_of_peak": byts / us / 1e3 / HBM_PEAK_GBPS,
                          "note": ("the north-star's >= 40 % of HBM peak on the heightmap gather is NOT met by the standalone kernel at this env "
                                   "count (a 14.5 MB launch is launch-bound: the same grid without gathers or stores takes 2.8 us); the product "
                                   "never launches K5 alone - it runs inside track_post_kernel, the roofline object above") if nn == N else
                                  "the standalone kernel where the launch is large enough to be bandwidth-bound"})
            del obs_big, rs_big, eo_big
        # the same fused kernel where its inputs do NOT sit in L2: the iter-0 stand-in of BASELINE configs[3] (1024 clips = 83 MB of clip
        # rows, 1504^2 heightfield = 9 MB), every env on its own clip / tile
        try:
            core0, clips0, tiled0 = workloads.build_core("iter0_1024clips", N, dev)
            us0 = time_launches(lambda: core0.post_step(full), 200)
            tj0, tsrc0 = committed_profile("iter0_1024clips", "pmc_traffic")
            t0b = tj0["traffic_bytes_corrected"] if tj0 is not None and tj0.get("envs") == N and tj0.get("workload") == "iter0_1024clips" else None
            extra.append({"kernel": "track_post_kernel on iter0_1024clips (BASELINE configs[3] stand-in: clip rows {:.0f} MB + heightfield {:.1f} MB, "
                                    "not L2-resident)".format(sum(c_["frames"].shape[0] for c_ in clips0) * 448 / 1e6, tiled0[0].nbytes / 1e6),
                          "envs": N, "us_per_launch": us0, "bound": "hbm", "algorithmic_bytes": alg_bytes, "achieved_GBps": alg_bytes / us0 / 1e3,
                          "frac_of_peak": alg_bytes / us0 / 1e3 / HBM_PEAK_GBPS, "traffic_bytes_per_launch": t0b,
                          "traffic_source": (tsrc0 + " (committed profile)") if t0b else None,
                          "frac_of_peak_measured_traffic": (t0b / us0 / 1e3 / HBM_PEAK_GBPS) if t0b else None})
            del core0, clips0, tiled0
        except Exception as exc:                      # a context line only: never let it take the metric down
            extra.append({"kernel": "track_post_kernel on iter0_1024clips", "error": repr(exc)})
        act = torch.zeros((N, env._sim_model.struct.dof_size), device=dev)
        state = (c.root_state.clone(), c.dof_state.clone())
        us = time_launches(lambda: L.parc_sim_step(_hip.stream(), env._sim_model.device_ptr(dev), c._terrain_struct, N, _hip.ptr(c.root_state),
                                                   _hip.ptr(c.dof_state), _hip.ptr(c.rigid_body_state), _hip.ptr(c.contact_forces),
                                                   _hip.ptr(c.env_offsets), _hip.ptr(act), _hip.ptr(env._action_bound_low),
                                                   _hip.ptr(env._action_bound_high), env._sim_steps * env._substeps, env._sim_h), 20)
        c.root_state.copy_(state[0])
        c.dof_state.copy_(state[1])
        extra.append({"kernel": "sim_step_bpl_kernel (articulated-body step, {} substeps)".format(env._sim_steps * env._substeps), "envs": N,
                      "us_per_launch": us, "us_per_launch_is": "20 launches replayed in one graph from the end-of-bench state with zero actions "
                                                               "(contact-rich: slower than in the rollout)",
                      "us_per_launch_in_rollout": prof["sim_step_us_mean"] if prof else None,
                      "bound": "instruction issue (one 256-VGPR wave per SIMD, VALU active 48 % of cycles); state traffic 1624 B/env",
                      "achieved_GBps": N * 1624 / us / 1e3})
        # the update phase is fp32 GEMMs (81 % of the iteration): the two largest shapes of a PPO minibatch against the dense fp32 MFMA peak
        mb = agent._batch_size * N
        for (m_, k_, n_, what) in ((mb, c.cfg.obs_dim, 2048, "layer-1 forward"), (mb, 2048, 1024, "layer-2 forward")):
            a_ = torch.randn((m_, k_), device=dev)
            w_ = torch.randn((n_, k_), device=dev)
            b_ = torch.zeros(n_, device=dev)
            us_g = time_launches(lambda: torch._addmm_activation(b_, a_, w_.t(), use_gelu=False), 10)
            tf = 2.0 * m_ * k_ * n_ / us_g / 1e6
            extra.append({"kernel": "policy / value MLP {} GEMM + bias + ReLU, {}x{}x{} (hipBLASLt fp32)".format(what, m_, k_, n_), "us_per_launch": us_g,
                          "bound": "mfma", "achieved_TFLOPs": tf, "peak_TFLOPs": MFMA_FP32_PEAK_TFLOPS, "frac_of_peak": tf / MFMA_FP32_PEAK_TFLOPS})
            del a_, w_, b_

    replay_bound = None
    try:
        ml_ = env.get_dm_env()._motion_lib
        w_ = ml_._motion_weights.double()
        steps_ = torch.clamp(ml_._motion_lengths.double() * 0.5 / env._timestep, max=float(env._cfg.struct.episode_length) / env._timestep)
        replay_bound = float((w_ * steps_).sum() / w_.sum())
    except Exception:           # noqa: BLE001
        replay_bound = None
    if rank == 0:
        total_env_steps = world * N * T * args.steps
        out = {
            "metric": "env-steps/sec (whole node), {} envs/GPU humanoid tracker at 1/2/4/8 MI355X".format(N),      # BASELINE.json's metric
            "value": total_env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            # weak: per-GPU work fixed; --scaling reference: the reference's rule, total samples per iteration fixed = "strong"
            "scaling": "weak" if args.scaling == "weak" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "{} envs/GPU tracker on procgen box heightfields, 64 synthetic clips (BASELINE.json configs[2])".format(N)
                       if args.workload == "boxes_64clips" else args.workload,
                       "envs_per_gpu": N, "env_steps_per_bench_step": N * T, "rollout_steps": T, "update_epochs": agent._update_epochs,
                       "minibatch": agent._batch_size * N, "sim_substeps": env._sim_steps * env._substeps, "parallelism": "dp{}".format(world),
                       "scaling_rule": ("per-rank work fixed (T = 32, minibatch = 4 x envs): weak scaling" if args.scaling == "weak" else
                                        "the reference's rule: T = ceil(32 / P) = {}, minibatch = ceil(4 / P) x envs = {} rows per rank "
                                        "(base_agent.py:179-180, ppo_agent.py:27-29)".format(T, agent._batch_size * N)),
                       "grad_allreduce": "per minibatch (reference cadence)" if args.grad_allreduce == "minibatch"
                       else "per PPO epoch (parameter + momentum averaging)"},
            "roofline": {"kernel": "track_post_kernel (fused K5 heightmap gather + K3 K2 K4 K6-K10)", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "achieved_is": "ALGORITHMIC bytes per launch (SURVEY.md 8d: 13 592 B/env) / us_per_launch = the launch IN THE ROLLOUT "
                                        "STEP, where the product issues it (behind the simulator launch; its clip rows and destination lines "
                                        "were evicted by the policy GEMMs since the previous step)",
                         "us_per_launch": kern_us,
                         "us_per_launch_source": ("dispatch begin -> end time stamps of {} launches: a child process ran the step's launch "
                                                  "sequence eagerly under rocprofv3 --kernel-trace during this run (tools/rollout_only.py "
                                                  "--eager, tools/rollout_trace_stats.py; the method of profiles/r04_rollout_kernel_stats.csv)"
                                                  .format(prof["full_launches"])) if prof else
                                                 "HIP events bound to the dispatch (hipExtLaunchKernel), this process; reads ~1 us more than the "
                                                 "dispatch's own begin -> end (rocprofv3 not available or --no-profiler-child)",
                         "profiler_child": prof,
                         "us_per_launch_by_events_bound_to_the_dispatch": kern_events_us,
                         "us_per_launch_by_events_min": float(np.min(in_rollout)) if in_rollout else None,
                         "us_per_launch_by_events_max": float(np.max(in_rollout)) if in_rollout else None,
                         "launches_event_timed": len(evs), "algorithmic_bytes_per_launch": alg_bytes,
                         "peak_measured_device_copy_GBps": copy_gbps,
                         "frac_of_measured_device_copy": (achieved / copy_gbps) if copy_gbps else None,
                         "frac_standalone": achieved_standalone / HBM_PEAK_GBPS, "us_per_launch_standalone": kern_graph_us,
                         "standalone_is": "200 identical launches replayed in one hipGraph: warm caches, nothing else on the chip "
                                          "(profiles/r04_post_step_boxes_64clips_kernel_stats.csv)",
                         "us_per_launch_eager_back_to_back": kern_b2b_us,
                         "traffic_source": (traffic_src + " (committed profile of the standalone launch, not measured in this run)") if traffic else None,
                         "frac_measured_traffic": (traffic / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "binding_roofline": "HBM by SURVEY's classification; measured: neither HBM (frac_measured_traffic) nor vector issue "
                                             "(valu.frac_of_valu_issue_peak) is saturated - the launch is bound by the dependent chains of its pose waves "
                                             "(6 waves per workgroup, 70 VGPRs) and the drain of its row stores (DESIGN.md section 3)",
                         "valu": valu},
            "rollout": ("one hipGraph replay per env step" + (", finished envs reset on the device inside the graph" if any(k[2] for k in agent._graphs)
                                                               else " + eager reset of finished envs")) if agent._graphs else "eager",
            "kernels": extra,
            "rollout_env_steps_per_s": world * N * T * args.steps / max(rollout_s[0], 1e-9),
            "rollout_fraction_of_time": rollout_s[0] / elapsed,
            "mean_episode_return": info["mean_return"] if info else None,
            # the return is reported absolute (no Isaac Gym run exists to compare with), next to the kinematic-replay upper
            # bound: a character that sits on the reference pose earns reward 1 per step (tests/test_env_gpu.py::
            # test_kinematic_replay_reward_upper_bound) from its uniformly drawn start phase to the end of its clip
            "mean_episode_return_kinematic_upper_bound": replay_bound,
            "mean_episode_length": info["mean_ep_len"] if info else None,
        }
        if not args.no_cpu_baseline and world == 1:       # (the CPU leg is a single-GPU datum: rank 0 at N=1 only)
            out["cpu_baseline"] = cpu_baseline(env, clips, tiled)
        print(json.dumps(out))
    if world > 1:
        barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
