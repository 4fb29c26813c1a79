#!/usr/bin/env python3
"""Rollout-only loop (no PPO update) for profiling the per-env-step kernel mix: python3 tools/rollout_only.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parc_amd import workloads
from parc_amd.util import mp_util

# --diag=0x...: timing-ablation bits ORed into every full post-step launch of the rollout (diagnostics library; results are garbage)
_bits = [int(a.split("=")[1], 0) for a in sys.argv[1:] if a.startswith("--diag=")]
if _bits:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import parc_diag
    parc_diag.install(_bits[0])
sys.argv = [a for a in sys.argv if not a.startswith("--diag=")]
dev = "cuda:0"
mp_util.init(0, 1, dev)
torch.manual_seed(0)
_wl = ([a.split("=")[1] for a in sys.argv if a.startswith("--workload=")] or ["boxes_64clips"])[0]
_ne = int(([a.split("=")[1] for a in sys.argv if a.startswith("--envs=")] or ["4096"])[0])
sys.argv = [a for a in sys.argv if not a.startswith(("--workload=", "--envs="))]
env, _, _ = workloads.build_env(_wl, _ne, dev, seed=0)
agent = workloads.build_agent(env, dev, mp_scale_rollout=False)
rollout = agent._rollout_train
if "--eager" in sys.argv:            # the same launches without the hipGraph (rocprofv3's per-kernel durations are only trustworthy for
    sys.argv.remove("--eager")      # eager launches: it adds ~3 us to every kernel node of a replayed graph, profiles/r04_rocprof_graph_node_inflation.txt)
    rollout = lambda k: workloads.eager_rollout_like_the_graph(agent, k)
agent._curr_obs, agent._curr_info = env.reset()
agent._init_train()
rollout(8)          # warm-up + graph capture
torch.cuda.synchronize()
timed = "--timed" in sys.argv       # (with --eager) every full post-step launch through parc_track_post_step_timed, like bench.py does
if timed:
    sys.argv.remove("--timed")
    env._core.timing_events = []
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
t0 = time.time()
rollout(n)
torch.cuda.synchronize()
dt = time.time() - t0
if timed:
    us = [p.elapsed_us() for p in env._core.timing_events]
    print("post-step launches timed by events bound to the dispatch: n %d mean %.2f us min %.2f max %.2f" % (len(us), sum(us) / len(us), min(us), max(us)))
print("rollout steps/s %.1f  ms/step %.3f  env-steps/s %.0f" % (n / dt, dt / n * 1e3, n * _ne / dt))
