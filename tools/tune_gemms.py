#!/usr/bin/env python3
"""Offline GEMM solution selection (PyTorch TunableOp over hipBLASLt / rocBLAS) for the shapes of one training iteration.
Run on the GPU box; writes gpurun_out/tunableop_results_<envs>.csv (merge its Gemm lines into parc_amd/tunableop_results.csv to ship them).
Prints the iteration time before (default heuristics) and after (tuned selections)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tunable
from parc_amd import workloads
from parc_amd.util import mp_util

dev = "cuda:0"
mp_util.init(0, 1, dev)
torch.manual_seed(0)
N_ENVS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096          # python3 tools/tune_gemms.py [envs per GPU]
env, _, _ = workloads.build_env("boxes_64clips", N_ENVS, dev, seed=0)
agent = workloads.build_agent(env, dev, mp_scale_rollout=False, tuned_gemms=False)
agent._curr_obs, agent._curr_info = env.reset()
agent._init_train()


def timed(n):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        agent._train_iter()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


agent._train_iter()
print("default heuristics: %.1f ms / iteration" % timed(3), flush=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tunableop_results_{}.csv".format(N_ENVS))
os.makedirs(os.path.dirname(out), exist_ok=True)
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(out, False)
THOROUGH = len(sys.argv) > 2 and sys.argv[2] == "thorough"      # python3 tools/tune_gemms.py 4096 thorough
tunable.set_max_tuning_duration(120 if THOROUGH else 30)        # ms per candidate solution
tunable.set_max_tuning_iterations(60 if THOROUGH else 10)
agent._use_hip_graph = False               # tuning launches candidates eagerly
agent._graphs.clear()
agent._graph_pool = None
t0 = time.time()
agent._train_iter()
torch.cuda.synchronize()
print("tuning pass took %.1f s, %d shapes" % (time.time() - t0, len(tunable.get_results())), flush=True)
tunable.tuning_enable(False)
agent._use_hip_graph = True
agent._graph_warm = 0
agent._train_iter()
print("tuned selections:   %.1f ms / iteration" % timed(3), flush=True)
for r in tunable.get_results():
    print(r)
