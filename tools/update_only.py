#!/usr/bin/env python3
"""PPO update phase only (one rollout to fill the buffer, then N x (_build_train_data + _update_model)) for profiling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parc_amd import workloads
from parc_amd.util import mp_util

dev = "cuda:0"
mp_util.init(0, 1, dev)
torch.manual_seed(0)
env, _, _ = workloads.build_env("boxes_64clips", 4096, dev, seed=0)
agent = workloads.build_agent(env, dev, mp_scale_rollout=False)
agent._curr_obs, agent._curr_info = env.reset()
agent._init_train()
agent._train_iter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.cuda.synchronize()
t0 = time.time()
for _ in range(n):
    agent._build_train_data()
    agent._update_model()
torch.cuda.synchronize()
print("update phase ms %.1f" % ((time.time() - t0) / n * 1e3))
