"""How long the host spends inside hipGraphLaunch for the captured rollout step (is the replay of a 27-node graph an asynchronous submission,
or does the host feed the nodes one by one while the GPU runs them?).  python tools/probe/replay_host_time.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parc_amd import workloads  # noqa: E402
from parc_amd.util import mp_util  # noqa: E402

dev = "cuda:0"
mp_util.init(0, 1, dev)
torch.manual_seed(0)
env, _, _ = workloads.build_env("boxes_64clips", 4096, dev, seed=0)
agent = workloads.build_agent(env, dev, mp_scale_rollout=False)
agent._curr_obs, agent._curr_info = env.reset()
agent._init_train()
agent._rollout_train(8)
torch.cuda.synchronize()
(g, done), = [v for v in agent._graphs.values()][:1]
host, total = [], []
for _ in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e6)
    total.append((t2 - t0) * 1e6)
host.sort()
total.sort()
# back to back: 32 replays without a sync in between
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(32):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"graphs_captured": len(agent._graphs), "host_us_inside_replay_median": round(host[len(host) // 2], 1),
                  "replay_plus_sync_us_median": round(total[len(total) // 2], 1),
                  "32_replays_host_us_each": round((t1 - t0) * 1e6 / 32, 1), "32_replays_total_us_each": round((t2 - t0) * 1e6 / 32, 1)}))
