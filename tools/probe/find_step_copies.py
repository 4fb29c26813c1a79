"""Which python lines of the rollout step issue aten::copy_ / aten::fill_ (small launches that could be folded away)?  One eager step body
under torch.profiler with stacks.  python tools/probe/find_step_copies.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
workloads.eager_rollout_like_the_graph(agent, 4)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    workloads.eager_rollout_like_the_graph(agent, 1)
torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::zeros",
                   "aten::empty_like", "aten::empty", "aten::index_put_", "aten::uniform_", "aten::normal_", "aten::randn_like"):
        stack = [s for s in (ev.stack or []) if "/parc_amd/" in s or "/tools/" in s]
        print(ev.name, "|", " <- ".join(s.split("/parc_amd/")[-1] for s in stack[:3]))
