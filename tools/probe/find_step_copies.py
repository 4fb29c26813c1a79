"""Which python lines of the rollout step issue small torch launches (copies, fills, casts) that could be folded away?  One eager step body
with the relevant Tensor methods wrapped to print their caller.  python tools/probe/find_step_copies.py"""
import os
import sys
import traceback

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
workloads.eager_rollout_like_the_graph(agent, 4)
torch.cuda.synchronize()
active = [False]


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        if active[0] and self.is_cuda:
            fr = [x for x in traceback.extract_stack()[:-1] if "/parc_amd/" in x.filename][-2:]
            print(name, tuple(self.shape), self.dtype, "|", " <- ".join("%s:%d %s" % (x.filename.split("/parc_amd/")[-1], x.lineno, x.name) for x in reversed(fr)), flush=True)
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)


for n in ("to", "contiguous", "copy_", "clone", "fill_", "zero_", "float", "int", "long", "__setitem__", "expand"):
    wrap(n)
active[0] = True
workloads.eager_rollout_like_the_graph(agent, 1)
active[0] = False
torch.cuda.synchronize()
