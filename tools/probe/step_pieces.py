"""Which part of the captured rollout step carries the >= 4.5 us per kernel node?  Each piece of the step is captured alone, 16 times in
one graph, and timed; the sum of the pieces is compared with the whole step.  python tools/probe/step_pieces.py"""
import json
import os
import sys

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
agent._rollout_train(6)
torch.cuda.synchronize()
env._info_snapshots = False
eb = agent._exp_buffer
eb.set_device_head(agent._head_t)
agent._in_graph_step = True


def timed(name, fn, reps=16):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    print(json.dumps({"piece": name, "us": round(best, 1)}), flush=True)
    return best


state = {}


def policy():
    state["a"], state["ai"] = agent._decide_action(agent._curr_obs, agent._curr_info)


policy()
a, ai = state["a"], state["ai"]
action = a.clone()


def pre():
    agent._record_data_pre_step(agent._curr_obs, agent._curr_info, a, ai)


def step():
    state["o"], state["r"], state["d"], state["i"] = env.step(action)


step()


def track():
    agent._train_return_tracker.update(state["i"], state["d"])


def post():
    agent._record_data_post_step(state["o"], state["r"], state["d"], state["i"])


def reset():
    env.reset_done(state["d"])


tot = 0.0
for name, fn in (("policy forward + action head", policy), ("record (pre step)", pre), ("env.step: simulator, post-step launch, step tail", step),
                 ("return tracker", track), ("record (post step)", post), ("reset_done (device-side restart)", reset)):
    tot += timed(name, fn)
whole = timed("the whole step body", lambda: agent._train_step_body(True))
print(json.dumps({"sum_of_pieces_us": round(tot, 1), "whole_step_us": round(whole, 1)}))
