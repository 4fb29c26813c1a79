#!/usr/bin/env python3
"""Short training run on a synthetic workload: prints mean episode return / length / losses per iteration
(evidence that the from-scratch simulator supports learning the tracking task)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--workload", default="flat_1clip")
    ap.add_argument("--out", default="")
    ap.add_argument("--save", default="", help="write the agent's state (model + normalisers) and the run counters here at the end")
    ap.add_argument("--resume", default="", help="continue from a state written by --save (momentum starts from zero, like the reference's resume)")
    ap.add_argument("--max-seconds", type=float, default=0.0, help="stop early when this much wall time has passed (a GPU call is time-limited)")
    args = ap.parse_args()
    from parc_amd import workloads
    from parc_amd.util import mp_util
    dev = "cuda:0"
    mp_util.init(0, 1, dev)
    torch.manual_seed(0)
    env, _, _ = workloads.build_env(args.workload, args.envs, dev, seed=0)
    agent = workloads.build_agent(env, dev)
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    it0, wall0, samples0 = 0, 0.0, 0
    if args.resume:
        agent.load(args.resume)
        meta = json.load(open(args.resume + ".json"))
        it0, wall0 = int(meta["iters_done"]), float(meta["wall_s"])
        samples0 = agent._sample_count = int(meta["samples"])        # the exploration anneal and the normaliser freeze read it
        if "fail_rates" in meta and env.has_dm_envs():
            env.get_dm_env()._motion_id_fail_rates[:] = torch.tensor(meta["fail_rates"], device=dev)
    t0 = time.time() - wall0
    rows = []
    for it in range(it0, it0 + args.iters):
        if args.max_seconds > 0 and time.time() - t0 - wall0 > args.max_seconds:
            break
        info = agent._train_iter()
        agent._sample_count = samples0 + agent._update_sample_count()
        row = {"iter": it, "samples": agent._sample_count, "mean_return": info["mean_return"], "mean_ep_len": info["mean_ep_len"],
               "episodes": info["num_eps"], "critic_loss": float(info["critic_loss"]), "actor_loss": float(info["actor_loss"]),
               "clip_frac": float(info["clip_frac"]), "pose_r": info["pose_r"], "root_pos_r": info["root_pos_r"], "wall_s": time.time() - t0}
        rows.append(row)
        if it % 5 == 0 or it == it0 + args.iters - 1:
            print(json.dumps(row), flush=True)
        if it % 10 == 9:
            agent._train_return_tracker.reset()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f)
    if args.save and rows:
        agent.save(args.save)
        meta = {"iters_done": rows[-1]["iter"] + 1, "samples": int(agent._sample_count), "wall_s": rows[-1]["wall_s"]}
        if env.has_dm_envs():
            meta["fail_rates"] = env.get_dm_env()._motion_id_fail_rates.tolist()
        json.dump(meta, open(args.save + ".json", "w"))


if __name__ == "__main__":
    main()
