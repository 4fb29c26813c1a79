"""Where and why a trained tracking policy loses its clip: every env starts the clip at t = 0 with the deterministic policy, and at the step an
episode ends the termination inputs are read back - which body left its pose tolerance (or the root distance / rotation, or a fall contact),
how far, at which clip time.   python tools/fail_probe.py <checkpoint> [workload] [envs]   -> one JSON summary"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd import workloads  # noqa: E402
from parc_amd.envs import base_env  # noqa: E402
from parc_amd.learning.dm_ppo_agent import AgentMode  # noqa: E402
from parc_amd.util import mp_util, torch_util  # noqa: E402


def main():
    ckpt = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else "civ_clip"
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    dev = "cuda:0"
    mp_util.init(0, 1, dev)
    torch.manual_seed(0)
    env, _, _ = workloads.build_env(workload, N, dev, seed=0)
    agent = workloads.build_agent(env, dev)
    agent.load(ckpt)
    agent.eval()
    agent.set_mode(AgentMode.TEST)
    env.set_rand_reset(False)                       # every episode starts at the clip's first frame
    env.set_rand_root_pos_offset_scale(0.0)
    obs, info = env.reset()
    km = env._kin_char_model
    names = km.get_body_names()
    cfg = env._cfg.struct
    tol = [float(cfg.pose_termination_dist[i]) for i in range(len(names) - 1)]
    ended = torch.zeros(N, dtype=torch.bool, device=dev)
    rec = []
    max_steps = int(env.get_dm_env()._motion_lib._motion_lengths.max().item() * env._control_freq) + 5
    with torch.no_grad():
        for step in range(max_steps):
            a, _ = agent._decide_action(obs, info)
            obs, r, done, info = env.step(a)
            new = (done != base_env.DoneFlags.NULL.value) & ~ended
            if new.any():
                ids = new.nonzero().flatten()
                bp, rbp = env._char_rigid_body_pos[ids], env._ref_body_pos[ids]
                rel = (bp[:, 1:] - bp[:, 0:1]) - (rbp[:, 1:] - rbp[:, 0:1])
                dist = torch.linalg.vector_norm(rel, dim=-1)                                   # [k, 14]
                over = dist / torch.tensor(tol, device=dev)
                root_d = torch.linalg.vector_norm(bp[:, 0] - rbp[:, 0], dim=-1)
                root_a = torch_util.quat_diff_angle(env._char_root_rot[ids], env._ref_root_rot[ids]).abs()
                cf = torch.linalg.vector_norm(env._char_contact_forces[ids], dim=-1)
                for k, e in enumerate(ids.tolist()):
                    worst = int(over[k].argmax())
                    rec.append({"env": e, "step": step + 1, "clip_time": round((step + 1) / env._control_freq, 3), "done": int(done[e]),
                                "worst_body": names[worst + 1], "worst_over_tol": round(float(over[k, worst]), 3),
                                "worst_dist": round(float(dist[k, worst]), 3), "root_dist": round(float(root_d[k]), 3),
                                "root_angle": round(float(root_a[k]), 3), "root_h": round(float(bp[k, 0, 2]), 3), "ref_root_h": round(float(rbp[k, 0, 2]), 3),
                                "contact_bodies": [names[b] for b in (cf[k] > 0.1).nonzero().flatten().tolist()],
                                "return_so_far": None})
                ended |= new
            if ended.all():
                break
            # finished envs keep stepping (their state no longer matters); do not reset: one episode per env
    steps = np.array([x["step"] for x in rec])
    by_body = {}
    for x in rec:
        key = x["worst_body"] if x["worst_over_tol"] > 1.0 else ("root_dist" if x["root_dist"] > float(cfg.root_pos_termination_dist) else
                                                                 ("root_angle" if x["root_angle"] > float(cfg.root_rot_termination_angle) else "other"))
        by_body[key] = by_body.get(key, 0) + 1
    out = {"workload": workload, "envs": N, "episodes": len(rec), "clip_steps": max_steps - 5,
           "end_step_quantiles": [int(v) for v in np.quantile(steps, [0.0, 0.1, 0.5, 0.9, 1.0])] if len(rec) else None,
           "done_codes": {str(c): int((np.array([x["done"] for x in rec]) == c).sum()) for c in (1, 2, 3)},
           "first_violation": dict(sorted(by_body.items(), key=lambda kv: -kv[1])), "examples": rec[:6]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
