"""Where do the full-size oracle outliers come from?  (round-2 review, "what's weak" 1)

Runs the failing case of round 2 -- iter0_1024clips at 4096 envs, reset + 3 steps -- and, for EVERY env (not a 64-env slice),
compares the device's reference pose / observation with the CPU oracle at the tight tolerance, then explains each outlier:
  * are the clip rows the device built equal to the oracle's, bit for bit? (a 1-ulp difference in a stored quaternion is the only way
    the two sides can see a different slerp cosine: both evaluate the cosine op by op in the same order)
  * per outlier element: the slerp cosine (in ulps below 1) of every quaternion the element depends on.
slerp (util/torch_util.py:443-468) is discontinuous at `cos >= 1 -> q0` (k = 0 | 1 ulps below one) and at `sin < 1e-3 -> average`
(k = 8 | 9: 1 - c*c = 2k * 2^-24 exactly for small k, 16 * 2^-24 = 9.5e-7 < 1e-6 < 18 * 2^-24).

    python tools/slerp_outliers.py [--workload iter0_1024clips] [--envs 4096] [--out gpurun_out/slerp_outliers.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="iter0_1024clips")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--out", default="gpurun_out/slerp_outliers.json")
    a = ap.parse_args()
    import smoke_impl
    from oracle import oracle as orc
    from parc_amd import workloads
    dev = "cuda:0"
    torch.manual_seed(0)
    env, clips, tiled = workloads.build_env(a.workload, a.envs, dev, seed=0)
    obs, info = env.reset()
    lo, hi = env._action_bound_low, env._action_bound_high
    mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
    for _ in range(3):
        obs, r, done, info = env.step(mid + 0.2 * half * torch.randn((a.envs, 28), device=dev))
    torch.cuda.synchronize()
    c = env._core
    char, mlib = smoke_impl.oracle_models(env, clips)
    rep = {"workload": a.workload, "envs": a.envs}
    # ---- 1. stored frames: device rows vs oracle arrays
    rows = c.mlib._rows.cpu().numpy()
    B = 15
    dq = np.concatenate([mlib.root_rot[:, None, :], mlib.joint_rot], axis=1).reshape(-1, 4 * B)
    gq = rows[:, 0:4 * B]
    diff = gq.view(np.int32).astype(np.int64) - dq.view(np.int32).astype(np.int64)
    rep["stored_quat_components"] = int(dq.size)
    rep["stored_quat_components_differing"] = int((diff != 0).sum())
    rep["stored_quat_max_ulp_diff"] = int(np.abs(diff).max())
    rep["stored_frames_with_a_differing_quat"] = int((diff != 0).any(axis=1).sum())
    rep["stored_frames"] = int(dq.shape[0])
    # ---- 2. every env through the comparison, collecting the explanation instead of asserting
    st = smoke_impl.oracle_compare(env, clips, tiled, obs, r, ids=None, report=True)
    rep.update(st)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({k: v for k, v in rep.items() if not isinstance(v, list)}, indent=1))


if __name__ == "__main__":
    main()
