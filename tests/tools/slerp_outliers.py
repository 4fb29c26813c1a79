"""Where do the full-size oracle outliers come from?  (round-2 review, "what's weak" 1)

Runs the failing case of round 2 -- iter0_1024clips at 4096 envs, reset + 3 steps -- and, for EVERY env (not a 64-env slice),
compares the device's reference pose / observation with the CPU oracle at the tight tolerance, then explains each outlier:
  * with the oracle sampling ITS OWN stored frames: how many elements are beyond 5e-5, and does each depend on a quaternion whose slerp
    cosine sits at a discontinuity?  (the two sides store frame quaternions built with two maths libraries: 1-ulp differences, which
    is the only way they can see a different cosine - both evaluate it op by op in the same order)
  * with the oracle sampling the frames the DEVICE stored (checked against its own at 5e-7): nothing may be beyond 5e-5.
slerp (util/torch_util.py:443-468) is discontinuous at `cos >= 1 -> q0` (k = 0 | 1 ulps below one) and at `sin < 1e-3 -> average`
(k = 8 | 9: 1 - c*c = 2k * 2^-24 exactly for small k, 16 * 2^-24 = 9.5e-7 < 1e-6 < 18 * 2^-24).

    python tests/tools/slerp_outliers.py [--workload iter0_1024clips] [--envs 4096] [--out gpurun_out/slerp_outliers.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


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
    # ---- 1. the oracle on ITS OWN stored frames: every element beyond the tight tolerance, and what it depends on
    rep["oracle_on_its_own_frames"] = smoke_impl.oracle_compare(env, clips, tiled, obs, r, ids=None, report=True, oracle_frames="own")
    # ---- 2. the oracle on the frames the DEVICE stored (checked against its own first): nothing may be beyond the tight tolerance
    rep["oracle_on_device_frames"] = smoke_impl.oracle_compare(env, clips, tiled, obs, r, ids=None, report=True, oracle_frames="device")
    own, devf = rep["oracle_on_its_own_frames"], rep["oracle_on_device_frames"]
    keys = [k for k, v in own.items() if isinstance(v, dict) and "beyond_tight" in v]
    rep["summary"] = {
        "tight_tolerance": smoke_impl.TIGHT,
        "own_frames: elements beyond tight": {k: own[k]["beyond_tight"] for k in keys if own[k]["beyond_tight"]},
        "own_frames: of them NOT downstream of a quaternion at a slerp discontinuity": sum(own[k]["beyond_tight_not_at_a_discontinuity"] for k in keys),
        "own_frames: largest error": max(own[k]["max_err"] for k in keys),
        "device_frames: elements beyond tight": sum(devf[k]["beyond_tight"] for k in keys),
        "device_frames: largest error": {k: devf[k]["max_err"] for k in keys},
        "done_mismatches": devf["done"]["mismatches"], "done_marginal": devf["done"]["marginal_decisions"],
    }
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep["summary"], indent=1))
    print(json.dumps(devf["stored_frames_device_vs_oracle"], indent=1))


if __name__ == "__main__":
    main()
