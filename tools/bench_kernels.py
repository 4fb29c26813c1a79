#!/usr/bin/env python3
"""Micro-benchmark of individual hot-path kernels on one GPU (HIP events on torch's current stream, which is the
stream the kernels are launched on).  Prints one JSON line per kernel."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd import _hip  # noqa: E402
from parc_amd.util import geom_util  # noqa: E402


def time_loop(fn, iters, warmup=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters   # us per launch (back-to-back, includes inter-kernel gap)


def time_single(fn, iters, warmup=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--hf", type=int, default=144)
    args = ap.parse_args()
    dev = "cuda:0"
    n, P = args.envs, 441
    g = torch.Generator().manual_seed(0)
    rays = geom_util.get_xy_points_cone(torch.zeros(2), 0.05, 2, 60, 3, 3, 0.26179938779).to(dev)
    hf = (torch.rand((args.hf, args.hf), generator=g) * 2.0).to(dev)
    half = args.hf * 0.4 / 2
    ter = _hip.terrain_struct(hf, [-half, -half], [0.4, 0.4])
    root_state = torch.zeros((n, 13))
    root_state[:, 0:2] = (torch.rand((n, 2), generator=g) - 0.5) * (2 * half - 8)
    root_state[:, 2] = 0.9
    q = torch.randn((n, 4), generator=g)
    root_state[:, 3:7] = q / q.norm(dim=-1, keepdim=True)
    root_state = root_state.to(dev)
    env_off = torch.zeros((n, 3), device=dev)
    obs = torch.zeros((n, 1312), device=dev)
    dst = _hip.c_vp(obs.data_ptr() + 4 * 871)
    L = _hip.lib()

    def k5():
        L.parc_refresh_obs_hfs(_hip.stream(), n, _hip.ptr(rays), P, _hip.ptr(root_state), _hip.ptr(env_off), ter, -3.0, 3.0, dst, 1312)

    us_loop = time_loop(k5, args.iters)
    us_med, us_min = time_single(k5, 100)
    alg_bytes = n * (16 + P * 4 + P * 4)
    print(json.dumps({"kernel": "hf_gather_kernel", "envs": n, "us_per_launch_back_to_back": us_loop, "us_single_median": us_med,
                      "us_single_min": us_min, "algorithmic_bytes": alg_bytes, "GBps_back_to_back": alg_bytes / us_loop / 1e3}))


if __name__ == "__main__":
    main()
