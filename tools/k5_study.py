#!/usr/bin/env python3
"""Standalone heightmap gather (K5, hf_gather_kernel) at the graded size: time per launch of every workgroup shape, next to the time
of the SAME grid with neither gathers nor stores (what a launch of that shape costs before a byte moves).  Launches are captured in
one hipGraph and replayed between two events, like bench.py's roofline figure.  One JSON line per variant."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd import _hip  # noqa: E402
from parc_amd.util import geom_util  # noqa: E402


def graph_time(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best


def main():
    dev = "cuda:0"
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    P = 441
    g = torch.Generator().manual_seed(0)
    rays = geom_util.get_xy_points_cone(torch.zeros(2), 0.05, 2, 60, 3, 3, 0.26179938779).to(dev)
    hf = (torch.rand((144, 144), generator=g) * 2.0).to(dev)
    ter = _hip.terrain_struct(hf, [-28.8, -28.8], [0.4, 0.4])
    rs = torch.zeros((n, 13))
    rs[:, 0:2] = (torch.rand((n, 2), generator=g) - 0.5) * 50.0
    rs[:, 2] = 0.9
    q = torch.randn((n, 4), generator=g)
    rs[:, 3:7] = q / q.norm(dim=-1, keepdim=True)
    rs = rs.to(dev)
    eo = torch.zeros((n, 3), device=dev)
    obs = torch.zeros((n, 1312), device=dev)
    dst = _hip.c_vp(obs.data_ptr() + 4 * 871)
    import parc_diag
    L = parc_diag.lib()          # the knobs exist in the diagnostics library only

    def k5():
        L.parc_refresh_obs_hfs(_hip.stream(), n, _hip.ptr(rays), P, _hip.ptr(rs), _hip.ptr(eo), ter, -3.0, 3.0, dst, 1312)
    byts = n * 3544
    ref = None
    for groups in (1, 2, 4, 8):
        L.parc_tune_hf_groups(groups)
        L.parc_tune_hf_ablation(0)
        obs.zero_()
        k5()
        torch.cuda.synchronize()
        if ref is None:
            ref = obs.clone()
        same = bool(torch.equal(obs, ref))
        us = graph_time(k5)
        L.parc_tune_hf_ablation(5)
        us_empty = graph_time(k5)
        L.parc_tune_hf_ablation(0)
        print(json.dumps({"kernel": "hf_gather_kernel", "envs": n, "workgroups": (n + 2 * groups - 1) // (2 * groups), "threads": 128 * groups,
                          "us_per_launch": round(us, 3), "us_same_grid_no_gather_no_store": round(us_empty, 3), "frac_of_8TBps": round(byts / us / 1e3 / 8000, 4),
                          "bit_identical_to_default": same}), flush=True)
    L.parc_tune_hf_groups(1)


if __name__ == "__main__":
    main()
