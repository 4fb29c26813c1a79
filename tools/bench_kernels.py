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

# --plain: the PRODUCT library and only launches of the product's own flags (what profiles/ *_kernel_stats / pmc files are taken from).
# Anything else (knob sweeps, role ablations) runs on the diagnostics library, tools/parc_diag.py.
PLAIN = "--plain" in sys.argv                    # one launch variant per process, no sweeps
if not PLAIN or any(a.startswith("--ablate=") or a.startswith("--lds-pad=") for a in sys.argv):
    import parc_diag  # noqa: E402
    parc_diag.install()


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

    # empty-ish kernel floor: K5 on 1 env (one workgroup)
    def k5_one():
        L.parc_refresh_obs_hfs(_hip.stream(), 1, _hip.ptr(rays), P, _hip.ptr(root_state), _hip.ptr(env_off), ter, -3.0, 3.0, dst, 1312)
    print(json.dumps({"kernel": "hf_gather_kernel(1 env)", "us_per_launch_back_to_back": time_loop(k5_one, args.iters)}))
    for nn in (256, 1024, 2048, 4096):
        if nn <= n:
            def k5n(nn=nn):
                L.parc_refresh_obs_hfs(_hip.stream(), nn, _hip.ptr(rays), P, _hip.ptr(root_state), _hip.ptr(env_off), ter, -3.0, 3.0, dst, 1312)
            print(json.dumps({"kernel": "hf_gather_kernel", "envs": nn, "us_per_launch_back_to_back": time_loop(k5n, args.iters)}))
    for epb in () if PLAIN else (1, 2, 4, 8):
        L.parc_tune_hf_envs_per_block(epb)
        print(json.dumps({"kernel": "hf_gather_kernel", "epb": epb, "envs": n, "us_per_launch_back_to_back": time_loop(k5, args.iters)}))
    if not PLAIN:
        L.parc_tune_hf_envs_per_block(2)
    for ab in () if PLAIN else (1, 2, 3, 4, 5, 0):
        L.parc_tune_hf_ablation(ab)
        print(json.dumps({"kernel": "hf_gather_kernel", "ablation": ab, "us": time_loop(k5, args.iters)}))
    us_loop = time_loop(k5, args.iters)
    us_med, us_min = time_single(k5, 100)
    alg_bytes = n * (16 + P * 4 + P * 4)
    print(json.dumps({"kernel": "hf_gather_kernel", "envs": n, "us_per_launch_back_to_back": us_loop, "us_single_median": us_med,
                      "us_single_min": us_min, "algorithmic_bytes": alg_bytes, "GBps_back_to_back": alg_bytes / us_loop / 1e3}))


def bench_post_step(n, iters, workload="boxes_64clips"):
    """Fused post-physics pass on a named synthetic workload (default: the 64-clip box terrains of BASELINE configs[2]; --workload=
    iter0_1024clips: the configs[3] stand-in whose clip database and heightfield do not fit in L2)."""
    from parc_amd import workloads
    core, clips, (hf, mn, dxdy, offs) = workloads.build_core(workload, n, "cuda:0")
    M = len(clips)
    # the product step's flags (the reference STATE is published by the step's tail launch, parc_step_tail, since round 3)
    full = _hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF | _hip.POST_TARGETS
    abl = [int(a.split("=")[1], 0) for a in sys.argv if a.startswith("--ablate=")]
    if abl:                      # PMC runs of one role ablation: every launch of the process uses it (diagnostics library)
        full |= abl[0]
    def graph_us_fn(fn, n=200):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            for _ in range(n):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            s_.record()
            gr.replay()
            e_.record()
            torch.cuda.synchronize()
            best = min(best, s_.elapsed_time(e_) * 1e3 / n)
        return best

    def graph_us(flags, n=200):
        """us per launch of n launches captured in ONE hipGraph and replayed (best of 3): the product's launch mode, and the only
        meaningful one for the cheap variants - an eager loop of this kernel is host-bound at ~8 us per launch (ctypes call with 2 KB
        of by-value structs), whatever the kernel does"""
        for _ in range(3):
            core.post_step(flags)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            for _ in range(n):
                core.post_step(flags)
        gr.replay()
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            s_.record()
            gr.replay()
            e_.record()
            torch.cuda.synchronize()
            best = min(best, s_.elapsed_time(e_) * 1e3 / n)
        return best
    pads = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--lds-pad=")]
    if pads:
        # occupancy probe (diagnostics library): the static LDS of the kernel lets 4 workgroups of 6 waves share a CU; padding it
        # leaves 3 / 2 / 1, i.e. the 1024 workgroups of 4096 envs run in more than one round and their load / store phases overlap
        for pad in pads:
            assert parc_diag.lib().parc_tune_post_lds_pad(pad) == 0
            print(json.dumps({"lds_pad_bytes": pad, "us_graph_replay": round(graph_us(full), 2)}), flush=True)
        parc_diag.lib().parc_tune_post_lds_pad(0)
    us_fused = time_loop(lambda: core.post_step(full), iters)
    us_graph = graph_us(full)
    # --plain runs are what the profiler passes wrap: ONE launch variant per process, so that a kernel-stats mean or a counter mean
    # is the full launch's and not a blend (round 3's files mixed 1123 full with 803 no-heightmap launches)
    us_nohf = None if PLAIN else graph_us(full & ~_hip.POST_HF)
    # timing diagnostics (bits of `what` the kernel honours for this purpose only): 0x10000 / 0x20000 / 0x40000 drop the target /
    # reference / character waves after the barrier, 0x80000 the heightmap wave, 0x100000 returns at entry, 0x200000 returns in front of the
    # barrier, 0x400000 / 0x800000 end a target wave after its slerp / its tree walk
    for name, bits in () if PLAIN else (("launch_only", 0x100000), ("up_to_the_barrier", 0x200000), ("none", 0x70000), ("only_char", 0x30000),
                                                         ("only_ref", 0x50000), ("only_tar", 0x60000), ("no_tar", 0x10000), ("no_ref", 0x20000), ("no_char", 0x40000),
                                                         ("no_heightmap_wave", 0x80000), ("only_tar_no_heightmap_wave", 0xe0000),
                                                         ("only_tar_no_stores", 0xe0000 | 0x1000000), ("all_but_target_stores", 0x1000000), ("only_tar_up_to_slerp", 0xe0000 | 0x400000), ("only_tar_up_to_tree_walk", 0xe0000 | 0x800000),
                                                         ("only_ref_no_heightmap_wave", 0xd0000), ("only_char_no_heightmap_wave", 0xb0000)):
        print(json.dumps({"ablation": name, "us_graph_replay": round(graph_us(full | bits), 2)}))
    if not PLAIN:
        # the step's tail launch: fail-rate EMA alone, and with the reference state co-scheduled
        fr = torch.full((M,), 0.5, device="cuda:0")
        us_fr = graph_us_fn(lambda: core.update_fail_rates(fr, 0.01))
        us_tail = graph_us_fn(lambda: core.step_tail(fr, 0.01))
        us_state = graph_us_fn(lambda: core.post_step(_hip.POST_REF))
        print(json.dumps({"kernel": "step tail", "us_fail_rate_kernel_alone": round(us_fr, 2), "us_step_tail_kernel (fail rates + reference state)": round(us_tail, 2),
                          "us_ref_state_kernel_alone": round(us_state, 2)}))
    # algorithmic bytes per env: K5 3544 + K3 7*760 + state 456 + obs cols [0,871) 3484 + bodies 780 + 8 out
    alg = n * (3544 + 7 * 760 + 456 + 3484 + 780 + 8)
    print(json.dumps({"kernel": "track_post_kernel(fused hf)", "workload": workload, "clips": M, "hf_cells": list(hf.shape),
                      "clip_row_bytes": int(sum(c["frames"].shape[0] for c in clips)) * 448, "envs": n, "us_per_launch_eager_back_to_back": us_fused,
                      "us_per_launch": us_graph, "algorithmic_bytes": alg, "GBps": alg / us_graph / 1e3, "frac_of_8TBps": alg / us_graph / 1e3 / 8000.0,
                      "us_without_hf": us_nohf, "mean_reward": core.reward.mean().item(),
                      "done_frac": (core.done != 0).float().mean().item()}))


def bench_sdf():
    """points_hf_sdf: B clips x N body sample points against X x Y columns (one box SDF per pair)."""
    import json
    from parc_amd.util import terrain_util
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    for B, N, X, Y in ((32, 304 * 20, 31, 31), (64, 304 * 60, 45, 45)):
        pts = (torch.rand((B, N, 3), generator=g) * 12.0).to(dev)
        hf = torch.rand((B, X, Y), generator=g).to(dev)
        mbc = torch.zeros((B, 2), device=dev)
        dxdy = torch.tensor([0.4, 0.4], device=dev)
        for _ in range(3):
            terrain_util.points_hf_sdf(pts, hf, mbc, dxdy)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            terrain_util.points_hf_sdf(pts, hf, mbc, dxdy)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 20
        pairs = B * N * X * Y
        print(json.dumps({"kernel": "points_hf_sdf_kernel", "batch": B, "points": N, "cells": X * Y, "us_per_call": us,
                          "G_point_cell_pairs_per_s": pairs / us / 1e3}))


if __name__ == "__main__":
    if "--sdf" in sys.argv:
        bench_sdf()
        sys.exit(0)
    if "--post" in sys.argv:
        ns = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--envs=")] or [4096]
        wl = [a.split("=")[1] for a in sys.argv if a.startswith("--workload=")] or ["boxes_64clips"]
        for nn in ns:
            bench_post_step(nn, 300 if nn <= 8192 else 60, wl[0])
        sys.exit(0)
    main()
