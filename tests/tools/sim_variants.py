"""Compiler-flag variants of the PRODUCT simulator kernel (sim_step_bpl_kernel, parc_sim.hip alone), timed and checked in one GPU call.

  python tests/tools/sim_variants.py build     # here (no GPU): tests/tools/_simvar/libsim_<name>.so
  python tests/tools/sim_variants.py run       # on the GPU box: us per 4096-env launch (20 launches replayed in one hipGraph, best of 3) and the
                                         # largest deviation from the g++ host build of the same equations after 3 env steps on 64 envs
"""
import ctypes
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "parc_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "tools", "_simvar")

VARIANTS = {
    "O3 with SLP vectorisation (the round-3 build)": "-O3",
    "O3 without SLP vectorisation (product build since round 4)": "-O3 -fno-slp-vectorize",
    "O2 without SLP": "-O2 -fno-slp-vectorize",
    "O3 without SLP, max-ILP scheduler": "-O3 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp",
    "O3 without SLP, no loop vectoriser": "-O3 -fno-slp-vectorize -fno-vectorize",
    "O3 without SLP, -ffp-contract=fast-honor-pragmas off (-ffp-contract=on)": "-O3 -fno-slp-vectorize -ffp-contract=on",
    "O3 without SLP, unroll threshold 400": "-O3 -fno-slp-vectorize -mllvm -unroll-threshold=400",
    "O3 without SLP + PARC_SIM_EXTRA (source-level variant under test)": "-O3 -fno-slp-vectorize " + os.environ.get("PARC_SIM_EXTRA", "-DPARC_SIM_VARIANT_NONE"),
}


def so_path(name):
    return os.path.join(OUT, "libsim_%s.so" % "".join(ch if ch.isalnum() else "_" for ch in name))


def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def one(item):
        name, flags = item
        subprocess.check_call([hipcc, "--offload-arch=gfx950"] + flags.split() + ["-std=c++17", "-fPIC", "-shared", "-o", so_path(name),
                                                                                  os.path.join(CSRC, "parc_sim.hip")])
        return name
    with ThreadPoolExecutor(max_workers=4) as ex:
        for name in ex.map(one, VARIANTS.items()):
            print("built", name, flush=True)


def run():
    import numpy as np
    import torch
    from oracle.sim_host import HostSim
    from parc_amd import _hip, _hip_sim, workloads
    from parc_amd.sim_model import SimModel
    dev = "cuda:0"
    env, _, _ = workloads.build_env("boxes_64clips", 4096, dev, seed=0)
    env.reset()
    a0 = torch.zeros((4096, 28), device=dev)
    for _ in range(3):
        env.step(a0)
    c = env._core
    state0 = (c.root_state.clone(), c.dof_state.clone())
    sm = SimModel(env._kin_char_model)
    n, steps = 64, 3
    rng = np.random.default_rng(3)
    hf = (rng.random((20, 20)) * 0.5).astype(np.float32)
    rs0 = np.zeros((n, 13), np.float32)
    rs0[:, 0:2] = rng.random((n, 2)) * 4.0 - 2.0
    rs0[:, 2] = 1.0 + 0.3 * rng.random(n)
    rs0[:, 6] = 1.0
    ds0 = (rng.standard_normal((n, 28, 2)) * 0.2).astype(np.float32)
    acts = (rng.standard_normal((steps, n, 28)) * 0.5).astype(np.float32)
    host = HostSim(sm.struct, n, hf, [-4.0, -4.0], [0.4, 0.4], variant="bpl")
    host.root_state[:], host.dof_state[:] = rs0, ds0
    for t in range(steps):
        host.step(acts[t], n_sub=4, h=1.0 / 120.0)
    T = lambda a: torch.tensor(a, device=dev)
    hf_t = T(hf)
    ter = _hip.terrain_struct(hf_t, [-4.0, -4.0], [0.4, 0.4])
    lo, hi = T(np.full(28, -10.0, np.float32)), T(np.full(28, 10.0, np.float32))
    eo = torch.zeros((n, 3), device=dev)
    for name in VARIANTS:
        if not os.path.exists(so_path(name)):
            continue
        L = ctypes.CDLL(so_path(name))
        _hip_sim.declare(L)
        rs, ds = T(rs0), T(ds0)
        rb, cf = torch.zeros((n, 15, 13), device=dev), torch.zeros((n, 15, 3), device=dev)
        for t in range(steps):
            a = T(acts[t])
            assert L.parc_sim_step(_hip.stream(), sm.device_ptr(dev), ter, n, _hip.ptr(rs), _hip.ptr(ds), _hip.ptr(rb), _hip.ptr(cf), _hip.ptr(eo),
                                   _hip.ptr(a), _hip.ptr(lo), _hip.ptr(hi), 4, 1.0 / 120.0) == 0
        torch.cuda.synchronize()
        dev_err = max(float(np.abs(rs.cpu().numpy()[:, 0:7] - host.root_state[:, 0:7]).max()), float(np.abs(ds.cpu().numpy()[..., 0] - host.dof_state[..., 0]).max()))

        def launch():
            L.parc_sim_step(_hip.stream(), env._sim_model.device_ptr(dev), c._terrain_struct, 4096, _hip.ptr(c.root_state), _hip.ptr(c.dof_state),
                            _hip.ptr(c.rigid_body_state), _hip.ptr(c.contact_forces), _hip.ptr(c.env_offsets), _hip.ptr(a0),
                            _hip.ptr(env._action_bound_low), _hip.ptr(env._action_bound_high), env._sim_steps * env._substeps, env._sim_h)
        best = 1e9
        for _ in range(3):
            c.root_state.copy_(state0[0])
            c.dof_state.copy_(state0[1])
            launch()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(20):
                    launch()
            c.root_state.copy_(state0[0])
            c.dof_state.copy_(state0[1])
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) * 1e3 / 20)
            del g
        print(json.dumps({"variant": name, "flags": VARIANTS[name], "us_per_4096_env_launch": round(best, 1),
                          "max_abs_deviation_from_host_build_after_3_steps": float("%.2e" % dev_err)}), flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
