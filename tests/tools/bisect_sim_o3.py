"""Localise the -O3 divergence of the one-env-per-lane simulator kernel (sim_step_kernel, parc_sim_core.h).

  python tests/tools/bisect_sim_o3.py build            # here (no GPU): variants of parc_sim.hip + parc_sim_ref.hip -> gpurun_out/../_bisect/*.so
  python tests/tools/bisect_sim_o3.py run              # on the GPU box: every variant against the g++ host build of the same source

A variant = optimisation flags + a set of loop tags kept rolled (-DPARC_BISECT -DPARC_ROLL_<k>, parc_sim_bisect.h).  `run` steps
the same random scene with every variant (both kernels: variant 0 = one env per lane, 1 = body per lane) and prints the largest
deviation from the host result, so one GPU call ranks all of them.
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
OUT = os.path.join(ROOT, "tools", "_bisect")
NTAGS = 14


def variants():
    v = {"O2": ("-O2", []), "O3": ("-O3", []), "O3_all_rolled": ("-O3", list(range(NTAGS))), "O3_fno_unroll": ("-O3 -fno-unroll-loops", []),
         "O3_fno_slp": ("-O3 -fno-slp-vectorize -fno-vectorize", [])}
    # which part of GVN (the pass -opt-bisect-limit names, see `optbisect`): its load elimination needs memory dependence analysis,
    # load-PRE moves loads across edges, scalar PRE does not touch memory; and the same build without the kernel's __restrict__
    v["O3_gvn_no_memdep"] = ("-O3 -mllvm -enable-gvn-memdep=false", [])
    v["O3_gvn_no_load_pre"] = ("-O3 -mllvm -enable-load-pre=false", [])
    v["O3_gvn_no_scalar_pre"] = ("-O3 -mllvm -enable-pre=false", [])
    v["O3_no_restrict"] = ("-O3 -D__restrict__=", [])
    for k in range(NTAGS):
        v["O3_only_%d_rolled" % k] = ("-O3", [k])
        v["O3_all_but_%d_rolled" % k] = ("-O3", [j for j in range(NTAGS) if j != k])
    extra = os.environ.get("PARC_BISECT_EXTRA")          # e.g. "name=-O3 -mllvm -foo;4,5"
    if extra:
        name, rest = extra.split("=", 1)
        flags, _, tags = rest.partition(";")
        v[name] = (flags, [int(t) for t in tags.split(",") if t])
    return v


def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def one(item):
        name, (flags, tags) = item
        so = os.path.join(OUT, "libsim_%s.so" % name)
        cmd = [hipcc, "--offload-arch=gfx950"] + flags.split() + ["-std=c++17", "-fPIC", "-shared", "-DPARC_BISECT"] + \
            ["-DPARC_ROLL_%d" % k for k in tags] + ["-I", os.path.dirname(os.path.abspath(__file__)), "-o", so,
                                                     os.path.join(CSRC, "parc_sim.hip"), os.path.join(CSRC, "parc_sim_ref.hip")]
        subprocess.check_call(cmd)
        return name
    with ThreadPoolExecutor(max_workers=6) as ex:
        for name in ex.map(one, variants().items()):
            print("built", name, flush=True)


def run():
    import numpy as np
    import torch
    from oracle.sim_host import HostSim
    from parc_amd import _hip, _hip_sim
    from parc_amd.anim.kin_char_model import KinCharModel
    from parc_amd.assets import humanoid_spec
    from parc_amd.sim_model import SimModel
    dev = "cuda:0"
    km = KinCharModel(dev)
    km.load_char_file(humanoid_spec.write_mjcf())
    sm = SimModel(km)
    n, steps = 64, 6
    rng = np.random.default_rng(3)
    hf = (rng.random((20, 20)) * 0.5).astype(np.float32)
    rs0 = np.zeros((n, 13), np.float32)
    rs0[:, 0:2] = rng.random((n, 2)) * 4.0 - 2.0
    rs0[:, 2] = 1.0 + 0.3 * rng.random(n)
    rs0[:, 6] = 1.0
    ds0 = (rng.standard_normal((n, 28, 2)) * 0.2).astype(np.float32)
    acts = (rng.standard_normal((steps, n, 28)) * 0.5).astype(np.float32)
    host = HostSim(sm.struct, n, hf, [-4.0, -4.0], [0.4, 0.4], variant="core")
    host.root_state[:], host.dof_state[:] = rs0, ds0
    href = []
    for t in range(steps):
        host.step(acts[t], n_sub=4, h=1.0 / 120.0)
        href.append((host.root_state.copy(), host.dof_state.copy()))
    T = lambda a: torch.tensor(a, device=dev)
    hf_t = T(hf)
    ter = _hip.terrain_struct(hf_t, [-4.0, -4.0], [0.4, 0.4])
    lo, hi = T(np.full(28, -10.0, np.float32)), T(np.full(28, 10.0, np.float32))
    eo = torch.zeros((n, 3), device=dev)
    results = {}
    for name in sorted(variants()):
        so = os.path.join(OUT, "libsim_%s.so" % name)
        if not os.path.exists(so):
            continue
        L = ctypes.CDLL(so)
        _hip_sim.declare(L)
        row = {}
        L.parc_diag_sim_step_env_per_lane.restype = ctypes.c_int
        L.parc_diag_sim_step_env_per_lane.argtypes = L.parc_sim_step.argtypes + [ctypes.c_int]
        for kern in (0, 1):
            step_fn, extra = (L.parc_diag_sim_step_env_per_lane, (64,)) if kern == 0 else (L.parc_sim_step, ())
            rs, ds = T(rs0), T(ds0)
            rb, cf = torch.zeros((n, 15, 13), device=dev), torch.zeros((n, 15, 3), device=dev)
            worst = 0.0
            for t in range(steps):
                a = T(acts[t])
                rc = step_fn(_hip.stream(), sm.device_ptr(dev), ter, n, _hip.ptr(rs), _hip.ptr(ds), _hip.ptr(rb), _hip.ptr(cf),
                             _hip.ptr(eo), _hip.ptr(a), _hip.ptr(lo), _hip.ptr(hi), 4, 1.0 / 120.0, *extra)
                assert rc == 0, rc
                torch.cuda.synchronize()
                d = max(float(np.abs(rs.cpu().numpy()[:, 0:7] - href[t][0][:, 0:7]).max()),
                        float(np.abs(ds.cpu().numpy()[..., 0] - href[t][1][..., 0]).max()))
                worst = max(worst, d if np.isfinite(d) else 1e9)
            row["one_env_per_lane" if kern == 0 else "body_per_lane"] = worst
        results[name] = row
        print("{:28s} one-env-per-lane {:.3e}   body-per-lane {:.3e}".format(name, row["one_env_per_lane"], row["body_per_lane"]), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "bisect_sim_o3.json"), "w") as f:
        json.dump(results, f, indent=1)


def optbisect():
    """On the GPU box: binary search over LLVM's -opt-bisect-limit for the first pass whose execution makes the -O3 build of
    sim_step_kernel diverge from the host result (every pass after the limit is skipped; skipping optional passes is always legal)."""
    import re
    import numpy as np
    import torch
    from oracle.sim_host import HostSim
    from parc_amd import _hip, _hip_sim
    from parc_amd.anim.kin_char_model import KinCharModel
    from parc_amd.assets import humanoid_spec
    from parc_amd.sim_model import SimModel
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    dev = "cuda:0"
    km = KinCharModel(dev)
    km.load_char_file(humanoid_spec.write_mjcf())
    sm = SimModel(km)
    n, steps = 64, 4
    rng = np.random.default_rng(3)
    hf = (rng.random((20, 20)) * 0.5).astype(np.float32)
    rs0 = np.zeros((n, 13), np.float32)
    rs0[:, 0:2] = rng.random((n, 2)) * 4.0 - 2.0
    rs0[:, 2] = 1.0 + 0.3 * rng.random(n)
    rs0[:, 6] = 1.0
    ds0 = (rng.standard_normal((n, 28, 2)) * 0.2).astype(np.float32)
    acts = (rng.standard_normal((steps, n, 28)) * 0.5).astype(np.float32)
    host = HostSim(sm.struct, n, hf, [-4.0, -4.0], [0.4, 0.4], variant="core")
    host.root_state[:], host.dof_state[:] = rs0, ds0
    href = []
    for t in range(steps):
        host.step(acts[t], n_sub=4, h=1.0 / 120.0)
        href.append((host.root_state.copy(), host.dof_state.copy()))
    T = lambda a: torch.tensor(a, device=dev)
    hf_t = T(hf)
    ter = _hip.terrain_struct(hf_t, [-4.0, -4.0], [0.4, 0.4])
    lo, hi = T(np.full(28, -10.0, np.float32)), T(np.full(28, 10.0, np.float32))
    eo = torch.zeros((n, 3), device=dev)

    def compile_and_run(limit):
        so = os.path.join(OUT, "libsim_bisect_%d.so" % limit)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-opt-bisect-limit=%d" % limit, "-o", so,
               os.path.join(CSRC, "parc_sim.hip"), os.path.join(CSRC, "parc_sim_ref.hip")]
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        # the device compilation is the one that names AMDGPU passes / the kernel; keep its numbered lines
        lines = [ln for ln in res.stderr.splitlines() if ln.startswith("BISECT:")]
        L = ctypes.CDLL(so)
        _hip_sim.declare(L)
        L.parc_diag_sim_step_env_per_lane.restype = ctypes.c_int
        L.parc_diag_sim_step_env_per_lane.argtypes = L.parc_sim_step.argtypes + [ctypes.c_int]
        rs, ds = T(rs0), T(ds0)
        rb, cf = torch.zeros((n, 15, 13), device=dev), torch.zeros((n, 15, 3), device=dev)
        worst = 0.0
        for t in range(steps):
            a = T(acts[t])
            rc = L.parc_diag_sim_step_env_per_lane(_hip.stream(), sm.device_ptr(dev), ter, n, _hip.ptr(rs), _hip.ptr(ds), _hip.ptr(rb), _hip.ptr(cf),
                                                   _hip.ptr(eo), _hip.ptr(a), _hip.ptr(lo), _hip.ptr(hi), 4, 1.0 / 120.0, 64)
            assert rc == 0
            torch.cuda.synchronize()
            d = max(float(np.abs(rs.cpu().numpy()[:, 0:7] - href[t][0][:, 0:7]).max()), float(np.abs(ds.cpu().numpy()[..., 0] - href[t][1][..., 0]).max()))
            worst = max(worst, d if np.isfinite(d) else 1e9)
        os.remove(so)
        return worst, lines
    w_all, lines = compile_and_run(-1)
    nums = [int(m.group(1)) for m in (re.search(r"\((\d+)\)", ln) for ln in lines) if m]
    total = max(nums)
    print("all passes: deviation {:.3e}, {} numbered passes (max over the host and device compilations)".format(w_all, total), flush=True)
    assert w_all > 1e-2, "the -O3 build does not diverge here"
    w0, _ = compile_and_run(0)
    print("no optional pass: deviation {:.3e}".format(w0), flush=True)
    assert w0 < 1e-3
    good, bad = 0, total
    while bad - good > 1:
        mid = (good + bad) // 2
        w, _ = compile_and_run(mid)
        print("limit {:6d}: deviation {:.3e}".format(mid, w), flush=True)
        if w < 1e-3:
            good = mid
        else:
            bad = mid
    _, lines = compile_and_run(bad)
    culprit = [ln for ln in lines if "(%d)" % bad in ln and "NOT running" not in ln]
    print("first pass whose execution makes the kernel diverge: limit", bad)
    for ln in culprit:
        print("  ", ln)
    with open(os.path.join(ROOT, "gpurun_out", "bisect_sim_o3_pass.txt"), "w") as f:
        f.write("limit {}\n".format(bad) + "\n".join(culprit) + "\n")


if __name__ == "__main__":
    {"build": build, "run": run, "optbisect": optbisect}[sys.argv[1]]()
