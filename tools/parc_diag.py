"""The diagnostics build of the HIP library: libparc_hip_diag.so = the product sources with -DPARC_DIAG_BUILD.

What it adds to the product library (tools/parc_diag.h):
  * the timing-ablation bits of `what` in parc_track_post_step (0x10000 ... 0x1000000: drop a role's waves, return at entry / in front
    of the barrier, ...) - results of such launches are garbage, the product library answers PARC_EINVAL to them;
  * parc_tune_hf_envs_per_block / parc_tune_hf_groups / parc_tune_hf_ablation (process-global, standalone heightmap kernel only);
  * parc_diag_sim_step_env_per_lane: the one-env-per-lane reference formulation of the simulator (parc_sim_ref.hip), lanes per
    workgroup as a per-call argument.

Only tools/ and tests/ import this module; `parc_amd` never loads the diagnostics library.  Tools that time the PRODUCT's python path on
the diagnostics kernels call install(): it swaps the handle `parc_amd._hip.lib()` returns for this process and, optionally, ORs ablation
bits into the rollout's full post-step launch.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd import _hip  # noqa: E402

_lib = None


def build(force=False, verbose=False):
    return _hip.build(force=force, verbose=verbose, diag=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_hip.DIAG_LIB_PATH) or _hip._stored_digest(_hip.DIAG_DIGEST_PATH) != _hip.source_digest():
            raise RuntimeError("libparc_hip_diag.so is missing or stale: build it with `python tools/parc_diag.py` (or __graft_entry__.build())")
        L = ctypes.CDLL(_hip.DIAG_LIB_PATH)
        _hip._declare(L)
        c_vp, c_int, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        for name in ("parc_tune_hf_envs_per_block", "parc_tune_hf_groups", "parc_tune_hf_ablation", "parc_tune_post_lds_pad"):
            getattr(L, name).argtypes = [c_int]
            getattr(L, name).restype = c_int
        L.parc_diag_sim_step_env_per_lane.restype = c_int
        L.parc_diag_sim_step_env_per_lane.argtypes = [c_vp, c_vp, _hip.TerrainS] + [c_int] + [c_vp] * 8 + [c_int, c_f, c_int]
        _lib = L
    return _lib


def install(post_bits=0):
    """Make `parc_amd` launch the diagnostics kernels in this process; post_bits: ablation bits ORed into every full post-step launch
    of the rollout (the one that computes reward / done for all envs)."""
    L = lib()
    _hip._lib = L
    if post_bits:
        from parc_amd import tracker_core
        orig = tracker_core.TrackerCore.post_step

        def post_step(self, what, env_ids=None, *a, **k):
            if env_ids is None and k.get("rows") is None and (what & _hip.POST_REWARD_DONE):
                what |= post_bits
            return orig(self, what, env_ids, *a, **k)
        tracker_core.TrackerCore.post_step = post_step
    return L


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
