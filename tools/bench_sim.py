#!/usr/bin/env python3
"""Simulator step kernel timing vs lanes per workgroup (run on the GPU box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parc_amd import _hip, workloads

dev = "cuda:0"
env, clips, tiled = workloads.build_env("boxes_64clips", 4096, dev, seed=0)
env.reset()
a = torch.zeros((4096, 28), device=dev)
L = _hip.lib()
ONLY_PRODUCT = "--plain" in sys.argv           # profiler passes: the product kernel alone
if not ONLY_PRODUCT:
    import parc_diag
    LD = parc_diag.lib()                           # one-env-per-lane reference kernel: diagnostics library
for th in (0,) if ONLY_PRODUCT else (64, 32, 0):
    for _ in range(3):
        env.step(a)
    torch.cuda.synchronize()
    c = env._core
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn, extra = (LD.parc_diag_sim_step_env_per_lane, (th,)) if th else (L.parc_sim_step, ())
    for _ in range(20):
        fn(_hip.stream(), env._sim_model.device_ptr(dev), c._terrain_struct, 4096, _hip.ptr(c.root_state), _hip.ptr(c.dof_state),
                        _hip.ptr(c.rigid_body_state), _hip.ptr(c.contact_forces), _hip.ptr(c.env_offsets), _hip.ptr(a),
                        _hip.ptr(env._action_bound_low), _hip.ptr(env._action_bound_high), env._sim_steps * env._substeps, env._sim_h, *extra)
    e.record()
    torch.cuda.synchronize()
    print(json.dumps({"kernel": "one env per lane, %d lanes/workgroup" % th if th else "body per lane (16 lanes/env)",
                      "us_per_step": s.elapsed_time(e) * 1e3 / 20}))
    env.reset()
