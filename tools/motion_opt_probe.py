"""Time the stage-2 motion optimiser on one synthetic clip at the size the kin-gen stage produces (PARC/kin_gen_default.yaml opt:
num_iters 3000, the default sample points; a 10 s clip at 30 fps on a 45 x 45-cell terrain) and run the batch driver end to end on a
file it writes itself.   python tools/motion_opt_probe.py [iters]   -> one JSON line"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd import synthetic  # noqa: E402
from parc_amd.anim import kin_char_model  # noqa: E402
from parc_amd.tools.motion_opt import motion_optimization as mo  # noqa: E402
from parc_amd.tools.motion_opt import optimize_motions  # noqa: E402
from parc_amd.util import geom_util, safe_pickle, terrain_util, torch_util  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    dev = "cuda:0"
    km = kin_char_model.KinCharModel(dev)
    km.load_char_file(kin_char_model.default_char_file())
    clip = synthetic.make_dataset(num_clips=1, seed=5, frames_range=(300, 300), tile_cells_range=(45, 45))[0]
    frames = torch.tensor(clip["frames"], dtype=torch.float32, device=dev)
    frames[:, 2] -= 0.03
    contacts = torch.tensor(clip["contacts"], dtype=torch.float32, device=dev)
    ter = terrain_util.SubTerrain.from_arrays(clip["hf"], clip["min_point"], clip["dxdy"], device=dev)
    pts = geom_util.get_char_point_samples(km)
    w = dict(w_root_pos=1.0, w_root_rot=10.0, w_joint_rot=1.0, w_smoothness=10.0, w_penetration=1000.0, w_contact=1000.0, w_sliding=10.0,
             w_body_constraints=1000.0, w_jerk=1000.0)
    t0 = time.time()
    bc = mo.compute_approx_body_constraints(frames[:, 0:3].contiguous(), torch_util.exp_map_to_quat(frames[:, 3:6]),
                                            km.dof_to_rot(frames[:, 6:].contiguous()), contacts, km, ter)
    torch.cuda.synchronize()
    t_bc = time.time() - t0
    res = {"frames": int(frames.shape[0]), "sample_points": int(sum(p.shape[0] for p in pts)), "terrain_cells": list(clip["hf"].shape),
           "body_constraints": int(sum(len(x) for x in bc)), "body_constraints_s": round(t_bc, 3), "iters": iters}
    only = os.environ.get("PROBE_ONLY")          # "graph" / "eager": that mode alone and no driver run (profiling)
    for mode in (False, True):
        if only and only != ("graph" if mode else "eager"):
            continue
        trace = []
        torch.cuda.synchronize()
        t0 = time.time()
        out = mo.motion_contact_optimization(src_frames=frames, contacts=contacts, body_points=pts, terrain=ter, char_model=km, num_iters=iters,
                                             step_size=0.001, body_constraints=bc, max_jerk=1000.0, exp_name="probe", use_wandb=False, log_file=None,
                                             use_graph=mode, verbose=False, loss_trace=trace, **w)
        torch.cuda.synchronize()
        dt = time.time() - t0
        tr = trace[0]
        key = "graph" if mode else "eager"
        res[key] = {"total_s": round(dt, 3), "ms_per_iter": round(1e3 * dt / iters, 4), "loss_first": float(tr[0]), "loss_last": float(tr[-1]),
                    "finite": bool(torch.isfinite(out).all())}
    if only:
        print(json.dumps(res))
        return
    # the batch driver on a file in the reference's format
    tmp = tempfile.mkdtemp(prefix="parc_mo_")
    cpu_t = terrain_util.SubTerrain.from_arrays(clip["hf"], clip["min_point"], clip["dxdy"], device="cpu")
    terrain_util.dump_reference_pickle({"fps": 30, "loop_mode": "CLAMP", "frames": frames.cpu(), "contacts": contacts.cpu(), "terrain": cpu_t},
                                       os.path.join(tmp, "clip.pkl"))
    cfg = dict(w, device=dev, num_iters=min(iters, 200), step_size=0.001, max_jerk=1000.0, auto_compute_body_constraints=True, frame_stride=1)
    path = optimize_motions.optimize_file(os.path.join(tmp, "clip.pkl"), cfg, km, pts, tmp, tmp)
    back = safe_pickle.load_motion_file_safe(path)
    res["driver_output"] = {"frames": list(np.asarray(back["frames"]).shape), "keys": sorted(k for k in back.keys())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
