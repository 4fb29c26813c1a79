"""Stage 4 with a policy that learned its clip: record_motions (parc_4_phys_record's call) WITHOUT the recorder's failure bypass, then
compare the recorded clip with the kinematic source and run it through the dataset builder (stage 5's first step).
python tools/record_trained.py <checkpoint> [workload]   -> one JSON line"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd import workloads  # noqa: E402
from parc_amd.util import create_dataset, mp_util, safe_pickle  # noqa: E402


def main():
    ckpt = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else "teaser_clip"
    dev = "cuda:0"
    mp_util.init(0, 1, dev)
    torch.manual_seed(0)
    env, clips, _ = workloads.build_env(workload, 4, dev, seed=0)
    out_dir = tempfile.mkdtemp(prefix="parc_rec_")
    env._output_motion_dir = os.path.join(out_dir, "recorded", "teaser") + "/"
    os.makedirs(env._output_motion_dir, exist_ok=True)
    agent = workloads.build_agent(env, dev)
    agent.load(ckpt)
    succ = agent.record_motions(max_steps=400)
    files = sorted(os.listdir(env._output_motion_dir))
    res = {"workload": workload, "envs": 4, "successful": int(sum(bool(s) for s in succ)), "files": files}
    if files:
        rec = safe_pickle.load_motion_file_safe(os.path.join(env._output_motion_dir, files[0]))
        fr = np.asarray(rec["frames"], np.float32)
        src = np.asarray(clips[0]["frames"], np.float32)
        n = min(len(fr), len(src))
        # the recorder localises the clip on its first frame (xy) and slices the terrain around it
        src_xy = src[:n, 0:2] - src[0, 0:2]
        res.update(frames=list(fr.shape), source_frames=list(src.shape), fps=int(rec["fps"]), terrain_cells=list(np.asarray(rec["terrain"]["hf"]).shape),
                   root_xy_err_mean=float(np.linalg.norm(fr[:n, 0:2] - src_xy, axis=-1).mean()),
                   root_xy_err_max=float(np.linalg.norm(fr[:n, 0:2] - src_xy, axis=-1).max()),
                   dof_err_mean=float(np.abs(fr[:n, 6:] - src[:n, 6:]).mean()),
                   contacts_shape=list(np.asarray(rec["contacts"]).shape), obs_shape=list(np.asarray(rec["obs"]).shape))
        entries = create_dataset.create_dataset_yaml([os.path.join(out_dir, "recorded")], os.path.join(out_dir, "motions.yaml"))
        res["dataset_entries"] = len(entries)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
