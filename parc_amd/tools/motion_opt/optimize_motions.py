"""Batch driver of the motion optimiser: every clip of a dataset YAML -> ``<name>_opt.pkl``.

Mirror of the reference's ``tools/motion_opt/optimize_motions.py`` (config keys of ``tools/motion_opt/config/motion_opt.yaml``:
motions_yaml_path, device, char_model, output_folder_path, num_iters, step_size, w_*, max_jerk, auto_compute_body_constraints,
frame_stride, char_point_samples{...}).  Input files are read with the non-executing reader, output files are written in the
reference's format (util.terrain_util.SubTerrain, tools.motion_opt.motion_optimization.BodyConstraint).

usage:  python -m parc_amd.tools.motion_opt.optimize_motions --config motion_opt.yaml
"""
import math
import os
import sys
import time

import numpy as np
import torch
import yaml

from ...anim import kin_char_model
from ...util import geom_util, safe_pickle, terrain_util, torch_util
from . import motion_optimization as moopt


def fetch_motion_files(motion_file):
    if os.path.splitext(motion_file)[1] == ".yaml":
        with open(motion_file, "r") as f:
            cfg = yaml.load(f, Loader=yaml.SafeLoader)
        files, weights = [], []
        for entry in cfg["motions"]:
            assert entry["weight"] >= 0
            files.append(entry["file"])
            weights.append(entry["weight"])
        return files, weights
    return [motion_file], [1.0]


def _load_clip(path, device):
    d = safe_pickle.load_motion_file_safe(path)
    t = d["terrain"]
    ter = terrain_util.SubTerrain.from_arrays(t["hf"], t["min_point"], t["dxdy"], t.get("hf_mask"), t.get("hf_maxmin"), device=device)
    frames = torch.as_tensor(np.asarray(d["frames"], np.float32), device=device)
    contacts = torch.as_tensor(np.asarray(d["contacts"], np.float32), device=device)
    fps = d.get("fps", 30)
    return frames, contacts, ter, int(fps.item() if isinstance(fps, np.ndarray) else fps)


def optimize_file(path, cfg, char_model, body_points, output_folder, log_folder):
    device = cfg["device"]
    frames, contacts, terrain, fps = _load_clip(path, device)
    stride = int(cfg.get("frame_stride", 1))
    name = os.path.basename(os.path.splitext(path)[0])
    body_constraints = None
    if cfg.get("auto_compute_body_constraints", False):
        body_constraints = moopt.compute_approx_body_constraints(root_pos=frames[:, 0:3].contiguous(), root_rot=torch_util.exp_map_to_quat(frames[:, 3:6]),
                                                                 joint_rot=char_model.dof_to_rot(frames[:, 6:].contiguous()), contacts=contacts,
                                                                 char_model=char_model, terrain=terrain)
        for lst in body_constraints:        # optimize_motions.py:143-147
            for c in lst:
                c.start_frame_idx = int(math.ceil(c.start_frame_idx / stride))
                c.end_frame_idx = int(math.floor(c.end_frame_idx // stride))
    frames, contacts = frames[::stride].contiguous(), contacts[::stride].contiguous()
    out_name = name + "_opt"
    w = {k: cfg[k] for k in ("w_root_pos", "w_root_rot", "w_joint_rot", "w_smoothness", "w_penetration", "w_contact", "w_sliding",
                             "w_body_constraints", "w_jerk")}
    opt = moopt.motion_contact_optimization(src_frames=frames, contacts=contacts, body_points=body_points, terrain=terrain, char_model=char_model,
                                            num_iters=cfg["num_iters"], step_size=cfg["step_size"], body_constraints=body_constraints,
                                            max_jerk=cfg["max_jerk"], exp_name=out_name, use_wandb=False,
                                            log_file=os.path.join(log_folder, "log_" + out_name + ".txt"), **w)
    cpu_t = terrain.torch_copy()
    cpu_t.set_device("cpu")
    data = {"fps": fps // stride, "loop_mode": "CLAMP", "frames": opt.cpu(), "contacts": contacts.cpu(), "terrain": cpu_t}
    if body_constraints is not None:
        for lst in body_constraints:
            for c in lst:
                c.constraint_point = c.constraint_point.cpu()
        data["opt:body_constraints"] = body_constraints
    out_path = os.path.join(output_folder, out_name + ".pkl")
    terrain_util.dump_reference_pickle(data, out_path)
    return out_path


def main(argv):
    cfg_path = argv[2] if len(argv) == 3 and argv[1] == "--config" else "tools/motion_opt/config/motion_opt.yaml"
    with open(cfg_path, "r") as f:
        cfg = yaml.safe_load(f)
    files, _ = fetch_motion_files(cfg["motions_yaml_path"])
    out_folder = cfg["output_folder_path"]
    log_folder = os.path.join(out_folder, "log")
    os.makedirs(log_folder, exist_ok=True)
    km = kin_char_model.KinCharModel(cfg["device"])
    km.load_char_file(cfg["char_model"])
    ps = cfg["char_point_samples"]
    body_points = geom_util.get_char_point_samples(km, sphere_num_subdivisions=ps["sphere_num_subdivisions"], box_num_slices=ps["box_num_slices"],
                                                   box_dim_x=ps["box_dim_x"], box_dim_y=ps["box_dim_y"],
                                                   capsule_num_circle_points=ps["capsule_num_circle_points"],
                                                   capsule_num_sphere_subdivisons=ps["capsule_num_sphere_subdivisions"],
                                                   capsule_num_cylinder_slices=ps["capsule_num_cylinder_slices"])
    t0 = time.time()
    for i, path in enumerate(files):
        print("OPTIMIZING MOTION:", os.path.basename(path), "{}/{}".format(i, len(files)))
        optimize_file(path, cfg, km, body_points, out_folder, log_folder)
    print("Total optimization time for", len(files), "motions:", time.time() - t0, "seconds.")


if __name__ == "__main__":
    main(sys.argv)
