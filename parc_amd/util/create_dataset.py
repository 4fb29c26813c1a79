"""Dataset YAML builder with class-balanced sampling weights (mirror of the reference's PARC/util/create_dataset.py:40-202,
the step between recorded motions and the tracker's ``dm.motion_file``).

Motion classes = first-level sub-folders of every input folder; every ``*.pkl`` below a class folder becomes one entry whose
weight is ``clip length x (intended class fraction / actual class fraction)``, so every class is sampled equally often
regardless of how many seconds of motion it holds.  Motion files are read with the non-executing reader
(``util.safe_pickle``): only ``frames``, ``fps``, the terrain's heightfield shape and an optional ``loss`` are needed.

The optional ``compute_preprocessing_data`` step (``terrain_util.compute_hf_extra_vals`` per clip, written back into the
motion file as ``hf_mask_inds`` + the updated terrain, PARC/util/create_dataset.py:147-160) runs the character through the FK
kernels, so it needs the GPU like the rest of the package; files that already carry ``hf_mask_inds`` are left alone.
"""
from pathlib import Path
from typing import List

import yaml

from . import safe_pickle


def _motion_summary(path):
    d = safe_pickle.load_motion_file_safe(str(path))
    frames = d["frames"]
    fps = d.get("fps", 30)                               # MotionData defaults (motion_edit_lib.py:43-46)
    ter = d.get("terrain")
    hf_shape = tuple(ter["hf"].shape) if isinstance(ter, dict) and hasattr(ter.get("hf"), "shape") else None
    loss = d.get("loss")
    loss = None if isinstance(loss, safe_pickle.Unresolved) or loss is None else float(loss)
    return frames.shape[0] / fps, hf_shape, loss


class _Preprocessor:
    """Per-clip heightfield preprocessing (hf_mask / hf_maxmin / per-frame cell lists) written back into the motion file."""
    Z_BUF, JUMP_BUF = 3.0, 0.8          # PARC/util/create_dataset.py:94-95

    def __init__(self, char_filepath, device="cuda:0"):
        from ..anim import kin_char_model
        from . import geom_util
        self._device = device
        self._char = kin_char_model.KinCharModel(device)
        self._char.load_char_file(char_filepath if char_filepath else kin_char_model.default_char_file())
        self._points = geom_util.get_char_point_samples(self._char)

    def run(self, path):
        import numpy as np
        import torch

        from . import terrain_util
        d = safe_pickle.load_motion_file_safe(str(path))
        if "hf_mask_inds" in d:
            return False
        t = d["terrain"]
        ter = terrain_util.SubTerrain.from_arrays(t["hf"], np.asarray(t["min_point"], np.float32), np.asarray(t["dxdy"], np.float32),
                                                  device=self._device)
        frames = torch.tensor(np.asarray(d["frames"], np.float32), device=self._device)
        inds = terrain_util.compute_hf_extra_vals(frames, ter, self._char, self._points, z_buf=self.Z_BUF, jump_buf=self.JUMP_BUF)
        out = {k: v for k, v in d.items() if isinstance(v, (np.ndarray, int, float, str, bool))}
        out["terrain"] = ter.numpy_copy()
        out["hf_mask_inds"] = [i.cpu() for i in inds]
        terrain_util.dump_reference_pickle(out, str(path))
        return True


def create_dataset_yaml(folder_paths: List[Path], save_path: Path, char_filepath: str = None, compute_preprocessing_data: bool = False,
                        cut_some_classes_in_half: bool = False, motion_classes_to_cut_in_half: List[str] = (),
                        max_terrain_dim_x: int = 45, max_terrain_dim_y: int = 45):
    prep = _Preprocessor(char_filepath) if compute_preprocessing_data else None
    folder_paths = [Path(p) for p in folder_paths]
    motion_classes = []
    proportions = dict()
    for folder_path in folder_paths:
        for folder in sorted(p for p in folder_path.iterdir() if p.is_dir()):
            if "ignore" in str(folder):
                continue
            motion_classes.append(folder.name)
            proportions[folder.name] = 1.0
    proportions_sum = sum(proportions.values())

    dirs = []
    for folder_path in folder_paths:
        dirs.extend(p for p in folder_path.rglob("*") if p.is_dir() and "ignore" not in str(p))

    motions = {c: [] for c in motion_classes}
    class_len = {c: 0.0 for c in motion_classes}
    for d in dirs:
        files = sorted(d.glob("*.pkl"))
        if cut_some_classes_in_half and any(c in str(d) for c in motion_classes_to_cut_in_half):
            files = files[::2]
        for fp in files:
            length, hf_shape, loss = _motion_summary(fp)
            if hf_shape is not None and (hf_shape[0] > max_terrain_dim_x or hf_shape[1] > max_terrain_dim_y):
                continue                                  # "Large terrain excluded"
            if loss is not None and loss > 20.0:
                continue                                  # bad generated motion
            if prep is not None:
                prep.run(fp)
            for c in motion_classes:
                if ("/" + c + "/") in str(fp):
                    motions[c].append((fp, length))
                    class_len[c] += length
                    break
            else:
                raise AssertionError("no motion class found in " + str(fp))

    total = sum(class_len.values())
    entries = []
    for c in motion_classes:
        fraction = class_len[c] / total
        factor = (proportions[c] / proportions_sum) / fraction
        for fp, length in motions[c]:
            entries.append({"file": str(fp), "weight": length * factor})
    Path(save_path).write_text(yaml.dump({"motions": entries}))
    return entries


def create_dataset_yaml_from_config(config):
    return create_dataset_yaml(folder_paths=[Path(p) for p in config["folder_paths"]], save_path=Path(config["save_path"]),
                               char_filepath=config.get("char_filepath"),
                               compute_preprocessing_data=config.get("compute_preprocessing_data", False),
                               cut_some_classes_in_half=config.get("cut_some_classes_in_half", False),
                               motion_classes_to_cut_in_half=config.get("motion_classes_to_cut_in_half", []),
                               max_terrain_dim_x=config.get("max_terrain_dim_x", 45), max_terrain_dim_y=config.get("max_terrain_dim_y", 45))
