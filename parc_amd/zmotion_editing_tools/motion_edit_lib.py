"""Motion files and the two clip edits that sit between stage 2's optimiser and the tracker's dataset.

Mirror of the part of the reference's ``zmotion_editing_tools/motion_edit_lib.py`` the stage scripts call: ``MotionData`` (:19-182),
``load_motion_file`` (:184-187), ``save_motion_data`` (:189-225), ``flip_motion_about_XZ_plane`` (:514-610: the mirrored copy
``parc_2_kin_gen.py:493-511`` adds for every optimised clip) and ``remove_hesitation_frames`` (:1242-1319).  The interactive editing
functions of that file (blending, stitching, retiming for the MOTION_FORGE GUI) are outside the tracker path.

Mechanics: the reference walks frames in Python (one dof->rotation->dof round trip per frame for the mirror, an O(T^2) loop of norms
for the hesitation search); here a clip goes through the pose kernels once (parc_dof_to_rot / parc_rot_to_dof / parc_forward_kinematics)
and the pairwise pose distances are one [T, T] matrix on the device.  Files are read with the non-executing reader unless
``unsafe_pickle=True`` and written in the reference's format (util.terrain_util.SubTerrain).  Checked against fixture G22.
"""
import numpy as np
import torch

from ..util import safe_pickle, terrain_util, torch_util


class MotionData:
    """dict-backed view of a motion file: frames [T, 34], contacts [T, 15], terrain, fps, loop_mode + optional extras"""

    def __init__(self, motion_data, device="cpu"):
        d = self._data = motion_data
        self._device = device
        for key in ("frames", "contacts", "floor_heights"):
            if key in d and not isinstance(d[key], torch.Tensor):
                d[key] = torch.tensor(np.asarray(d[key]), dtype=torch.float32, device=device)
        if "terrain" in d:
            t = d["terrain"]
            if isinstance(t, dict):          # what the non-executing reader returns for a SubTerrain
                d["terrain"] = t = terrain_util.SubTerrain.from_arrays(t["hf"], t["min_point"], t["dxdy"], t.get("hf_mask"), t.get("hf_maxmin"),
                                                                        name=t.get("terrain_name", "terrain"), device=device)
            t.update_old()
            t.to_torch(device)
        if "path_nodes" in d:
            d["path_nodes"] = torch.as_tensor(d["path_nodes"]).to(device=device)
        d.setdefault("fps", 30)
        d.setdefault("loop_mode", "CLAMP")
        if 29 < self.get_fps() < 31:
            self.set_fps(30)

    def set_hf_mask_inds_device(self, device):
        inds = self._data["hf_mask_inds"]
        for k in range(len(inds)):
            inds[k] = inds[k].to(device=device)

    def get_fps(self):
        return self._data["fps"]

    def set_fps(self, fps):
        self._data["fps"] = int(fps)

    def get_loop_mode(self):
        return self._data["loop_mode"]

    def get_frames(self):
        return self._data["frames"]

    def set_frames(self, motion_frames):
        self._data["frames"] = motion_frames

    def has_contacts(self):
        return "contacts" in self._data

    def get_contacts(self):
        return self._data["contacts"]

    def set_contacts(self, contacts):
        self._data["contacts"] = contacts

    def has_hf_mask_inds(self):
        return "hf_mask_inds" in self._data

    def get_hf_mask_inds(self):
        return self._data["hf_mask_inds"]

    def set_hf_mask_inds(self, hf_mask_inds):
        self._data["hf_mask_inds"] = hf_mask_inds

    def has_terrain(self):
        return "terrain" in self._data

    def get_terrain(self):
        return self._data["terrain"]

    def set_terrain(self, terrain):
        self._data["terrain"] = terrain

    def remove_terrain(self):
        del self._data["terrain"]

    def has_opt_body_constraints(self):
        return "opt:body_constraints" in self._data

    def get_opt_body_constraints(self):
        return self._data["opt:body_constraints"]

    def set_opt_body_constraints(self, body_constraints):
        self._data["opt:body_constraints"] = body_constraints

    def remove_opt_body_constraints(self):
        del self._data["opt:body_constraints"]

    def save_to_file(self, motion_filepath, verbose=True):
        d = self._data
        for key in ("frames", "contacts", "floor_heights"):
            if key in d and isinstance(d[key], torch.Tensor):
                d[key] = d[key].cpu().numpy().astype(np.float32)
        if "terrain" in d and isinstance(d["terrain"].hf, torch.Tensor):
            d["terrain"] = d["terrain"].numpy_copy()
        if "hf_mask_inds" in d:
            self.set_hf_mask_inds_device("cpu")
        if self.has_opt_body_constraints():
            for per_body in self.get_opt_body_constraints():
                for c in per_body:
                    c.constraint_point = c.constraint_point.to(device="cpu")
        terrain_util.dump_reference_pickle(d, motion_filepath)
        if verbose:
            print("wrote motion data to", motion_filepath)


def load_motion_file(motion_filepath, device="cpu", unsafe_pickle=False):
    if unsafe_pickle:
        data = safe_pickle.load_executing(motion_filepath)
    else:
        data = dict(safe_pickle.load_motion_file_safe(motion_filepath))
    return MotionData(data, device=device)


def save_motion_data(motion_filepath, motion_frames, contact_frames, terrain, fps, loop_mode, **kwargs):
    data = dict()
    if motion_frames is not None:
        data["frames"] = motion_frames.cpu().numpy().astype(np.float32) if isinstance(motion_frames, torch.Tensor) else motion_frames
    if contact_frames is not None:
        data["contacts"] = contact_frames.cpu().numpy().astype(np.float32) if isinstance(contact_frames, torch.Tensor) else contact_frames
    if terrain is not None:
        data["terrain"] = terrain.numpy_copy() if isinstance(terrain.hf, torch.Tensor) else terrain
    if fps is not None:
        data["fps"] = fps
    if loop_mode is not None:
        data["loop_mode"] = loop_mode
    for key, value in kwargs.items():
        data[key] = value.cpu() if isinstance(value, torch.Tensor) else value
    terrain_util.dump_reference_pickle(data, motion_filepath)
    print("wrote motion data to", motion_filepath)


# dof slices / body rows of the humanoid that trade places under the mirror (right <-> left)
_DOF_PAIRS = ((slice(6, 9), slice(10, 13)), (slice(9, 10), slice(13, 14)), (slice(14, 17), slice(21, 24)), (slice(17, 18), slice(24, 25)),
              (slice(18, 21), slice(25, 28)))
_BODY_PAIRS = ((3, 6), (4, 7), (5, 8), (9, 12), (10, 13), (11, 14))


def flip_motion_about_XZ_plane(motion_frames, char_model, contact_frames=None):
    """The clip mirrored in the plane y = 0: root position and rotation reflected, every joint rotation reflected ((x, y, z, w) ->
    (x, -y, z, -w), the same rotation as (-x, y, -z, w)) and mapped back to dofs, then right and left limbs exchanged."""
    dev_in = motion_frames.device
    f = motion_frames.to(device=char_model._device, dtype=torch.float32).clone()
    f[:, 1] *= -1.0
    f[:, 3] *= -1.0            # exponential map: y component reflected, then the sense of rotation reversed
    f[:, 5] *= -1.0
    jr = char_model.dof_to_rot(f[:, 6:].contiguous())
    jr = jr * jr.new_tensor([1.0, -1.0, 1.0, -1.0])
    dof = char_model.rot_to_dof(jr.contiguous())
    out = dof.clone()
    for a, b in _DOF_PAIRS:
        out[:, a], out[:, b] = dof[:, b], dof[:, a]
    f[:, 6:] = out
    f = f.to(dev_in)
    if contact_frames is None:
        return f
    c = contact_frames.clone()
    for a, b in _BODY_PAIRS:
        c[:, a], c[:, b] = contact_frames[:, b], contact_frames[:, a]
    return f, c


def remove_hesitation_frames(motion_frames, contact_frames, char_model, hesitation_val=0.15, hesitation_min_seq_len=4, verbose=False):
    """Drop stretches where the character dithers: walking the frames in order, every later frame whose body positions lie within
    `hesitation_val` (Frobenius norm over all bodies) of a kept frame is marked; marked runs of at least `hesitation_min_seq_len`
    consecutive frames are removed.  The marking is order dependent (a marked frame is no anchor), so it runs on the host over the
    device-computed distance matrix."""
    dev = char_model._device
    f = motion_frames.to(device=dev, dtype=torch.float32)
    body_pos, _ = char_model.forward_kinematics(f[:, 0:3].contiguous(), torch_util.exp_map_to_quat(f[:, 3:6]), char_model.dof_to_rot(f[:, 6:].contiguous()))
    T = int(body_pos.shape[0])
    flat = body_pos.reshape(T, -1)
    close = (torch.linalg.vector_norm(flat.unsqueeze(0) - flat.unsqueeze(1), dim=-1) < hesitation_val).cpu().numpy()
    marked = np.zeros(T, dtype=bool)
    for i in range(T):
        if not marked[i]:
            marked[i + 1:] |= close[i, i + 1:]
    drop = np.zeros(T, dtype=bool)
    i = 0
    while i < T:
        if marked[i]:
            j = i
            while j + 1 < T and marked[j + 1]:
                j += 1
            if j + 1 - i >= hesitation_min_seq_len:
                drop[i:j + 1] = True
            if verbose:
                print("hesitation run", i, j, "dropped" if drop[i] else "kept")
            i = j + 1
        else:
            i += 1
    keep = torch.as_tensor(~drop)
    return motion_frames[keep.to(motion_frames.device)], contact_frames[keep.to(contact_frames.device)]
