"""Clip database resident in HBM + frame sampling through the HIP kernels.

Host-side mirror of the reference's ``anim/motion_lib.py`` (MotionLib :21-525): same constructor
arguments, the per-clip tensors it exposes (``_motion_weights``, ``_motion_lengths``, ``_motion_num_frames``,
``_motion_loop_modes``, ``_motion_root_pos_delta`` ...) and ``calc_motion_frame`` / ``sample_motions`` /
``sample_time`` with the same meaning.  Instead of seven flat frame arrays one frame is one 448-byte row
(layout: include/parc_hip.h parc_motion_lib_t); rows are derived on the GPU by parc_motion_lib_build and
sampled by parc_calc_motion_frame / the fused post-step kernel.  No CPU fallback.

Motion files are read with the non-executing reader (parc_amd.util.safe_pickle) unless
``unsafe_pickle=True`` is passed, in which case ``pickle.load`` is used exactly as the reference does.
"""
import enum
import os

import numpy as np
import torch
import yaml

from .. import _hip
from ..util import safe_pickle
from ..util import terrain_util


class LoopMode(enum.Enum):
    CLAMP = 0
    WRAP = 1


def extract_pose_data(frame):
    return frame[..., 0:3], frame[..., 3:6], frame[..., 6:]


def _row_layout(num_bodies, dof_size):
    off_pos = 4 * num_bodies
    off_contacts = off_pos + 3
    off_root_vel = off_contacts + num_bodies
    off_root_ang_vel = off_root_vel + 3
    off_dof_vel = off_root_ang_vel + 3
    stride = (off_dof_vel + dof_size + 3) // 4 * 4
    return dict(off_pos=off_pos, off_contacts=off_contacts, off_root_vel=off_root_vel,
                off_root_ang_vel=off_root_ang_vel, off_dof_vel=off_dof_vel, row_stride=stride)


class MotionLib:
    def __init__(self, motion_input, kin_char_model, device, init_type="motion_file", loop_mode=None, fps=None,
                 contact_info=False, contacts=None, unsafe_pickle=False):
        self._device = device
        self._kin_char_model = kin_char_model
        self._contact_info = contact_info
        self._unsafe_pickle = unsafe_pickle
        self._hf_mask_inds = None
        if init_type == "motion_file":
            clips = self._read_motion_files(motion_input)
        elif init_type == "motion_frames":
            if isinstance(motion_input, torch.Tensor) and motion_input.is_cuda:
                # generated plans (the motion-generator sub-env rebuilds its library at every replan): stays on the device
                self._motion_names = self._motion_files = None
                self._terrains = []
                self._build_uniform(motion_input, contacts if contact_info else None, fps, loop_mode)
                return
            clips = self._clips_from_frames(motion_input, loop_mode, fps, contacts)
            self._build(clips)
            self._apply_motion_frames_dof_vel_rule(fps)
            return
        elif init_type == "clips":
            # in-memory clip dicts (parc_amd.synthetic.make_dataset): frames, contacts, fps, loop, weight[, name, hf...]
            clips = list(motion_input)
            self._motion_names = [c.get("name", "clip_%d" % i) for i, c in enumerate(clips)]
            self._motion_files = list(self._motion_names)
            self._terrains = []
            for c in clips:
                if "hf" in c:
                    self._terrains.append(terrain_util.SubTerrain.from_arrays(c["hf"], c["min_point"], c["dxdy"], device=device))
                else:
                    self._terrains.append(None)
        else:
            raise NotImplementedError("init_type {!r} (the diffusion loader is outside the tracker hot path)".format(init_type))
        self._build(clips)

    # ------------------------------------------------------------------ loading (reference :204-403)
    def _fetch_motion_files(self, motion_file):
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

    def _load_one(self, path):
        if self._unsafe_pickle:
            data = safe_pickle.load_executing(path)
            ter = data.get("terrain")
            return data, ter
        data = safe_pickle.load_motion_file_safe(path)
        ter = data.get("terrain")
        if isinstance(ter, dict):
            ter = terrain_util.SubTerrain.from_arrays(ter["hf"], ter["min_point"], ter["dxdy"], ter.get("hf_mask"),
                                                      ter.get("hf_maxmin"), name=ter.get("terrain_name", "terrain"), device="cpu")
        return data, ter

    def _read_motion_files(self, motion_file):
        files, weights = self._fetch_motion_files(motion_file)
        D = self._kin_char_model.get_dof_size()
        B = self._kin_char_model.get_num_joints()
        clips = []
        self._motion_files, self._motion_names, self._terrains, self._motion_extras = [], [], [], []
        for path, w in zip(files, weights):
            data, ter = self._load_one(path)
            fps = data.get("fps", 30)
            if isinstance(fps, np.ndarray):
                fps = fps.item()
            loop = data.get("loop_mode", "CLAMP")
            frames = data.get("frames")
            if frames is None:
                frames = np.zeros((3, 6 + D), dtype=np.float32)
            frames = np.asarray(frames, dtype=np.float32)
            if frames.ndim == 3 and frames.shape[0] == 1:
                frames = frames[0]
            con = data.get("contacts") if self._contact_info else None
            if con is not None:
                con = np.asarray(con, dtype=np.float32)
                if con.ndim == 3 and con.shape[0] == 1:
                    con = con[0]
            elif self._contact_info:
                con = np.zeros((frames.shape[0], B), dtype=np.float32)
            name = os.path.basename(os.path.splitext(path)[0])
            assert name not in self._motion_names, "motion names must be unique"
            self._motion_names.append(name)
            self._motion_files.append(path)
            self._motion_extras.append(data.get("extra"))
            if ter is not None:
                ter.update_old()
                ter.to_torch(self._device)
            self._terrains.append(ter)
            clips.append(dict(frames=frames, contacts=con, fps=float(fps), loop=LoopMode[loop].value, weight=float(w)))
        return clips

    def _clips_from_frames(self, motion_frames, loop_mode, fps, contacts):
        mf = motion_frames.detach().cpu().numpy() if isinstance(motion_frames, torch.Tensor) else np.asarray(motion_frames)
        if mf.ndim == 2:
            mf = mf[None]
        con = None
        if contacts is not None:
            con = contacts.detach().cpu().numpy() if isinstance(contacts, torch.Tensor) else np.asarray(contacts)
            if con.ndim == 2:
                con = con[None]
        return [dict(frames=mf[i].astype(np.float32), contacts=None if con is None else con[i].astype(np.float32),
                     fps=float(fps), loop=loop_mode.value, weight=1.0) for i in range(mf.shape[0])]

    def _build(self, clips):
        km = self._kin_char_model
        B, D = km.get_num_joints(), km.get_dof_size()
        dev = self._device
        M = len(clips)
        nf = np.array([c["frames"].shape[0] for c in clips], dtype=np.int64)
        fps = np.array([c["fps"] for c in clips], dtype=np.float64)
        start = np.concatenate([[0], np.cumsum(nf)[:-1]]).astype(np.int64)
        total = int(nf.sum())
        frames = np.concatenate([c["frames"] for c in clips], axis=0).astype(np.float32)
        has_con = all(c["contacts"] is not None for c in clips)
        contacts = np.concatenate([c["contacts"] for c in clips], axis=0).astype(np.float32) if has_con else None
        frame_clip = np.repeat(np.arange(M, dtype=np.int32), nf)

        w = torch.tensor([c["weight"] for c in clips], dtype=torch.float32, device=dev)
        self._motion_weights = w / w.sum()
        self._motion_fps = torch.tensor(fps, dtype=torch.float32, device=dev)
        self._motion_dt = torch.tensor(1.0 / fps, dtype=torch.float32, device=dev)
        self._motion_num_frames = torch.tensor(nf, dtype=torch.long, device=dev)
        self._motion_lengths = torch.tensor(1.0 / fps * (nf - 1), dtype=torch.float32, device=dev)   # :275
        self._motion_loop_modes = torch.tensor([c["loop"] for c in clips], dtype=torch.int, device=dev)
        self._motion_start_idx = torch.tensor(start, dtype=torch.long, device=dev)
        self._motion_ids = torch.arange(M, dtype=torch.long, device=dev)
        self._motion_frames = torch.tensor(frames, dtype=torch.float32, device=dev)
        delta = np.stack([c["frames"][-1, 0:3] - c["frames"][0, 0:3] for c in clips]).astype(np.float32)
        delta[:, 2] = 0.0                                                                             # :278-279
        self._motion_root_pos_delta = torch.tensor(delta, dtype=torch.float32, device=dev)

        # device-side tables of the C struct (int32 where the kernels index)
        self._d_num_frames = self._motion_num_frames.to(torch.int32).contiguous()
        self._d_start_idx = self._motion_start_idx.to(torch.int32).contiguous()
        self._layout = _row_layout(B, D)
        self._rows = torch.empty((total, self._layout["row_stride"]), dtype=torch.float32, device=dev)
        d_contacts = torch.tensor(contacts, dtype=torch.float32, device=dev) if contacts is not None else None
        d_clip = torch.tensor(frame_clip, dtype=torch.int32, device=dev)
        self._c_struct = None
        _hip.check(_hip.lib().parc_motion_lib_build(_hip.stream(), km.c_struct(), self.c_struct(), total,
                                                    _hip.ptr(self._motion_frames), _hip.ptr(d_contacts), _hip.ptr(d_clip),
                                                    _hip.ptr(self._motion_fps), _hip.ptr(self._rows)), "parc_motion_lib_build")
        torch.cuda.current_stream().synchronize()   # the temporaries above must outlive the launch

    def _build_uniform(self, frames, contacts, fps, loop_mode):
        """M clips of the same length from device tensors frames [M, F, 6 + D] (contacts [M, F, B] or None), reference
        _load_motion_frames :137-203: no per-clip weights (unit), same derived quantities as file clips.  Nothing is read back."""
        km = self._kin_char_model
        B, D = km.get_num_joints(), km.get_dof_size()
        dev = self._device
        if frames.dim() == 2:
            frames = frames.unsqueeze(0)
        if contacts is not None and contacts.dim() == 2:
            contacts = contacts.unsqueeze(0)
        M, F = int(frames.shape[0]), int(frames.shape[1])
        assert frames.shape[2] == 6 + D and F >= 1
        total = M * F
        f32, i64 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.long, device=dev)
        self._motion_weights = torch.ones(M, **f32)
        self._motion_fps = torch.full((M,), float(fps), **f32)
        self._motion_dt = torch.full((M,), 1.0 / fps, **f32)
        self._motion_num_frames = torch.full((M,), F, **i64)
        self._motion_lengths = torch.full((M,), 1.0 / fps * (F - 1), **f32)
        self._motion_loop_modes = torch.full((M,), loop_mode.value, dtype=torch.int, device=dev)
        self._motion_start_idx = torch.arange(M, **i64) * F
        self._motion_ids = torch.arange(M, **i64)
        self._motion_frames = torch.empty((total, 6 + D), **f32)
        self._motion_root_pos_delta = torch.zeros((M, 3), **f32)
        self._d_num_frames = self._motion_num_frames.to(torch.int32).contiguous()
        self._d_start_idx = self._motion_start_idx.to(torch.int32).contiguous()
        self._layout = _row_layout(B, D)
        self._rows = torch.empty((total, self._layout["row_stride"]), **f32)
        self._d_contacts = torch.empty((total, B), **f32) if contacts is not None else None
        self._d_clip = torch.arange(M, dtype=torch.int32, device=dev).repeat_interleave(F)
        self._uniform_shape = (M, F)
        self._uniform_fps = float(fps)
        self._c_struct = None
        self.update_frames(frames, contacts)

    def _apply_motion_frames_dof_vel_rule(self, fps):
        """Libraries built from frame tensors carry dof velocities divided by fps^2: the reference's _load_motion_frames hands `fps` to
        compute_frame_dof_vel where the file loader hands `dt` (anim/motion_lib.py:181 against :283; kin_char_model.py:543-581 divides
        by that argument).  Restated as it is - the velocity reward term and the re-spawn state of the generator sub-env see these values."""
        o = self._layout["off_dof_vel"]
        self._rows[:, o:o + self._kin_char_model.get_dof_size()] *= 1.0 / (float(fps) * float(fps))

    def update_frames(self, frames, contacts=None):
        """Replace the frames of a library built from device tensors (same [M, F, .] shape) and rebuild its rows in place: every
        pointer a kernel or a captured graph holds stays valid."""
        M, F = self._uniform_shape
        km = self._kin_char_model
        assert tuple(frames.shape[0:2]) == (M, F)
        self._motion_frames.copy_(frames.reshape(M * F, -1))
        if self._d_contacts is not None:
            assert contacts is not None
            self._d_contacts.copy_(contacts.reshape(M * F, -1))
        d = self._motion_root_pos_delta
        torch.sub(frames[:, -1, 0:3], frames[:, 0, 0:3], out=d)
        d[:, 2] = 0.0
        _hip.check(_hip.lib().parc_motion_lib_build(_hip.stream(), km.c_struct(), self.c_struct(), M * F, _hip.ptr(self._motion_frames),
                                                    _hip.ptr(self._d_contacts), _hip.ptr(self._d_clip), _hip.ptr(self._motion_fps),
                                                    _hip.ptr(self._rows)), "parc_motion_lib_build")
        self._apply_motion_frames_dof_vel_rule(self._uniform_fps)

    # ------------------------------------------------------------------ C-ABI view
    def c_struct(self):
        if self._c_struct is None:
            km = self._kin_char_model
            L = self._layout
            self._c_struct = _hip.MotionLibS(
                self.num_motions(), km.get_num_joints(), km.get_dof_size(), L["row_stride"], L["off_pos"], L["off_contacts"],
                L["off_root_vel"], L["off_root_ang_vel"], L["off_dof_vel"],
                _hip.ptr(self._d_num_frames), _hip.ptr(self._d_start_idx), _hip.ptr(self._motion_lengths),
                _hip.ptr(self._motion_loop_modes), _hip.ptr(self._motion_root_pos_delta), _hip.ptr(self._rows))
        return self._c_struct

    # ------------------------------------------------------------------ flat per-frame views (reference attribute names)
    def _col(self, off, n):
        return self._rows[:, off:off + n]

    @property
    def _frame_root_rot(self):
        return self._rows[:, 0:4]

    @property
    def _frame_joint_rot(self):
        B = self._kin_char_model.get_num_joints()
        return self._rows[:, 4:4 * B].reshape(-1, B - 1, 4)

    @property
    def _frame_root_pos(self):
        return self._col(self._layout["off_pos"], 3)

    @property
    def _frame_contacts(self):
        return self._col(self._layout["off_contacts"], self._kin_char_model.get_num_joints())

    @property
    def _frame_root_vel(self):
        return self._col(self._layout["off_root_vel"], 3)

    @property
    def _frame_root_ang_vel(self):
        return self._col(self._layout["off_root_ang_vel"], 3)

    @property
    def _frame_dof_vel(self):
        return self._col(self._layout["off_dof_vel"], self._kin_char_model.get_dof_size())

    # ------------------------------------------------------------------ reference API
    def num_motions(self):
        return self._motion_lengths.shape[0]

    def get_total_length(self):
        return torch.sum(self._motion_lengths).item()

    def sample_motions(self, n, motion_weights=None):
        if motion_weights is None:
            motion_weights = self._motion_weights
        return torch.multinomial(motion_weights, num_samples=n, replacement=True)

    def sample_time(self, motion_ids, truncate_time=None):
        phase = torch.rand(motion_ids.shape, device=self._device)
        motion_len = self._motion_lengths[motion_ids]
        if truncate_time is not None:
            assert truncate_time >= 0.0
            motion_len = motion_len - truncate_time
        return phase * motion_len

    def get_motion_length(self, motion_ids):
        return self._motion_lengths[motion_ids]

    def get_motion_loop_mode(self, motion_ids):
        return self._motion_loop_modes[motion_ids]

    def get_motion_loop_mode_enum(self, motion_id):
        return LoopMode(self._motion_loop_modes[motion_id].item())

    def get_motion_names(self):
        return self._motion_names

    def calc_motion_frame(self, motion_ids, motion_times):
        """(root_pos, root_rot, root_vel, root_ang_vel, joint_rot, dof_vel[, contacts])   reference :80-112"""
        km = self._kin_char_model
        B, D = km.get_num_joints(), km.get_dof_size()
        ids = motion_ids.reshape(-1).to(torch.int64).contiguous()
        times = motion_times.reshape(-1).to(torch.float32).contiguous()
        Q = ids.shape[0]
        dev = ids.device
        rp = torch.empty((Q, 3), dtype=torch.float32, device=dev)
        rr = torch.empty((Q, 4), dtype=torch.float32, device=dev)
        rv = torch.empty((Q, 3), dtype=torch.float32, device=dev)
        rav = torch.empty((Q, 3), dtype=torch.float32, device=dev)
        jr = torch.empty((Q, B - 1, 4), dtype=torch.float32, device=dev)
        dv = torch.empty((Q, D), dtype=torch.float32, device=dev)
        co = torch.empty((Q, B), dtype=torch.float32, device=dev)
        _hip.check(_hip.lib().parc_calc_motion_frame(_hip.stream(), self.c_struct(), Q, _hip.ptr(ids), _hip.ptr(times), _hip.ptr(rp),
                                                     _hip.ptr(rr), _hip.ptr(rv), _hip.ptr(rav), _hip.ptr(jr), _hip.ptr(dv),
                                                     _hip.ptr(co)), "parc_calc_motion_frame")
        ret = [rp, rr, rv, rav, jr, dv]
        if self._contact_info:
            ret.append(co)
        return tuple(ret)

    def joint_rot_to_dof(self, joint_rot):
        return self._kin_char_model.rot_to_dof(joint_rot)

    def calc_motion_phase(self, motion_ids, times):
        motion_len = self._motion_lengths[motion_ids]
        phase = times / motion_len
        wrap = self._motion_loop_modes[motion_ids] == LoopMode.WRAP.value
        phase = torch.where(wrap, phase - torch.floor(phase), phase)
        return torch.clip(phase, 0.0, 1.0)
