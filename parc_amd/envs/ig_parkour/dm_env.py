"""DeepMimic sub-environment: clip database, terrain tiling, reset sampling, per-clip failure rates.

Host-side mirror of the reference's ``envs/ig_parkour/dm_env.py`` (DeepMimicEnv :23-116, 188-356, 493-684, 786-875)
and of the ``RefCharEnv`` parts it inherits (envs/ig_parkour/mgdm_dm_util.py:23-136,232-276).  The per-step
arithmetic (reference pose sampling, target observations, termination, failure-rate EMA) runs in the HIP
kernels; this class owns the bookkeeping tensors and the sampling logic that the reference also does with torch
RNG calls (multinomial / rand) at reset time.
"""
import os

import numpy as np
import torch

from ...anim import motion_lib
from ...util import terrain_util
from .. import base_env


class DeepMimicEnv:
    def __init__(self, config, num_envs, device, visualize, char_model, core=None, motion_input=None):
        env_config = config["env"]
        dm = env_config["dm"]
        self._num_envs = num_envs
        self._device = device
        self._visualize = visualize
        self._kin_char_model = char_model
        self._timestep = 1.0 / env_config["control_freq"]
        self._rand_root_pos_offset_scale = env_config["rand_root_pos_offset_scale"]
        self._root_pos_offset = None
        self._root_rot_offset = None
        self._root_vel_offset = None
        self._root_ang_vel_offset = None
        self._dof_pos_offset = None
        self._dof_vel_offset = None
        self._max_obs_h = env_config["max_obs_h"]
        self._min_obs_h = env_config["min_obs_h"]
        self._random_reset_pos = dm.get("random_reset_pos", False)
        self._min_motion_weight = dm.get("min_motion_weight", 0.01)
        self._demo_mode = env_config["demo_mode"]
        self._rand_reset = env_config["rand_reset"]
        self._target_xy_future_time_max = dm["target_xy_future_time_max"]
        self._target_xy_future_time_min = dm["target_xy_future_time_min"]
        self._ignore_fail_rates = dm.get("ignore_fail_rates", False)
        self._terrains_per_motion = dm["terrains_per_motion"]
        self._fail_rate_quantiles = torch.tensor(dm["fail_rate_quantiles"], dtype=torch.float32, device=device)
        self._one_motion_mode = False
        self._selected_motion_id = 0
        self._terrain_build_mode = dm.get("terrain_build_mode", "square")
        self._build_tile_meshes = bool(dm.get("build_tile_meshes", True))     # the reference always builds them (for PhysX)
        self._all_terrain_verts, self._all_terrain_tris = [], []
        self._motion_classes = dm.get("motion_classes", [])
        self._has_motion_classes = False

        if motion_input is not None:            # in-memory clips (parc_amd.synthetic)
            self._motion_lib = motion_lib.MotionLib(motion_input, char_model, device, init_type="clips", contact_info=True)
        else:
            self._motion_lib = motion_lib.MotionLib(dm["motion_file"], char_model, device, contact_info=True,
                                                    unsafe_pickle=bool(dm.get("unsafe_pickle", False)))
        M = self._motion_lib.num_motions()
        if "fail_rates_path" in dm:
            self._motion_id_fail_rates = torch.load(dm["fail_rates_path"], weights_only=True).to(dtype=torch.float32, device=device)
        else:
            self._motion_id_fail_rates = torch.ones(M, dtype=torch.float32, device=device)
        self._ema_weight = 0.01
        self._motion_start_time_fraction = torch.zeros(num_envs, dtype=torch.float32, device=device)
        self._core = core
        self._dm_motion_offsets = None
        self._terrain = None

    # ------------------------------------------------------------------ buffers shared with the kernels
    def attach(self, core):
        self._core = core
        self._motion_ids = core.motion_ids
        self._motion_terrain_ids = core.motion_terrain_ids
        self._motion_time_offsets = core.motion_time_offsets
        self._time_buf = core.time_buf
        self._timestep_buf = core.timestep_buf
        self._done_buf = core.done
        self._ray_hfs = core.ray_hfs
        self._ray_xy_points = core.ray_xy_points
        self._all_ids = torch.arange(self._num_envs, device=self._device, dtype=torch.int64)

    # ------------------------------------------------------------------ terrain (reference :118-126,188-356,493-507)
    def build_terrain(self, env_config, terrain_save_path, x_offset=0.0, y_offset=0.0):
        if self._terrain_build_mode == "file":
            return self.load_motion_terrain_file(env_config, terrain_save_path)
        if self._terrain_build_mode == "wide":
            return self.build_terrain_wide(env_config, terrain_save_path, x_offset, y_offset)
        if self._terrain_build_mode != "square":
            raise AssertionError("unsupported terrain build mode")
        return self.build_terrain_square(env_config, terrain_save_path, x_offset, y_offset)

    @staticmethod
    def _tile_mesh(hf, min_x, min_y, dx, padding=0):
        """Collision mesh of one terrain tile as the reference builds it for the simulator (dm_env.py:257-286): two triangles
        for a flat tile, the voxelised column mesh otherwise.  The MI355X simulator collides with the heightfield columns
        directly; the meshes are produced for the return value / cache file of the reference's API."""
        hf = np.asarray(hf, np.float32)
        if abs(float(hf.max()) - float(hf.min())) < 1e-5:
            max_x, max_y = min_x + dx * (hf.shape[0] - 1), min_y + dx * (hf.shape[1] - 1)
            z = hf[0, 0]
            verts = np.array([[min_x - dx / 2, min_y - dx / 2, z], [max_x + dx / 2, min_y - dx / 2, z], [min_x - dx / 2, max_y + dx / 2, z],
                              [max_x + dx / 2, max_y + dx / 2, z]], dtype=np.float32)
            return verts, np.array([[0, 1, 2], [1, 3, 2]], dtype=np.uint32)
        return terrain_util.convert_heightfield_to_voxelized_trimesh(hf, min_x, min_y, dx, padding=padding)

    def _save_terrain_cache(self, terrain_save_path, all_verts=None, all_tris=None):
        self._all_terrain_verts = all_verts if all_verts is not None else []
        self._all_terrain_tris = all_tris if all_tris is not None else []
        if not terrain_save_path:
            return
        os.makedirs(os.path.dirname(terrain_save_path) or ".", exist_ok=True)
        # reference format (dm_env.py:344-354): the SubTerrain and the offsets as torch tensors (its load_terrain calls
        # set_device / .to on them), the per-clip meshes as numpy arrays; written from the CPU so any host can open it
        cpu_t = self._terrain.torch_copy()
        cpu_t.set_device("cpu")
        terrain_util.dump_reference_pickle({"terrain": cpu_t, "terrains_per_motion": self._terrains_per_motion,
                                            "motion_offsets": self._dm_motion_offsets.detach().cpu(),
                                            "all_terrain_verts": self._all_terrain_verts, "all_terrain_tris": self._all_terrain_tris},
                                           terrain_save_path)

    def load_motion_terrain_file(self, env_config, terrain_save_path):
        """terrain_build_mode "file" (reference :128-186): ONE shared terrain named by the motion YAML's `terrain:` key,
        per-clip offsets from the clips' optional `min_point_offset`."""
        import yaml
        from ...util import safe_pickle
        with open(env_config["dm"]["motion_file"], "r") as f:
            my = yaml.safe_load(f)
        loader = safe_pickle.load_executing if env_config["dm"].get("unsafe_pickle", False) else safe_pickle.load_motion_file_safe
        t = loader(my["terrain"])["terrain"]
        if isinstance(t, dict):
            t = terrain_util.SubTerrain.from_arrays(t["hf"], t["min_point"], t["dxdy"], t.get("hf_mask"), t.get("hf_maxmin"), device=self._device)
        else:
            t.to_torch(self._device)
        self._terrain = t
        self._terrains_per_motion = 1
        offs = []
        for elem in my["motions"]:
            d = loader(elem["file"])
            mpo = d.get("min_point_offset") if hasattr(d, "get") else None
            offs.append(np.zeros(2, np.float32) if mpo is None or isinstance(mpo, safe_pickle.Unresolved) else np.asarray(mpo, np.float32).reshape(2))
        self._dm_motion_offsets = torch.tensor(np.stack(offs), dtype=torch.float32, device=self._device).unsqueeze(1)
        # reference :141-156: one voxelised mesh of the shared terrain, min point / cell size as float32 .item() values
        verts, tris = terrain_util.convert_heightfield_to_voxelized_trimesh(t.hf, t.min_point[0].item(), t.min_point[1].item(),
                                                                            t.dxdy[0].item(), padding=0)
        self._save_terrain_cache(terrain_save_path, [[verts]], [[tris]])
        return self._all_terrain_verts, self._all_terrain_tris

    def build_terrain_wide(self, env_config, terrain_save_path, x_offset=0.0, y_offset=0.0):
        """terrain_build_mode "wide" (reference :362-491): clips side by side along x, `terrains_per_motion` copies along y,
        `padding` metres between neighbours."""
        hm = env_config["dm"]["heightmap"]
        dx = float(hm["horizontal_scale"])
        padding = float(hm["padding"])
        npad = int(round(padding / dx))
        ters = self._motion_lib._terrains
        M, R = self._motion_lib.num_motions(), self._terrains_per_motion
        offsets = torch.zeros((M, R, 2), dtype=torch.float32, device=self._device)
        gmin = np.zeros(2)
        gmax = np.zeros(2)
        xo = x_offset
        for i in range(M):
            t = ters[i]
            yo = y_offset
            for j in range(R):
                offsets[i, j, 0] = xo - t.min_point[0]
                offsets[i, j, 1] = yo - t.min_point[1]
                gmin = np.minimum(gmin, [xo, yo])
                gmax = np.maximum(gmax, [xo + t.dims[0].item() * dx, yo + t.dims[1].item() * dx])
                yo += dx * t.dims[1].item() + padding * 2.0
            xo += dx * t.dims[0].item() + padding * 2.0
        dims = np.round((gmax - gmin) / dx).astype(int)
        glob = terrain_util.SubTerrain("heightmap", int(dims[0]), int(dims[1]), dx, dx, float(gmin[0]), float(gmin[1]), device=self._device)
        sx = 0
        for i in range(M):
            t = ters[i]
            sy = 0
            for j in range(R):
                glob.hf[sx:sx + t.dims[0], sy:sy + t.dims[1]] = t.hf
                glob.hf_mask[sx:sx + t.dims[0], sy:sy + t.dims[1]] = t.hf_mask
                sy += int(t.dims[1].item()) + 2 * npad
            sx += int(t.dims[0].item()) + 2 * npad
        self._terrain = glob
        self._dm_motion_offsets = offsets
        self._save_terrain_cache(terrain_save_path)
        return [], []

    def build_terrain_square(self, env_config, terrain_save_path, x_offset=0.0, y_offset=0.0):
        hm = env_config["dm"]["heightmap"]
        dx = float(hm["horizontal_scale"])
        npad = hm["padding"] / dx
        assert abs(round(npad) - npad) < 1e-5
        npad = int(round(npad))
        assert self._terrains_per_motion == 1
        ters = self._motion_lib._terrains
        M = self._motion_lib.num_motions()
        assert all(t is not None for t in ters), "every clip needs its terrain for the tiled layout"
        dim_x = max(int(t.dims[0].item()) for t in ters) + 2 * npad
        dim_y = max(int(t.dims[1].item()) for t in ters) + 2 * npad
        n_side = int(np.ceil(np.sqrt(M)))
        first_x = -dim_x * n_side * dx / 2.0 + x_offset
        first_y = -dim_y * n_side * dx / 2.0 + y_offset
        glob = terrain_util.SubTerrain("heightmap", dim_x * n_side, dim_y * n_side, dx, dx, first_x, first_y, device=self._device)
        offsets = torch.zeros((M, self._terrains_per_motion, 2), dtype=torch.float32, device=self._device)
        k = 0
        all_verts, all_tris = [], []
        for i in range(n_side):
            for j in range(n_side):
                if k >= M:
                    break
                t = ters[k]
                t.pad(npad, torch.min(t.hf).item())                 # :245-249 pads with the clip terrain's minimum
                xo, yo = first_x + i * dim_x * dx, first_y + j * dim_y * dx
                if self._build_tile_meshes:
                    v, tr = self._tile_mesh(t.hf.cpu().numpy(), xo, yo, dx)
                    all_verts.append([v])
                    all_tris.append([tr])
                offsets[k, 0, 0] = xo - t.min_point[0]
                offsets[k, 0, 1] = yo - t.min_point[1]
                sx, sy = i * dim_x, j * dim_y
                glob.hf[sx:sx + t.hf.shape[0], sy:sy + t.hf.shape[1]] = t.hf
                glob.hf_mask[sx:sx + t.hf.shape[0], sy:sy + t.hf.shape[1]] = t.hf_mask
                k += 1
        self._terrain = glob
        self._dm_motion_offsets = offsets
        self._save_terrain_cache(terrain_save_path, all_verts, all_tris)
        return self._all_terrain_verts, self._all_terrain_tris

    def load_terrain(self, terrain_save_path):
        """reference :493-507.  Read with the non-executing reader, so a cache written by the reference itself (device
        tensors inside) opens too."""
        from ...util import safe_pickle
        data = safe_pickle.load_motion_file_safe(terrain_save_path)
        t, offs = data["terrain"], data.get("motion_offsets")
        if not isinstance(t, dict) or not isinstance(offs, (np.ndarray, torch.Tensor)):
            raise RuntimeError("{} is not a terrain cache this reader can open without executing it; delete it to "
                               "rebuild".format(terrain_save_path))
        self._terrain = terrain_util.SubTerrain.from_arrays(t["hf"], t["min_point"], t["dxdy"], t.get("hf_mask"), t.get("hf_maxmin"),
                                                            device=self._device)
        self._terrains_per_motion = int(data["terrains_per_motion"])
        self._dm_motion_offsets = torch.as_tensor(offs, dtype=torch.float32).to(self._device)
        self._all_terrain_verts = data.get("all_terrain_verts", [])
        self._all_terrain_tris = data.get("all_terrain_tris", [])
        return self._all_terrain_verts, self._all_terrain_tris

    def set_tiled(self, hf, min_point, dxdy, motion_offsets):
        """Install an already tiled global heightfield (parc_amd.synthetic.tile_square)."""
        self._terrain = terrain_util.SubTerrain.from_arrays(hf, min_point, dxdy, device=self._device)
        self._dm_motion_offsets = torch.tensor(motion_offsets, dtype=torch.float32, device=self._device)

    # ------------------------------------------------------------------ reset (reference :517-568,656-684)
    def _get_motion_times(self, env_ids=None):
        if env_ids is None:
            return self._time_buf + self._motion_time_offsets
        return self._time_buf[env_ids] + self._motion_time_offsets[env_ids]

    def sample_reset(self, env_ids):
        """Pick clip, tile and start time for the given envs and write the bookkeeping buffers.  The reference
        state itself is then produced by the post-step kernel (PARC_POST_REF) for these envs."""
        n = len(env_ids)
        ml = self._motion_lib
        if self._demo_mode:
            motion_ids = env_ids % ml.num_motions()
        elif self._one_motion_mode:
            motion_ids = torch.full_like(env_ids, self._selected_motion_id)
        elif self._ignore_fail_rates:
            motion_ids = ml.sample_motions(n)
        else:
            w = torch.clamp(self._motion_id_fail_rates, min=self._min_motion_weight) * ml._motion_weights
            motion_ids = ml.sample_motions(n, w)
        terrain_ids = torch.randint(high=self._terrains_per_motion, size=(n,), dtype=torch.int64, device=self._device)
        if self._rand_reset:
            motion_times = ml.sample_time(motion_ids)
        else:
            motion_times = ml.get_motion_length(motion_ids) * self._motion_start_time_fraction[env_ids]
        self._motion_ids[env_ids] = motion_ids
        self._motion_terrain_ids[env_ids] = terrain_ids
        self._motion_time_offsets[env_ids] = motion_times
        self._core.motion_xy_offset[env_ids] = self._dm_motion_offsets[motion_ids, terrain_ids]
        self._timestep_buf[env_ids] = 0
        self._time_buf[env_ids] = 0.0
        self._done_buf[env_ids] = base_env.DoneFlags.NULL.value

    def sample_reset_all(self):
        """Candidate (clip, tile, start time) for EVERY env, with fixed shapes and no host round trip (device-side reset:
        parc_reset_apply keeps them only where an episode ended).  Same distributions as sample_reset."""
        n = self._num_envs
        ml = self._motion_lib
        M = ml.num_motions()
        if self._demo_mode:
            motion_ids = self._all_ids % M
        elif self._one_motion_mode:
            motion_ids = torch.full_like(self._all_ids, self._selected_motion_id)
        else:
            w = ml._motion_weights
            if not self._ignore_fail_rates:
                w = torch.clamp(self._motion_id_fail_rates, min=self._min_motion_weight) * w
            cdf = torch.cumsum(w, dim=0)
            u = torch.rand(n, device=self._device, dtype=cdf.dtype) * cdf[-1]
            motion_ids = torch.searchsorted(cdf, u, right=True).clamp_(max=M - 1)      # multinomial with replacement
        terrain_ids = torch.randint(high=self._terrains_per_motion, size=(n,), dtype=torch.int64, device=self._device)
        if self._rand_reset:
            motion_times = ml.sample_time(motion_ids)
        else:
            motion_times = ml.get_motion_length(motion_ids) * self._motion_start_time_fraction
        return motion_ids, terrain_ids, motion_times

    # ------------------------------------------------------------------ accessors used by agent / recorder
    def set_rand_root_pos_offset_scale(self, val):
        self._rand_root_pos_offset_scale = val

    # fixed per-env offsets applied to the character state at reset (RefCharEnv.set_*_offset / apply_offsets_to_char_state,
    # mgdm_dm_util.py:138-157,236-276; the GUI caller sets position and heading offsets, ig_parkour_env.py:414-427)
    def _set_offset(self, name, val, width):
        if isinstance(val, torch.Tensor):
            assert val.shape[0] == self._num_envs and val.shape[1] == width
        setattr(self, name, val)

    def set_root_pos_offset(self, val=None):
        self._set_offset("_root_pos_offset", val, 3)

    def set_root_rot_offset(self, val=None):
        self._set_offset("_root_rot_offset", val, 4)

    def set_root_vel_offset(self, val=None):
        self._set_offset("_root_vel_offset", val, 3)

    def set_root_ang_vel_offset(self, val=None):
        self._set_offset("_root_ang_vel_offset", val, 3)

    def set_dof_pos_offset(self, val=None):
        self._set_offset("_dof_pos_offset", val, self._kin_char_model.get_dof_size())

    def set_dof_vel_offset(self, val=None):
        self._set_offset("_dof_vel_offset", val, self._kin_char_model.get_dof_size())

    def has_state_offsets(self):
        return any(getattr(self, n) is not None for n in ("_root_pos_offset", "_root_rot_offset", "_root_vel_offset", "_root_ang_vel_offset",
                                                          "_dof_pos_offset", "_dof_vel_offset"))

    def set_demo_mode(self, val=None):
        self._demo_mode = (not self._demo_mode) if val is None else val
        return self._demo_mode

    def set_motion_start_time_fraction(self, val):
        self._motion_start_time_fraction = val

    def get_env_motion_length(self, env_ids):
        return self._motion_lib.get_motion_length(self._motion_ids[env_ids])

    def get_env_motion_time(self, env_ids):
        return self._get_motion_times(env_ids)

    def get_env_motion_name(self, env_id):
        return self._motion_lib.get_motion_names()[self._motion_ids[env_id].item()]

    def post_test_update(self):
        return

    def get_extra_log_info(self):
        """Per-clip failure rates and their quantiles (reference :786-845)."""
        names = self._motion_lib.get_motion_names()
        fr = self._motion_id_fail_rates
        top, _ = torch.sort(fr, descending=True)
        q = torch.quantile(top, self._fail_rate_quantiles)
        # one transfer for everything (the reference reads a scalar per clip)
        nq = int(self._fail_rate_quantiles.shape[0])
        vals = torch.cat([fr.to(torch.float32), top[0:1], q.to(torch.float32), self._fail_rate_quantiles.to(torch.float32)]).tolist()
        M = len(names)
        info = {"MOTION_FAIL_RATES": {names[i]: vals[i] * 100.0 for i in range(M)},
                "Misc": {"top fail rate": vals[M] * 100.0}}
        for i in range(nq):
            key = "Fail Rate at " + str(round(vals[M + 1 + nq + i] * 100.0)) + "% Quantile"
            info["Misc"][key] = vals[M + 1 + i] * 100.0
        return info
