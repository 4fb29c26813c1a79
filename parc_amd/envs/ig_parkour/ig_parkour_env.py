"""Vectorised humanoid tracking environment on MI355X.

Drop-in for the reference's ``envs/ig_parkour/ig_parkour_env.py`` IGParkourEnv (+ its bases
``envs/ig_char_env.py`` IGCharEnv and ``envs/ig_env.py`` IGEnv): same constructor, ``reset`` / ``step`` /
``get_obs_space`` / ``get_action_space`` surface, the same persistent buffers returned by reference, and the
attribute names the agent / recorder reach into.  Isaac Gym is replaced by the HIP simulator
(include/parc_sim.h); reference-pose sampling, observations, reward and termination are one fused HIP launch
per step (include/parc_hip.h parc_track_post_step).

``fraction_dm_envs`` splits the env rows like the reference (ig_parkour_env.py:65-67): rows [0, n_dm) follow dataset clips (the
DeepMimic sub-env, the tracker default 1.0), rows [n_dm, N) follow plans of a motion generator (mgdm_env.py).  Each sub-env has its
own clip library and heightfield; every launch of a step (simulator, fused post-step) is issued once per sub-env on its row range, so
the default all-DeepMimic configuration runs exactly the launches it ran before.
"""
import math
import os
import pickle
import time
from collections import OrderedDict

import numpy as np
import torch

from ... import _hip
from ...anim import kin_char_model
from ...gym_spaces import Box
from ...sim_model import SimModel, action_bounds_pd
from ...tracker_core import TrackerConfig, TrackerCore
from ...util import geom_util, terrain_util, torch_util
from .. import base_env
from . import dm_env, mgdm_env

SIM_CHAR_IDX = 0


class IGParkourEnv(base_env.BaseEnv):
    NAME = "ig_parkour"

    def __init__(self, config, num_envs, device, visualize, motion_input=None, tiled_terrain=None):
        super().__init__(visualize=visualize)
        if visualize:
            raise NotImplementedError("the MI355X build has no viewer; run with visualize=False")
        self._start_compute_time = time.time()
        env_config = config["env"]
        self._config = config
        self._num_envs = num_envs
        self._device = device
        self._episode_length = env_config["episode_length"]
        self._info_snapshots = True
        self._env_spacing = env_config.get("env_spacing", 5)
        self._global_obs = env_config["global_obs"]
        self._fraction_dm_envs = env_config["fraction_dm_envs"]
        self._num_dm_envs = min(int(self._fraction_dm_envs * num_envs), num_envs)
        self._num_mgdm_envs = num_envs - self._num_dm_envs
        # the plan clock as one more observation column, for every row, when generator rows exist (ig_parkour_env.py:77,1227-1233)
        self._enable_replan_timer_obs = bool(env_config.get("enable_replan_timer_obs", False)) and self._num_mgdm_envs > 0
        self._output_motion_dir = env_config.get("output_motion_dir", "output/_motions/recorded_motions/")
        self._never_done = env_config.get("never_done", False)
        self._report_tracking_error = env_config.get("report_tracking_error", False)
        self._use_heightmap = True
        self._use_contact_info = env_config["use_contact_info"]
        self._enable_tar_obs = env_config.get("enable_tar_obs", True)
        self._rand_reset = env_config.get("rand_reset", True)
        self._demo_mode = env_config["demo_mode"]
        self._bypass_record_fail = False
        self._write_agent_states_flag = False
        self._record_obs = False

        # sim timing (envs/ig_env.py:100-118, YAML `sim:` block)
        sim_freq = env_config.get("sim_freq", 60)
        control_freq = env_config.get("control_freq", 10)
        assert sim_freq >= control_freq and sim_freq % control_freq == 0
        self._control_freq = control_freq
        self._timestep = 1.0 / control_freq
        self._sim_steps = int(sim_freq / control_freq)
        self._substeps = int(config.get("sim", {}).get("substeps", 2))
        self._sim_h = 1.0 / (sim_freq * self._substeps)

        # character
        self._kin_char_model = kin_char_model.KinCharModel(device)
        self._kin_char_model.load_char_file(env_config["char_file"])
        km = self._kin_char_model
        assert env_config["control_mode"] == "pd", "only the PD control mode of the tracker config is implemented"
        self._sim_model = SimModel(km)
        low, high = action_bounds_pd(km)
        self._action_space = Box(low=low.astype(np.float32), high=high.astype(np.float32))
        self._action_bound_low = torch.tensor(low, dtype=torch.float32, device=device)
        self._action_bound_high = torch.tensor(high, dtype=torch.float32, device=device)

        # heightmap ray fan (ig_parkour_env.py:139-155)
        self._ray_xy_points = geom_util.get_xy_points_cone(
            center=torch.zeros(2), dx=env_config["ray_dx"], num_neg=env_config["ray_points_behind"],
            num_pos=env_config["ray_points_ahead"], num_rays_neg=env_config["ray_num_left"],
            num_rays_pos=env_config["ray_num_right"], angle_between_rays=env_config["ray_angle"]).to(device)

        # sub-envs: dataset clips on rows [0, n_dm), generated plans on rows [n_dm, N) (ig_parkour_env.py:160-163)
        n_dm, n_mg = self._num_dm_envs, self._num_mgdm_envs
        self._dm_env = dm_env.DeepMimicEnv(config, n_dm, device, visualize, km, motion_input=motion_input) if n_dm > 0 else None
        self._mgdm_env = mgdm_env.MotionGenDeepMimicEnv(config, n_mg, device, visualize, km) if n_mg > 0 else None
        self._cfg = TrackerConfig(env_config, km, self._ray_xy_points.shape[0])
        self._core = TrackerCore(num_envs, device, km, self._dm_env._motion_lib if n_dm > 0 else None, self._cfg, self._ray_xy_points)
        if n_dm > 0:
            self._dm_env.attach(self._core)
        self._dm_ids = torch.arange(n_dm, device=device, dtype=torch.long) if n_mg > 0 else None     # None: a launch covers every row
        self._build_terrains(env_config, tiled_terrain)
        if n_dm > 0:
            self._core.set_terrain(self._dm_env._terrain)

        # env placement (ig_parkour_env.py:492-501)
        n_row = int(np.sqrt(num_envs))
        idx = torch.arange(num_envs, device=device)
        self._env_offsets = self._core.env_offsets
        self._env_offsets[:, 0] = self._env_spacing * 2 * (idx % n_row)
        self._env_offsets[:, 1] = self._env_spacing * 2 * torch.div(idx, n_row, rounding_mode="floor")

        self._build_sim_tensors(env_config)
        self._build_data_buffers()
        if n_mg > 0:
            self._mgdm_env.attach(self._core, n_dm)
            self._mgdm_env.replan()          # the first plans exist before the first observation (ig_parkour_env.py:796-798)
        self.set_write_agent_states_flag(env_config.get("write_agent_states", False))
        if self.is_writing_agent_states():
            self.build_agent_states_dict()

    # ------------------------------------------------------------------ construction helpers
    def _build_terrains(self, env_config, tiled_terrain):
        """ig_parkour_env.py:587-634: the generator's square first, the dataset tiles 30 m below it in y when both exist.  The
        simulator collides every sub-env's rows with that sub-env's own heightfield, so the two need not share a lattice."""
        x_off = y_off = 0.0
        if self.has_mgdm_envs():
            mg = self._mgdm_env
            path = env_config["mgdm"].get("terrain_save_path")
            if path and os.path.exists(path):
                _, _, min_point = mg.load_terrain(path)
            else:
                _, _, min_point = mg.build_terrain(env_config, path)
            x_off, y_off = min_point[0].item(), min_point[1].item() - 30.0
        if not self.has_dm_envs():
            return
        dm = self._dm_env
        if tiled_terrain is not None:
            dm.set_tiled(*tiled_terrain)
            return
        path = env_config["dm"].get("terrain_save_path")
        if path and os.path.exists(path):
            dm.load_terrain(path)
        else:
            dm.build_terrain(env_config, path, x_off, y_off)

    def _build_sim_tensors(self, env_config):
        """Views with the reference's names onto the Isaac-Gym-layout state tensors (ig_char_env.py:166-217,
        ig_parkour_env.py:685-718)."""
        c = self._core
        N, B, D = self._num_envs, self._cfg.num_bodies, self._cfg.dof_size
        self._root_state, self._dof_state = c.root_state, c.dof_state
        self._rigid_body_state, self._contact_forces = c.rigid_body_state, c.contact_forces
        self._char_root_pos = c.root_state[:, 0:3]
        self._char_root_rot = c.root_state[:, 3:7]
        self._char_root_vel = c.root_state[:, 7:10]
        self._char_root_ang_vel = c.root_state[:, 10:13]
        ds = c.dof_state.view(N, D, 2)
        self._char_dof_pos = ds[..., 0]
        self._char_dof_vel = ds[..., 1]
        rb = c.rigid_body_state.view(N, B, 13)
        self._char_rigid_body_pos = rb[..., 0:3]
        self._char_rigid_body_rot = rb[..., 3:7]
        self._char_rigid_body_vel = rb[..., 7:10]
        self._char_rigid_body_ang_vel = rb[..., 10:13]
        self._char_contact_forces = c.contact_forces.view(N, B, 3)
        self._ref_root_pos, self._ref_root_rot = c.ref_root_pos, c.ref_root_rot
        self._ref_root_vel, self._ref_root_ang_vel = c.ref_root_vel, c.ref_root_ang_vel
        self._ref_body_pos, self._ref_joint_rot = c.ref_body_pos, c.ref_joint_rot
        self._ref_dof_pos, self._ref_dof_vel, self._ref_contacts = c.ref_dof_pos, c.ref_dof_vel, c.ref_contacts
        self._key_body_ids = torch.tensor(self._cfg.key_body_ids, dtype=torch.long, device=self._device)
        self._contact_body_ids = torch.tensor(self._cfg.contact_body_ids, dtype=torch.long, device=self._device)
        self._num_rbs = B
        self._action_buffer = torch.zeros((N, D), dtype=torch.float32, device=self._device)
        init_pose = env_config.get("init_pose", None)
        self._init_pose = torch.tensor(init_pose if init_pose is not None else [0.0] * (6 + D), dtype=torch.float32, device=self._device)

    def _build_data_buffers(self):
        c = self._core
        N = self._num_envs
        self._reward_buf, self._done_buf = c.reward, c.done
        self._timestep_buf, self._time_buf = c.timestep_buf, c.time_buf
        self._ep_num_buf = torch.zeros(N, device=self._device, dtype=torch.int64)
        # The kernels write the fused row of cfg.obs_dim columns.  In the tracker's default configuration that IS the handed-out row;
        # a configuration with other segments (has_target_xy_obs, global_root_height_obs, enable_tar_obs / use_contact_info off, the
        # replan timer of a generator sub-env: ig_parkour_env.py:1163-1239) hands out a buffer of its own that one gather launch
        # (parc_assemble_obs) fills from the fused rows, the per-env extras and the plan clock.
        # state of the env's own random generator (step_randoms): [step counter, ticket] and the policy-noise buffer it fills
        self._rng_seed = None
        self._rng_state = torch.zeros(2, dtype=torch.int64, device=self._device)
        self._action_noise = torch.empty((N, self._cfg.dof_size), dtype=torch.float32, device=self._device)
        self._obs_shapes, cols = self._cfg.obs_layout(self._enable_replan_timer_obs)
        self._obs_cols = None if cols is None else torch.tensor(cols, dtype=torch.int32, device=self._device)
        self._obs_buf = c.obs if cols is None else torch.zeros((N, len(cols)), dtype=torch.float32, device=self._device)
        self._ray_hfs = c.ray_hfs
        self._target_xy, self._next_target_xy_time = c.target_xy, c.next_target_xy_time
        self._info = dict()
        self._all_env_ids = torch.arange(N, device=self._device, dtype=torch.long)
        names = ["pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty", "task_r1", "task_r2", "total_task_r"]
        self._reward_term_views = OrderedDict((n, c.reward_terms[i]) for i, n in enumerate(names))
        self._reward_all_names = ["total_r"] + names      # row order of c.reward_all

    # ------------------------------------------------------------------ env API (envs/base_env.py, envs/ig_env.py:51-98)
    def get_num_envs(self):
        return self._num_envs

    def get_reward_bounds(self):
        return (0.0, 1.0)

    def get_obs_space(self):
        return Box(low=-np.inf, high=np.inf, shape=[int(self._obs_buf.shape[1])], dtype=np.float32)

    def _publish_obs(self, env_ids=None):
        """rows the kernels just wrote -> the handed-out buffer (only for a configuration whose row is not the fused row)"""
        if self._obs_cols is None:
            return
        t = self._mgdm_env.get_mgdm_time_buf() if self._enable_replan_timer_obs else None
        self._core.assemble_obs(self._obs_cols, self._obs_buf, scalar=t, env_ids=env_ids)

    def step_randoms(self, action_dim, tick=None):
        """Every random number of one rollout step in ONE launch (parc_rng_step, counter-based Philox keyed by torch's seed at the first
        call): returns the N(0, 1) action noise [N, action_dim] of the policy and refills the env's uniform pool (xy-target resample,
        restart sampling) that the step() which follows consumes (the key is drawn from torch's host generator at the first call) - instead of two launches of torch's generator, which inside a replayed
        hipGraph also cost two fills of its seed / offset cells per replay."""
        c = self._core
        if self._rng_seed is None:
            # the key comes out of torch's (host) generator at the first call: torch.manual_seed() fixes it like it fixes torch's own
            # draws, and two envs of one process get different keys.  (A host-side draw: the device cells - step counter, noise buffer -
            # exist since construction, so a first call inside a hipGraph capture allocates and fills nothing that a replay would repeat.)
            self._rng_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        assert action_dim == self._action_noise.shape[1]
        # tick = (device int64 cell, modulus): the caller's per-step counter (the experience buffer's write row), moved on by this launch
        cell, mod = (None, 0) if tick is None else tick
        _hip.check(_hip.lib().parc_rng_step(_hip.stream(), self._rng_seed, _hip.ptr(self._rng_state), _hip.ptr(c.rand_pool), c.rand_pool.numel(),
                                            _hip.ptr(self._action_noise), self._action_noise.numel(), _hip.ptr(cell), int(mod)), "parc_rng_step")
        c.rand_pool_prefilled = True
        return self._action_noise

    def _draw_step_uniforms(self):
        """the uniform pool of this step: already drawn together with the policy's noise (step_randoms), or one torch launch"""
        c = self._core
        if getattr(c, "rand_pool_prefilled", False):
            c.rand_pool_prefilled = False
        else:
            c.rand_pool.uniform_()
        c.rand_pool_fresh = True

    def _finish_reward(self):
        """rel_task_w > 0: the task term multiplies the tracking reward (ig_parkour_env.py:1399-1404; the kernel left deepmimic_r in
        the reward buffer and total_task_r in its term row, TrackerConfig)"""
        if self._cfg.rel_task_w > 0:
            c = self._core
            torch.mul(c.reward, c.reward_terms[8], out=c.reward)

    def has_dm_envs(self):
        return self._num_dm_envs > 0

    def has_mgdm_envs(self):
        return self._num_mgdm_envs > 0

    def get_dm_env(self):
        return self._dm_env

    def get_mgdm_env(self):
        return self._mgdm_env

    # (replanning belongs to the motion-generator sub-env; a pure tracker reports zeros - persistent tensors, not a fill per call)
    def get_replan_time_buf(self):
        if self.has_mgdm_envs():
            return self._mgdm_env._replan_time_buf
        if getattr(self, "_replan_time_zero", None) is None:
            self._replan_time_zero = torch.zeros(1, dtype=torch.float32, device=self._device)
        return self._replan_time_zero

    def get_replan_counter(self):
        if self._num_dm_envs == 0:
            return self._mgdm_env.get_replan_counter()
        if getattr(self, "_replan_counter_zero", None) is None:
            self._replan_counter_zero = torch.zeros(self._num_dm_envs, dtype=torch.int64, device=self._device)
        if not self.has_mgdm_envs():
            return self._replan_counter_zero
        return torch.cat([self._replan_counter_zero, self._mgdm_env.get_replan_counter()], dim=0)

    def apply_hard_reset(self):
        if self.has_mgdm_envs():
            self._mgdm_env.apply_hard_reset()

    # ``_episode_length`` is written by callers (dm_ppo_agent.record_motions of the reference sets it to 1000 s): the value the
    # kernels see lives in the config struct, so the attribute writes through
    @property
    def _episode_length(self):
        return self._episode_length_value

    @_episode_length.setter
    def _episode_length(self, val):
        self._episode_length_value = val
        cfg = getattr(self, "_cfg", None)
        if cfg is not None:
            cfg.struct.episode_length = float(val)

    def host_step_signature(self):
        """The host-side parameters a step / reset bakes into its kernel launches (a captured hipGraph of the step is only valid
        while they keep these values; learning/dm_ppo_agent keys its graphs by this tuple)."""
        dm = self._dm_env

        def sig(v):
            return ("tensor", v.data_ptr()) if torch.is_tensor(v) else v        # device-resident values are read by the graph itself
        out = (float(self._cfg.struct.episode_length),)
        if dm is not None:
            out += (dm._rand_reset, dm._demo_mode, sig(dm._rand_root_pos_offset_scale), sig(dm._motion_start_time_fraction), dm.has_state_offsets())
        mg = self._mgdm_env
        if mg is not None:         # (a replan may rebuild the plans' clip library with another shape: its table pointers are launch arguments)
            ml = mg._motion_lib
            # the plans' clip library enters the launches as pointers AND shapes: a library rebuilt at the same address with other
            # dimensions must not replay a graph captured for the old one
            out += ("mgdm", mg._demo_mode, None if ml is None else (ml._rows.data_ptr(), tuple(ml._rows.shape), ml.num_motions()),
                    mg._dont_auto_update_targets)
        return out

    def host_step_state(self):
        """what a step changes on the HOST (the generator sub-env's fp32 plan clock and its replan flag): snapshot / restore around a
        hipGraph capture that fails - the capture runs the step's host code once, the eager step that follows runs it again"""
        mg = self._mgdm_env
        return None if mg is None else (mg._plan_time_host, mg._replan_flag)

    def restore_host_step_state(self, state):
        if state is not None:
            self._mgdm_env._plan_time_host, self._mgdm_env._replan_flag = state

    def set_rand_reset(self, val=None):
        val = (not self._rand_reset) if val is None else val
        if self.has_dm_envs():
            self._dm_env._rand_reset = val
        self._rand_reset = val

    def set_demo_mode(self, val):
        self._demo_mode = val
        for sub in (self._dm_env, self._mgdm_env):
            if sub is not None:
                sub._demo_mode = val

    def set_rand_root_pos_offset_scale(self, val):
        for sub in (self._dm_env, self._mgdm_env):
            if sub is not None:
                sub.set_rand_root_pos_offset_scale(val)

    def get_extra_log_info(self):
        return self._dm_env.get_extra_log_info() if self.has_dm_envs() else {}

    def post_test_update(self):
        if self.has_dm_envs():
            self._dm_env.post_test_update()

    # ------------------------------------------------------------------ reset (ig_parkour_env.py:1012-1041, dm_env.py:656-684)
    def reset(self, env_ids=None):
        if env_ids is None:
            env_ids = self._all_env_ids
        env_ids = env_ids.to(torch.long)
        if self.has_mgdm_envs():
            n_dm = self._num_dm_envs
            if self.has_dm_envs():
                self._reset_dm(env_ids[env_ids < n_dm])
            self._reset_mgdm(env_ids[env_ids >= n_dm] - n_dm)
        else:
            self._reset_dm(env_ids)
        self._update_info()
        return self._obs_buf, self._info

    def _refresh_bodies(self, env_ids):
        c = self._core
        _hip.check(_hip.lib().parc_sim_refresh_bodies(_hip.stream(), self._sim_model.device_ptr(self._device), self._num_envs,
                                                      _hip.ptr(env_ids), int(env_ids.shape[0]), _hip.ptr(c.root_state),
                                                      _hip.ptr(c.dof_state), _hip.ptr(c.rigid_body_state), _hip.ptr(c.contact_forces)),
                   "parc_sim_refresh_bodies")

    def _reset_mgdm(self, rel_ids):
        """The sub-env decides (mgdm_env.reset: a pending replan re-plans ALL its envs, otherwise the given ones restart on their
        plan's current frame) and flags the rows whose simulator state it rewrote; like the reference (ig_parkour_env.py:1031-1039)
        only those rows get new body poses, new observations and an episode count."""
        mg = self._mgdm_env
        mg.reset(rel_ids)
        changed = mg._need_refresh.nonzero().flatten()
        if len(changed) > 0:
            mg._need_refresh[:] = False
            ids = changed + self._num_dm_envs
            self._refresh_bodies(ids)
            mg._post(_hip.POST_OBS | _hip.POST_HF, changed)
            self._publish_obs(ids)
            self._ep_num_buf[ids] += 1

    def _reset_dm(self, env_ids):
        c = self._core
        if len(env_ids) > 0:
            dm = self._dm_env
            dm.sample_reset(env_ids)
            c.post_step(_hip.POST_REF, env_ids)                       # reference state at the sampled clip time
            # RefCharEnv._char_state_init_from_ref + add_noise_to_char_state (mgdm_dm_util.py:119-136)
            self._char_root_pos[env_ids] = c.ref_root_pos[env_ids]
            self._char_root_rot[env_ids] = c.ref_root_rot[env_ids]
            self._char_root_vel[env_ids] = c.ref_root_vel[env_ids]
            self._char_root_ang_vel[env_ids] = c.ref_root_ang_vel[env_ids]
            self._char_dof_pos[env_ids] = c.ref_dof_pos[env_ids]
            self._char_dof_vel[env_ids] = c.ref_dof_vel[env_ids]
            if dm._rand_root_pos_offset_scale != 0.0:
                noise = torch.rand((env_ids.shape[0], 2), device=self._device, dtype=torch.float32) * 2.0 - 1.0
                self._char_root_pos[env_ids, 0:2] += dm._rand_root_pos_offset_scale * noise
            # RefCharEnv.apply_offsets_to_char_state (mgdm_dm_util.py:138-157): fixed offsets on top of the reference state
            if dm._root_pos_offset is not None:
                self._char_root_pos[env_ids] += dm._root_pos_offset[env_ids]
            if dm._root_rot_offset is not None:
                self._char_root_rot[env_ids] = torch_util.quat_mul(dm._root_rot_offset[env_ids], self._char_root_rot[env_ids])
            if dm._root_vel_offset is not None:
                self._char_root_vel[env_ids] += dm._root_vel_offset[env_ids]
            if dm._root_ang_vel_offset is not None:
                self._char_root_ang_vel[env_ids] += dm._root_ang_vel_offset[env_ids]
            if dm._dof_pos_offset is not None:
                self._char_dof_pos[env_ids] += dm._dof_pos_offset[env_ids]
            if dm._dof_vel_offset is not None:
                self._char_dof_vel[env_ids] += dm._dof_vel_offset[env_ids]
            self._next_target_xy_time[env_ids] = 0.0
            # publish body poses of the new state, then observations for these envs only
            self._refresh_bodies(env_ids)
            c.target_rand.uniform_()
            c.post_step(_hip.POST_OBS | _hip.POST_HF | _hip.POST_TARGETS, env_ids)      # + xy target resample for these envs
            self._publish_obs(env_ids)
            self._ep_num_buf[env_ids] += 1

    # ------------------------------------------------------------------ device-side reset of finished envs
    def supports_device_reset(self):
        """may the rows that finish in the NEXT step restart on the device (reset_done)?  Dataset rows: always, unless the GUI's state
        offsets are set.  Generator rows: unless the reset after that step has to call the generator (replan) - a host decision the
        fp32 mirror of the plan clock answers without a read-back - and only in the training configuration of the dataset rows."""
        dm, mg = self._dm_env, self._mgdm_env
        if dm is not None and (dm.has_state_offsets() or dm._dm_motion_offsets is None):
            return False
        if mg is not None:
            if mg.replan_pending_after_next_step(self._timestep):
                return False
            if dm is not None and (dm._demo_mode or dm._one_motion_mode or not dm._rand_reset):
                return False
        return True

    def supports_graph_step(self):
        """every launch of a step is fixed-shape, for generator rows too (targets drawn for all rows and taken under a mask, the
        replan-time rule decided on the device); what stays on the host is the replan itself, which happens in a reset, not in a step"""
        return True

    def host_step_replayed(self):
        """host-side bookkeeping of a step that ran as a replayed hipGraph (learning/dm_ppo_agent calls it after every replay)"""
        if self._mgdm_env is not None:
            self._mgdm_env.host_step_replayed(self._timestep)

    def reset_done(self, done=None):
        """reset(nonzero(done)) without the nonzero: the same state changes as ``reset(env_ids)`` for every env whose
        done flag is set, as fixed-shape device work (no host sync, capturable in the rollout hipGraph).  Candidates are
        drawn for all envs and applied where the flag is set (parc_reset_apply), the reference pose / character state /
        observations of those envs come from masked launches of the post-step kernel."""
        c, dm = self._core, self._dm_env
        L = _hip.lib()
        done = self._done_buf if done is None else done
        mixed = self.has_mgdm_envs()
        if mixed:
            c.reset_mask.copy_(done != base_env.DoneFlags.NULL.value)       # all rows; the dataset rows' launch rewrites its part
        # the dataset rows are rows [0, N) of every buffer: a mixed env launches the same kernels on that row range
        N = self._num_dm_envs
        rows = (0, N) if mixed else None
        if N == 0:
            return self._reset_done_mgdm_rows()
        ml = dm._motion_lib
        if not (dm._demo_mode or dm._one_motion_mode) and dm._rand_reset and done.dtype == torch.int32 and done.is_contiguous():
            # the training configuration: sampling + bookkeeping in two launches from the step's uniform pool (no torch ops at all)
            if not c.rand_pool_fresh:
                c.rand_pool.uniform_()             # reset_done outside a step (tests, tools)
            c.rand_pool_fresh = False
            M = ml.num_motions()
            if c.reset_cdf is None or c.reset_cdf.numel() < M:
                c.reset_cdf = torch.empty(M, dtype=torch.float32, device=self._device)
            offs = dm._dm_motion_offsets
            assert offs.is_contiguous() and offs.dim() == 3
            fr = None if dm._ignore_fail_rates else dm._motion_id_fail_rates
            _hip.check(L.parc_reset_sample_apply(_hip.stream(), N, _hip.ptr(done), _hip.ptr(c.reset_mask), _hip.ptr(c.reset_uniforms), M,
                                                 _hip.ptr(ml._motion_weights), _hip.ptr(fr), float(dm._min_motion_weight),
                                                 _hip.ptr(ml._motion_lengths), _hip.ptr(offs), int(offs.shape[1]),
                                                 float(dm._rand_root_pos_offset_scale), _hip.ptr(c.reset_cdf), _hip.ptr(c.motion_ids),
                                                 _hip.ptr(c.motion_terrain_ids), _hip.ptr(c.motion_time_offsets), _hip.ptr(c.motion_xy_offset),
                                                 _hip.ptr(c.timestep_buf), _hip.ptr(c.time_buf), _hip.ptr(c.done), _hip.ptr(c.next_target_xy_time),
                                                 _hip.ptr(self._ep_num_buf), _hip.ptr(c.init_noise_xy)), "parc_reset_sample_apply")
            c.post_step(_hip.POST_REF | _hip.POST_INIT_CHAR | _hip.POST_MASKED, rows=rows)
            _hip.check(L.parc_sim_refresh_bodies_masked(_hip.stream(), self._sim_model.device_ptr(self._device), N, _hip.ptr(c.reset_mask),
                                                        _hip.ptr(c.root_state), _hip.ptr(c.dof_state), _hip.ptr(c.rigid_body_state),
                                                        _hip.ptr(c.contact_forces)), "parc_sim_refresh_bodies_masked")
            c.post_step(_hip.POST_OBS | _hip.POST_HF | _hip.POST_MASKED | _hip.POST_TARGETS, reset_rand=True, rows=rows)
            if mixed:
                return self._reset_done_mgdm_rows()
            self._publish_obs()
            self._update_info()
            return self._obs_buf, self._info
        assert not mixed, "generator rows restart on the device only in the training configuration (supports_device_reset)"
        c.reset_mask.copy_(done != base_env.DoneFlags.NULL.value)
        new_mid, new_tid, new_t = dm.sample_reset_all()
        offs = dm._dm_motion_offsets
        assert offs.is_contiguous() and offs.dim() == 3
        _hip.check(L.parc_reset_apply(_hip.stream(), N, _hip.ptr(c.reset_mask), _hip.ptr(new_mid), _hip.ptr(new_tid),
                                      _hip.ptr(new_t.contiguous()), _hip.ptr(offs), int(offs.shape[1]), _hip.ptr(c.motion_ids),
                                      _hip.ptr(c.motion_terrain_ids), _hip.ptr(c.motion_time_offsets), _hip.ptr(c.motion_xy_offset),
                                      _hip.ptr(c.timestep_buf), _hip.ptr(c.time_buf), _hip.ptr(c.done), _hip.ptr(c.next_target_xy_time),
                                      _hip.ptr(self._ep_num_buf)), "parc_reset_apply")
        sc = dm._rand_root_pos_offset_scale
        if sc != 0.0:
            c.init_noise_xy.uniform_(-sc, sc)
        else:
            c.init_noise_xy.zero_()
        c.post_step(_hip.POST_REF | _hip.POST_INIT_CHAR | _hip.POST_MASKED)
        _hip.check(L.parc_sim_refresh_bodies_masked(_hip.stream(), self._sim_model.device_ptr(self._device), N, _hip.ptr(c.reset_mask),
                                                    _hip.ptr(c.root_state), _hip.ptr(c.dof_state), _hip.ptr(c.rigid_body_state),
                                                    _hip.ptr(c.contact_forces)), "parc_sim_refresh_bodies_masked")
        c.target_rand.uniform_()
        c.post_step(_hip.POST_OBS | _hip.POST_HF | _hip.POST_MASKED | _hip.POST_TARGETS)
        self._publish_obs()
        self._update_info()
        return self._obs_buf, self._info

    def _reset_done_mgdm_rows(self):
        """the generator rows' share of reset_done (c.reset_mask holds done != NULL for every row): a soft reset - the character goes back
        onto the current frame of its plan - as masked, fixed-shape work; only reached while no replan is pending (supports_device_reset)"""
        c, mg = self._core, self._mgdm_env
        e0, n = self._num_dm_envs, self._num_mgdm_envs
        B, D = self._cfg.num_bodies, self._cfg.dof_size
        mask = c.reset_mask[e0:]
        mg.reset_rows_masked(mask != 0)
        p = _hip.ptr
        _hip.check(_hip.lib().parc_sim_refresh_bodies_masked(_hip.stream(), self._sim_model.device_ptr(self._device), n, p(mask), p(c.root_state[e0:]),
                                                             p(c.dof_state[e0 * D:]), p(c.rigid_body_state[e0 * B:]), p(c.contact_forces[e0 * B:])),
                   "parc_sim_refresh_bodies_masked")
        mg._post(_hip.POST_OBS | _hip.POST_HF | _hip.POST_MASKED)
        self._ep_num_buf[e0:] += mask
        self._publish_obs()
        self._update_info()
        return self._obs_buf, self._info

    # ------------------------------------------------------------------ step (ig_env.py:68-86,839-848)
    def step(self, action):
        c = self._core
        act = action.to(dtype=torch.float32).contiguous()
        if self.has_mgdm_envs():
            return self._step_sub_envs(act)
        # _pre_physics_step + _physics_step: PD targets = clipped action, sim_steps x substeps at h; _update_time (ig_env.py:862-865:
        # timestep += 1, time = timestep * dt) rides in the same launch
        L = _hip.lib()
        sim_args = (_hip.stream(), self._sim_model.device_ptr(self._device), c._terrain_struct, self._num_envs, _hip.ptr(c.root_state),
                    _hip.ptr(c.dof_state), _hip.ptr(c.rigid_body_state), _hip.ptr(c.contact_forces), _hip.ptr(c.env_offsets), _hip.ptr(act),
                    _hip.ptr(self._action_bound_low), _hip.ptr(self._action_bound_high), self._sim_steps * self._substeps, self._sim_h)
        _hip.check(L.parc_sim_step_tick(*sim_args, _hip.ptr(self._timestep_buf), _hip.ptr(self._time_buf), float(self._timestep)),
                   "parc_sim_step_tick")
        # _update_misc (incl. the xy target resample) / _update_observations / _update_reward / _update_done in one launch
        self._draw_step_uniforms()      # all uniforms of this step and of the restarts that follow it (tracker_core.rand_pool)
        # (the reference STATE - ref_* buffers - rides in the fail-rate launch below: nothing in the fused launch reads it)
        c.post_step(_hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF | _hip.POST_TARGETS)
        c.step_tail(self._dm_env._motion_id_fail_rates, self._dm_env._ema_weight)
        self._finish_reward()
        self._publish_obs()
        if self._never_done:
            self._done_buf[:] = base_env.DoneFlags.NULL.value
        self._update_info(step=True)
        if self._write_agent_states_flag:
            self.write_agent_states()
        return self._obs_buf, self._reward_buf, self._done_buf, self._info

    def _sim_rows(self, act, e0, n, terrain_struct):
        """the simulator launch (+ env clock) for rows [e0, e0 + n) against one sub-env's heightfield"""
        c = self._core
        B, D = self._cfg.num_bodies, self._cfg.dof_size
        p = _hip.ptr
        _hip.check(_hip.lib().parc_sim_step_tick(_hip.stream(), self._sim_model.device_ptr(self._device), terrain_struct, n, p(c.root_state[e0:]),
                                                 p(c.dof_state[e0 * D:]), p(c.rigid_body_state[e0 * B:]), p(c.contact_forces[e0 * B:]),
                                                 p(c.env_offsets[e0:]), p(act[e0:]), p(self._action_bound_low), p(self._action_bound_high),
                                                 self._sim_steps * self._substeps, self._sim_h, p(self._timestep_buf[e0:]), p(self._time_buf[e0:]),
                                                 float(self._timestep)), "parc_sim_step_tick")

    def _step_sub_envs(self, act):
        """The same step when rows are split between the two sub-envs: every launch once per sub-env on its rows, in the reference's
        order (ig_env.py:839-848 with ig_parkour_env.py:996-1010,1521-1535): pre-physics hooks, simulate, clocks, update_misc, then
        reference pose / observations / reward / termination, then the generator sub-env's own termination rules."""
        c, mg = self._core, self._mgdm_env
        n_dm, n_mg = self._num_dm_envs, self._num_mgdm_envs
        mg.pre_physics_step()
        if n_dm > 0:
            self._sim_rows(act, 0, n_dm, c._terrain_struct)
        self._sim_rows(act, n_dm, n_mg, mg.terrain_struct())
        mg.update_time(self._timestep)
        mg.update_misc(fixed_shape=True)
        if n_dm > 0:
            self._draw_step_uniforms()
            c.post_step(_hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF | _hip.POST_TARGETS, rows=(0, n_dm))
            c.step_tail(self._dm_env._motion_id_fail_rates, self._dm_env._ema_weight, rows=(0, n_dm))     # fail rates + reference state
        mg._post(_hip.POST_REF | _hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF)
        mg.update_done_extra(fixed_shape=True)
        self._finish_reward()
        self._publish_obs()
        if self._never_done:
            self._done_buf[:] = base_env.DoneFlags.NULL.value
        self._update_info(step=True)
        if self._write_agent_states_flag:
            self.write_agent_states()
        return self._obs_buf, self._reward_buf, self._done_buf, self._info

    def _update_motion_targets(self):
        """DeepMimicEnv._update_motion_targets (dm_env.py:617-654) as torch ops: every env draws a candidate, only those whose
        timer expired take it.  The env itself now does this inside the post-step launch (PARC_POST_TARGETS); this version is
        kept as the readable statement of the rule (tests compare the two).  Feeds the logged task reward only."""
        c = self._core
        dm = self._dm_env
        N = self._num_envs
        due = self._time_buf >= self._next_target_xy_time
        fut = torch.rand(N, dtype=torch.float32, device=self._device)
        fut = fut * (dm._target_xy_future_time_max - dm._target_xy_future_time_min) + dm._target_xy_future_time_min
        times = dm._get_motion_times() + fut
        root_pos = dm._motion_lib.calc_motion_frame(c.motion_ids, times)[0]
        xy = root_pos[:, 0:2] + c.motion_xy_offset - c.env_offsets[:, 0:2] + torch.randn((N, 2), device=self._device) * 0.05
        self._target_xy[:] = torch.where(due.unsqueeze(-1), xy, self._target_xy)
        self._next_target_xy_time[:] = torch.where(due, self._time_buf + fut, self._next_target_xy_time)

    def _update_info(self, step=False):
        info = self._info
        # snapshots, as the reference hands out (ig_parkour_env.py:1428,1543-1547) - except inside a captured rollout step, whose only
        # consumer (the agent's record / return-tracker kernels) reads the entries before anything overwrites the buffers
        snap = (lambda t: t.clone()) if self._info_snapshots else (lambda t: t)
        info["timestep"] = snap(self._timestep_buf)
        info["ep_num"] = snap(self._ep_num_buf)
        info["compute_time"] = time.time() - self._start_compute_time
        info["char_contact_forces"] = snap(self._char_contact_forces)
        if step:
            r = dict(self._reward_term_views)
            r["total_r"] = snap(self._reward_buf)
            info["rewards"] = r
            info["rewards_all"] = (self._reward_all_names, self._core.reward_all)     # same data as one [10, N] block
            if self._report_tracking_error:
                info["tracking_error"] = self._compute_tracking_error()

    def _compute_tracking_error(self):
        """compute_tracking_error (mgdm_dm_util.py:578-611) with torch ops on the published state (test-time metric)."""
        km = self._kin_char_model
        c = self._core
        jr = km.dof_to_rot(self._char_dof_pos.contiguous())
        bp, br = km.forward_kinematics(self._char_root_pos.contiguous(), self._char_root_rot.contiguous(), jr)
        rbp, rbr = km.forward_kinematics(c.ref_root_pos, c.ref_root_rot, c.ref_joint_rot)

        qangle = torch_util.quat_diff_angle          # 2 atan2(|v|, w) of the w >= 0 difference, like the reference (accurate near 0)
        pose_err = qangle(br, rbr).mean(dim=-1)
        root_pos_err = torch.linalg.vector_norm(c.ref_root_pos - self._char_root_pos, dim=-1)
        body_err = torch.linalg.vector_norm((rbp - c.ref_root_pos.unsqueeze(1)) - (bp - self._char_root_pos.unsqueeze(1)), dim=-1).mean(dim=-1)
        root_rot_err = qangle(self._char_root_rot, c.ref_root_rot)
        dof_vel_err = (c.ref_dof_vel - self._char_dof_vel).abs().mean(dim=-1)
        rv = (c.ref_root_vel - self._char_root_vel).abs().mean(dim=-1)
        rav = (c.ref_root_ang_vel - self._char_root_ang_vel).abs().mean(dim=-1)
        return torch.stack([root_pos_err, root_rot_err, body_err, pose_err, dof_vel_err, rv, rav], dim=-1)

    # ------------------------------------------------------------------ observation helpers used by the agent
    def _compute_obs(self, env_ids=None, ret_obs_shapes=False):
        """Observation rows for env_ids (all if None); with ret_obs_shapes the ordered segment table the agent uses to
        build the normaliser's index set (ig_parkour_env.py:1163-1239, learning/dm_ppo_agent.py:91-109)."""
        if ret_obs_shapes:
            return OrderedDict((k, dict(v)) for k, v in self._obs_shapes.items())
        if self.has_mgdm_envs():
            ids = self._all_env_ids if env_ids is None else env_ids.to(torch.long)
            n_dm = self._num_dm_envs
            if n_dm > 0:
                self._core.post_step(_hip.POST_OBS | _hip.POST_HF, ids[ids < n_dm])
            self._mgdm_env._post(_hip.POST_OBS | _hip.POST_HF, ids[ids >= n_dm] - n_dm)
            self._publish_obs(env_ids)
        else:
            self._core.post_step(_hip.POST_OBS | _hip.POST_HF, env_ids)
            self._publish_obs(env_ids)
        return self._obs_buf if env_ids is None else self._obs_buf[env_ids]

    # ------------------------------------------------------------------ motion recording (ig_parkour_env.py:850-995,1594-1620)
    def set_write_agent_states_flag(self, val):
        self._write_agent_states_flag = val

    def is_writing_agent_states(self):
        return self._write_agent_states_flag

    def is_writing_env_state(self, env_id):
        return self._writing_env_state[env_id]

    def set_writing_env_state(self, env_id, val):
        self._writing_env_state[env_id] = val
        self._writing_dirty = True                # the device mask is refreshed once, at the next write_agent_states

    def set_env_success_state(self, env_id, val):
        self._env_success_state[env_id] = val

    def get_env_success_states(self):
        return self._env_success_state

    # ---- motion recording (ig_parkour_env.py:895-995,1594-1620 of the reference).  The reference appends every env's state to
    # Python lists on every step (a loop over all envs with device reads); here the states are scattered into device buffers
    # [steps, env, ...] with a per-env write row, and the host only touches the envs that finish (one small transfer per step).
    def build_agent_states_dict(self, name_suffix="", record_obs=False):
        obs_shapes = self._compute_obs(ret_obs_shapes=True) if record_obs else None
        self._dm_agent_motion = []
        for _ in range(self._num_envs):
            d = {"fps": int(self._control_freq), "loop_mode": "CLAMP"}
            if record_obs:
                d["obs_shapes"] = OrderedDict((k, {"use_normalizer": v["use_normalizer"], "shape": tuple(v["shape"])})
                                              for k, v in obs_shapes.items())
            self._dm_agent_motion.append(d)
        self._record_obs = record_obs
        self.set_write_agent_states_flag(True)
        self._writing_env_state = [True] * self._num_envs
        self._env_success_state = [False] * self._num_envs
        self._save_motion_name_suffix = name_suffix
        os.makedirs(self._output_motion_dir, exist_ok=True)
        N, dev = self._num_envs, self._device
        # rows: the longest clip at the control rate + the frame written at reset + slack; row `cap` is a dump row for envs that
        # are not recording (so the scatter needs no compaction)
        # (a dataset row's recording ends with its clip.  A generator row records until it FAILS, across time-outs and replans, like the
        # reference's lists (ig_parkour_env.py:957-995): its buffers start at one episode and double whenever the number of steps
        # written since this call - a host counter, no read-back - reaches the capacity)
        cap = 8 + (int(math.ceil(float(self._dm_env._motion_lib._motion_lengths.max().item()) * self._control_freq)) if self.has_dm_envs()
                   else int(math.ceil(float(self._episode_length) * self._control_freq)))
        if getattr(self, "_rec_cap", -1) != cap or self._rec_has_obs != record_obs:
            self._rec_frames = torch.empty((cap + 1, N, 34), dtype=torch.float32, device=dev)
            self._rec_contacts = torch.empty((cap + 1, N, self._char_contact_forces.shape[1]), dtype=torch.float32, device=dev)
            self._rec_obs = torch.empty((cap + 1, N, self._obs_buf.shape[1]), dtype=torch.float32, device=dev) if record_obs else None
            self._rec_cap, self._rec_has_obs = cap, record_obs
        self._rec_steps = 0
        self._rec_len = torch.zeros(N, dtype=torch.long, device=dev)
        self._rec_arange = torch.arange(N, device=dev)
        self._writing_dev = torch.ones(N, dtype=torch.bool, device=dev)
        self._writing_dirty = False

    def _get_char_state_all(self):
        """[N, 34] frames (root pos | root exp map | dofs) and [N, 15] binary contacts of every env in one go."""
        q = self._char_root_rot
        q = torch.where(q[:, 3:4] < 0, -q, q)
        l = torch.linalg.vector_norm(q[:, 0:3], dim=-1, keepdim=True)
        ang = 2.0 * torch.atan2(l, q[:, 3:4])
        axis = torch.where(l > 1e-5, q[:, 0:3] / l.clamp_min(1e-12), torch.tensor([0.0, 0.0, 1.0], device=q.device).expand_as(q[:, 0:3]))
        em = torch.where(l > 1e-5, ang, torch.zeros_like(ang)) * axis
        frames = torch.cat([self._char_root_pos, em, self._char_dof_pos], dim=-1)
        contacts = (torch.linalg.vector_norm(self._char_contact_forces, dim=-1) > 1e-5).to(torch.float32)
        return frames, contacts

    def write_agent_states(self):
        if not self.is_writing_agent_states():
            return
        if self._writing_dirty:
            self._writing_dev = torch.tensor(self._writing_env_state, dtype=torch.bool, device=self._device)
            self._writing_dirty = False
        if self.has_mgdm_envs() and self._rec_steps >= self._rec_cap:
            self._grow_recorder()
        self._rec_steps += 1
        frames, contacts = self._get_char_state_all()
        w = self._writing_dev
        row = torch.where(w & (self._rec_len < self._rec_cap), self._rec_len, torch.full_like(self._rec_len, self._rec_cap))
        ar = self._rec_arange
        self._rec_frames[row, ar] = frames
        self._rec_contacts[row, ar] = contacts
        if self._record_obs:
            self._rec_obs[row, ar] = self._obs_buf
        self._rec_len += w
        fin = w & (self._done_buf == base_env.DoneFlags.FAIL.value)
        self._writing_dev = w & ~fin
        n_fin, n_writing = torch.stack([fin.sum(), self._writing_dev.sum()]).tolist()      # the step's one host read
        if n_fin > 0:
            dm, n_dm = self._dm_env, self._num_dm_envs
            ids = fin.nonzero().flatten()
            dm_ids = ids[ids < n_dm]
            if len(dm_ids) > 0:
                mlen = dm._motion_lib._motion_lengths[dm._motion_ids[dm_ids]].tolist()
                mtime = dm._get_motion_times()[dm_ids].tolist()
            for k, e in enumerate(ids.tolist()):
                self._writing_env_state[e] = False
                if e >= n_dm:                  # a generator row: whatever it recorded, under the default name (ig_parkour_env.py:993-994)
                    self.save_agent_states_to_file(e)
                    continue
                name = dm.get_env_motion_name(e)
                if not self._bypass_record_fail and mtime[k] < mlen[k] - self._timestep * 2.0:
                    print("env", e, "failed to track motion", name)
                    continue
                self._env_success_state[e] = True
                self.save_agent_states_to_file(e, name + self._save_motion_name_suffix)
        self.set_write_agent_states_flag(n_writing > 0)

    def _grow_recorder(self):
        """double the recorder's rows (generator rows record until they fail); the dump row moves to the new end"""
        old, new = self._rec_cap, 2 * self._rec_cap

        def grow(buf):
            if buf is None:
                return None
            out = torch.empty((new + 1,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
            out[:old] = buf[:old]
            return out
        self._rec_frames, self._rec_contacts, self._rec_obs = grow(self._rec_frames), grow(self._rec_contacts), grow(self._rec_obs)
        self._rec_cap = new

    def save_agent_states_to_file(self, env_id, output_motion_name=None):
        rec = self._dm_agent_motion[env_id]
        T = min(int(self._rec_len[env_id].item()), self._rec_cap)
        frames = self._rec_frames[:T, env_id].cpu().numpy().astype(np.float32)
        is_dm = env_id < self._num_dm_envs
        if is_dm:                              # dataset rows: env -> global xy; generator rows stay as they are (ig_parkour_env.py:905-916)
            frames[:, 0:2] += self._env_offsets[env_id, 0:2].cpu().numpy()
        out = dict(rec)
        out["contacts"] = self._rec_contacts[:T, env_id].cpu().numpy().astype(np.float32)
        if self._record_obs:
            out["obs"] = self._rec_obs[:T, env_id].cpu().numpy().astype(np.float32)
        ter = self._dm_env._terrain if is_dm else self._mgdm_env._terrain
        pad = round(1.0 // ter.dxdy[0].item()) * ter.dxdy[0].item()
        sliced, frames = terrain_util.slice_terrain_around_motion(frames, ter, padding=pad)
        out["terrain"] = sliced.numpy_copy()
        out["frames"] = frames
        if output_motion_name is None:
            output_motion_name = "dm_motion_" + str(env_id).zfill(3)
        path = os.path.join(self._output_motion_dir, output_motion_name + ".pkl")
        terrain_util.dump_reference_pickle(out, path)
        print("wrote motion data to", path, "num frames =", frames.shape[0])
