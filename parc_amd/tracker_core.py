"""Device buffers + launch glue of the tracker's post-physics pass.

Owns the tensors the reference's env classes own (IGEnv state tensors envs/ig_env.py:764-780,
ref_* buffers ig_parkour_env.py:694-704, data buffers :771-811) and hands their raw pointers to the
HIP kernels (include/parc_hip.h).  The env class (parc_amd/envs/ig_parkour/ig_parkour_env.py) exposes views of
these tensors under the reference's attribute names.
"""
import numpy as np

import torch

from . import _hip


class TrackerConfig:
    """Scalars the kernels need, parsed from the env YAML (PARC/tracker_config/dm_env_default.yaml)."""

    def __init__(self, env_config, kin_char_model, num_ray_points):
        km = kin_char_model
        B, D = km.get_num_joints(), km.get_dof_size()
        J = B - 1
        # Variants of IGParkourEnv._compute_obs / _update_reward (ig_parkour_env.py:1054-1244,1275-1404) beyond the tracker's defaults:
        #   has_target_xy_obs, global_root_height_obs, enable_tar_obs = False, use_contact_info = False  -> column layout (obs_layout)
        #   track_root_h = False, use_contact_info = False, rel_task_w > 0                             -> reward (kernel flags / one multiply)
        #   global_obs, track_root = False                                                               -> kernel flags
        self.has_target_xy_obs = bool(env_config.get("has_target_xy_obs", False))
        self.global_root_height_obs = bool(env_config.get("global_root_height_obs", False))
        self.enable_tar_obs = bool(env_config.get("enable_tar_obs", True))
        self.use_contact_info = bool(env_config.get("use_contact_info", True))
        self.rel_task_w = float(env_config.get("rel_task_w", 0.0))

        self.timestep = 1.0 / env_config["control_freq"]
        steps = list(env_config.get("tar_obs_steps", [1]))
        keys = [km.get_body_id(n) for n in env_config.get("key_bodies", [])]
        assert len(steps) <= _hip.MAX_TAR_STEPS and len(keys) <= _hip.MAX_KEY_BODIES
        s = _hip.TrackCfgS()
        s.num_tar_steps = len(steps)
        ts = torch.tensor(steps, dtype=torch.int) * self.timestep   # mgdm_dm_util.py:289: timestep * tar_obs_steps (fp32)
        for i, v in enumerate(ts.to(torch.float32).tolist()):
            s.tar_dt[i] = v
        s.num_key_bodies = len(keys)
        for i, k in enumerate(keys):
            s.key_body_ids[i] = k
        jw = env_config.get("joint_err_w", None)
        jw = [1.0] * J if jw is None else list(jw)
        assert len(jw) == J
        dw = np.zeros(D, dtype=np.float32)
        for j in range(1, B):       # ig_parkour_env.py:1573-1592
            dd = km.get_joint_dof_dim(j)
            if dd > 0:
                di = km.get_joint_dof_idx(j)
                dw[di:di + dd] = jw[j - 1]
        for i in range(J):
            s.joint_err_w[i] = jw[i]
        for i in range(D):
            s.dof_err_w[i] = float(dw[i])
        cw = env_config["contact_weights"]
        for i in range(B):
            s.contact_w[i] = float(cw[i])
        w = np.array([env_config["pose_w"], env_config["vel_w"], env_config["root_pos_w"], env_config["root_vel_w"],
                      env_config["key_pos_w"]], dtype=np.float64)
        w = w / w.sum()             # ig_parkour_env.py:122-132
        for i in range(5):
            s.reward_w[i] = float(w[i])
        # rel_task_w > 0: reward = deepmimic_r * task_r, without rel_deepmimic_w (ig_parkour_env.py:1399-1404); the kernel then writes
        # deepmimic_r and the env multiplies by the task term the same launch produced
        s.rel_deepmimic_w = float(env_config["rel_deepmimic_w"]) if self.rel_task_w <= 0 else 1.0
        s.track_root_h = int(bool(env_config.get("track_root_h", True)))
        s.use_contact_info = int(self.use_contact_info)
        s.global_obs = int(bool(env_config.get("global_obs", False)))
        ptd = env_config["pose_termination_dist"]
        for i in range(J):
            s.pose_termination_dist[i] = float(ptd[i])
        s.pose_termination = int(bool(env_config.get("pose_termination", False)))
        s.enable_early_termination = int(bool(env_config["enable_early_termination"]))
        s.track_root = int(bool(env_config["track_root"]))
        s.root_pos_termination_dist = float(env_config["root_pos_termination_dist"])
        s.root_rot_termination_angle = float(env_config["root_rot_termination_angle"])
        s.termination_height = float(env_config["termination_height"])
        cb = [km.get_body_id(n) for n in env_config.get("contact_bodies", [])]
        s.num_contact_bodies = len(cb)
        for b in cb:
            s.contact_body_mask[b] = 1
        s.episode_length = float(env_config["episode_length"])
        s.contact_eps = 1e-5        # IGParkourEnv._get_char_contact_state default eps (ig_parkour_env.py:841)
        s.min_obs_h = float(env_config["min_obs_h"])
        s.max_obs_h = float(env_config["max_obs_h"])
        s.num_ray_points = int(num_ray_points)
        s.task1_w = float(env_config.get("task1_w", 0.7))
        s.task2_w = float(env_config.get("task2_w", 0.3))
        s.target_radius = float(env_config.get("target_radius", 1.0))
        s.target_future_min = float(env_config["dm"]["target_xy_future_time_min"])
        s.target_future_max = float(env_config["dm"]["target_xy_future_time_max"])
        K, S = len(keys), len(steps)
        self.char_obs_dim = 12 + 6 * J + D + 3 * K
        self.tar_obs_dim = 9 + 6 * J + 3 * K
        self.obs_dim = self.char_obs_dim + S * self.tar_obs_dim + S * B + B + num_ray_points
        s.obs_dim = self.obs_dim
        self.struct = s
        self.key_body_ids = keys
        self.contact_body_ids = cb
        self.tar_obs_steps = steps
        self.num_bodies, self.dof_size = B, D
        self.num_ray_points = int(num_ray_points)

    def obs_layout(self, replan_timer=False):
        """(shapes, col_map): the segment table of IGParkourEnv._compute_obs(ret_obs_shapes=True) (ig_parkour_env.py:1163-1239) for this
        configuration, and for each column of the handed-out row its source in the virtual row [ fused row (obs_dim) | aux: root
        height, local target x, y, 0 | plan clock ] that parc_assemble_obs gathers from - None when the handed-out row IS the fused
        row (the tracker's default configuration: no gather, the kernels' buffer is handed out)."""
        from collections import OrderedDict
        B, S, P = self.num_bodies, len(self.tar_obs_steps), self.num_ray_points
        Wc, Wt, D0 = self.char_obs_dim, self.tar_obs_dim, self.obs_dim
        shapes, cols = OrderedDict(), []
        char_cols = list(range(0, Wc))
        if self.global_root_height_obs:                # ig_char_env.py:618-620: root height in front of the character block
            char_cols = [D0 + 0] + char_cols
        shapes["char_obs"] = {"use_normalizer": True, "shape": torch.Size([len(char_cols)])}
        cols += char_cols
        o = Wc
        if self.enable_tar_obs:
            shapes["tar_obs"] = {"use_normalizer": True, "shape": torch.Size([S, Wt])}
            cols += list(range(o, o + S * Wt))
        o += S * Wt
        if self.use_contact_info and self.enable_tar_obs:
            shapes["tar_contacts"] = {"use_normalizer": False, "shape": torch.Size([S, B])}
            cols += list(range(o, o + S * B))
        o += S * B
        if self.use_contact_info:
            shapes["char_contacts"] = {"use_normalizer": False, "shape": torch.Size([B])}
            cols += list(range(o, o + B))
        o += B
        shapes["hf"] = {"use_normalizer": False, "shape": torch.Size([P])}
        cols += list(range(o, o + P))
        assert o + P == D0
        if self.has_target_xy_obs:                     # ig_parkour_env.py:1212-1224
            shapes["target_xy"] = {"use_normalizer": True, "shape": torch.Size([2])}
            cols += [D0 + 1, D0 + 2]
        if replan_timer:                               # ig_parkour_env.py:1226-1232
            shapes["replan_t"] = {"use_normalizer": False, "shape": torch.Size([1])}
            cols += [D0 + 4]
        native = cols == list(range(D0))
        return shapes, (None if native else cols)


class TrackerCore:
    def __init__(self, num_envs, device, kin_char_model, motion_lib, cfg, ray_xy_points):
        self.N = N = num_envs
        self.device = device
        self.km = kin_char_model
        self.mlib = motion_lib
        self.cfg = cfg
        B, D = cfg.num_bodies, cfg.dof_size
        f32 = dict(dtype=torch.float32, device=device)
        z = torch.zeros
        # Isaac Gym state tensor layouts (envs/ig_env.py:764-780)
        self.root_state = z((N, 13), **f32)
        self.root_state[:, 6] = 1.0
        self.dof_state = z((N * D, 2), **f32)
        self.rigid_body_state = z((N * B, 13), **f32)
        self.rigid_body_state[:, 6] = 1.0
        self.contact_forces = z((N * B, 3), **f32)
        self.env_offsets = z((N, 3), **f32)
        self.motion_ids = z((N,), dtype=torch.int64, device=device)
        self.motion_terrain_ids = z((N,), dtype=torch.int64, device=device)
        self.motion_time_offsets = z((N,), **f32)
        self.motion_xy_offset = z((N, 2), **f32)
        self.time_buf = z((N,), **f32)
        self.timestep_buf = z((N,), dtype=torch.int, device=device)
        self.ref_root_pos = z((N, 3), **f32)
        self.ref_root_rot = z((N, 4), **f32)
        self.ref_root_rot[:, 3] = 1.0
        self.ref_root_vel = z((N, 3), **f32)
        self.ref_root_ang_vel = z((N, 3), **f32)
        self.ref_joint_rot = z((N, B - 1, 4), **f32)
        self.ref_joint_rot[..., 3] = 1.0
        self.ref_dof_vel = z((N, D), **f32)
        self.ref_dof_pos = z((N, D), **f32)
        self.ref_contacts = z((N, B), **f32)
        self.ref_body_pos = z((N, B, 3), **f32)
        self.obs = z((N, cfg.obs_dim), **f32)
        # values of the optional observation columns (root height, localised xy target; parc_env_buffers_t.obs_aux), allocated only
        # for configurations whose handed-out row has them
        self.obs_aux = z((N, 4), **f32) if (cfg.has_target_xy_obs or cfg.global_root_height_obs) else None
        # row 0 = total reward, rows 1..9 = the logged terms: one [10, N] block so the return tracker adds it in one op
        self.reward_all = z((10, N), **f32)
        self.reward = self.reward_all[0]
        self.reward_terms = self.reward_all[1:10]
        self.target_xy = z((N, 2), **f32)
        self.next_target_xy_time = z((N,), **f32)
        self.done = z((N,), dtype=torch.int, device=device)
        self.done_kind = z((N,), dtype=torch.int, device=device)
        # device-side reset (PARC_POST_MASKED / PARC_POST_INIT_CHAR)
        self.reset_mask = z((N,), dtype=torch.int, device=device)
        self.init_noise_xy = z((N, 2), **f32)
        # every uniform an env step consumes, drawn by ONE launch per step (IGParkourEnv.step): [N,3] xy-target resample of the step
        # (PARC_POST_TARGETS), [N,3] the same for the envs that restart, [5,N] reset sampling (clip, tile, phase, xy noise)
        self.rand_pool = z((11 * N,), **f32)
        self.target_rand = self.rand_pool[0:3 * N].view(N, 3)
        self.target_rand_reset = self.rand_pool[3 * N:6 * N].view(N, 3)
        self.reset_uniforms = self.rand_pool[6 * N:11 * N].view(5, N)
        self.rand_pool_fresh = False
        self.reset_cdf = None
        self.ray_xy_points = ray_xy_points.to(device=device, dtype=torch.float32).contiguous()
        P = self.ray_xy_points.shape[0]
        assert P == cfg.struct.num_ray_points
        # the heightmap columns of the observation row ARE the _ray_hfs buffer (no copy)
        self.ray_hfs = self.obs[:, cfg.obs_dim - P:]
        self.terrain = None
        self._terrain_struct = None
        self._buf_struct = None
        self._buf_struct_reset = None
        self.timing_events = None      # bench.py: list of _hip.HipEventPair bound to the full post-step launches of a rollout

    def set_terrain(self, terrain):
        self.terrain = terrain
        hf = terrain.hf.to(device=self.device, dtype=torch.float32).contiguous()
        self._hf = hf
        self._terrain_struct = _hip.terrain_struct(hf, terrain.min_point.tolist(), terrain.dxdy.tolist())

    def buffers(self, reset=False, rows=None):
        """parc_env_buffers_t of this core; reset=True: the variant whose target uniforms are the restart slice of the pool;
        rows=(e0, n): the same buffers seen as a core of n envs starting at row e0 (every pointer advanced by e0 rows, reward_terms keeps
        the allocation's row length) - how a sub-env launches on its contiguous row range with plain, coalesced env indexing and with
        the per-env mask of the device-side reset, neither of which an env-id list allows."""
        if rows is not None:
            key = (bool(reset), int(rows[0]), int(rows[1]))
            cache = self.__dict__.setdefault("_buf_struct_rows", {})
            if key not in cache:
                e0, n = key[1], key[2]
                assert 0 <= e0 and e0 + n <= self.N
                B, D = self.cfg.num_bodies, self.cfg.dof_size
                s = _hip.EnvBuffersS.from_buffer_copy(self.buffers(reset))
                rowlen = dict(root_state=13, dof_state=2 * D, rigid_body_state=13 * B, contact_forces=3 * B, env_offsets=3, motion_ids=2,
                              motion_time_offsets=1, motion_xy_offset=2, time_buf=1, target_xy=2, ref_root_pos=3, ref_root_rot=4, ref_root_vel=3,
                              ref_root_ang_vel=3, ref_joint_rot=4 * (B - 1), ref_dof_vel=D, ref_dof_pos=D, ref_contacts=B, ref_body_pos=3 * B,
                              obs=self.cfg.obs_dim, reward=1, reward_terms=1, done=1, done_kind=1, env_mask=1, init_noise_xy=2,
                              next_target_time=1, target_rand=3, obs_aux=4)          # in 4-byte words (motion_ids: int64 = 2 words)
                for name, words in rowlen.items():
                    base = getattr(s, name)
                    if base:
                        setattr(s, name, base + 4 * words * e0)
                s.num_envs = n
                s.reward_terms_stride = self.N
                cache[key] = s
            return cache[key]
        if reset:
            if self._buf_struct_reset is None:
                self._buf_struct_reset = _hip.EnvBuffersS.from_buffer_copy(self.buffers())
                self._buf_struct_reset.target_rand = _hip.ptr(self.target_rand_reset)
            return self._buf_struct_reset
        if self._buf_struct is None:
            p = _hip.ptr
            self._buf_struct = _hip.EnvBuffersS(
                self.N, p(self.root_state), p(self.dof_state), p(self.rigid_body_state), p(self.contact_forces),
                p(self.env_offsets), p(self.motion_ids), p(self.motion_time_offsets), p(self.motion_xy_offset), p(self.time_buf),
                p(self.target_xy),
                p(self.ref_root_pos), p(self.ref_root_rot), p(self.ref_root_vel), p(self.ref_root_ang_vel),
                p(self.ref_joint_rot), p(self.ref_dof_vel), p(self.ref_dof_pos), p(self.ref_contacts), p(self.ref_body_pos),
                p(self.obs), p(self.reward), p(self.reward_terms), p(self.done), p(self.done_kind),
                p(self.reset_mask), p(self.init_noise_xy), p(self.next_target_xy_time), p(self.target_rand), p(self.obs_aux), 0)
        return self._buf_struct

    # ---- K5 (IGParkourEnv._refresh_obs_hfs)
    def refresh_obs_hfs(self):
        c = self.cfg.struct
        P = c.num_ray_points
        out = _hip.c_vp(self.obs.data_ptr() + 4 * (self.cfg.obs_dim - P))
        _hip.check(_hip.lib().parc_refresh_obs_hfs(_hip.stream(), self.N, _hip.ptr(self.ray_xy_points), P, _hip.ptr(self.root_state),
                                                  _hip.ptr(self.env_offsets), self._terrain_struct, c.min_obs_h, c.max_obs_h, out,
                                                  self.cfg.obs_dim), "parc_refresh_obs_hfs")

    # ---- fused K3/K2/K4/K6-K10
    def post_step(self, what, env_ids=None, reset_rand=False, mlib=None, terrain_struct=None, rows=None):
        """mlib / terrain_struct: the clip library and heightfield of a sub-env other than the default one (the motion-generator
        sub-env launches on its own rows with its own); rows=(e0, n): launch on that contiguous row range (see buffers)"""
        assert rows is None or env_ids is None
        mlib = self.mlib if mlib is None else mlib
        terrain_struct = self._terrain_struct if terrain_struct is None else terrain_struct
        if env_ids is not None:
            env_ids = env_ids.to(torch.int64).contiguous()
            n = int(env_ids.shape[0])
            if n == 0:
                return
            ids = _hip.ptr(env_ids)
        else:
            n, ids = 0, _hip.c_vp(0)
        timed = self.timing_events is not None and env_ids is None and rows is None and (what & _hip.POST_REWARD_DONE)
        if timed:
            # bench.py: the launch with a pair of events bound to its dispatch (parc_track_post_step_timed) - the kernel's own duration
            pair = _hip.HipEventPair()
            _hip.check(_hip.lib().parc_track_post_step_timed(_hip.stream(), self.km.c_struct(), mlib.c_struct(), terrain_struct, self.cfg.struct,
                                                             self.buffers(reset_rand, rows), ids, n, what, _hip.ptr(self.ray_xy_points),
                                                             pair.start, pair.stop), "parc_track_post_step_timed")
            self.timing_events.append(pair)
            return
        _hip.check(_hip.lib().parc_track_post_step(_hip.stream(), self.km.c_struct(), mlib.c_struct(), terrain_struct,
                                                   self.cfg.struct, self.buffers(reset_rand, rows), ids, n, what, _hip.ptr(self.ray_xy_points)),
                   "parc_track_post_step")

    def assemble_obs(self, col_map, out, scalar=None, env_ids=None):
        """rows of a non-default observation layout (TrackerConfig.obs_layout) gathered from the fused rows, obs_aux and one scalar"""
        if env_ids is not None:
            env_ids = env_ids.to(torch.int64).contiguous()
            if env_ids.numel() == 0:
                return
        _hip.check(_hip.lib().parc_assemble_obs(_hip.stream(), self.N, _hip.ptr(self.obs), self.cfg.obs_dim, _hip.ptr(self.obs_aux),
                                                _hip.ptr(scalar), _hip.ptr(col_map), _hip.ptr(out), int(out.shape[1]),
                                                _hip.ptr(env_ids), 0 if env_ids is None else int(env_ids.numel())), "parc_assemble_obs")

    def step_tail(self, fail_rates, ema_w, publish_ref_state=True, rows=None):
        """fail-rate EMA of the step + (optionally) the per-step publication of the reference state, co-scheduled in one launch;
        rows=(0, n): the dataset rows of a split env (they start at row 0, so done_kind needs no offset)"""
        assert rows is None or rows[0] == 0
        _hip.check(_hip.lib().parc_step_tail(_hip.stream(), self.km.c_struct(), self.mlib.c_struct(), self.buffers(False, rows),
                                             _hip.POST_REF if publish_ref_state else 0, self.mlib.num_motions(), _hip.ptr(self.done_kind),
                                             float(ema_w), _hip.ptr(fail_rates)), "parc_step_tail")

    def update_fail_rates(self, fail_rates, ema_w):
        _hip.check(_hip.lib().parc_update_fail_rates(_hip.stream(), self.N, self.mlib.num_motions(), _hip.ptr(self.motion_ids),
                                                     _hip.ptr(self.done_kind), float(ema_w), _hip.ptr(fail_rates)),
                   "parc_update_fail_rates")
