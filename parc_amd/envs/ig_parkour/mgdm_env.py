"""Motion-generator sub-environment: closed-loop replanning around the tracker.

Mirror of the reference's ``envs/ig_parkour/mgdm_env.py`` (MotionGenDeepMimicEnv :37-861, ReplanFlags :28-30): every
``plan_length`` seconds ALL envs of the sub-env ask a motion generator for a new plan (a clip per env, continuing from the character's
current pose towards a random xy target), the plans become the clip library the tracker follows, and envs that failed / left the
terrain / used up ``max_replans`` are re-spawned at a random place instead ("hard reset").

How it sits on the MI355X path:
* the sub-env owns a contiguous range of the env's rows in the tracker core (parc_amd/tracker_core.py); its clip library is a
  ``MotionLib`` of one fixed-length clip per env that lives on the device and is rebuilt in place at every replan
  (``MotionLib.update_frames`` -> parc_motion_lib_build; the kernels' pointers never change);
* the per-step work is the SAME fused launch as for dataset clips (parc_track_post_step on the sub-env's rows, with this library,
  this terrain and PARC_POST_PLAN_CLOCK): clip id = row, clip time = the global plan clock (handed over in ``motion_time_offsets``),
  generated frames are env-local (``motion_xy_offset = env_offsets``);
* what the reference adds around it - out-of-bounds / too-high / replan-time termination, target picking, the one-frame state
  histories, spawn sampling - is fixed-shape torch work on the sub-env's rows; the plan clock is mirrored on the host in fp32, so the
  replan decision costs no read-back.

The generator is a callable ``generator(target_xy, prev_frames, terrain, char_model, settings) -> MotionFrames`` with the attributes
the reference reads from its model object (``_num_prev_states``, ``_sequence_fps``, ``_dx/_dy/_num_x_neg/...``).  It is the ONLY entry:
this package imports nothing of the reference's ``diffusion`` package and opens no model file; the adapter that wraps stage 1's trained
``diffusion.mdm.MDM`` is user-side code, and ``mgdm.model_path`` without a generator is an error.
Checked against fixture G21 (the reference's class driven on CPU with a recorded stand-in generator).
"""
import enum
import os
import random
import time

import numpy as np
import torch

from ... import _hip
from ...anim import motion_lib
from ...util import geom_util, motion_util, terrain_util, torch_util
from ...util.motion_util import MotionFrames
from .. import base_env

SIM_CHAR_IDX = 0
REF_CHAR_IDX = 1


class ReplanFlags(enum.Enum):
    REPLAN = 0
    HARD_RESET = 1


class MDMGenSettings:
    """The per-call options the env sets for the generator (field names of diffusion/gen_util.py MDMGenSettings :12-33)."""
    use_prev_state = True
    use_cfg = True
    cfg_scale = 0.65
    prev_state_ind_key = True
    target_condition_key = True
    feature_vector_key = True
    use_ddim = True
    ddim_stride = 10


class MotionGenDeepMimicEnv:
    def __init__(self, config, num_envs, device, visualize, char_model, generator=None, rand_fn=None):
        env_config = config["env"]
        mg = env_config["mgdm"]
        self._num_envs = num_envs
        self._device = device
        self._visualize = visualize
        self._kin_char_model = char_model
        self._timestep = 1.0 / env_config["control_freq"]
        self._rand_root_pos_offset_scale = env_config["rand_root_pos_offset_scale"]
        self._max_obs_h, self._min_obs_h = env_config["max_obs_h"], env_config["min_obs_h"]
        self._demo_mode = env_config["demo_mode"]
        self._plan_length = mg["plan_length"]
        self._ddim_stride = mg["ddim_stride"]
        self._max_replans = mg["max_replans"]
        self._replan_flag = True                 # reset() is the first call: it must plan
        self._cfg_scale = mg["cfg_scale"]
        self._target_dist_max, self._target_dist_min = mg["target_dist_max"], mg["target_dist_min"]
        self._target_dur_max, self._target_dur_min = mg["target_dur_max"], mg["target_dur_min"]
        self._target_radius = env_config["target_radius"]
        self._target_heading_scale = mg["target_heading_scale"]
        self._dont_auto_update_targets = mg.get("dont_auto_update_targets", False)
        self._true_tensor = torch.ones(num_envs, dtype=torch.bool, device=device)
        # every uniform draw goes through this hook, in the reference's order and shapes (tests replay recorded draws)
        self._rand = rand_fn if rand_fn is not None else (lambda n: torch.rand(n, dtype=torch.float32, device=device))

        gen = generator if generator is not None else mg.get("generator")
        if gen is None:
            # reference :32-35 load_mdm unpickles a model object of its diffusion package here; this package does neither
            raise RuntimeError("rows under the motion-generator sub-env (fraction_dm_envs < 1) need a planner: pass a callable "
                               "generator(target_xy, prev_frames, terrain, char_model, settings) -> MotionFrames as mgdm.generator"
                               + (" -- mgdm.model_path ({!r}) is not opened by this package; wrap the reference's trained model with the "
                                  "user-side adapter of INTEGRATION.md section 3 and hand it over as mgdm.generator".format(mg["model_path"])
                                  if mg.get("model_path") else ""))
        self._mgen = gen
        self._num_prev_states = gen._num_prev_states
        assert self._num_prev_states > 1, "the env keeps the one-frame state histories the generator continues from"
        self._build_obs_hfs(env_config, num_envs, device)
        self._motion_ids = torch.arange(num_envs, device=device, dtype=torch.int64)
        self._motion_lib = None
        # the plan clock: one device element (what observers read) + its fp32 image on the host (what decides)
        self._mgdm_time_buf = torch.zeros(1, device=device, dtype=torch.float32)
        self._replan_time_buf = self._mgdm_time_buf
        self._plan_time_host = np.float32(0.0)
        self._replan_buf = torch.full((num_envs,), ReplanFlags.HARD_RESET.value, device=device, dtype=torch.int64)
        self._replan_counter = torch.zeros(num_envs, device=device, dtype=torch.int64)
        self._core = None
        self._terrain = None

    # ------------------------------------------------------------------ wiring into the tracker core
    def attach(self, core, first_env):
        """Views of rows [first_env, first_env + num_envs) of the core's buffers under the reference's attribute names
        (RefCharEnv.get_sim_tensor_views / get_data_buffer_views, mgdm_dm_util.py:45-98)."""
        n, e0 = self._num_envs, first_env
        e1 = e0 + n
        B, D = self._kin_char_model.get_num_joints(), self._kin_char_model.get_dof_size()
        self._core, self._first_env = core, e0
        self._rows = (e0, n)
        self._abs_ids = torch.arange(e0, e1, device=self._device, dtype=torch.int64)
        rs = core.root_state[e0:e1]
        self._char_root_pos, self._char_root_rot = rs[:, 0:3], rs[:, 3:7]
        self._char_root_vel, self._char_root_ang_vel = rs[:, 7:10], rs[:, 10:13]
        ds = core.dof_state.view(core.N, D, 2)[e0:e1]
        self._char_dof_pos, self._char_dof_vel = ds[..., 0], ds[..., 1]
        rb = core.rigid_body_state.view(core.N, B, 13)[e0:e1]
        self._char_rigid_body_pos, self._char_rigid_body_vel, self._char_rigid_body_ang_vel = rb[..., 0:3], rb[..., 7:10], rb[..., 10:13]
        self._char_contact_forces = core.contact_forces.view(core.N, B, 3)[e0:e1]
        for name in ("ref_root_pos", "ref_root_rot", "ref_root_vel", "ref_root_ang_vel", "ref_body_pos", "ref_joint_rot", "ref_dof_pos",
                     "ref_dof_vel", "ref_contacts"):
            setattr(self, "_" + name, getattr(core, name)[e0:e1])
        self._reward_buf, self._done_buf = core.reward[e0:e1], core.done[e0:e1]
        self._time_buf, self._timestep_buf = core.time_buf[e0:e1], core.timestep_buf[e0:e1]
        self._target_xy, self._next_target_xy_time = core.target_xy[e0:e1], core.next_target_xy_time[e0:e1]
        self._env_offsets = core.env_offsets[e0:e1]
        self._ray_hfs = core.ray_hfs[e0:e1]
        self._ray_xy_points = core.ray_xy_points
        self._key_body_ids = torch.tensor(core.cfg.key_body_ids, dtype=torch.long, device=self._device)
        self._need_refresh = torch.zeros(n, dtype=torch.bool, device=self._device)        # the reference's _actors_need_reset column
        self._agent_state_hist = self._get_char_motion_frames(ref=False, null=True)
        self._ref_state_hist = self._get_char_motion_frames(ref=True, null=True)
        # clip id = row of the sub-env; generated frames are env-local (no tile offset)
        core.motion_ids[e0:e1] = self._motion_ids
        self._motion_time_offsets = core.motion_time_offsets[e0:e1]

    def sync_core_rows(self):
        """the two per-env inputs of the fused launch that are not constants: clip time = plan clock, xy offset = env offset"""
        self._motion_time_offsets.copy_(self._mgdm_time_buf.expand(self._num_envs))
        self._core.motion_xy_offset[self._first_env:self._first_env + self._num_envs] = self._env_offsets[:, 0:2]

    def terrain_struct(self):
        if getattr(self, "_terrain_struct", None) is None:
            t = self._terrain
            self._hf_dev = t.hf.to(device=self._device, dtype=torch.float32).contiguous()
            self._terrain_struct = _hip.terrain_struct(self._hf_dev, t.min_point.tolist(), t.dxdy.tolist())
        return self._terrain_struct

    def _post(self, what, env_ids=None):
        """the fused launch on this sub-env's rows: all of them as a contiguous row range (plain env indexing, may be masked by the
        core's reset mask), or the listed ones"""
        self.sync_core_rows()
        if env_ids is None:
            self._core.post_step(what | _hip.POST_PLAN_CLOCK, rows=self._rows, mlib=self._motion_lib, terrain_struct=self.terrain_struct())
        else:
            self._core.post_step(what | _hip.POST_PLAN_CLOCK, self._abs_ids[env_ids], mlib=self._motion_lib, terrain_struct=self.terrain_struct())

    # ------------------------------------------------------------------ env protocol (reference :89-128)
    def reset(self, env_ids):
        if self._replan_flag:
            if self._demo_mode:
                self._replan_buf[:] = ReplanFlags.REPLAN.value
            self.replan()
        elif len(env_ids) > 0:
            self._timestep_buf[env_ids] = 0
            self._time_buf[env_ids] = 0
            self._done_buf[env_ids] = base_env.DoneFlags.NULL.value
            self._reset_char(env_ids)

    def pre_physics_step(self):
        # the character's state before the simulator overwrites it: the generator continues from it (stored in place: the history
        # tensors keep their addresses, a captured rollout step writes the same ones on every replay)
        self._agent_state_hist.store(self._get_char_motion_frames(ref=False))

    def update_time(self, timestep):
        self._mgdm_time_buf[0] += timestep
        self._plan_time_host = np.float32(self._plan_time_host + np.float32(timestep))

    def host_step_replayed(self, timestep):
        """What a step changes on the HOST, for a step that ran as a replayed hipGraph (the device part is in the graph): the plan
        clock's fp32 mirror advances, and once it passes the plan length the next reset has to replan (update_done_extra)."""
        self._plan_time_host = np.float32(self._plan_time_host + np.float32(timestep))
        if self._plan_time_host > np.float32(self._plan_length):
            self._replan_flag = True

    def replan_pending_after_next_step(self, timestep):
        """True if the reset that follows the NEXT step has to call the generator (host decision: such a step is followed by the
        eager reset path, every other step restarts its finished rows on the device)"""
        return self._replan_flag or np.float32(self._plan_time_host + np.float32(timestep)) > np.float32(self._plan_length)

    def update_misc(self, fixed_shape=False):
        """fixed_shape: targets are drawn for every row and taken where the timer ran out (no index list, no read-back: capturable;
        same distribution, different consumption of the random stream than the reference's `nonzero` + per-id draws, which the
        id-list path keeps for the fixture replay)"""
        self._ref_state_hist.store(self._get_char_motion_frames(ref=True))
        if fixed_shape:
            self.pick_new_xy_targets_masked(self._time_buf > self._next_target_xy_time)
            return
        due = (self._time_buf > self._next_target_xy_time).nonzero()
        if len(due) > 0:
            self.pick_new_xy_targets(due.squeeze(-1))

    def _update_ref_motion(self):
        self._post(_hip.POST_REF)

    def compute_tar_obs(self, tar_obs_steps, env_ids=None):
        """-> (root_pos [n, S, 3], root_rot [n, S, 4], joint_rot [n, S, J, 4], key_pos [n, S, K, 3], contacts [n, S, B]) of the plan at
        plan time + steps * dt (reference :130-157).  The env's own step does this inside the fused launch; this is the API."""
        ids = self._motion_ids if env_ids is None else self._motion_ids[env_ids]
        n, S = int(ids.shape[0]), int(tar_obs_steps.shape[0])
        times = self._mgdm_time_buf.expand(n).unsqueeze(-1) + self._timestep * tar_obs_steps.to(self._device)
        rp, rr, _, _, jr, _, con = self._motion_lib.calc_motion_frame(ids.unsqueeze(-1).expand(n, S).reshape(-1), times.reshape(-1))
        bp, _ = self._kin_char_model.forward_kinematics(rp, rr, jr)
        key = bp[:, self._key_body_ids, :].reshape(n, S, -1, 3) if len(self._key_body_ids) > 0 else torch.zeros([0], device=self._device)
        return rp.reshape(n, S, 3), rr.reshape(n, S, 4), jr.reshape(n, S, -1, 4), key, con.reshape(n, S, -1)

    # ------------------------------------------------------------------ termination (reference :159-204)
    def update_done(self, **unused):
        """RefCharEnv.update_done (the fused launch, with the tracker config the core was built with) + the sub-env's own rules"""
        self._post(_hip.POST_REWARD_DONE)
        self.update_done_extra()

    def update_done_extra(self, fixed_shape=False):
        t = self._terrain
        NULL, FAIL, TIME = (base_env.DoneFlags.NULL.value, base_env.DoneFlags.FAIL.value, base_env.DoneFlags.TIME.value)
        HARD = ReplanFlags.HARD_RESET.value
        gxy = self._char_root_pos[..., 0:2] + self._env_offsets[..., 0:2]
        lo = t.min_point + self._oob_region
        hi = t.min_point + t.dims * t.dxdy - self._oob_region
        oob = (gxy[..., 0] < lo[0]) | (gxy[..., 0] > hi[0]) | (gxy[..., 1] < lo[1]) | (gxy[..., 1] > hi[1])
        done, rb = self._done_buf, self._replan_buf
        done[:] = torch.where(oob, torch.full_like(done, TIME), done)                         # walked off the terrain: not a failure
        rb[:] = torch.where(oob, torch.full_like(rb, HARD), rb)
        done[:] = torch.where(self._char_root_pos[..., 2] > 3.0, torch.full_like(done, FAIL), done)      # launched into the air
        rb[:] = torch.where(done == FAIL, torch.full_like(rb, HARD), rb)                      # a failed env re-spawns at the next replan
        if fixed_shape:
            # the same rule with the decision on the device (the plan clock is a device element): nothing here depends on the host's
            # branch, so a captured step stays valid across the replan boundary; the host's own consequence - the next reset replans -
            # is drawn from the fp32 mirror of the clock (here, or in host_step_replayed for a replayed step)
            due = self._mgdm_time_buf > self._plan_length                                     # [1], broadcasts over the rows
            rb[:] = torch.where(due & (self._replan_counter >= self._max_replans), torch.full_like(rb, HARD), rb)
            done[:] = torch.where(due & (done == NULL) & (rb == HARD), torch.full_like(done, TIME), done)
            if self._plan_time_host > np.float32(self._plan_length):
                self._replan_flag = True
        elif self._plan_time_host > np.float32(self._plan_length):
            self._replan_flag = True
            hard = self._compute_hard_reset_envs_mask()
            done[:] = torch.where((done == NULL) & hard, torch.full_like(done, TIME), done)   # re-spawning envs end their episode

    def _compute_hard_reset_envs_mask(self):
        rb = self._replan_buf
        rb[:] = torch.where(self._replan_counter >= self._max_replans, torch.full_like(rb, ReplanFlags.HARD_RESET.value), rb)
        return rb == ReplanFlags.HARD_RESET.value

    # ------------------------------------------------------------------ terrain (reference :206-366)
    def build_terrain(self, env_config, terrain_save_path):
        """A square of alternating platforms (checkerboard of `num_segments`^2 blocks, heights drawn from `platform_heights`), or the
        terrain of `mgdm.terrain_file`; spawn bounds; voxelised mesh; cache file in the reference's format."""
        start = time.perf_counter()
        hm = env_config["mgdm"]["heightmap"]
        dx, safety = hm["horizontal_scale"], hm["safety_region"]
        if "terrain_file" in env_config["mgdm"]:
            from ...util import safe_pickle
            path = env_config["mgdm"]["terrain_file"]
            if os.path.splitext(path)[1] == ".yaml":
                import yaml
                self._hard_motion_lib = motion_lib.MotionLib(path, self._kin_char_model, self._device)
                with open(path, "r") as f:
                    path = yaml.safe_load(f)["terrain"]
            t = safe_pickle.load_motion_file_safe(path)["terrain"]
            ter = terrain_util.SubTerrain.from_arrays(t["hf"], t["min_point"], t["dxdy"], t.get("hf_mask"), t.get("hf_maxmin"), device=self._device)
            min_x, min_y = ter.min_point[0].item(), ter.min_point[1].item()
            x_length, y_length = ter.dims[0].item() * ter.dxdy[0].item(), ter.dims[1].item() * ter.dxdy[1].item()
        else:
            x_length = np.round(np.sqrt(2048)) * hm["sq_m_per_env"] + safety * 2.0       # the reference sizes it for 2048 envs whatever N is
            y_length = x_length
            gx, gy = int(x_length / dx), int(y_length / dx)
            min_x, min_y = -x_length / 2.0, -y_length / 2.0
            ter = terrain_util.SubTerrain(x_dim=gx, y_dim=gy, dx=dx, dy=dx, min_x=min_x, min_y=min_y, device=self._device)
            S = hm["num_segments"]
            heights = hm["platform_heights"]

            def cuts(dim):
                c = [i * (dim // S) for i in range(S + 1)]
                c[S] += dim % S
                return c
            cx, cy = cuts(ter.dims[0].item()), cuts(ter.dims[1].item())
            for i in range(S):
                for j in range(S):
                    raised = (i % 2 == 0) == (j % 2 == 0)
                    val = heights[random.randint(0, len(heights) - 1)] if raised else 0.0
                    ter.hf[cx[i]:cx[i + 1], cy[j]:cy[j + 1]] = val
        nt = ter.numpy_copy()
        verts, tris = terrain_util.convert_heightfield_to_voxelized_trimesh(nt.hf, min_x=nt.min_point[0], min_y=nt.min_point[1], dx=dx)
        self._oob_region = safety / 10
        self._spawn_min_x, self._spawn_min_y = min_x + safety, min_y + safety
        self._spawn_max_x, self._spawn_max_y = min_x + x_length - safety, min_y + y_length - safety
        self._terrain = ter
        self._terrain_struct = None
        if terrain_save_path:
            os.makedirs(os.path.dirname(terrain_save_path) or ".", exist_ok=True)
            cpu_t = ter.torch_copy()
            cpu_t.set_device("cpu")
            terrain_util.dump_reference_pickle({"oob_region": self._oob_region, "spawn_min_x": self._spawn_min_x, "spawn_max_x": self._spawn_max_x,
                                                "spawn_min_y": self._spawn_min_y, "spawn_max_y": self._spawn_max_y, "terrain": cpu_t,
                                                "vertices": verts, "triangles": tris}, terrain_save_path)
        print("building mgdm heightfield and mesh time:", time.perf_counter() - start, " seconds.")
        return verts, tris, ter.min_point

    def load_terrain(self, terrain_save_path):
        from ...util import safe_pickle
        d = safe_pickle.load_motion_file_safe(terrain_save_path)
        t = d["terrain"]
        if not isinstance(t, dict):
            raise RuntimeError("{} is not a terrain cache this reader can open without executing it; delete it to rebuild".format(terrain_save_path))
        self._oob_region = float(d["oob_region"])
        self._spawn_min_x, self._spawn_max_x = float(d["spawn_min_x"]), float(d["spawn_max_x"])
        self._spawn_min_y, self._spawn_max_y = float(d["spawn_min_y"]), float(d["spawn_max_y"])
        self._terrain = terrain_util.SubTerrain.from_arrays(t["hf"], t["min_point"], t["dxdy"], t.get("hf_mask"), t.get("hf_maxmin"), device=self._device)
        self._terrain_struct = None
        return d["vertices"], d["triangles"], self._terrain.min_point

    # ------------------------------------------------------------------ the generator's local height grid (reference :368-428)
    def _build_obs_hfs(self, env_config, num_envs, device):
        g = self._mgen
        self._mgdm_local_xy_points = geom_util.get_xy_grid_points(center=torch.zeros(2, dtype=torch.float32, device=device), dx=g._dx, dy=g._dy,
                                                                  num_x_neg=g._num_x_neg, num_x_pos=g._num_x_pos, num_y_neg=g._num_y_neg,
                                                                  num_y_pos=g._num_y_pos)
        num_points = (1 + g._num_x_neg + g._num_x_pos) * (1 + g._num_y_neg + g._num_y_pos)
        self._mgdm_hfs = torch.zeros((num_envs, num_points), dtype=torch.float32, device=device)
        self._mgdm_floor_heights = torch.zeros(num_envs, dtype=torch.float32, device=device)

    def refresh_obs_hfs(self, char_root_pos_xyz, char_heading):
        """Heights on the generator's grid around every character (relative to the root) and the floor height under it; positions are
        global.  The ray fan of the observation (RefCharEnv._refresh_ray_obs_hfs) is part of the fused launch and not repeated here.
        The reference refreshes this grid on every step although only a replan could read it; here the env calls it at replans."""
        gx, gy = self._mgdm_local_xy_points.shape[0], self._mgdm_local_xy_points.shape[1]
        n = char_root_pos_xyz.shape[0]
        xy = char_root_pos_xyz[..., 0:2]
        pts = torch_util.rotate_2d_vec(self._mgdm_local_xy_points.unsqueeze(0).expand(n, -1, -1, -1),
                                       char_heading.unsqueeze(-1).unsqueeze(-1).expand(-1, gx, gy)) + xy.unsqueeze(1).unsqueeze(1)
        hfs = terrain_util.get_local_hf_from_terrain(pts.reshape(n * gx * gy, 2), self._terrain).view(n, gx * gy)
        self._mgdm_hfs = hfs - self._char_root_pos[..., 2].unsqueeze(-1)
        self._mgdm_floor_heights = terrain_util.get_local_hf_from_terrain(xy, self._terrain)

    # ------------------------------------------------------------------ targets (reference :430-474)
    def pick_new_xy_targets(self, env_ids=None):
        """A point at U[dist_min, dist_max] in a direction within +-pi * heading_scale of the character's heading, and the time at which
        the next one is due."""
        if env_ids is None:
            env_ids = torch.arange(self._num_envs, dtype=torch.int64, device=self._device)
        n = len(env_ids)
        heading = (self._rand(n) * (torch.pi * 2) - torch.pi) * self._target_heading_scale
        dist = self._rand(n) * (self._target_dist_max - self._target_dist_min) + self._target_dist_min
        rel = torch.zeros((n, 2), dtype=torch.float32, device=self._device)
        rel[:, 0] = 1.0
        rel = torch_util.rotate_2d_vec(rel * dist.unsqueeze(-1), heading)
        rel = torch_util.rotate_2d_vec(rel, torch_util.calc_heading(self._char_root_rot[env_ids]))
        self._target_xy[env_ids] = self._char_root_pos[env_ids, 0:2] + rel
        nxt = self._rand(n) * (self._target_dur_max - self._target_dur_min) + self._target_dur_min + self._time_buf[env_ids]
        self._next_target_xy_time[env_ids] = 100000.0 if self._dont_auto_update_targets else nxt

    def pick_new_xy_targets_masked(self, mask):
        """pick_new_xy_targets for the rows where `mask` [n] is set: three draws for ALL rows, taken under the mask"""
        n = self._num_envs
        heading = (self._rand(n) * (torch.pi * 2) - torch.pi) * self._target_heading_scale
        dist = self._rand(n) * (self._target_dist_max - self._target_dist_min) + self._target_dist_min
        rel = torch.zeros((n, 2), dtype=torch.float32, device=self._device)
        rel[:, 0] = 1.0
        rel = torch_util.rotate_2d_vec(rel * dist.unsqueeze(-1), heading)
        rel = torch_util.rotate_2d_vec(rel, torch_util.calc_heading(self._char_root_rot))
        new_xy = self._char_root_pos[:, 0:2] + rel
        nxt = self._rand(n) * (self._target_dur_max - self._target_dur_min) + self._target_dur_min + self._time_buf
        if self._dont_auto_update_targets:
            nxt = torch.full_like(nxt, 100000.0)
        self._target_xy.copy_(torch.where(mask.unsqueeze(-1), new_xy, self._target_xy))
        self._next_target_xy_time.copy_(torch.where(mask, nxt, self._next_target_xy_time))

    def reset_rows_masked(self, mask):
        """reset(env_ids) without a pending replan (reference :89-128 second branch, _reset_char :499-505) for the rows whose bool mask
        is set, as fixed-shape work: clocks and flag cleared, character put on the current frame of its plan, history restarted.
        The caller refreshes body poses / observations of those rows (masked launches) and counts the episode."""
        m1, m2 = mask, mask.unsqueeze(-1)
        self._timestep_buf.masked_fill_(m1, 0)
        self._time_buf.masked_fill_(m1, 0.0)
        self._done_buf.masked_fill_(m1, base_env.DoneFlags.NULL.value)
        self._char_rigid_body_vel.masked_fill_(m2.unsqueeze(-1), 0.0)
        self._char_rigid_body_ang_vel.masked_fill_(m2.unsqueeze(-1), 0.0)
        for dst, src in ((self._char_root_pos, self._ref_root_pos), (self._char_root_rot, self._ref_root_rot), (self._char_root_vel, self._ref_root_vel),
                         (self._char_root_ang_vel, self._ref_root_ang_vel), (self._char_dof_pos, self._ref_dof_pos), (self._char_dof_vel, self._ref_dof_vel)):
            dst.copy_(torch.where(m2, src, dst))
        self._agent_state_hist.set_vals_masked(self._ref_state_hist, mask)

    # ------------------------------------------------------------------ state helpers (reference :499-565)
    def _reset_char(self, env_ids):
        """a soft reset puts the character on the CURRENT frame of its plan (plans change at replans only)"""
        if len(env_ids) > 0:
            self._char_state_init_from_ref(env_ids)
            self._need_refresh[env_ids] = True
            self._agent_state_hist.set_vals(self._ref_state_hist, env_ids)

    def _char_state_init_from_ref(self, env_ids):
        self._char_rigid_body_vel[env_ids] = 0.0
        self._char_rigid_body_ang_vel[env_ids] = 0.0
        self._char_root_pos[env_ids] = self._ref_root_pos[env_ids]
        self._char_root_rot[env_ids] = self._ref_root_rot[env_ids]
        self._char_root_vel[env_ids] = self._ref_root_vel[env_ids]
        self._char_root_ang_vel[env_ids] = self._ref_root_ang_vel[env_ids]
        self._char_dof_pos[env_ids] = self._ref_dof_pos[env_ids]
        self._char_dof_vel[env_ids] = self._ref_dof_vel[env_ids]

    def _get_char_motion_frames(self, eps=1e-5, ref=False, null=False):
        if null:
            ret = MotionFrames()
            ret.init_blank_frames(self._kin_char_model, 1, self._num_envs)
            ret.body_pos = ret.body_rot = None
            return ret
        km = self._kin_char_model
        if ref:
            root_pos, root_rot, contacts = self._ref_root_pos, self._ref_root_rot, self._ref_contacts
            joint_rot = km.dof_to_rot(self._ref_dof_pos.contiguous())
        else:
            root_pos, root_rot = self._char_root_pos, self._char_root_rot
            joint_rot = km.dof_to_rot(self._char_dof_pos.contiguous())
            contacts = (torch.linalg.vector_norm(self._char_contact_forces, dim=-1) > eps).to(torch.float32)
        return MotionFrames(root_pos=root_pos.clone(), root_rot=root_rot.clone(), joint_rot=joint_rot.clone(), contacts=contacts.clone()).unsqueeze(1)

    def _get_state_dict_from_motion_lib(self, t, motion_ids):
        times = torch.ones(motion_ids.shape, dtype=torch.float, device=self._device) * t
        rp, rr, _, _, jr, _, con = self._motion_lib.calc_motion_frame(motion_ids, times)
        return MotionFrames(root_pos=rp, root_rot=rr, joint_rot=jr, contacts=con).unsqueeze(1)

    # ------------------------------------------------------------------ replan (reference :575-826)
    @torch.no_grad()
    def replan(self):
        dev = self._device
        hard_mask = self._compute_hard_reset_envs_mask()
        hard_ids = hard_mask.nonzero().squeeze(dim=-1)
        replan_ids = (~hard_mask).nonzero().squeeze(dim=-1)
        H = int(hard_ids.shape[0])
        if H > 0:
            # re-spawn: a uniform place inside the spawn square, new targets from there, 0.7-0.9 m above the floor
            self._timestep_buf[hard_ids] = 0
            self._time_buf[hard_ids] = 0
            self._done_buf[hard_ids] = base_env.DoneFlags.NULL.value
            new_x = self._rand(H) * (self._spawn_max_x - self._spawn_min_x) + self._spawn_min_x
            new_y = self._rand(H) * (self._spawn_max_y - self._spawn_min_y) + self._spawn_min_y
            gxy = torch.stack([new_x, new_y], dim=-1)
            new_heading = self._rand(H) * torch.pi * 2.0 - torch.pi
            gx, gy = self._mgdm_local_xy_points.shape[0], self._mgdm_local_xy_points.shape[1]
            local = torch_util.rotate_2d_vec(self._mgdm_local_xy_points.unsqueeze(0).expand(H, -1, -1, -1), new_heading.unsqueeze(-1).unsqueeze(-1))
            grid = (gxy.unsqueeze(1).unsqueeze(1) + local).reshape(H * gx * gy, 2)
            self._mgdm_hfs[hard_ids] = terrain_util.get_local_hf_from_terrain(grid, self._terrain).view(H, gx * gy)
            floor = terrain_util.get_local_hf_from_terrain(gxy, self._terrain)
            self._mgdm_floor_heights[hard_ids] = floor
            self._char_root_pos[hard_ids, 0] = new_x - self._env_offsets[hard_ids, 0]
            self._char_root_pos[hard_ids, 1] = new_y - self._env_offsets[hard_ids, 1]
            self.pick_new_xy_targets(hard_ids)
            self._char_root_pos[hard_ids, 2] = floor + self._rand(H) * 0.2 + 0.7
        prev_frames = motion_util.cat_motion_frames([self._agent_state_hist, self._get_char_motion_frames()])
        settings = MDMGenSettings()
        settings.ddim_stride = 100
        settings.use_cfg = True
        settings.use_prev_state = torch.ones(self._num_envs, dtype=torch.bool, device=dev)
        settings.use_prev_state[hard_ids] = False
        settings.prev_state_ind_key = torch.ones(self._num_envs, dtype=torch.bool, device=dev)
        settings.prev_state_ind_key[hard_ids] = False
        plan = self._mgen(self._target_xy, prev_frames, self._terrain, self._kin_char_model, settings)
        frames = torch.cat([plan.root_pos, torch_util.quat_to_exp_map(plan.root_rot), self._kin_char_model.rot_to_dof(plan.joint_rot.contiguous())], dim=-1)
        if self._motion_lib is None or tuple(frames.shape[0:2]) != self._motion_lib._uniform_shape:
            self._motion_lib = motion_lib.MotionLib(frames.contiguous(), self._kin_char_model, dev, init_type="motion_frames",
                                                    loop_mode=motion_lib.LoopMode.CLAMP, fps=self._mgen._sequence_fps, contact_info=True,
                                                    contacts=plan.contacts.contiguous())
        else:
            self._motion_lib.update_frames(frames, plan.contacts)
        frame_zero = self._get_state_dict_from_motion_lib(0.0, self._motion_ids)
        self._ref_state_hist.store(frame_zero)
        if H > 0:
            # the re-spawned characters start on the plan's frame at one control step
            times = torch.ones(H, dtype=torch.float, device=dev) * self._timestep
            rp, rr, rv, rav, jr, dv, _ = self._motion_lib.calc_motion_frame(self._motion_ids[hard_ids], times)
            self._char_rigid_body_vel[hard_ids] = 0.0
            self._char_rigid_body_ang_vel[hard_ids] = 0.0
            self._char_root_pos[hard_ids] = rp
            self._char_root_rot[hard_ids] = rr
            self._char_root_vel[hard_ids] = rv
            self._char_root_ang_vel[hard_ids] = rav
            self._char_dof_pos[hard_ids] = self._kin_char_model.rot_to_dof(jr)
            self._char_dof_vel[hard_ids] = dv
            self._agent_state_hist.set_vals(frame_zero, hard_ids)
            self._need_refresh[hard_ids] = True
        # with two previous states the character sits at one step into the plan and so does the reference pose
        start = self._timestep * (self._num_prev_states - 1)
        self._mgdm_time_buf[0] = start
        self._plan_time_host = np.float32(start)
        self._replan_buf[:] = ReplanFlags.REPLAN.value
        self._replan_counter[replan_ids] += 1
        self._replan_counter[hard_ids] = 1
        self._replan_flag = False
        self._update_ref_motion()

    # ------------------------------------------------------------------ accessors (reference :828-861)
    def get_target_dim(self):
        name = getattr(getattr(self._mgen, "_target_type", None), "name", "XY_DIR")
        return {"XY_POS": 2, "XY_POS_AND_HEADING": 3, "XY_DIR": 2}[name]

    def get_mgdm_time_buf(self):
        return self._mgdm_time_buf

    def get_replan_counter(self):
        return self._replan_counter

    def apply_hard_reset(self):
        self._replan_buf[:] = ReplanFlags.HARD_RESET.value
        self._replan_flag = True

    def set_rand_root_pos_offset_scale(self, val):
        self._rand_root_pos_offset_scale = val
