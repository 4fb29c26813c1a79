"""PPO learner of the tracker (DeepMimic-style imitation).

Host-side mirror of the reference's agent stack -- learning/base_agent.py BaseAgent, learning/ppo_agent.py PPOAgent and
learning/dm_ppo_agent.py DMPPOAgent -- as one class with the same method names, config keys, experience-buffer layout
(learning/experience_buffer.py, buffers added at base_agent.py:224-253, ppo_agent.py:62-80, dm_ppo_agent.py:281-313), loss
(ppo_agent.py:212-330) and checkpoint keys.  What changes is where the work runs:
  * TD(lambda) and the advantage normalisation are HIP kernels (parc_amd/learning/rl_util.py) instead of a Python loop;
  * the minibatch sampler gathers only the six buffers the loss reads;
  * the large-critic-loss guard is evaluated on the device and the NaN trap reads one stack of losses per update epoch (no ``.item()``
    per minibatch; the offending minibatch is dumped and training stops, like ppo_agent.py:242-252);
  * gradients live in one flat buffer that RCCL all-reduces in place (learning/mp_optimizer.py).
"""
import enum
import os
import time

import numpy as np
import torch

from ..envs import base_env
from ..gym_spaces import Box
from ..util import mp_util
from ..util.logger import Logger
from . import dm_ppo_model, experience_buffer, mp_optimizer, normalizer, rl_util
from .dm_ppo_return_tracker import DMPPOReturnTracker
from .tracking_error_tracker import TrackingErrorTracker


class AgentMode(enum.Enum):
    TRAIN = 0
    TEST = 1


# "norm_obs" = the normalised observations, written once per iteration by _build_train_data (the normaliser statistics do not
# change during the update), so a minibatch is one gather instead of gather + subtract + divide + clamp
_LOSS_KEYS = ["norm_obs", "action", "a_logp", "tar_val", "adv", "rand_action_mask"]


def enable_tuned_gemms():
    """Use the offline GEMM solution selection shipped in parc_amd/tunableop_results.csv (PyTorch TunableOp over hipBLASLt /
    rocBLAS, produced by tools/tune_gemms.py on an MI355X for the shapes of the 4096-env training iteration: the default
    heuristic leaves 10-15 % on the first-layer GEMMs).  Selection only - nothing is tuned at run time; shapes that are not
    in the file, or a file recorded for another library version / architecture, fall back to the default heuristic."""
    import torch.cuda.tunable as tunable
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tunableop_results.csv")
    if not os.path.exists(path) or tunable.is_enabled():
        return
    tunable.set_filename(path, False)
    tunable.tuning_enable(False)
    if hasattr(tunable, "write_file_on_exit"):
        tunable.write_file_on_exit(False)   # selection only: never rewrite the shipped file (N ranks share it)
    tunable.enable(True)


class DMPPOAgent(torch.nn.Module):
    NAME = "DM_PPO"

    def __init__(self, config, env, device):
        super().__init__()
        self._env = env
        self._device = device
        self._iter = 0
        self._sample_count = 0
        self._config = config
        self._is_terrain_runner = env.NAME in ("ig_terrain_runner", "ig_parkour")
        self._has_target_task = env.NAME in ("ig_terrain_runner", "ig_deepmimic_terrain", "ig_parkour")
        self._load_params(config)
        self._build_normalizers()
        self._model = dm_ppo_model.DMPPOModel(config["model"], env)
        self.to(self._device)
        params = [p for p in self.parameters() if p.requires_grad]
        self._optimizer = mp_optimizer.MPOptimizer(config["optimizer"], params)
        self._build_exp_buffer(config)
        self._train_return_tracker = DMPPOReturnTracker(self.get_num_envs(), self._device, self._has_target_task)
        self._test_return_tracker = DMPPOReturnTracker(self.get_num_envs(), self._device, self._has_target_task)
        self._mode = AgentMode.TRAIN
        self._curr_obs = None
        self._curr_info = None
        self._ppo_cfg = None
        if str(self._device).startswith("cuda") and config.get("tuned_gemms", True):
            enable_tuned_gemms()
        # hipGraph rollout: the fixed-shape part of one env step (policy forward, record, simulator, post-step kernel,
        # return tracker, record) is captured once and replayed; only the data-dependent reset of finished envs stays eager
        self._use_hip_graph = bool(config.get("hip_graph_rollout", True)) and str(self._device).startswith("cuda")
        self._graphs = dict()
        self._graph_pool = None
        self._graph_warm = 0
        self._exp_prob_t = torch.ones([1, 1], dtype=torch.float32, device=self._device)
        self._head_t = torch.zeros([1], dtype=torch.int64, device=self._device)
        self._head_dev, self._exp_prob_dev = 0, 1.0      # what those two device cells hold (host mirror)
        self._compute_times = []                         # (row, seconds) of the steps of this rollout, written in one go at its end
        self._replan_time_rows, self._replan_time_src = None, None
        self._ones_mask = None
        if getattr(self._env, "_report_tracking_error", False):
            self._test_tracking_error_tracker = TrackingErrorTracker(self.get_num_envs(), self._device)

    # ------------------------------------------------------------------ config (base_agent.py:170-182, ppo_agent.py:21-51)
    @staticmethod
    def rollout_shape(config, num_procs):
        """(steps_per_iter, batch_size) of one rank.  The reference divides the rollout length and the minibatch by the world size
        (base_agent.py:179-180, ppo_agent.py:27-29: ceil(32 / P) steps, ceil(4 / P) x num_envs rows per minibatch) so that the samples
        per iteration and the number of optimizer steps stay what they are on one GPU; "mp_scale_rollout: false" keeps the per-rank
        work fixed instead (weak scaling)."""
        P = int(num_procs) if config.get("mp_scale_rollout", True) else 1
        return int(np.ceil(config["steps_per_iter"] / P)), int(np.ceil(config["batch_size"] / P))

    def _load_params(self, config):
        self._discount = config["discount"]
        self._iters_per_output = config["iters_per_output"]
        self._iters_per_checkpoint = config["iters_per_checkpoint"]
        self._normalizer_samples = config.get("normalizer_samples", np.inf)
        self._test_episodes = config["test_episodes"]
        self._steps_per_iter, self._batch_size = self.rollout_shape(config, mp_util.get_num_procs())
        self._update_epochs = config["update_epochs"]
        self._td_lambda = config["td_lambda"]
        self._ppo_clip_ratio = config["ppo_clip_ratio"]
        self._norm_adv_clip = config["norm_adv_clip"]
        self._action_bound_weight = config["action_bound_weight"]
        self._action_entropy_weight = config["action_entropy_weight"]
        self._action_reg_weight = config["action_reg_weight"]
        self._critic_loss_weight = config["critic_loss_weight"]
        self._exp_anneal_samples = config.get("exp_anneal_samples", np.inf)
        self._exp_prob_beg = config.get("exp_prob_beg", 1.0)
        self._exp_prob_end = config.get("exp_prob_end", 1.0)
        self._clip_grad_norm = config.get("clip_grad_norm", False)
        self._max_grad_norm = config.get("max_grad_norm", 0.5)
        self._critic_loss_type = config.get("critic_loss_type", "L2")

    def _build_normalizers(self):
        """Observation normaliser that leaves the contact / heightmap segments untouched (dm_ppo_agent.py:78-117),
        action normaliser from the action bounds (base_agent.py:195-203)."""
        obs_space = self._env.get_obs_space()
        shapes = self._env._compute_obs(ret_obs_shapes=True)
        idx, cur = [], 0
        for key in shapes:
            shape = shapes[key]["shape"]
            flat = shape[0] * shape[1] if len(shape) >= 2 else shape[0]
            if not shapes[key]["use_normalizer"]:
                idx.append(torch.arange(cur, cur + flat, dtype=torch.int64, device=self._device))
            cur += flat
        non_norm = torch.cat(idx) if idx else None
        self._obs_norm = normalizer.Normalizer(obs_space.shape, device=self._device, dtype=torch.float32, non_norm_indices=non_norm,
                                               clip=self._config["norm_obs_clip"])
        a_space = self._env.get_action_space()
        assert isinstance(a_space, Box)
        a_mean = torch.tensor(0.5 * (a_space.high + a_space.low), device=self._device, dtype=torch.float32)
        a_std = torch.tensor(0.5 * (a_space.high - a_space.low), device=self._device, dtype=torch.float32)
        self._a_norm = normalizer.Normalizer(a_mean.shape, device=self._device, init_mean=a_mean, init_std=a_std, dtype=torch.float32)

    def _build_exp_buffer(self, config):
        T, N, dev = self._steps_per_iter, self.get_num_envs(), self._device
        self._exp_buffer = experience_buffer.ExperienceBuffer(buffer_length=T, batch_size=N, device=dev)
        obs_dim = list(self._env.get_obs_space().shape)
        a_dim = list(self._env.get_action_space().shape)
        B = self._env._cfg.num_bodies if hasattr(self._env, "_cfg") else 15

        def add(name, shape, dtype=torch.float32):
            self._exp_buffer.add_buffer(name, torch.zeros([T, N] + shape, device=dev, dtype=dtype))
        add("obs", obs_dim)
        add("next_obs", obs_dim)
        add("action", a_dim)
        add("reward", [])
        add("done", [], torch.int)
        add("a_logp", [])
        add("tar_val", [])
        add("adv", [])
        add("rand_action_mask", [])
        add("timestep", [], torch.int)
        add("ep_num", [], torch.int)
        add("compute_time", [])
        add("prev_char_contact_forces", [B, 3])
        add("next_char_contact_forces", [B, 3])
        add("env_id", [], torch.int64)
        add("norm_obs", obs_dim)          # not in the reference's buffer set: update-phase cache, see _LOSS_KEYS
        # likewise: the per-sample inputs of the loss as ONE record per sample, [Normalizer.normalize(action) of ppo_agent.py:214 (A) | a_logp |
        # adv | rand_action_mask | tar_val | pad], written once per iteration: a minibatch then gathers two arrays instead of six
        self._loss_rec_w = ((a_dim[0] if isinstance(a_dim, (list, tuple)) else int(a_dim)) + 4 + 3) // 4 * 4
        add("loss_rec", [self._loss_rec_w])
        self._env_ids = torch.arange(0, N, 1, device=dev, dtype=torch.int64)
        if self._is_terrain_runner:
            add("replan_timer", [])
            add("replan_counter", [], torch.int64)

    # ------------------------------------------------------------------ small helpers
    def get_num_envs(self):
        return self._env.get_num_envs()

    def get_action_size(self):
        return int(np.prod(self._env.get_action_space().shape))

    def calc_num_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def set_mode(self, mode):
        self._mode = mode
        self._env.set_mode(base_env.EnvMode.TRAIN if mode == AgentMode.TRAIN else base_env.EnvMode.TEST)

    def save(self, out_file):
        if mp_util.is_root_proc():
            torch.save(self.state_dict(), out_file)

    def load(self, in_file):
        self.load_state_dict(torch.load(in_file, map_location=self._device, weights_only=True))
        self._optimizer.sync()
        Logger.print("Loaded model parameters from {:s}".format(in_file))

    def _get_exp_prob(self):
        if np.isfinite(self._exp_anneal_samples):
            l = float(np.clip(float(self._sample_count) / self._exp_anneal_samples, 0.0, 1.0))
            return (1.0 - l) * self._exp_prob_beg + l * self._exp_prob_end
        return self._exp_prob_beg

    def _set_exp_prob(self, exp_prob):
        """The annealed exploration probability (ppo_agent.py:97-99) lives in a device scalar that the Bernoulli draw reads.
        A fill_ with a Python float inside a captured step would bake the capture-time value into the graph, so inside a
        captured step nothing is written here: _train_step_graph fills the scalar eagerly before every replay."""
        if not getattr(self, "_in_graph_step", False):
            self._exp_prob_t.fill_(exp_prob)
            self._exp_prob_dev = exp_prob

    def _need_normalizer_update(self):
        return self._sample_count < self._normalizer_samples

    # ------------------------------------------------------------------ acting (ppo_agent.py:87-119)
    @torch.no_grad()
    def _decide_action(self, obs, info):
        self._obs_ingested = self._action_recorded = False
        if getattr(self, "_in_graph_step", False) and self._mode == AgentMode.TRAIN and self._obs_norm.can_ingest(obs) \
                and obs.shape[0] == self.get_num_envs() and self._exp_buffer._device_head is not None:
            # captured step: normalisation for the forward pass, the raw copy into the experience buffer's row and the normaliser's
            # moment sums in ONE pass over the observation rows (_record_data_pre_step then skips those two)
            norm_obs = self._obs_norm.ingest(obs, record=self._need_normalizer_update(),
                                             copy_into=(self._exp_buffer.get_data("obs"), self._exp_buffer._device_head))
            self._obs_ingested = True
        else:
            norm_obs = self._obs_norm.normalize(obs)
        dist = self._model.eval_actor(norm_obs)
        if obs.is_cuda and dist.logstd.dim() == 2 and dist.logstd.stride(0) == 0 and self._config.get("fused_action_head", True):
            return self._decide_action_fused(dist, obs.shape[0])
        if self._mode == AgentMode.TRAIN:
            exp_prob = self._get_exp_prob()
            if exp_prob >= 1.0:
                norm_a = dist.sample()
                mask = torch.ones_like(norm_a[..., 0])
            else:
                self._set_exp_prob(exp_prob)
                mask = torch.bernoulli(self._exp_prob_t.expand(obs.shape[0], 1))
                norm_a = torch.where(mask == 1.0, dist.sample(), dist.mode)
                mask = mask.squeeze(-1)
        else:
            norm_a = dist.mode
            mask = torch.zeros_like(norm_a[..., 0])
        logp = dist.log_prob(norm_a)
        a = self._a_norm.unnormalize(norm_a)
        return a, {"a_logp": logp, "rand_action_mask": mask}

    def _decide_action_fused(self, dist, n):
        """Same rule as the branches of _decide_action, with sampling, log-probability and un-normalisation in one launch
        (parc_action_head) after the random draws."""
        from .. import _hip
        mean = dist.mean.contiguous()
        A = mean.shape[1]
        if self._mode == AgentMode.TRAIN:
            exp_prob = self._get_exp_prob()
            if exp_prob >= 1.0:
                if self._ones_mask is None or self._ones_mask.shape[0] != n:
                    self._ones_mask = torch.ones(n, dtype=torch.float32, device=mean.device)
                mask = self._ones_mask                     # read-only downstream (action head, record): one persistent tensor
            else:
                self._set_exp_prob(exp_prob)
                mask = torch.bernoulli(self._exp_prob_t.expand(n, 1)).squeeze(-1).contiguous()
            # inside a captured step the env has drawn the step's random numbers in one launch of its own counter-based generator
            # (_train_step_body); torch's generator otherwise
            noise = self._step_noise if (getattr(self, "_step_noise", None) is not None and self._step_noise.shape == mean.shape) else torch.randn_like(mean)
        else:
            mask = torch.zeros(n, dtype=torch.float32, device=mean.device)
            noise = mean                   # unused where the mask is 0
        a = torch.empty_like(mean)
        logp = torch.empty(n, dtype=torch.float32, device=mean.device)
        p = _hip.ptr
        eb = self._exp_buffer
        self._action_recorded = False
        if (getattr(self, "_in_graph_step", False) and self._mode == AgentMode.TRAIN and A <= 32 and eb._device_head is not None and n == self.get_num_envs()
                and hasattr(self._env, "_char_contact_forces")):
            # captured step: the head also writes action / a_logp / rand_action_mask / the contact forces of this moment into the
            # experience buffer's row (_record_data_pre_step without a launch of its own)
            f = self._env._char_contact_forces
            nf = f[0].numel()
            _hip.check(_hip.lib().parc_action_head_record(_hip.stream(), n, A, p(mean), p(dist.logstd[0].contiguous()), p(noise), p(mask),
                                                          p(self._a_norm.get_mean()), p(self._a_norm.get_std()), p(a), p(logp),
                                                          p(eb.get_data("action")), p(eb.get_data("a_logp")), p(eb.get_data("rand_action_mask")),
                                                          p(f.contiguous()), p(eb.get_data("prev_char_contact_forces")), nf, p(eb._device_head)),
                       "parc_action_head_record")
            self._action_recorded = True
            return a, {"a_logp": logp, "rand_action_mask": mask}
        _hip.check(_hip.lib().parc_action_head(_hip.stream(), n, A, p(mean), p(dist.logstd[0].contiguous()), p(noise), p(mask),
                                               p(self._a_norm.get_mean()), p(self._a_norm.get_std()), p(a), p(logp)), "parc_action_head")
        return a, {"a_logp": logp, "rand_action_mask": mask}

    def step(self):
        action, action_info = self._decide_action(self._curr_obs, self._curr_info)
        next_obs, r, done, next_info = self._env.step(action)
        return next_obs, r, done, next_info, action, action_info

    def _record_data_pre_step(self, obs, info, action, action_info):
        eb = self._exp_buffer
        if getattr(self, "_in_graph_step", False):
            # captured step: the pre-step buffers in one launch (K15), sources at stable addresses; the observation rows and their
            # moment sums were already taken care of by the pass that normalised them (_decide_action)
            ingested = getattr(self, "_obs_ingested", False)
            if self._need_normalizer_update() and not ingested:
                self._obs_norm.record(obs)
            items = [] if ingested else [("obs", obs)]
            if not getattr(self, "_action_recorded", False):     # (else the action head wrote them itself: _decide_action_fused)
                items += [("action", action), ("a_logp", action_info["a_logp"]), ("rand_action_mask", action_info["rand_action_mask"]),
                          ("prev_char_contact_forces", self._env._char_contact_forces)]
            if items:
                eb.record_group(items)
            return
        eb.record("obs", obs)
        eb.record("action", action)
        if self._need_normalizer_update():
            self._obs_norm.record(obs)
        eb.record("a_logp", action_info["a_logp"])
        eb.record("rand_action_mask", action_info["rand_action_mask"])
        # (inside a captured step the info snapshot of the previous eager reset is not a stable address: read the env's
        # own buffer, which holds the same values until the simulator runs)
        eb.record("prev_char_contact_forces", self._env._char_contact_forces if getattr(self, "_in_graph_step", False)
                  else info["char_contact_forces"])

    def _record_data_post_step(self, next_obs, r, done, next_info):
        eb = self._exp_buffer
        if getattr(self, "_in_graph_step", False):
            items = [("next_obs", next_obs), ("reward", r), ("done", done), ("timestep", next_info["timestep"]),
                     ("ep_num", next_info["ep_num"]), ("next_char_contact_forces", next_info["char_contact_forces"]),
                     ("env_id", self._env_ids)]
            if self._is_terrain_runner:
                # (the plan clock is ONE device value: the record launch writes it to every env of the row)
                items += [("replan_timer", self._env.get_replan_time_buf()), ("replan_counter", self._env.get_replan_counter())]
            eb.record_group(items)
            return
        eb.record("next_obs", next_obs)
        eb.record("reward", r)
        eb.record("done", done)
        eb.record("timestep", next_info["timestep"])
        eb.record("ep_num", next_info["ep_num"])
        if not getattr(self, "_in_graph_step", False):
            eb.get_data("compute_time")[eb._buffer_head].fill_(float(next_info["compute_time"]))
        eb.record("next_char_contact_forces", next_info["char_contact_forces"])
        eb.record("env_id", self._env_ids)
        if self._is_terrain_runner:
            eb.record("replan_timer", self._env.get_replan_time_buf().expand(self.get_num_envs()))
            eb.record("replan_counter", self._env.get_replan_counter())

    def _reset_done_envs(self, done):
        done_indices = (done != base_env.DoneFlags.NULL.value).nonzero(as_tuple=False).flatten()
        return self._env.reset(done_indices)

    def _device_tick(self):
        """does the captured step advance the device's write row itself?  (the env's one-launch random draw carries the tick)"""
        return hasattr(self._env, "step_randoms") and self._config.get("device_step_randoms", True)

    def _train_step_body(self, device_reset=False):
        self._step_noise = None
        if getattr(self, "_in_graph_step", False) and self._device_tick():
            # first launch of the step: every random number it needs (policy noise + the env's uniform pool) and the write row's tick
            self._step_noise = self._env.step_randoms(self.get_action_size(), tick=(self._head_t, self._exp_buffer._buffer_length))
        action, action_info = self._decide_action(self._curr_obs, self._curr_info)
        self._record_data_pre_step(self._curr_obs, self._curr_info, action, action_info)
        next_obs, r, done, next_info = self._env.step(action)
        self._train_return_tracker.update(next_info, done)
        self._record_data_post_step(next_obs, r, done, next_info)
        if device_reset:
            # finished envs restart on the device (masked kernels): no nonzero(), the step stays sync-free
            self._curr_obs, self._curr_info = self._env.reset_done(done)
        return done

    def _device_reset_ok(self):
        env = self._env
        return self._graph_ok() and hasattr(env, "reset_done") and env.supports_device_reset()

    def _graph_ok(self):
        env = self._env
        return (self._use_hip_graph and self._mode == AgentMode.TRAIN and not getattr(env, "_write_agent_states_flag", False)
                and hasattr(env, "_char_contact_forces") and getattr(getattr(env, "_core", None), "timing_events", None) is None
                and getattr(env, "supports_graph_step", lambda: True)())

    def _train_step_graph(self, device_reset=False):
        """One rollout step through a captured hipGraph (torch.cuda.CUDAGraph = hipGraph on ROCm).  Graphs are keyed by
        the host-side branches inside the step; step inputs are the env's persistent buffers, the write row of the
        experience buffer is the device scalar ``_head_t``."""
        eb = self._exp_buffer
        exp_prob = self._get_exp_prob()
        sig = self._env.host_step_signature() if hasattr(self._env, "host_step_signature") else ()
        key = (self._need_normalizer_update(), exp_prob >= 1.0, device_reset) + tuple(sig)
        # the write row and the exploration probability live on the device; the host only re-writes them when they differ from what
        # the device holds - no fill per step.  With an env that draws the step's random numbers itself (step_randoms) the captured
        # step's FIRST launch moves the row on by one, so between steps the cell holds the row of the step before.
        want = (eb._buffer_head - 1) % eb._buffer_length if self._device_tick() else eb._buffer_head
        if self._head_dev != want:
            self._head_t.fill_(want)
            self._head_dev = want
        if self._exp_prob_dev != exp_prob:
            self._exp_prob_t.fill_(exp_prob)         # read by the captured Bernoulli draw: the value of THIS step, not of the capture
            self._exp_prob_dev = exp_prob
        g = self._graphs.get(key)
        if g is None:
            if self._graph_warm < 2:                 # library handles / workspaces are created by eager steps first
                self._graph_warm += 1
                return self._train_step_body(device_reset)
            eb.set_device_head(self._head_t)
            self._in_graph_step = True
            snapshots = getattr(self._env, "_info_snapshots", None)
            if snapshots is not None:
                self._env._info_snapshots = False
            count0 = self._obs_norm._new_count
            host0 = self._env.host_step_state() if hasattr(self._env, "host_step_state") else None
            captured = False
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread_local: calls made by other threads of the process (e.g. the RCCL watchdog of a multi-GPU job) must
                # not invalidate the capture
                with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
                    done = self._train_step_body(device_reset)
                if self._graph_pool is None:
                    self._graph_pool = g.pool()
                captured = True
            except Exception as exc:                 # keep training: eager launches are always available
                Logger.print("hipGraph capture of the rollout step failed ({}); continuing with eager launches".format(exc))
                self._use_hip_graph = False
            finally:
                eb.set_device_head(None)
                self._in_graph_step = False
                if snapshots is not None:
                    self._env._info_snapshots = snapshots
            self._obs_norm._new_count = count0       # capture enqueues nothing; the replay below is this step
            if not captured:
                torch.cuda.synchronize()
                if hasattr(self._env, "restore_host_step_state"):
                    self._env.restore_host_step_state(host0)     # the failed capture ran the step's host code; the eager step runs it again
                done = self._train_step_body(False)
                self._graph_fallback_reset = True
                return done
            self._graphs[key] = (g, done)
            fresh = True          # the capture ran the step's host code once already (plan-clock mirror etc.): the replay below is that step
        else:
            fresh = False
        g, done = self._graphs[key]
        g.replay()
        if self._device_tick():
            self._head_dev = (self._head_dev + 1) % eb._buffer_length  # the replayed step moved the device's write row on
        if not fresh and hasattr(self._env, "host_step_replayed"):
            self._env.host_step_replayed()           # what the step changes on the host (the device part is the graph)
        if key[0]:
            self._obs_norm._new_count += self.get_num_envs()
        self._compute_times.append((eb._buffer_head, time.time() - self._env._start_compute_time))
        return done

    def _flush_compute_times(self):
        """info["compute_time"] of the rollout's graph-replayed steps into the experience buffer: one small copy per rollout instead of
        one fill per step"""
        if self._compute_times:
            rows = torch.tensor([r for r, _ in self._compute_times], dtype=torch.long, device=self._device)
            vals = torch.tensor([v for _, v in self._compute_times], dtype=torch.float32, device=self._device)
            buf = self._exp_buffer.get_data("compute_time")
            buf[rows] = vals.unsqueeze(1).expand(-1, buf.shape[1])
            self._compute_times = []

    def _rollout_train(self, num_steps):
        for _ in range(num_steps):
            if self._graph_ok():
                dev_reset = self._device_reset_ok()
                done = self._train_step_graph(dev_reset)
                if getattr(self, "_graph_fallback_reset", False):     # capture failed: this step ran eagerly without the device reset
                    self._graph_fallback_reset = False
                    dev_reset = False
            else:
                dev_reset = False
                done = self._train_step_body()
            if not dev_reset:
                self._curr_obs, self._curr_info = self._reset_done_envs(done)
            self._exp_buffer.inc()
        self._flush_compute_times()

    def _rollout_test(self, num_episodes):
        self._test_return_tracker.reset()
        report_te = getattr(self._env, "_report_tracking_error", False)
        if report_te:
            self._test_tracking_error_tracker.reset()
        if num_episodes == 0:
            return {"mean_return": 0.0, "mean_ep_len": 0.0, "num_eps": 0}
        min_eps_per_env = int(np.ceil(num_episodes / self.get_num_envs()))
        while True:
            action, _ = self._decide_action(self._curr_obs, self._curr_info)
            _, _, done, next_info = self._env.step(action)
            self._test_return_tracker.update(next_info, done)
            if report_te and "tracking_error" in next_info:
                self._test_tracking_error_tracker.update(next_info["tracking_error"], done)
            self._curr_obs, self._curr_info = self._reset_done_envs(done)
            if torch.all(self._test_return_tracker.get_eps_per_env() > min_eps_per_env - 1):
                break
        out = {"mean_return": self._test_return_tracker.get_mean_return().item(),
               "mean_ep_len": self._test_return_tracker.get_mean_ep_len().item(),
               "num_eps": self._test_return_tracker.get_episodes()}
        if report_te:
            t = self._test_tracking_error_tracker
            out.update({"test_mean_root_pos_tracking_err": t.get_mean_root_pos_err().item(),
                        "test_mean_root_rot_tracking_err": t.get_mean_root_rot_err().item(),
                        "test_mean_body_pos_tracking_err": t.get_mean_body_pos_err().item(),
                        "test_mean_body_rot_tracking_err": t.get_mean_body_rot_err().item(),
                        "test_mean_dof_vel_tracking_err": t.get_mean_dof_vel_err().item(),
                        "test_mean_root_vel_tracking_err": t.get_mean_root_vel_err().item(),
                        "test_mean_root_ang_vel_tracking_err": t.get_mean_root_ang_vel_err().item()})
        return out

    def hard_reset_envs(self):
        if self._is_terrain_runner:
            self._env.apply_hard_reset()
        self._curr_obs, self._curr_info = self._env.reset()

    def test_model(self, num_episodes):
        self.eval()
        self.set_mode(AgentMode.TEST)
        self.hard_reset_envs()
        P = mp_util.get_num_procs()
        return self._rollout_test(int(np.ceil(num_episodes / P)))

    @torch.no_grad()
    def _critic_values(self):
        """(V(obs), V(next_obs)) over the whole rollout buffer, [T, N] each; also fills the "norm_obs" cache."""
        eb = self._exp_buffer
        obs, next_obs, done = eb.get_data("obs"), eb.get_data("next_obs"), eb.get_data("done")
        norm_obs = self._obs_norm.normalize(obs, out=eb.get_data("norm_obs"))
        vals = self._model.eval_critic(norm_obs).squeeze(-1)
        # V(next_obs[t]) == V(obs[t+1]) wherever env did not finish at t (the two rows hold the same observation), so the
        # second critic pass of the reference (ppo_agent.py:146-150) only has to run on the last step and on finished envs
        T = obs.shape[0]
        need = done != base_env.DoneFlags.NULL.value
        need[T - 1] = True
        next_vals = torch.empty_like(vals)
        next_vals[:T - 1] = vals[1:]
        idx = need.flatten().nonzero().flatten()                      # one host sync per iteration
        sel = next_obs.flatten(0, 1)[idx]
        next_vals.flatten()[idx] = self._model.eval_critic(self._obs_norm.normalize(sel)).squeeze(-1)
        return vals, next_vals

    # ------------------------------------------------------------------ train data (dm_ppo_agent.py:343-411)
    @torch.no_grad()
    def _build_train_data(self):
        self.eval()
        eb = self._exp_buffer
        obs, next_obs = eb.get_data("obs"), eb.get_data("next_obs")
        r, done = eb.get_data("reward"), eb.get_data("done")
        mask = eb.get_data("rand_action_mask")
        vals, next_vals = self._critic_values()
        r_min, r_max = self._env.get_reward_bounds()
        next_vals = torch.clamp(next_vals, r_min / (1.0 - self._discount), r_max / (1.0 - self._discount))
        succ_val = self._env.get_reward_succ() / (1.0 - self._discount)
        fail_val = self._env.get_reward_fail() / (1.0 - self._discount)
        next_vals = torch.where(done == base_env.DoneFlags.SUCC.value, torch.full_like(next_vals, succ_val), next_vals)
        next_vals = torch.where(done == base_env.DoneFlags.FAIL.value, torch.full_like(next_vals, fail_val), next_vals)
        new_vals = rl_util.compute_td_lambda_return(r, next_vals, done, self._discount, self._td_lambda)      # K16 (HIP)
        norm_adv, mean_std = rl_util.normalize_advantage(new_vals, vals, mask, self._norm_adv_clip)            # K17 (HIP)
        eb.set_data("tar_val", new_vals)
        eb.set_data("adv", norm_adv)
        if eb.has_buffer("loss_rec"):             # (the action normaliser is constant: bounds of the action space)
            rec = eb.get_data("loss_rec")
            A = eb.get_data("action").shape[-1]
            self._a_norm.normalize(eb.get_data("action"), out=rec[..., 0:A])
            rec[..., A] = eb.get_data("a_logp")
            rec[..., A + 1] = norm_adv
            rec[..., A + 2] = mask
            rec[..., A + 3] = new_vals
        return {"adv_mean": mean_std[0], "adv_std": mean_std[1]}

    # ------------------------------------------------------------------ update (ppo_agent.py:186-330)
    def _compute_loss(self, batch):
        norm_obs = batch["norm_obs"] if "norm_obs" in batch else self._obs_norm.normalize(batch["obs"])
        norm_a = self._a_norm.normalize(batch["action"])
        pred = self._model.eval_critic(norm_obs).squeeze(-1)
        if norm_obs.is_cuda and self._config.get("fused_ppo_loss", True):
            fused = self._compute_loss_fused(batch, norm_obs, norm_a, pred)
            if fused is not None:
                return fused
        diff = batch["tar_val"] - pred
        critic_loss = torch.mean(torch.square(diff)) if self._critic_loss_type == "L2" else torch.mean(torch.abs(diff))
        m = (batch["rand_action_mask"] == 1.0).to(torch.float32)
        cnt = m.sum().clamp_min(1.0)
        # masked means == the reference's boolean-index selection of random-action samples, without the host sync
        a_dist = self._model.eval_actor(norm_obs)
        a_logp = a_dist.log_prob(norm_a)
        ratio = torch.exp(a_logp - batch["a_logp"])
        adv = batch["adv"]
        l0 = adv * ratio
        l1 = adv * torch.clamp(ratio, 1.0 - self._ppo_clip_ratio, 1.0 + self._ppo_clip_ratio)
        actor_loss = -(torch.minimum(l0, l1) * m).sum() / cnt
        info = {"critic_loss": critic_loss.detach(), "clip_frac": (((torch.abs(ratio - 1.0) > self._ppo_clip_ratio).float() * m).sum() / cnt).detach(),
                "imp_ratio": ((ratio * m).sum() / cnt).detach()}
        if self._action_bound_weight != 0:
            vmin = torch.clamp_max(a_dist.mode + 1.0, 0.0)
            vmax = torch.clamp_min(a_dist.mode - 1.0, 0.0)
            viol = torch.sum(torch.square(vmin), dim=-1) + torch.sum(torch.square(vmax), dim=-1)
            abl = (viol * m).sum() / cnt
            actor_loss = actor_loss + self._action_bound_weight * abl
            info["action_bound_loss"] = abl.detach()
        if self._action_entropy_weight != 0:
            ent = (a_dist.entropy() * m).sum() / cnt
            actor_loss = actor_loss - self._action_entropy_weight * ent
            info["action_entropy"] = ent.detach()
        if self._action_reg_weight != 0:
            reg = (a_dist.param_reg() * m).sum() / cnt
            actor_loss = actor_loss + self._action_reg_weight * reg
            info["action_reg_loss"] = reg.detach()
        info["actor_loss"] = actor_loss.detach()
        # "LARGE CRITIC LOSS" guard (ppo_agent.py:225-238): stop the actor gradient when the critic is off, on device
        actor_term = torch.where(critic_loss.detach() > 20.0, actor_loss.detach(), actor_loss)
        loss = actor_term + self._critic_loss_weight * critic_loss
        info["loss"] = loss
        return info

    def _compute_loss_fused(self, batch, norm_obs, norm_a, pred):
        """Same loss through parc_ppo_loss (one pass for value + gradient); None when the policy's log-std depends on the
        state (StdType.VARIABLE), which the kernel does not cover."""
        a_dist = self._model.eval_actor(norm_obs)
        logstd = a_dist.logstd
        if logstd.dim() != 2 or logstd.stride(0) != 0:
            return None
        loss, out = rl_util.ppo_loss(a_dist.mean, logstd[0], pred, norm_a, batch["a_logp"], batch["adv"], batch["rand_action_mask"],
                                     batch["tar_val"], self._ppo_clip_ratio, self._action_bound_weight, self._action_entropy_weight,
                                     self._action_reg_weight, self._critic_loss_weight, 20.0, self._critic_loss_type != "L2")
        # the logged scalars are slots of ONE device vector: _update_model accumulates that vector (one add per minibatch instead of
        # one per scalar) and names the slots at the end
        slots = {"loss": 0, "critic_loss": 1, "actor_loss": 2, "clip_frac": 3, "imp_ratio": 4}
        if self._action_bound_weight != 0:
            slots["action_bound_loss"] = 5
        if self._action_entropy_weight != 0:
            slots["action_entropy"] = 6
        if self._action_reg_weight != 0:
            slots["action_reg_loss"] = 7
        return {"loss": loss, "_packed": out, "_slots": slots}

    def _explicit_update_ok(self):
        return (str(self._device).startswith("cuda") and self._config.get("explicit_backward", True) and self._config.get("fused_ppo_loss", True)
                and self._model.supports_explicit_backward())

    def _minibatch_step_explicit(self, batch, acc):
        """One PPO minibatch (ppo_agent.py:195-204: loss, backward, optimizer step) without an autograd graph: forward GEMMs, the fused
        loss kernel (value + dLoss/d outputs), DMPPOModel.train_backward writing every gradient once into the optimizer's flat buffer,
        clip + SGD.  Same arithmetic as _compute_loss + MPOptimizer.step (tests compare the two)."""
        norm_obs = batch["norm_obs"]
        mean, logstd, pred, saved = self._model.train_forward(norm_obs)
        if self._ppo_cfg is None:
            self._ppo_cfg = rl_util.ppo_cfg(self._ppo_clip_ratio, self._action_bound_weight, self._action_entropy_weight, self._action_reg_weight,
                                            self._critic_loss_weight, 20.0, self._critic_loss_type != "L2")
        if "loss_rec" in batch:
            out, g_mean, g_logstd, g_pred = rl_util.ppo_loss_and_grads_packed(mean, logstd, pred, batch["loss_rec"], self._ppo_cfg)
        else:
            out, g_mean, g_logstd, g_pred = rl_util.ppo_loss_and_grads(mean, logstd, pred, self._a_norm.normalize(batch["action"]), batch["a_logp"],
                                                                        batch["adv"], batch["rand_action_mask"], batch["tar_val"], self._ppo_cfg)
        write = lambda grad_of, done: self._model.train_backward(saved, g_mean, g_logstd, g_pred, grad_of, done)
        if self._clip_grad_norm:
            self._optimizer.step_explicit(write, model=self._model, max_norm=self._max_grad_norm)
        else:
            self._optimizer.step_explicit(write)
        # logged scalars: the packed vector of this minibatch; _update_model stacks an epoch's vectors (one launch per epoch, and the
        # NaN trap reads that stack)
        acc.append(out)

    def _nan_trap(self, loss_rows, batches, epoch):
        """The reference stops at the first minibatch whose critic or actor loss is NaN and dumps that batch (ppo_agent.py:242-252).
        Here the losses of an epoch's minibatches are looked at together - ONE host read per epoch instead of one per minibatch -: at
        the first NaN row the minibatch that produced it is written to output/debug_batch.pkl and training stops; at most the rest
        of that epoch (<= 7 optimizer steps) ran on it."""
        bad = torch.isnan(loss_rows).any(dim=1).tolist()
        if not any(bad):
            return
        k = bad.index(True)
        import pickle
        batch_file = os.path.join("output", "debug_batch.pkl")
        os.makedirs("output", exist_ok=True)
        with open(batch_file, "wb") as f:
            pickle.dump({name: v.detach().cpu() for name, v in batches[k].items()}, f)
        Logger.print("NAN LOSS in minibatch {} of update epoch {}: wrote debug batch file to {}".format(k, epoch, batch_file))
        raise FloatingPointError("NaN loss in minibatch {} of update epoch {} (batch dumped to {}; the reference prints, dumps and exits "
                                 "at this point)".format(k, epoch, batch_file))

    def _update_model(self):
        self.train()
        N = self.get_num_envs()
        num_samples = self._exp_buffer.get_sample_count()
        batch_size = self._batch_size * N
        num_batches = int(np.ceil(float(num_samples) / batch_size))
        acc = dict()
        if self._explicit_update_ok():
            slots = {"loss": 0, "critic_loss": 1, "actor_loss": 2, "clip_frac": 3, "imp_ratio": 4}
            for name, w, i in (("action_bound_loss", self._action_bound_weight, 5), ("action_entropy", self._action_entropy_weight, 6),
                               ("action_reg_loss", self._action_reg_weight, 7)):
                if w != 0:
                    slots[name] = i
            keys = ["norm_obs", "loss_rec"] if self._exp_buffer.has_buffer("loss_rec") else _LOSS_KEYS
            total = None
            for epoch in range(self._update_epochs):
                outs, batches = [], []
                for _ in range(num_batches):
                    batches.append(self._exp_buffer.sample(batch_size, keys=keys))
                    self._minibatch_step_explicit(batches[-1], outs)
                stack = torch.stack(outs)                               # [minibatches, 16]: loss, critic loss, actor loss, ...
                self._nan_trap(stack[:, 0:3], batches, epoch)
                total = stack.sum(dim=0) if total is None else total + stack.sum(dim=0)
                del batches
                self._optimizer.end_epoch()
            packed = total / (self._update_epochs * num_batches)
            return {k: packed[i] for k, i in slots.items()}
        for epoch in range(self._update_epochs):
            losses, batches = [], []
            for _ in range(num_batches):
                batch = self._exp_buffer.sample(batch_size, keys=_LOSS_KEYS)
                info = self._compute_loss(batch)
                losses.append(info["loss"].detach())
                batches.append(batch)
                if self._clip_grad_norm:
                    self._optimizer.step(info["loss"], model=self._model, max_norm=self._max_grad_norm)
                else:
                    self._optimizer.step(info["loss"])
                if "_packed" in info:
                    slots = info["_slots"]
                    v = info["_packed"].detach()
                    acc["_packed"] = acc["_packed"] + v if "_packed" in acc else v.clone()
                else:
                    for k, v in info.items():
                        v = v.detach()
                        acc[k] = acc[k] + v if k in acc else v.clone()
            self._nan_trap(torch.stack(losses).reshape(-1, 1), batches, epoch)
            del batches
            self._optimizer.end_epoch()          # exchange point of the per-epoch cadence (optimizer: grad_allreduce "epoch")
        steps = self._update_epochs * num_batches
        if "_packed" in acc:
            packed = acc.pop("_packed") / steps
            return {k: packed[i] for k, i in slots.items()}
        return {k: v / steps for k, v in acc.items()}

    def _train_iter(self):
        self._exp_buffer.reset()
        self.eval()
        self.set_mode(AgentMode.TRAIN)
        self._rollout_train(self._steps_per_iter)
        data_info = self._build_train_data()
        train_info = self._update_model()
        if self._need_normalizer_update():
            self._obs_norm.update()
        info = {**train_info, **data_info}
        tr = self._train_return_tracker
        info.update(tr.summary())           # one device->host copy for all tracked means
        return info

    def _init_train(self):
        self._iter = 0
        self._sample_count = 0
        self._exp_buffer.clear()
        self._train_return_tracker.reset()
        self._test_return_tracker.reset()

    def _update_sample_count(self):
        return mp_util.reduce_sum(self._exp_buffer.get_total_samples())

    def _build_logger(self, log_file):
        log = Logger()
        log.set_step_key("Samples")
        if mp_util.is_root_proc():
            log.configure_output_file(log_file)
        return log

    def _log_train_info(self, train_info, test_info, start_time):
        L = self._logger
        L.log("Iteration", self._iter, collection="1_Info")
        L.log("Wall_Time", (time.time() - start_time) / 3600.0, collection="1_Info")
        L.log("Samples", self._sample_count, collection="1_Info")
        L.log("Test_Return", test_info["mean_return"], collection="0_Main")
        L.log("Test_Episode_Length", test_info["mean_ep_len"], collection="0_Main", quiet=True)
        L.log("Test_Episodes", mp_util.reduce_sum(test_info["num_eps"]), collection="1_Info", quiet=True)
        L.log("Train_Return", train_info.pop("mean_return"), collection="0_Main")
        L.log("Train_Episode_Length", train_info.pop("mean_ep_len"), collection="0_Main", quiet=True)
        L.log("Train_Episodes", mp_util.reduce_sum(train_info.pop("num_eps")), collection="1_Info", quiet=True)
        for k, v in train_info.items():
            if k == "loss":
                v = v.detach()
            L.log(k.title(), v)
        L.log("Exp_Prob", self._get_exp_prob())
        if self._is_terrain_runner:
            L.log("replan timer", self._env.get_replan_time_buf().item())

    def _output_train_model(self, it, out_model_file, int_output_dir):
        self.save(out_model_file)
        if int_output_dir != "":
            self.save(os.path.join(int_output_dir, "model_{:010d}.pt".format(it)))
            if mp_util.is_root_proc() and self._env.has_dm_envs():
                torch.save(self._env.get_dm_env()._motion_id_fail_rates, os.path.join(int_output_dir, "fail_rates_{:010d}.pt".format(it)))

    def train_model(self, max_samples, out_model_file, int_output_dir, log_file, logger_type=None):
        """Outer loop (dm_ppo_agent.py:230-272)."""
        start_time = time.time()
        self._curr_obs, self._curr_info = self._env.reset()
        self._logger = self._build_logger(log_file)
        self._init_train()
        test_info = {"mean_return": 0.0, "mean_ep_len": 0.0, "num_eps": 0}
        while self._sample_count < max_samples:
            train_info = self._train_iter()
            output_iter = (self._iter % self._iters_per_output == 0)
            if output_iter:
                test_info = self.test_model(self._test_episodes)
                extra = self._env.get_extra_log_info()
                for coll in extra:
                    for k, v in extra[coll].items():
                        self._logger.log(k, v, collection=coll, quiet=True)
                self._env.post_test_update()
            self._sample_count = self._update_sample_count()
            self._log_train_info(train_info, test_info, start_time)
            self._logger.print_log()
            if output_iter:
                self._logger.write_log()
                self._train_return_tracker.reset()
                self.hard_reset_envs()
            if self._iter % self._iters_per_checkpoint == 0:
                self._output_train_model(self._iter, out_model_file, int_output_dir)
            self._iter += 1

    # ------------------------------------------------------------------ motion recording (dm_ppo_agent.py:414-533)
    def record_motions(self, max_steps=None):
        """Deterministic rollout of every env on its own clip (demo mode: clip id = env id mod M) from t=0; clips
        tracked to their end are written to the env's output_motion_dir.  Clips that failed are retried from start
        fractions 0.1 .. 0.5 (shorter than 2 s remaining: skipped), like the reference.  Returns the success flags
        (the reference prints the statistics and calls exit())."""
        self.eval()
        self.set_mode(AgentMode.TEST)
        env = self._env
        env.set_rand_reset(False)
        env.set_demo_mode(True)
        env.set_rand_root_pos_offset_scale(0.0)
        env._episode_length = 1000.0
        N = self.get_num_envs()

        def helper(prev_successful=None):
            self._curr_obs, self._curr_info = env.reset()
            env.build_agent_states_dict("_dm", record_obs=True)
            env.write_agent_states()
            if prev_successful is not None:
                for e in range(N):
                    env.set_writing_env_state(e, not prev_successful[e])
            steps = 0
            while env.is_writing_agent_states() and (max_steps is None or steps < max_steps):
                action, _ = self._decide_action(self._curr_obs, self._curr_info)
                _, _, done, _ = self._env.step(action)
                self._curr_obs, self._curr_info = self._reset_done_envs(done)
                steps += 1

        helper()
        successful = list(env.get_env_success_states())
        counts = [sum(successful)]
        dm = env.get_dm_env()
        clip_len = dm._motion_lib._motion_lengths.cpu().numpy()            # one transfer, not one per env and retry
        M = dm._motion_lib.num_motions()
        for frac in [0.1, 0.2, 0.3, 0.4, 0.5]:
            if all(successful):
                break
            for e in range(N):
                if (1.0 - frac) * float(clip_len[e % M]) < 2.0:
                    successful[e] = True
            dm.set_motion_start_time_fraction(torch.full([N], frac, dtype=torch.float32, device=self._device))
            helper(prev_successful=successful)
            new = list(env.get_env_success_states())
            counts.append(sum(new))
            successful = [a or b for a, b in zip(successful, new)]
        Logger.print("Successful motions at 0 percent start time: {} / {}".format(counts[0], N))
        Logger.print("Total successful motions: {}".format(sum(counts)))
        return successful


# the reference's class hierarchy base_agent.BaseAgent -> ppo_agent.PPOAgent -> dm_ppo_agent.DMPPOAgent is one class here;
# parc_amd.install_reference_aliases() maps the three module names to this module
BaseAgent = PPOAgent = DMPPOAgent
