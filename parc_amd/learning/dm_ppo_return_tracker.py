"""Per-term episodic return tracking (mirror of the reference's learning/dm_ppo_return_tracker.py:6-99).
get_mean_return() of "total_r" is the "mean episode return" of the north-star metric.  The running means are kept
on the device and updated with masked arithmetic: no per-step nonzero()/host sync."""
import torch

from ..envs import base_env

_KEYS = ["total_r", "pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty"]
_TASK_KEYS = ["task_r1", "task_r2", "total_task_r"]


class DMPPOReturnTracker:
    def __init__(self, num_envs, device, target_task=False):
        self._device = device
        keys = _KEYS + (_TASK_KEYS if target_task else [])
        self._episodes_t = torch.zeros([1], device=device, dtype=torch.float64)
        self._mean_ep_len = torch.zeros([1], device=device, dtype=torch.float32)
        self._ep_len_buf = torch.zeros([num_envs], device=device, dtype=torch.long)
        self._eps_per_env_buf = torch.zeros([num_envs], device=device, dtype=torch.long)
        self._return_bufs = {k: torch.zeros([num_envs], device=device, dtype=torch.float32) for k in keys}
        self._mean_returns = {k: torch.zeros([1], device=device, dtype=torch.float32) for k in keys}

    def get_mean_return(self):
        return self._mean_returns["total_r"]

    def get_specific_mean_return(self, key):
        return self._mean_returns[key]

    def get_mean_ep_len(self):
        return self._mean_ep_len

    def get_episodes(self):
        return int(self._episodes_t.item())

    def get_eps_per_env(self):
        return self._eps_per_env_buf

    def reset(self):
        self._episodes_t.zero_()
        self._eps_per_env_buf.zero_()
        self._mean_ep_len.zero_()
        self._ep_len_buf.zero_()
        for k in self._return_bufs:
            self._return_bufs[k].zero_()
            self._mean_returns[k].zero_()

    def update(self, info, done):
        rewards = info["rewards"]
        for k in self._return_bufs:
            assert k in rewards, k
            self._return_bufs[k] += rewards[k]
        self._ep_len_buf += 1
        mask = done != base_env.DoneFlags.NULL.value
        maskf = mask.to(torch.float32)
        n_new = maskf.sum().to(torch.float64)                      # device scalar, may be 0
        new_count = self._episodes_t + n_new
        w_new = torch.where(new_count > 0, n_new / new_count.clamp_min(1.0), torch.zeros_like(new_count)).to(torch.float32)
        w_old = 1.0 - w_new
        denom = maskf.sum().clamp_min(1.0)
        new_len = (self._ep_len_buf.to(torch.float32) * maskf).sum() / denom
        self._mean_ep_len = torch.where(n_new > 0, w_new * new_len + w_old * self._mean_ep_len, self._mean_ep_len)
        for k in self._return_bufs:
            new_mean = (self._return_bufs[k] * maskf).sum() / denom
            self._mean_returns[k] = torch.where(n_new > 0, w_new * new_mean + w_old * self._mean_returns[k], self._mean_returns[k])
            self._return_bufs[k] *= (1.0 - maskf)
        self._episodes_t = new_count
        self._ep_len_buf *= (~mask).to(torch.long)
        self._eps_per_env_buf += mask.to(torch.long)
