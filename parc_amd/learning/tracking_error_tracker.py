"""Episode-averaged tracking errors at test time (mirror of the reference's learning/tracking_error_tracker.py): the seven
columns of compute_tracking_error (mgdm_dm_util.py:578-611) are accumulated per env, divided by the episode length when
the episode ends and folded into running means weighted by episode count.  Device-side masked arithmetic, no host sync."""
import torch

from ..envs import base_env

NAMES = ["root_pos", "root_rot", "body_pos", "body_rot", "dof_vel", "root_vel", "root_ang_vel"]


class TrackingErrorTracker:
    def __init__(self, num_envs, device):
        self._device = device
        self._episodes = torch.zeros([1], device=device, dtype=torch.float64)
        self._mean = torch.zeros([7], device=device, dtype=torch.float32)
        self._buf = torch.zeros([num_envs, 7], device=device, dtype=torch.float32)
        self._ep_len_buf = torch.zeros([num_envs], device=device, dtype=torch.long)

    def reset(self):
        self._episodes.zero_()
        self._mean.zero_()
        self._buf.zero_()
        self._ep_len_buf.zero_()

    def update(self, tracking_error, done):
        assert tracking_error.shape == self._buf.shape and done.shape[0] == self._buf.shape[0]
        self._buf += tracking_error
        self._ep_len_buf += 1
        mask = done != base_env.DoneFlags.NULL.value
        maskf = mask.to(torch.float32)
        n_new = maskf.sum().to(torch.float64)
        ep_mean = self._buf / self._ep_len_buf.clamp_min(1).to(torch.float32).unsqueeze(-1)
        new_mean = (ep_mean * maskf.unsqueeze(-1)).sum(dim=0) / maskf.sum().clamp_min(1.0)
        total = self._episodes + n_new
        w_new = torch.where(total > 0, n_new / total.clamp_min(1.0), torch.zeros_like(total)).to(torch.float32)
        self._mean = torch.where(n_new > 0, w_new * new_mean + (1.0 - w_new) * self._mean, self._mean)
        self._episodes = total
        self._buf *= (1.0 - maskf).unsqueeze(-1)
        self._ep_len_buf *= (~mask).to(torch.long)

    def _get(self, i):
        return self._mean[i:i + 1]

    def get_mean_root_pos_err(self):
        return self._get(0)

    def get_mean_root_rot_err(self):
        return self._get(1)

    def get_mean_body_pos_err(self):
        return self._get(2)

    def get_mean_body_rot_err(self):
        return self._get(3)

    def get_mean_dof_vel_err(self):
        return self._get(4)

    def get_mean_root_vel_err(self):
        return self._get(5)

    def get_mean_root_ang_vel_err(self):
        return self._get(6)
