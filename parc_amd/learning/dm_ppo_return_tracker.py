"""Per-term episodic return tracking (mirror of the reference's learning/dm_ppo_return_tracker.py:6-99).
get_mean_return() of "total_r" is the "mean episode return" of the north-star metric.  All terms live in ONE [K, N]
device tensor and the running means in one [K] tensor, updated with masked arithmetic: a dozen launches per env step
instead of six per term, and no per-step nonzero()/host sync."""
import torch

from ..envs import base_env

_KEYS = ["total_r", "pose_r", "vel_r", "root_pos_r", "root_vel_r", "key_pos_r", "contact_penalty"]
_TASK_KEYS = ["task_r1", "task_r2", "total_task_r"]


class DMPPOReturnTracker:
    def __init__(self, num_envs, device, target_task=False):
        self._device = device
        self._keys = _KEYS + (_TASK_KEYS if target_task else [])
        K = len(self._keys)
        self._episodes_t = torch.zeros([1], device=device, dtype=torch.float64)
        self._mean_ep_len = torch.zeros([1], device=device, dtype=torch.float32)
        self._ep_len_buf = torch.zeros([num_envs], device=device, dtype=torch.long)
        self._eps_per_env_buf = torch.zeros([num_envs], device=device, dtype=torch.long)
        self._return_buf = torch.zeros([K, num_envs], device=device, dtype=torch.float32)
        self._mean_return = torch.zeros([K], device=device, dtype=torch.float32)
        self._use_kernel = True
        self._workspace = None

    def get_mean_return(self):
        return self._mean_return[0:1]

    def get_specific_mean_return(self, key):
        i = self._keys.index(key)
        return self._mean_return[i:i + 1]

    def get_mean_ep_len(self):
        return self._mean_ep_len

    def get_episodes(self):
        return int(self._episodes_t.item())

    def get_eps_per_env(self):
        return self._eps_per_env_buf

    def summary(self):
        """{"mean_return", "mean_ep_len", "num_eps", <every tracked term>: mean} with a single host transfer."""
        v = torch.cat([self._mean_return.to(torch.float64), self._mean_ep_len.to(torch.float64), self._episodes_t]).tolist()
        K = len(self._keys)
        out = {"mean_return": v[0], "mean_ep_len": v[K], "num_eps": int(v[K + 1])}
        for i, k in enumerate(self._keys):
            out[k] = v[i]
        return out

    def reset(self):
        self._episodes_t.zero_()
        self._eps_per_env_buf.zero_()
        self._mean_ep_len.zero_()
        self._ep_len_buf.zero_()
        self._return_buf.zero_()
        self._mean_return.zero_()

    def _stack_rewards(self, info):
        K = len(self._keys)
        if "rewards_all" in info:
            names, block = info["rewards_all"]
            if list(names[:K]) == self._keys:
                return block[:K]
        rewards = info["rewards"]
        for k in self._keys:
            assert k in rewards, k
        return torch.stack([rewards[k] for k in self._keys], dim=0)

    def update(self, info, done):
        block = self._stack_rewards(info)
        if block.is_cuda and block.stride(-1) == 1 and done.dtype == torch.int32 and self._use_kernel:
            # K21 in one launch (parc_return_tracker_update); the torch expression below is the same rule
            from .. import _hip
            p = _hip.ptr
            if self._workspace is None:
                n_ws = int(_hip.lib().parc_return_tracker_workspace_floats(int(done.shape[0])))
                self._workspace = torch.zeros(n_ws, dtype=torch.float32, device=block.device)
            _hip.check(_hip.lib().parc_return_tracker_update(_hip.stream(), int(done.shape[0]), len(self._keys), p(block), int(block.stride(0)),
                                                             p(done), p(self._return_buf), p(self._ep_len_buf), p(self._eps_per_env_buf),
                                                             p(self._mean_return), p(self._mean_ep_len), p(self._episodes_t), p(self._workspace)),
                       "parc_return_tracker_update")
            return
        self._return_buf += block
        self._ep_len_buf += 1
        mask = done != base_env.DoneFlags.NULL.value
        maskf = mask.to(torch.float32)
        n_newf = maskf.sum()
        n_new = n_newf.to(torch.float64)                       # device scalar, may be 0
        new_count = self._episodes_t + n_new
        w_new = torch.where(new_count > 0, n_new / new_count.clamp_min(1.0), torch.zeros_like(new_count)).to(torch.float32)
        denom = n_newf.clamp_min(1.0)
        any_new = n_newf > 0
        new_len = torch.dot(self._ep_len_buf.to(torch.float32), maskf) / denom
        # (in-place: the tensors keep their addresses, which a captured rollout graph relies on)
        self._mean_ep_len.copy_(torch.where(any_new, torch.lerp(self._mean_ep_len, new_len, w_new), self._mean_ep_len))
        new_mean = torch.mv(self._return_buf, maskf) / denom    # [K] mean return of the episodes that ended this step
        self._mean_return.copy_(torch.where(any_new, torch.lerp(self._mean_return, new_mean, w_new), self._mean_return))
        keep = ~mask
        self._return_buf *= keep
        self._episodes_t.copy_(new_count)
        self._ep_len_buf *= keep
        self._eps_per_env_buf += mask
