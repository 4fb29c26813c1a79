"""Running mean/std normaliser with a non-normalised index set (mirror of the reference's learning/normalizer.py:7-124;
``_count/_mean/_std`` are Parameters so they live in the state_dict under the same keys)."""
import numpy as np
import torch

from ..util import mp_util


class Normalizer(torch.nn.Module):
    def __init__(self, shape, device, init_mean=None, init_std=None, min_std=1e-4, clip=np.inf, dtype=torch.float, non_norm_indices=None):
        super().__init__()
        self._min_var = min_std * min_std
        self._clip = clip
        self.dtype = dtype
        self._non_norm_indices = non_norm_indices
        self._count = torch.nn.Parameter(torch.zeros([1], device=device, dtype=torch.long), requires_grad=False)
        self._mean = torch.nn.Parameter(torch.zeros(shape, device=device, dtype=dtype), requires_grad=False)
        self._std = torch.nn.Parameter(torch.ones(shape, device=device, dtype=dtype), requires_grad=False)
        if init_mean is not None:
            self._mean[:] = init_mean
        if init_std is not None:
            self._std[:] = init_std
        self._mean_sq = None
        self._new_count = 0
        self._new_sum = torch.zeros_like(self._mean)
        self._new_sum_sq = torch.zeros_like(self._mean)

    def record(self, x):
        shape = self._mean.shape
        assert len(x.shape) > len(shape)
        x = x.flatten(start_dim=0, end_dim=len(x.shape) - len(shape) - 1)
        self._new_count += x.shape[0]
        self._new_sum += torch.sum(x, dim=0)
        self._new_sum_sq += torch.sum(torch.square(x), dim=0)

    def update(self):
        if self._mean_sq is None:
            self._mean_sq = (torch.square(self._std) + torch.square(self._mean)).type(self.dtype)
        self._new_count = mp_util.reduce_sum(self._new_count)
        mp_util.reduce_inplace_sum(self._new_sum)
        mp_util.reduce_inplace_sum(self._new_sum_sq)
        new_count = self._new_count
        new_mean = self._new_sum / new_count
        new_mean_sq = self._new_sum_sq / new_count
        new_total = self._count + new_count
        w_old = self._count.type(torch.float) / new_total.type(torch.float)
        w_new = float(new_count) / new_total.type(torch.float)
        self._mean[:] = w_old * self._mean + w_new * new_mean
        self._mean_sq[:] = w_old * self._mean_sq + w_new * new_mean_sq
        self._count[:] = new_total
        var = torch.clamp_min(self._mean_sq - torch.square(self._mean), self._min_var)
        self._std[:] = torch.sqrt(var).type(self.dtype)
        self._new_count = 0
        self._new_sum[:] = 0
        self._new_sum_sq[:] = 0
        if self._non_norm_indices is not None:
            self._mean[self._non_norm_indices] = 0.0
            self._std[self._non_norm_indices] = 1.0

    def get_shape(self):
        return self._mean.shape

    def get_count(self):
        return self._count

    def get_mean(self):
        return self._mean

    def get_std(self):
        return self._std

    def normalize(self, x, out=None):
        if (x.is_cuda and x.dtype == torch.float32 and self.dtype == torch.float32 and x.is_contiguous() and self._mean.dim() == 1
                and self._mean.shape[0] % 4 == 0 and x.shape[-1] == self._mean.shape[0] and x.data_ptr() % 16 == 0 and np.isfinite(self._clip)):
            # one pass (parc_normalize_clamp), same fp32 operations as the expression below
            from .. import _hip
            if out is None:
                out = torch.empty_like(x)
            _hip.check(_hip.lib().parc_normalize_clamp(_hip.stream(), x.numel() // x.shape[-1], int(x.shape[-1]), _hip.ptr(x), _hip.ptr(self._mean),
                                                       _hip.ptr(self._std), float(self._clip), _hip.ptr(out)), "parc_normalize_clamp")
            return out
        res = torch.clamp((x - self._mean) / self._std, -self._clip, self._clip).type(self.dtype)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def unnormalize(self, norm_x):
        return (norm_x * self._std + self._mean).type(self.dtype)
