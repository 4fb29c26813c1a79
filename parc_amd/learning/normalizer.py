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
        # statistics gathered since the last update(): row 0 = sum x, row 1 = sum x^2 (one buffer, so a record() is two
        # reductions into adjacent rows and update() exchanges it between ranks in ONE all-reduce together with the count)
        self._new_count = 0
        self._acc = torch.zeros((2,) + tuple(self._mean.shape), device=device, dtype=dtype)
        self._scratch = None

    @property
    def _new_sum(self):
        return self._acc[0]

    @property
    def _new_sum_sq(self):
        return self._acc[1]

    def record(self, x):
        lead = x.dim() - self._mean.dim()
        assert lead > 0
        rows = x.reshape((-1,) + tuple(self._mean.shape))
        self._new_count += rows.shape[0]           # (host-side int: the graph rollout accounts for it itself)
        if (rows.is_cuda and rows.dtype == torch.float32 and self.dtype == torch.float32 and rows.is_contiguous() and self._mean.dim() == 1
                and self._mean.shape[0] % 4 == 0 and rows.data_ptr() % 16 == 0):
            # one pass, fixed summation order (parc_moments_accumulate)
            from .. import _hip
            L = _hip.lib()
            need = int(L.parc_moments_workspace_floats(rows.shape[0], rows.shape[1]))
            if self._scratch is None or self._scratch.numel() < need:
                self._scratch = torch.empty(need, dtype=torch.float32, device=rows.device)
            _hip.check(L.parc_moments_accumulate(_hip.stream(), rows.shape[0], rows.shape[1], _hip.ptr(rows), _hip.ptr(self._acc),
                                                 _hip.ptr(self._scratch)), "parc_moments_accumulate")
            return
        self._acc[0] += rows.sum(dim=0)
        self._acc[1] += (rows * rows).sum(dim=0)

    def update(self):
        """Fold the recorded batch into the running moments (count-weighted average of E[x] and E[x^2], learning/normalizer.py:36-62
        of the reference); with several ranks the batch statistics are summed over all of them first."""
        if self._mean_sq is None:
            self._mean_sq = (self._std * self._std + self._mean * self._mean).type(self.dtype)
        n_new = self._new_count
        if mp_util.enable_mp():
            # one exchange: [sum | sum of squares | count] in float64 (exact for the count, no loss for the fp32 sums)
            packed = torch.cat([self._acc.reshape(-1).double(), torch.tensor([float(n_new)], dtype=torch.float64, device=self._acc.device)])
            mp_util.reduce_inplace_sum(packed)
            self._acc.copy_(packed[:-1].reshape(self._acc.shape))
            n_new = int(round(packed[-1].item()))
        n_old = self._count
        n_tot = n_old + n_new
        keep = n_old.type(torch.float) / n_tot.type(torch.float)
        take = float(n_new) / n_tot.type(torch.float)
        batch = self._acc / n_new                       # row 0 = batch mean, row 1 = batch mean of squares
        self._mean[:] = keep * self._mean + take * batch[0]
        self._mean_sq[:] = keep * self._mean_sq + take * batch[1]
        self._count[:] = n_tot
        self._std[:] = (self._mean_sq - self._mean * self._mean).clamp_min(self._min_var).sqrt().type(self.dtype)
        if self._non_norm_indices is not None:          # columns that must pass through unchanged
            self._mean[self._non_norm_indices] = 0.0
            self._std[self._non_norm_indices] = 1.0
        self._new_count = 0
        self._acc.zero_()

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

    def can_ingest(self, x):
        return (x.is_cuda and x.dtype == torch.float32 and self.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2 and self._mean.dim() == 1
                and self._mean.shape[0] % 4 == 0 and x.shape[-1] == self._mean.shape[0] and x.data_ptr() % 16 == 0 and np.isfinite(self._clip))

    def ingest(self, x, record=False, copy_into=None, out=None):
        """normalize(x) - and, in the SAME pass over the rows (parc_obs_ingest), record(x) when `record` and a raw copy of the rows into
        time row `head` of a [T, N, D] buffer when copy_into = (buffer, head device int64): what a rollout step does with its
        observations in three launches otherwise.  Results equal the three separate calls bit for bit."""
        from .. import _hip
        assert self.can_ingest(x)
        L = _hip.lib()
        if out is None:
            out = torch.empty_like(x)
        ws = None
        if record:
            self._new_count += x.shape[0]
            need = int(L.parc_moments_workspace_floats(x.shape[0], x.shape[1]))
            if self._scratch is None or self._scratch.numel() < need:
                self._scratch = torch.empty(need, dtype=torch.float32, device=x.device)
            ws = self._scratch
        dst, row = (None, None) if copy_into is None else copy_into
        if dst is not None:
            assert dst.is_contiguous() and dst.dtype == torch.float32 and tuple(dst.shape[1:]) == tuple(x.shape) and dst.data_ptr() % 16 == 0
        _hip.check(L.parc_obs_ingest(_hip.stream(), x.shape[0], x.shape[1], _hip.ptr(x), _hip.ptr(self._mean), _hip.ptr(self._std), float(self._clip),
                                     _hip.ptr(out), _hip.ptr(dst), _hip.ptr(row), _hip.ptr(self._acc) if record else None, _hip.ptr(ws)),
                   "parc_obs_ingest")
        return out

    def unnormalize(self, norm_x):
        return (norm_x * self._std + self._mean).type(self.dtype)
