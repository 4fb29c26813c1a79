"""Time-major rollout storage [T, N, ...] (mirror of the reference's learning/experience_buffer.py:3-115: same buffer
names, same flat [T*N, ...] view, same randperm-walking sampler).  ``sample`` takes an optional ``keys`` list so the
update gathers only the six buffers the PPO loss reads instead of all seventeen (11.9 KB/sample in the reference)."""
import torch


class ExperienceBuffer:
    def __init__(self, buffer_length, batch_size, device):
        self._buffer_length = buffer_length
        self._batch_size = batch_size
        self._device = device
        self._buffer_head = 0
        self._total_samples = 0
        self._buffers = dict()
        self._flat_buffers = dict()
        self._sample_buf = torch.randperm(buffer_length * batch_size, device=device, dtype=torch.long)
        self._sample_buf_head = 0
        self._device_head = None

    def has_buffer(self, name):
        return name in self._buffers

    def add_buffer(self, name, buffer):
        assert len(buffer.shape) >= 2 and buffer.shape[0] == self._buffer_length and buffer.shape[1] == self._batch_size
        assert name not in self._buffers
        self._buffers[name] = buffer
        self._flat_buffers[name] = buffer.view([buffer.shape[0] * buffer.shape[1]] + list(buffer.shape[2:]))

    def reset(self):
        self._buffer_head = 0
        self._reset_sample_buf()

    def clear(self):
        self.reset()
        self._total_samples = 0

    def inc(self):
        self._buffer_head = (self._buffer_head + 1) % self._buffer_length
        self._total_samples += self._batch_size

    def get_total_samples(self):
        return self._total_samples

    def get_sample_count(self):
        return min(self._total_samples, self._buffer_length * self._batch_size)

    def record(self, name, data):
        assert data.shape[0] == self._batch_size
        if self._device_head is not None:
            # hipGraph rollout: the write position is a device scalar, so one captured graph serves every step
            buf = self._buffers[name]
            buf.index_copy_(0, self._device_head, data.to(buf.dtype).unsqueeze(0))
        else:
            self._buffers[name][self._buffer_head] = data

    def record_group(self, items):
        """record() for several buffers in ONE launch (K15, parc_record_step): items = [(name, tensor [N, ...]), ...].
        GPU only, with a device head (the captured rollout step); falls back to record() otherwise."""
        if self._device_head is None or not items[0][1].is_cuda:
            for name, data in items:
                self.record(name, data)
            return
        from .. import _hip
        arr = (_hip.RecordFieldS * len(items))()
        keep = []
        for i, (name, data) in enumerate(items):
            buf = self._buffers[name]
            if data.numel() == 1 and self._batch_size > 1 and buf[0, 0].numel() == 1 and data.dtype == buf.dtype and data.element_size() == 4:
                keep.append(data)              # one value for every env of the row: broadcast inside the launch
                arr[i] = _hip.RecordFieldS(data.data_ptr(), buf.data_ptr(), 4, 2)
                continue
            assert data.shape[0] == self._batch_size
            conv = 1 if (data.dtype == torch.int64 and buf.dtype == torch.int32) else 0
            if not conv and data.dtype != buf.dtype:
                data = data.to(buf.dtype)
            data = data.contiguous()
            keep.append(data)
            row = data[0].numel() * data.element_size()
            assert row % 4 == 0 and (conv == 1 or row == buf[0, 0].numel() * buf.element_size())
            arr[i] = _hip.RecordFieldS(data.data_ptr(), buf.data_ptr(), row, conv)
        _hip.check(_hip.lib().parc_record_step(_hip.stream(), self._batch_size, _hip.ptr(self._device_head), len(items), arr),
                   "parc_record_step")

    def set_device_head(self, head_t):
        """head_t: int64 device tensor [1] mirroring ``_buffer_head`` (None switches back to host indexing)."""
        self._device_head = head_t

    def get_data(self, name):
        return self._buffers[name]

    def get_data_flat(self, name):
        return self._flat_buffers[name]

    def set_data(self, name, data):
        buf = self._buffers[name]
        assert buf.shape[0] == data.shape[0] and buf.shape[1] == data.shape[1]
        buf[:] = data

    def set_data_flat(self, name, data):
        self._flat_buffers[name][:] = data

    def sample(self, n, keys=None):
        idx = self._sample_rand_idx(n)
        names = self._flat_buffers.keys() if keys is None else keys
        return {k: self._flat_buffers[k][idx] for k in names}

    def _reset_sample_buf(self):
        self._sample_buf[:] = torch.randperm(self._buffer_length * self._batch_size, device=self._device, dtype=torch.long)
        self._sample_buf_head = 0

    def _sample_rand_idx(self, n):
        L = self._sample_buf.shape[0]
        assert n <= L
        if self._sample_buf_head + n <= L:
            idx = self._sample_buf[self._sample_buf_head:self._sample_buf_head + n]
            self._sample_buf_head += n
        else:
            # NOT cloned, like the reference (experience_buffer.py:102-110): the tail is a view, the new permutation is written
            # in place, so a wrapping call returns [new[head:], new[:rem]] - indices of ONE permutation, no duplicates (fixture G25)
            head = self._sample_buf[self._sample_buf_head:]
            rem = n - (L - self._sample_buf_head)
            self._reset_sample_buf()
            idx = torch.cat([head, self._sample_buf[:rem]], dim=0)
            self._sample_buf_head = rem
        count = self.get_sample_count()
        if count == L:                      # full buffer (every training iteration): the permutation already indexes it
            return idx
        return torch.remainder(idx, count)
