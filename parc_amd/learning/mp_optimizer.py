"""Data-parallel optimizer step: SGD/AdamW on one process per GPU with ONE RCCL all-reduce per step.

Mirror of the reference's learning/mp_optimizer.py:5-90 (MPOptimizer.step/sync/_check_synced).  Difference in
mechanics, not semantics: every parameter's ``.grad`` is a view into one persistent flat fp32 buffer, so the
all-reduce (mean over ranks) runs in place on that buffer with no gather/scatter copies (the reference packs with
parameters_to_vector and unpacks again each step)."""
import torch

from ..util import mp_util


class MPOptimizer:
    CHECK_SYNC_STEPS = 1000

    def __init__(self, config, param_list):
        self._param_list = param_list
        self._steps = 0
        # "minibatch": all-reduce the gradient on every step (the reference's cadence, default).
        # "epoch": the north-star's cadence - one exchange per PPO epoch: ranks step locally and end_epoch() averages the
        # parameters and the optimizer's moment buffers in one flat all-reduce (local SGD with periodic averaging).
        self._cadence = config.get("grad_allreduce", "minibatch")
        assert self._cadence in ("minibatch", "epoch")
        lr = float(config["learning_rate"])
        wd = float(config.get("weight_decay", 0.0))
        # one multi-tensor launch per step where torch offers it (device parameters): same update rule, 1 kernel instead of 3
        fused = {"fused": True} if (param_list[0].is_cuda and config.get("fused_optimizer", True)) else {}
        will_be_flat = (config["type"] == "SGD" and param_list[0].is_cuda and param_list[0].dtype == torch.float32 and bool(config.get("flat_sgd", True)))
        if will_be_flat:
            self._optimizer = None            # the flat SGD step below replaces it: no torch optimizer object that is never stepped
        elif config["type"] == "SGD":
            self._optimizer = torch.optim.SGD(param_list, lr, momentum=0.9, weight_decay=wd, **fused)
        elif config["type"] == "Adam":
            self._optimizer = torch.optim.AdamW(param_list, lr, weight_decay=wd, **fused)
        else:
            raise AssertionError("Unsupported optimizer type: " + config["type"])
        n = sum(p.numel() for p in param_list)
        self._flat_grad = torch.zeros(n, dtype=param_list[0].dtype, device=param_list[0].device)
        off = 0
        for p in param_list:
            p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        # SGD on the GPU: parameters and momentum live in flat buffers too (every parameter is a view, like its gradient), and clip +
        # momentum + update are two passes over them (parc_sgd_momentum_step) instead of norm, scale, multi-tensor SGD and a fill
        self._flat_sgd = (config["type"] == "SGD" and param_list[0].is_cuda and param_list[0].dtype == torch.float32
                          and bool(config.get("flat_sgd", True)))
        if self._flat_sgd:
            from .. import _hip
            self._flat_param = torch.empty(n, dtype=torch.float32, device=param_list[0].device)
            off = 0
            with torch.no_grad():
                for p in param_list:
                    view = self._flat_param[off:off + p.numel()].view_as(p)
                    view.copy_(p)
                    p.data = view
                    off += p.numel()
            self._flat_mom = torch.zeros_like(self._flat_param)
            self._sgd_ws = torch.empty(int(_hip.lib().parc_sgd_workspace_floats()), dtype=torch.float32, device=self._flat_param.device)
            self._grad_norm = torch.zeros(1, dtype=torch.float32, device=self._flat_param.device)
            self._lr, self._momentum, self._wd = lr, 0.9, wd
        self.sync()
        # Overlap of the per-minibatch exchange with backward: the flat gradient is cut into a few contiguous buckets at
        # parameter boundaries; a bucket's all-reduce starts (asynchronously, on RCCL's stream) as soon as the last of its
        # gradients has been accumulated, while autograd is still producing the others.  Parameters are registered actor
        # first, critic second and backward visits each network last-layer-first, so with 4 buckets only the first critic
        # layer's 10.7 MB are exchanged after backward ends instead of all 42.6 MB.
        self._overlap = bool(config.get("overlap_allreduce", True)) and self._cadence == "minibatch" and \
            hasattr(param_list[0], "register_post_accumulate_grad_hook")
        self._buckets = []
        self._in_backward = False
        if self._overlap:
            self._build_buckets(int(config.get("allreduce_buckets", 4)))

    def _build_buckets(self, n_buckets):
        total = self._flat_grad.numel()
        cap = total / float(max(n_buckets, 1))
        off, start, count = 0, 0, 0
        for i, p in enumerate(self._param_list):
            off += p.numel()
            count += 1
            last = i == len(self._param_list) - 1
            # close after a bias (1-D parameter) so a layer's weight and bias, whose gradients arrive together, share a bucket
            if last or (off - start >= cap * 0.95 and p.dim() == 1):
                self._buckets.append({"lo": start, "hi": off, "n": count, "pending": 0, "work": None})
                start, count = off, 0
        b = 0
        seen = 0
        for p in self._param_list:
            bucket = self._buckets[b]
            p.register_post_accumulate_grad_hook(lambda _p, bucket=bucket: self._grad_ready(bucket))
            seen += 1
            if seen == bucket["n"]:
                b, seen = b + 1, 0

    def _grad_ready(self, bucket):
        if not self._in_backward:
            return
        bucket["pending"] -= 1
        if bucket["pending"] == 0:
            bucket["work"] = torch.distributed.all_reduce(self._flat_grad[bucket["lo"]:bucket["hi"]], op=torch.distributed.ReduceOp.SUM,
                                                          async_op=True)

    def _backward_and_exchange(self, loss):
        """loss.backward() followed by (or, with buckets, overlapped with) the all-reduce(mean) of the flat gradient."""
        mp = mp_util.enable_mp() and self._cadence == "minibatch"
        if mp and self._overlap:
            for bk in self._buckets:
                bk["pending"], bk["work"] = bk["n"], None
            self._in_backward = True
            try:
                loss.backward()
            finally:
                self._in_backward = False
            for bk in self._buckets:
                if bk["work"] is None:       # a parameter of this bucket received no gradient in this graph: exchange it now
                    torch.distributed.all_reduce(self._flat_grad[bk["lo"]:bk["hi"]], op=torch.distributed.ReduceOp.SUM)
                else:
                    bk["work"].wait()
            self._flat_grad /= mp_util.get_num_procs()
            return
        loss.backward()
        if mp:
            torch.distributed.all_reduce(self._flat_grad, op=torch.distributed.ReduceOp.SUM)
            self._flat_grad /= mp_util.get_num_procs()

    def step_explicit(self, write_grads, **kwargs):
        """The same step when the caller computes the gradients itself: write_grads(grad_of, done) must OVERWRITE the gradient of
        every parameter (grad_of(p) = p's slice of the flat gradient) and call done(p) when p's gradient is complete, in the order a
        backward pass would (the bucketed exchange is started from it exactly like from autograd's hooks).  Nothing is zeroed or
        accumulated: a step costs the gradient kernels themselves."""
        mp = mp_util.enable_mp() and self._cadence == "minibatch"
        written = set()

        def grad_of(p):
            written.add(id(p))
            return p.grad
        if mp and self._overlap:
            if not hasattr(self, "_bucket_of"):
                self._bucket_of, b, seen = dict(), 0, 0
                for p in self._param_list:
                    self._bucket_of[id(p)] = self._buckets[b]
                    seen += 1
                    if seen == self._buckets[b]["n"]:
                        b, seen = b + 1, 0
            for bk in self._buckets:
                bk["pending"], bk["work"] = bk["n"], None
            self._in_backward = True
            try:
                write_grads(grad_of, lambda p: self._grad_ready(self._bucket_of[id(p)]))
            finally:
                self._in_backward = False
            for bk in self._buckets:
                if bk["work"] is None:
                    torch.distributed.all_reduce(self._flat_grad[bk["lo"]:bk["hi"]], op=torch.distributed.ReduceOp.SUM)
                else:
                    bk["work"].wait()
            self._flat_grad /= mp_util.get_num_procs()
        else:
            write_grads(grad_of, lambda p: None)
            if mp:
                torch.distributed.all_reduce(self._flat_grad, op=torch.distributed.ReduceOp.SUM)
                self._flat_grad /= mp_util.get_num_procs()
        if self._steps == 0:
            missing = [i for i, p in enumerate(self._param_list) if id(p) not in written]
            assert not missing, "explicit backward left parameters {} without a gradient".format(missing)
        self._finish_step(**kwargs)

    def step(self, loss, **kwargs):
        self._flat_grad.zero_()
        self._backward_and_exchange(loss)
        self._finish_step(**kwargs)

    def reset_state(self):
        """Forget the optimizer's moments (tests restart from saved weights)."""
        if self._flat_sgd:
            self._flat_mom.zero_()
        else:
            self._optimizer.state.clear()

    CHECK_ALIAS_STEPS = 200

    def _check_aliasing(self):
        """Every parameter and its gradient must still BE the slice of the flat buffers they were bound to at construction: the flat
        SGD step (and the in-place gradient exchange) update the flat buffers only, so anything that rebinds `p.data` / `p.grad`
        afterwards - load_state_dict(assign=True), module.to() / .float() onto another device or dtype, zero_grad(set_to_none=True) -
        would leave the model reading tensors that training no longer updates, silently.  Pointer compares only: no device work."""
        off = 0
        for p in self._param_list:
            nb = p.element_size()
            if self._flat_sgd and p.data_ptr() != self._flat_param.data_ptr() + off * nb:
                raise RuntimeError("a parameter of shape {} no longer aliases the optimizer's flat parameter buffer (something rebound "
                                   "p.data after the optimizer was built): rebuild the MPOptimizer".format(tuple(p.shape)))
            if p.grad is None or p.grad.data_ptr() != self._flat_grad.data_ptr() + off * nb:
                raise RuntimeError("the gradient of a parameter of shape {} no longer aliases the flat gradient buffer (e.g. "
                                   "zero_grad(set_to_none=True) was called on the model): rebuild the MPOptimizer".format(tuple(p.shape)))
            off += p.numel()

    def _finish_step(self, **kwargs):
        if self._steps % self.CHECK_ALIAS_STEPS == 0:
            self._check_aliasing()
        if self._flat_sgd:
            from .. import _hip
            max_norm = float(kwargs["max_norm"]) if "model" in kwargs else -1.0
            _hip.check(_hip.lib().parc_sgd_momentum_step(_hip.stream(), self._flat_param.numel(), _hip.ptr(self._flat_param), _hip.ptr(self._flat_grad),
                                                         _hip.ptr(self._flat_mom), max_norm, float(self._lr), float(self._momentum), float(self._wd),
                                                         _hip.ptr(self._sgd_ws), _hip.ptr(self._grad_norm)), "parc_sgd_momentum_step")
            if mp_util.enable_mp() and self._cadence == "minibatch" and self._steps % self.CHECK_SYNC_STEPS == 0:
                assert self._check_synced(), "Network parameters desynchronized"
            self._steps += 1
            return
        if "model" in kwargs:
            # the gradient norm of the flat buffer == norm over model parameters (all trainable params are in it)
            max_norm = kwargs["max_norm"]
            norm = torch.linalg.vector_norm(self._flat_grad)
            if self._flat_grad.is_cuda and self._flat_grad.dtype == torch.float32:
                from .. import _hip
                _hip.check(_hip.lib().parc_scale_by_clipped_norm(_hip.stream(), self._flat_grad.numel(), _hip.ptr(self._flat_grad),
                                                                 _hip.ptr(norm.reshape(1)), float(max_norm)), "parc_scale_by_clipped_norm")
            else:
                self._flat_grad *= torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self._optimizer.step()
        if mp_util.enable_mp() and self._cadence == "minibatch" and self._steps % self.CHECK_SYNC_STEPS == 0:
            assert self._check_synced(), "Network parameters desynchronized"
        self._steps += 1

    def end_epoch(self):
        """Per-epoch exchange of the "epoch" cadence: average parameters and float optimizer state over the ranks with ONE
        all-reduce of a flat buffer (no-op for the per-minibatch cadence or a single process)."""
        if not (mp_util.enable_mp() and self._cadence == "epoch"):
            return
        if self._flat_sgd:
            with torch.no_grad():
                for flat in (self._flat_param, self._flat_mom):
                    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
                    flat /= mp_util.get_num_procs()
            return
        with torch.no_grad():
            tensors = list(self._param_list)
            for p in self._param_list:
                st = self._optimizer.state.get(p, {})
                tensors += [v for v in st.values() if torch.is_tensor(v) and v.is_floating_point() and v.numel() == p.numel()]
            flat = torch.cat([t.reshape(-1) for t in tensors])
            torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
            flat /= mp_util.get_num_procs()
            off = 0
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()

    def get_steps(self):
        return self._steps

    def _flat_view_of_params(self):
        """(flat tensor holding every parameter, scatter-back function or None): the flat parameter buffer itself where the parameters
        are views of it (flat SGD), a packed copy otherwise"""
        if self._flat_sgd:
            return self._flat_param, None
        flat = torch.cat([p.detach().reshape(-1) for p in self._param_list])

        def scatter(src):
            off = 0
            with torch.no_grad():
                for p in self._param_list:
                    p.copy_(src[off:off + p.numel()].view_as(p))
                    off += p.numel()
        return flat, scatter

    def sync(self):
        """rank 0's parameters to every rank: ONE broadcast of the flat buffer (the reference sends one tensor per parameter,
        mp_optimizer.py:69-74: 16 collectives for the two MLPs)"""
        if not mp_util.enable_mp():
            return
        with torch.no_grad():
            flat, scatter = self._flat_view_of_params()
            torch.distributed.broadcast(flat, src=mp_util.ROOT_PROC_RANK)      # in place on the flat parameter buffer
            if scatter is not None:
                scatter(flat)

    def _check_synced(self):
        """do all ranks hold rank 0's parameters, bit for bit?  One broadcast of a flat copy + one 4-byte MIN-reduce
        (mp_optimizer.py:76-90 of the reference: one broadcast per parameter)"""
        if not mp_util.enable_mp():
            return True
        with torch.no_grad():
            flat, _ = self._flat_view_of_params()
            ref = flat.clone()
            torch.distributed.broadcast(ref, src=mp_util.ROOT_PROC_RANK)
            synced = torch.equal(flat, ref)
        buf = torch.tensor([int(synced)], dtype=torch.int, device=self._param_list[0].device)
        mp_util.reduce_inplace_min(buf)
        return buf.item() != 0
