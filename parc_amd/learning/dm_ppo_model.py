"""Actor / critic networks of the tracker: 1312 -> 2048 -> 1024 -> 512 ReLU MLPs, Gaussian policy head with fixed
log-std.  Mirror of the reference's learning/dm_ppo_model.py + ppo_model.py + nets/fc_3layers_2048units.py +
distribution_gaussian_diag.py with identical module names, so state_dict keys match
(``_actor_layers.{0,2,4}``, ``_action_dist._mean_net``, ``_action_dist._logstd_net``, ``_critic_layers``, ``_critic_out``).
The GEMMs are plain fp32 library GEMMs (hipBLASLt through torch)."""
import enum

import numpy as np
import torch


class StdType(enum.Enum):
    FIXED = 0
    CONSTANT = 1
    VARIABLE = 2


class DistributionGaussianDiag:
    def __init__(self, mean, logstd):
        self._mean = mean
        self._logstd = logstd
        self._std_cache = None          # exp(logstd) on first use (the fused action head never needs it)
        self._dim = mean.shape[-1]

    @property
    def _std(self):
        if self._std_cache is None:
            self._std_cache = torch.exp(self._logstd)
        return self._std_cache

    @property
    def stddev(self):
        return self._std

    @property
    def logstd(self):
        return self._logstd

    @property
    def mean(self):
        return self._mean

    @property
    def mode(self):
        return self._mean

    def sample(self):
        return self._mean + self._std * torch.randn_like(self._mean)

    def log_prob(self, x):
        diff = x - self._mean
        logp = -0.5 * torch.sum(torch.square(diff / self._std), dim=-1)
        return logp + (-0.5 * self._dim * np.log(2.0 * np.pi) - torch.sum(self._logstd, dim=-1))

    def entropy(self):
        return torch.sum(self._logstd, dim=-1) + 0.5 * self._dim * np.log(2.0 * np.pi * np.e)

    def param_reg(self):
        return torch.sum(torch.square(self._mean), dim=-1)


class DistributionGaussianDiagBuilder(torch.nn.Module):
    def __init__(self, in_size, out_size, std_type, init_std, init_output_scale=0.01):
        super().__init__()
        self._std_type = std_type
        self._mean_net = torch.nn.Linear(in_size, out_size)
        torch.nn.init.uniform_(self._mean_net.weight, -init_output_scale, init_output_scale)
        torch.nn.init.zeros_(self._mean_net.bias)
        logstd = float(np.log(init_std))
        if std_type == StdType.FIXED:
            self._logstd_net = torch.nn.Parameter(torch.full((out_size,), logstd, dtype=torch.float32), requires_grad=False)
        elif std_type == StdType.CONSTANT:
            self._logstd_net = torch.nn.Parameter(torch.full((out_size,), logstd, dtype=torch.float32), requires_grad=True)
        else:
            self._logstd_net = torch.nn.Linear(in_size, out_size)
            torch.nn.init.uniform_(self._logstd_net.weight, -init_output_scale, init_output_scale)
            torch.nn.init.constant_(self._logstd_net.bias, logstd)

    def forward(self, x):
        mean = self._mean_net(x)
        if self._std_type == StdType.VARIABLE:
            logstd = self._logstd_net(x)
        else:
            logstd = torch.broadcast_to(self._logstd_net, mean.shape)
        return DistributionGaussianDiag(mean=mean, logstd=logstd)


_NETS = {"fc_3layers_2048units": [2048, 1024, 512], "fc_3layers_1024units": [1024, 1024, 512], "fc_2layers_1024units": [1024, 512]}


class _LinearReLU(torch.autograd.Function):
    """relu(x W^T + b) with the bias + ReLU applied in the GEMM epilogue (hipBLASLt RELU_BIAS through
    torch._addmm_activation), so the forward pass writes the activation once instead of GEMM output + clamp.  The op
    has no registered derivative in torch 2.10, hence the explicit backward (the same three kernels autograd uses)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        g = torch.ops.aten.threshold_backward(gy, y, 0.0)
        gx = g.mm(weight) if ctx.needs_input_grad[0] else None
        gw = g.t().mm(x) if ctx.needs_input_grad[1] else None
        gb = g.sum(dim=0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


class FusedMLP(torch.nn.Sequential):
    """Sequential(Linear, ReLU, Linear, ReLU, ...) with the reference's parameter names (``0.weight``, ``2.weight`` ...);
    on the GPU every Linear+ReLU pair runs as one fused GEMM."""

    def forward(self, x):
        mods = list(self)
        if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32):
            return super().forward(x)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, torch.nn.Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.ReLU):
                if torch.is_grad_enabled() and (x.requires_grad or m.weight.requires_grad):
                    x = _LinearReLU.apply(x, m.weight, m.bias)
                else:
                    x = torch._addmm_activation(m.bias, x, m.weight.t(), use_gelu=False)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


def build_net(name, in_size, activation):
    if name not in _NETS:
        raise NotImplementedError("net '{}' (the tracker default is fc_3layers_2048units)".format(name))
    layers = []
    for out_size in _NETS[name]:
        lin = torch.nn.Linear(in_size, out_size)
        torch.nn.init.zeros_(lin.bias)
        layers += [lin, activation()]
        in_size = out_size
    return FusedMLP(*layers), in_size


class DMPPOModel(torch.nn.Module):
    def __init__(self, config, env):
        super().__init__()
        self._activation = torch.nn.ReLU
        obs_dim = int(np.prod(env.get_obs_space().shape))
        a_dim = int(np.prod(env.get_action_space().shape))
        self._actor_layers, h = build_net(config["actor_net"], obs_dim, self._activation)
        self._action_dist = DistributionGaussianDiagBuilder(h, a_dim, std_type=StdType[config["actor_std_type"]],
                                                            init_std=config["action_std"],
                                                            init_output_scale=config["actor_init_output_scale"])
        self._critic_layers, hc = build_net(config["critic_net"], obs_dim, self._activation)
        self._critic_out = torch.nn.Linear(hc, 1)
        torch.nn.init.zeros_(self._critic_out.bias)

    def eval_actor(self, obs):
        return self._action_dist(self._actor_layers(obs))

    # ------------------------------------------------------------------ explicit training step (no autograd graph)
    # The update phase runs the same two fixed MLPs 40 times per iteration.  Going through autograd costs, per minibatch, a
    # threshold_backward + a separate bias reduction per layer, an accumulate-into-.grad per parameter and the zeroing of the flat
    # gradient first.  Spelled out, every weight / bias gradient is written ONCE, straight into its slice of the optimizer's flat
    # gradient (GEMM with out=), and the ReLU mask + bias gradient of a layer are one pass (parc_relu_bwd_bias_grad).  Same GEMMs,
    # same values as autograd (tests/test_learner_gpu.py compares the two paths); used by DMPPOAgent when the policy's log-std is
    # not state dependent.
    def supports_explicit_backward(self):
        def plain(seq):
            mods = list(seq)
            return len(mods) % 2 == 0 and all(isinstance(m, torch.nn.Linear) for m in mods[0::2]) and all(isinstance(m, torch.nn.ReLU) for m in mods[1::2]) \
                and all(m.out_features % 4 == 0 for m in mods[0::2])
        return plain(self._actor_layers) and plain(self._critic_layers) and self._action_dist._std_type != StdType.VARIABLE

    @staticmethod
    def _trunk_forward(seq, x):
        acts = [x]
        for lin in list(seq)[0::2]:
            x = torch._addmm_activation(lin.bias, x, lin.weight.t(), use_gelu=False)
            acts.append(x)
        return acts

    @torch.no_grad()
    def train_forward(self, norm_obs):
        """-> (mean [B, A], logstd [A], pred [B], saved activations)"""
        a = self._trunk_forward(self._actor_layers, norm_obs)
        c = self._trunk_forward(self._critic_layers, norm_obs)
        mnet = self._action_dist._mean_net
        mean = torch.addmm(mnet.bias, a[-1], mnet.weight.t())
        pred = torch.addmm(self._critic_out.bias, c[-1], self._critic_out.weight.t()).squeeze(-1)
        return mean, self._action_dist._logstd_net, pred, (a, c)

    def _relu_bwd(self, d, y, db):
        from .. import _hip
        L = _hip.lib()
        need = int(L.parc_relu_bwd_workspace_floats(d.shape[0], d.shape[1]))
        ws = getattr(self, "_relu_ws", None)
        if ws is None or ws.numel() < need or ws.device != d.device:
            ws = self._relu_ws = torch.empty(max(need, 1), dtype=torch.float32, device=d.device)
        _hip.check(L.parc_relu_bwd_bias_grad(_hip.stream(), d.shape[0], d.shape[1], _hip.ptr(d), _hip.ptr(y), _hip.ptr(db), _hip.ptr(ws)),
                   "parc_relu_bwd_bias_grad")

    def _weighted_colsum(self, x, w, out):
        from .. import _hip
        L = _hip.lib()
        need = int(L.parc_relu_bwd_workspace_floats(x.shape[0], x.shape[1]))
        ws = getattr(self, "_relu_ws", None)
        if ws is None or ws.numel() < need or ws.device != x.device:
            ws = self._relu_ws = torch.empty(max(need, 1), dtype=torch.float32, device=x.device)
        _hip.check(L.parc_weighted_colsum(_hip.stream(), x.shape[0], x.shape[1], _hip.ptr(x), _hip.ptr(w.contiguous()), _hip.ptr(out), _hip.ptr(ws)),
                   "parc_weighted_colsum")

    def _trunk_backward(self, seq, acts, d, grad_of, done):
        """d = dLoss/d(last activation) [B, h]; writes every layer's weight / bias gradient, last layer first."""
        lins = list(seq)[0::2]
        for k in range(len(lins) - 1, -1, -1):
            lin = lins[k]
            self._relu_bwd(d, acts[k + 1], grad_of(lin.bias))           # d <- d * (y > 0) in place, bias gradient
            torch.mm(d.t(), acts[k], out=grad_of(lin.weight))
            done(lin.weight)
            done(lin.bias)
            if k > 0:
                d = torch.mm(d, lin.weight)

    @torch.no_grad()
    def train_backward(self, saved, g_mean, g_logstd, g_pred, grad_of, done=lambda p: None):
        """Gradients of the loss w.r.t. every parameter, given dLoss/d(mean, logstd, pred) from the fused loss kernel.  grad_of(p) = the
        tensor to write p's gradient into (overwritten); done(p) is called once p's gradient is complete, in the order autograd
        would finish them (actor before critic, last layer first) -- the data-parallel optimizer starts bucket exchanges from it."""
        a, c = saved
        mnet = self._action_dist._mean_net
        torch.mm(g_mean.t(), a[-1], out=grad_of(mnet.weight))
        torch.sum(g_mean, dim=0, out=grad_of(mnet.bias))
        done(mnet.weight)
        done(mnet.bias)
        if self._action_dist._std_type == StdType.CONSTANT:
            grad_of(self._action_dist._logstd_net).copy_(g_logstd)
            done(self._action_dist._logstd_net)
        self._trunk_backward(self._actor_layers, a, torch.mm(g_mean, mnet.weight), grad_of, done)
        gp = g_pred.unsqueeze(-1)
        if self._critic_out.weight.shape[0] == 1:
            # a [1, h] weight gradient is a weighted column sum (as a GEMM with M = 1 the library takes 62 us for it, as a gemv 311 us)
            self._weighted_colsum(c[-1], g_pred, grad_of(self._critic_out.weight).view(-1))
        else:
            torch.mm(gp.t(), c[-1], out=grad_of(self._critic_out.weight))
        torch.sum(gp, dim=0, out=grad_of(self._critic_out.bias))
        done(self._critic_out.weight)
        done(self._critic_out.bias)
        self._trunk_backward(self._critic_layers, c, torch.mm(gp, self._critic_out.weight), grad_of, done)

    def eval_critic(self, obs):
        return self._critic_out(self._critic_layers(obs))
