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

    def eval_critic(self, obs):
        return self._critic_out(self._critic_layers(obs))
