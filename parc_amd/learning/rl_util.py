"""TD(lambda) returns and advantage normalisation on the GPU.

Mirror of the reference's ``learning/rl_util.py`` (compute_td_lambda_return :6-29) and of the advantage
block of ``DMPPOAgent._build_train_data`` (learning/dm_ppo_agent.py:393-403), as HIP kernels
(parc_td_lambda_return / parc_adv_normalize).  No CPU fallback.
"""
import torch

from .. import _hip

_workspace = {}


def compute_td_lambda_return(r, next_vals, done, discount, td_lambda):
    """r, next_vals [T,N] fp32, done [T,N] int32 (DoneFlags) -> returns [T,N]."""
    assert r.shape == next_vals.shape and r.dim() == 2
    r = r.contiguous().float()
    nv = next_vals.contiguous().float()
    dn = done.contiguous().to(torch.int32)
    T, N = r.shape
    ret = torch.empty_like(r)
    _hip.check(_hip.lib().parc_td_lambda_return(_hip.stream(), T, N, _hip.ptr(r), _hip.ptr(nv), _hip.ptr(dn), float(discount),
                                                float(td_lambda), _hip.ptr(ret)), "parc_td_lambda_return")
    return ret


def normalize_advantage(ret, vals, rand_action_mask, clip):
    """clamp((ret - vals - mean) / max(std, 1e-5), +-clip) with mean/std (unbiased) over rand_action_mask == 1.
    Returns (norm_adv, mean_std) where mean_std is a 2-element device tensor."""
    ret = ret.contiguous().float()
    vals = vals.contiguous().float()
    mask = rand_action_mask.contiguous().float()
    dev = ret.device
    ws = _workspace.get(dev)
    if ws is None:
        ws = torch.empty(3 * 1024, dtype=torch.float64, device=dev)
        _workspace[dev] = ws
    out = torch.empty_like(ret)
    ms = torch.empty(2, dtype=torch.float32, device=dev)
    _hip.check(_hip.lib().parc_adv_normalize(_hip.stream(), ret.numel(), _hip.ptr(ret), _hip.ptr(vals), _hip.ptr(mask), float(clip),
                                             _hip.ptr(out), _hip.ptr(ms), _hip.ptr(ws)), "parc_adv_normalize")
    return out, ms


def ppo_loss_and_grads(mean, logstd, pred, norm_a, old_logp, adv, mask, tar_val, cfg):
    """parc_ppo_loss without autograd: -> (info[16] with info[0] = loss, dLoss/d mean [B, A], dLoss/d logstd [A], dLoss/d pred [B])"""
    from .. import _hip
    B, A = mean.shape
    mean, norm_a = mean.contiguous(), norm_a.contiguous()
    g_mean = torch.empty_like(mean)
    g_logstd = torch.empty_like(logstd)
    g_pred = torch.empty(B, dtype=torch.float32, device=mean.device)
    out = torch.empty(16, dtype=torch.float32, device=mean.device)
    ws = torch.empty(_hip.lib().parc_ppo_workspace_floats(B), dtype=torch.float32, device=mean.device)
    p = _hip.ptr
    _hip.check(_hip.lib().parc_ppo_loss(_hip.stream(), B, A, p(mean), p(logstd.contiguous()), p(norm_a), p(old_logp.contiguous()),
                                        p(adv.contiguous()), p(mask.contiguous()), p(pred.contiguous()), p(tar_val.contiguous()), cfg,
                                        p(g_mean), p(g_logstd), p(g_pred), p(out), p(ws)), "parc_ppo_loss")
    return out, g_mean, g_logstd, g_pred


def ppo_loss_and_grads_packed(mean, logstd, pred, rec, cfg):
    """ppo_loss_and_grads on packed per-sample records rec[B, W] = [norm_action (A) | a_logp | adv | rand_action_mask | tar_val | pad]"""
    from .. import _hip
    B, A = mean.shape
    assert rec.dim() == 2 and rec.shape[0] == B and rec.shape[1] >= A + 4 and rec.is_contiguous()
    mean = mean.contiguous()
    g_mean = torch.empty_like(mean)
    g_logstd = torch.empty_like(logstd)
    g_pred = torch.empty(B, dtype=torch.float32, device=mean.device)
    out = torch.empty(16, dtype=torch.float32, device=mean.device)
    ws = torch.empty(_hip.lib().parc_ppo_workspace_floats(B), dtype=torch.float32, device=mean.device)
    p = _hip.ptr
    _hip.check(_hip.lib().parc_ppo_loss_packed(_hip.stream(), B, A, p(mean), p(logstd.contiguous()), p(rec), int(rec.shape[1]), p(pred.contiguous()),
                                               cfg, p(g_mean), p(g_logstd), p(g_pred), p(out), p(ws)), "parc_ppo_loss_packed")
    return out, g_mean, g_logstd, g_pred


def ppo_cfg(clip_ratio, bound_w, entropy_w, reg_w, critic_w, large_critic_loss=20.0, critic_l1=False):
    from .. import _hip
    return _hip.PPOCfgS(float(clip_ratio), float(bound_w), float(entropy_w), float(reg_w), float(critic_w), float(large_critic_loss),
                        int(bool(critic_l1)))


class _PPOLossFn(torch.autograd.Function):
    """loss, info = ppo_loss(mean, logstd, pred | norm_a, old_logp, adv, mask, tar_val, cfg): forward runs the three
    kernels of parc_ppo_loss, which also produce dloss/d(mean, logstd, pred); backward hands them out."""

    @staticmethod
    def forward(ctx, mean, logstd, pred, norm_a, old_logp, adv, mask, tar_val, cfg):
        out, g_mean, g_logstd, g_pred = ppo_loss_and_grads(mean, logstd, pred, norm_a, old_logp, adv, mask, tar_val, cfg)
        ctx.save_for_backward(g_mean, g_logstd, g_pred)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        g_mean, g_logstd, g_pred = ctx.saved_tensors
        return g_mean * g_loss, g_logstd * g_loss, g_pred * g_loss, None, None, None, None, None, None


def ppo_loss(mean, logstd, pred, norm_a, old_logp, adv, mask, tar_val, clip_ratio, bound_w, entropy_w, reg_w, critic_w,
             large_critic_loss=20.0, critic_l1=False):
    """Fused PPO loss of PPOAgent._compute_loss (learning/ppo_agent.py:186-330) on the GPU: returns (loss, info[11]) with
    info = loss, critic_loss, actor_loss, clip_frac, imp_ratio, action_bound_loss, entropy, reg_loss, cnt, ..."""
    cfg = ppo_cfg(clip_ratio, bound_w, entropy_w, reg_w, critic_w, large_critic_loss, critic_l1)
    return _PPOLossFn.apply(mean, logstd, pred, norm_a, old_logp, adv, mask, tar_val, cfg)
