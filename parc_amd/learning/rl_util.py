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
