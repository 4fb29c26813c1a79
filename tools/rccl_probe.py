"""Exercise every collective of the data-parallel path on the REAL backend (RCCL = torch.distributed "nccl") with the one GPU a test box
has: a process group of world size 1, with the package's "more than one process" switch forced on for this probe only.  A mean over one
rank is the identity, so every result must equal the single-process result bit for bit - what this checks is that the calls themselves
(asynchronous bucket all-reduces on views of the flat gradient started from gradient callbacks, waits, broadcasts, MIN reductions,
per-epoch averaging of the flat parameter / momentum buffers, logger aggregation) are accepted by RCCL and ordered correctly against
the compute stream.   Run:  python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tools/rccl_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd import workloads  # noqa: E402
from parc_amd.util import mp_util  # noqa: E402


def run(force_mp, cadence):
    torch.manual_seed(0)
    mp_util.enable_mp = (lambda: True) if force_mp else (lambda: False)
    env, _, _ = workloads.build_env("boxes_64clips", 64, "cuda:0", seed=1)
    agent = workloads.build_agent(env, "cuda:0", steps_per_iter=8, update_epochs=2, batch_size=2)
    agent._optimizer._cadence = cadence                 # what bench.py --grad-allreduce sets
    agent._curr_obs, agent._curr_info = env.reset()
    agent._init_train()
    torch.manual_seed(1)
    infos = [agent._train_iter() for _ in range(2)]
    opt = agent._optimizer
    synced = opt._check_synced() if force_mp else True
    flat = torch.cat([p.detach().reshape(-1) for p in opt._param_list]).clone()
    return flat, float(infos[-1]["critic_loss"]), synced, bool(opt._overlap), len(opt._buckets)


def main():
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)))
    assert torch.distributed.get_world_size() == 1 and torch.distributed.get_backend() == "nccl"
    mp_util.init(0, 1, "cuda:0")
    out = {}
    for cadence in ("minibatch", "epoch"):
        ref, loss_ref, _, _, _ = run(False, cadence)
        got, loss, synced, overlap, buckets = run(True, cadence)
        out[cadence] = {"max_abs_diff": float((ref - got).abs().max()), "loss": loss, "loss_single": loss_ref, "synced": synced,
                        "overlap": overlap, "buckets": buckets, "finite": bool(torch.isfinite(got).all())}
    # the helper collectives on their own
    x = torch.arange(5, dtype=torch.float32, device="cuda:0")
    mp_util.enable_mp = lambda: True
    out["helpers"] = {"broadcast": bool(torch.equal(mp_util.broadcast(x), x)), "sum": bool(torch.equal(mp_util.reduce_sum(x), x)),
                      "min": float(mp_util.reduce_min(3.5)), "mean": bool(torch.equal(mp_util.reduce_mean(x), x))}
    torch.distributed.barrier()
    torch.cuda.synchronize()
    torch.distributed.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
