"""Process-group helpers: one process per GPU, RCCL over xGMI (torch.distributed backend "nccl" on ROCm), gloo on CPU.

Mirror of the reference's util/mp_util.py:10-132 (same function names and meaning).  ``init`` also accepts an already
initialised default group (torchrun / bench.py launch)."""
import os

import torch

ROOT_PROC_RANK = 0
global_mp_device = None
global_num_procs = 1


def init(rank, num_procs, device, master_port=None):
    global global_mp_device, global_num_procs
    global_mp_device = device
    global_num_procs = num_procs
    assert num_procs > 0
    if num_procs > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if master_port is not None:
            os.environ["MASTER_PORT"] = str(master_port)
        backend = "gloo" if device == "cpu" else "nccl"
        torch.distributed.init_process_group(backend, rank=rank, world_size=num_procs)
        assert torch.distributed.get_world_size() == num_procs


def get_num_procs():
    return global_num_procs


def get_proc_rank():
    return torch.distributed.get_rank() if enable_mp() else 0


def is_root_proc():
    return get_proc_rank() == ROOT_PROC_RANK


def enable_mp():
    return global_num_procs > 1


def get_device():
    return global_mp_device


def broadcast(x):
    if enable_mp():
        data = x.clone()
        torch.distributed.broadcast(data, src=ROOT_PROC_RANK)
        return data
    return x


def reduce_all(x, op):
    if not enable_mp():
        return x
    is_tensor = torch.is_tensor(x)
    buf = x.clone() if is_tensor else torch.tensor(x, device=get_device())
    torch.distributed.all_reduce(buf, op=op)
    return buf if is_tensor else buf.item()


def reduce_sum(x):
    return reduce_all(x, torch.distributed.ReduceOp.SUM)


def reduce_min(x):
    return reduce_all(x, torch.distributed.ReduceOp.MIN)


def reduce_max(x):
    return reduce_all(x, torch.distributed.ReduceOp.MAX)


def reduce_mean(x):
    return reduce_sum(x) / get_num_procs()


def reduce_inplace_all(x, op):
    if enable_mp():
        torch.distributed.all_reduce(x, op=op)


def reduce_inplace_sum(x):
    reduce_inplace_all(x, torch.distributed.ReduceOp.SUM)


def reduce_inplace_mean(x):
    reduce_inplace_sum(x)
    x /= get_num_procs()


def reduce_inplace_min(x):
    reduce_inplace_all(x, torch.distributed.ReduceOp.MIN)


def reduce_inplace_max(x):
    reduce_inplace_all(x, torch.distributed.ReduceOp.MAX)
