"""Process-group helpers: one process per GPU, RCCL over xGMI (torch.distributed backend "nccl" on ROCm), gloo on CPU.

Mirror of the reference's util/mp_util.py:10-132 (same function names and meaning).  ``init`` also accepts an already
initialised default group (torchrun / bench.py launch)."""
import os

import torch

ROOT_PROC_RANK = 0
global_mp_device = None
global_num_procs = 1
global_requested_device = None


def visible_device_count():
    """GPUs this process may use, WITHOUT touching the HIP runtime: HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES if one
    is set, else the KFD topology nodes that have SIMDs (CPU nodes report simd_count 0).  bench.py's parent process and the
    per-rank device mapping use it (a parent that has initialised the GPU must not spawn / exec ranks on this pool)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for node in os.listdir(root):
            props = {}
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    for line in f:
                        parts = line.split()
                        if len(parts) == 2:
                            props[parts[0]] = parts[1]
            except OSError:
                continue
            if int(props.get("simd_count", "0")) <= 0:
                continue
            # a container may see all of the host's nodes in sysfs but be allowed to open only some render nodes
            minor = int(props.get("drm_render_minor", "-1"))
            if minor >= 0 and os.path.exists("/dev/dri") and not os.access("/dev/dri/renderD{}".format(minor), os.R_OK | os.W_OK):
                continue
            n += 1
    except OSError:
        return 0
    return n


def rank_device(rank, num_procs, device, num_devices=None, local_rank=None):
    """One process per GPU.  The reference's launcher hands the SAME --device string (default "cuda:0") to every rank it spawns
    (run.py:98,150-162) and leaves the mapping to the user; with this package under its aliases `--num_workers 8` would put 8 ranks
    on GPU 0.  So: with more than one process, a bare "cuda" / "cuda:0" means "this rank's GPU" = LOCAL_RANK (torchrun) or the rank,
    modulo the number of visible devices (two ranks on a one-GPU box share it).  Any other string ("cuda:3", "cpu") is taken as given."""
    if num_procs > 1 and device in ("cuda", "cuda:0"):
        if num_devices is None:             # inside a rank the runtime's own count is the authority (it may be initialised here)
            num_devices = torch.cuda.device_count() if torch.cuda.is_available() else visible_device_count()
        if local_rank is None:
            local_rank = int(os.environ.get("LOCAL_RANK", rank))
        return "cuda:{}".format(local_rank % max(int(num_devices), 1))
    return device


def init(rank, num_procs, device, master_port=None):
    global global_mp_device, global_num_procs, global_requested_device
    global_requested_device = device
    device = rank_device(rank, num_procs, device)
    global_mp_device = device
    global_num_procs = num_procs
    assert num_procs > 0
    if str(device).startswith("cuda") and torch.cuda.is_available() and torch.device(device).index is not None:
        torch.cuda.set_device(device)            # RCCL binds a communicator to the current device (a bare "cuda" keeps the current one)
    if num_procs > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if master_port is not None:
            os.environ["MASTER_PORT"] = str(master_port)
        backend = os.environ.get("PARC_MP_BACKEND") or ("gloo" if device == "cpu" else "nccl")
        torch.distributed.init_process_group(backend, rank=rank, world_size=num_procs)
        assert torch.distributed.get_world_size() == num_procs


def resolve_device(device):
    """The device a builder should use for `device` as the launcher passed it: this rank's device if it is the very string init() was
    given (the reference's run.py passes that same string on to build_env / build_agent, run.py:111-117), else `device` itself."""
    if global_mp_device is not None and device == global_requested_device:
        return global_mp_device
    return device


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
