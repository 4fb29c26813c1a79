"""What the fused post-step launch costs when it does not follow itself: graph-replayed pairs (other kernel, track_post_kernel) against the
other kernel alone.  The standalone loops of tools/bench_kernels.py leave the clip rows, the heightfield, the kernel's own code and the
row addresses hot; inside the rollout step the launch follows the simulator and 0.3 ms of policy GEMMs.
python tools/post_step_cold.py [workload]  -> one JSON line per neighbour"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd import _hip, workloads  # noqa: E402


def graph_us(fn, n=64):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "boxes_64clips"
    core, clips, _ = workloads.build_core(workload, 4096, "cuda:0")
    full = _hip.POST_OBS | _hip.POST_REWARD_DONE | _hip.POST_HF | _hip.POST_TARGETS
    post = lambda: core.post_step(full)
    dev = "cuda:0"
    a64, b64 = torch.empty(16 << 20, device=dev), torch.empty(16 << 20, device=dev)          # 64 MB each
    a8, b8 = torch.empty(2 << 20, device=dev), torch.empty(2 << 20, device=dev)              # 8 MB each
    x, w, y = torch.randn(4096, 2048, device=dev), torch.randn(2048, 1024, device=dev), torch.empty(4096, 1024, device=dev)
    xs, ws, ys = torch.randn(256, 256, device=dev), torch.randn(256, 256, device=dev), torch.empty(256, 256, device=dev)
    state = [core.root_state, core.dof_state, core.rigid_body_state, core.contact_forces]
    tmp = [torch.empty_like(t) for t in state]

    def touch_state():                         # the simulator's role: the launch's per-env inputs were just written by another kernel
        for t, u in zip(state, tmp):
            u.copy_(t)
            t.copy_(u)
    neighbours = {"itself (back to back)": None, "copy 8 MB": lambda: b8.copy_(a8), "copy 64 MB": lambda: b64.copy_(a64),
                  "fp32 GEMM 256^3 (library kernel, little data)": lambda: torch.mm(xs, ws, out=ys),
                  "fp32 GEMM 4096x2048x1024 (the rollout's layer 2)": lambda: torch.mm(x, w, out=y),
                  "rewrite of the simulator outputs (8 small copies)": touch_state}
    for name, other in neighbours.items():
        if other is None:
            print(json.dumps({"neighbour": name, "post_us": round(graph_us(post), 2)}))
            continue
        alone = graph_us(other)
        both = graph_us(lambda: (other(), post()))
        print(json.dumps({"neighbour": name, "neighbour_alone_us": round(alone, 2), "pair_us": round(both, 2), "post_us": round(both - alone, 2)}))


if __name__ == "__main__":
    main()
