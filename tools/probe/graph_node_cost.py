"""What a kernel node costs inside a replayed hipGraph, by what surrounds it (the rollout step's 27 nodes take >= 4.5 us each although an
empty kernel repeated in a graph takes 1.6 us).  python tools/probe/graph_node_cost.py  -> one JSON line per composition"""
import json

import torch

dev = "cuda:0"
f = torch.zeros(1024, device=dev)
l = torch.zeros(1024, dtype=torch.int64, device=dev)
g2 = torch.zeros(1024, device=dev)
x, w, y = torch.randn(4096, 2048, device=dev), torch.randn(2048, 1024, device=dev), torch.empty(4096, 1024, device=dev)
big_a, big_b = torch.empty(8 << 20, device=dev), torch.empty(8 << 20, device=dev)


def run(name, ops, reps):
    for _ in range(2):
        for op in ops:
            op()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(reps):
            for op in ops:
                op()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3)
    return {"graph": name, "nodes": reps * len(ops), "us_per_replay": round(best, 1), "us_per_node": round(best / (reps * len(ops)), 2)}


tiny = [lambda: f.fill_(1.0), lambda: l.fill_(3), lambda: g2.add_(1.0), lambda: f.mul_(0.5), lambda: l.add_(1), lambda: g2.copy_(f),
        lambda: torch.maximum(f, g2, out=g2), lambda: f.neg_()]
gemm = lambda: torch.mm(x, w, out=y)
copy = lambda: big_b.copy_(big_a)
out = [run("one tiny kernel repeated (fill)", [tiny[0]], 64),
       run("8 different tiny kernels in turn", tiny, 8),
       run("GEMM 4096x2048x1024 alone", [gemm], 16)]
g_alone = out[-1]["us_per_node"]
r = run("GEMM + 8 different tiny kernels", [gemm] + tiny, 8)
r["us_per_tiny_node_after_subtracting_the_GEMM"] = round((r["us_per_replay"] / 8 - g_alone) / 8, 2)
out.append(r)
c_alone = run("32 MB copy alone", [copy], 16)
out.append(c_alone)
r = run("32 MB copy + 8 different tiny kernels", [copy] + tiny, 8)
r["us_per_tiny_node_after_subtracting_the_copy"] = round((r["us_per_replay"] / 8 - c_alone["us_per_node"]) / 8, 2)
out.append(r)
# the same 8 tiny kernels, one replay per graph of 8 nodes, 64 replays back to back (the rollout replays a 27-node graph per step)
for op in tiny:
    op()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    for op in tiny:
        op()
g.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    s.record()
    for _ in range(64):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) * 1e3)
out.append({"graph": "8 different tiny kernels, 64 replays of the 8-node graph", "nodes": 512, "us_per_replay": round(best / 64, 1), "us_per_node": round(best / 512, 2)})
for o in out:
    print(json.dumps(o))
