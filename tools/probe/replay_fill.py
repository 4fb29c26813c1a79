"""Does torch.cuda.CUDAGraph.replay() itself launch a fill when the captured graph holds no generator op?  Run under rocprofv3 --kernel-trace
--stats and count FillFunctor<long> launches: 20 replays of a graph of 4 adds.  python tools/probe/replay_fill.py"""
import torch

f = torch.zeros(1024, device="cuda:0")
for _ in range(3):
    f.add_(1.0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    for _ in range(4):
        f.add_(1.0)
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("done", float(f[0]))
