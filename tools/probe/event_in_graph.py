"""Can a pair of timing events be captured INTO a hipGraph and read back after a replay?  (bench.py wants the duration of one node of the
rollout step's graph without a profiler.)  Tries torch's events, then hipEventRecord / hipEventRecordWithFlags through ctypes.
python tools/probe/event_in_graph.py -> one JSON line per attempt"""
import ctypes
import json

import torch

dev = "cuda:0"
a, b = torch.empty(8 << 20, device=dev), torch.empty(8 << 20, device=dev)
f = torch.zeros(1024, device=dev)


def body(rec0, rec1):
    f.fill_(1.0)
    rec0()
    b.copy_(a)          # ~10 us
    rec1()
    f.add_(1.0)


def attempt(name, make, rec, elapsed):
    try:
        e0, e1 = make(), make()
        body(lambda: rec(e0), lambda: rec(e1))
        torch.cuda.synchronize()
        eager = elapsed(e0, e1)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            body(lambda: rec(e0), lambda: rec(e1))
        vals = []
        for _ in range(5):
            g.replay()
            torch.cuda.synchronize()
            vals.append(round(elapsed(e0, e1), 2))
        print(json.dumps({"attempt": name, "eager_us": round(eager, 2), "in_graph_us": vals}), flush=True)
    except Exception as exc:      # noqa: BLE001
        print(json.dumps({"attempt": name, "error": repr(exc)[:300]}), flush=True)
        torch.cuda.synchronize()


attempt("torch.cuda.Event(enable_timing=True)", lambda: torch.cuda.Event(enable_timing=True), lambda e: e.record(),
        lambda x, y: x.elapsed_time(y) * 1e3)
try:
    attempt("torch.cuda.Event(enable_timing=True, external=True)", lambda: torch.cuda.Event(enable_timing=True, external=True),
            lambda e: e.record(), lambda x, y: x.elapsed_time(y) * 1e3)
except TypeError as exc:
    print(json.dumps({"attempt": "external=True", "error": repr(exc)}))

hip = ctypes.CDLL("libamdhip64.so")


def mk():
    ev = ctypes.c_void_p()
    assert hip.hipEventCreate(ctypes.byref(ev)) == 0
    return ev


def el(x, y):
    ms = ctypes.c_float()
    rc = hip.hipEventElapsedTime(ctypes.byref(ms), x, y)
    if rc != 0:
        raise RuntimeError("hipEventElapsedTime -> %d" % rc)
    return ms.value * 1e3


def rec_plain(ev):
    rc = hip.hipEventRecord(ev, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError("hipEventRecord -> %d" % rc)


def rec_ext(ev):
    rc = hip.hipEventRecordWithFlags(ev, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_uint(1))   # hipEventRecordExternal
    if rc != 0:
        raise RuntimeError("hipEventRecordWithFlags -> %d" % rc)


attempt("hipEventRecord (ctypes)", mk, rec_plain, el)
if hasattr(hip, "hipEventRecordWithFlags"):
    attempt("hipEventRecordWithFlags(external) (ctypes)", mk, rec_ext, el)
else:
    print(json.dumps({"attempt": "hipEventRecordWithFlags", "error": "symbol absent"}))
