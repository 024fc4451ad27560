#!/usr/bin/env python3
"""Timing experiment: the config-2 forward as ONE launch over 256 clips against TWO launches over 128 clips each on two streams
(joined by events), and against two launches in one stream.  usage: DMEL_LIB=... python tools/split_launch.py [c2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import capi, synth
from bench import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
B, L, sr, lam, hop, M = CONFIGS[name]
T = L // hop + 1
x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out)
plans = [capi.Plan(L, hop, M, sr, max_batch=B) for _ in range(4)]
main = torch.cuda.current_stream()
side = [torch.cuda.Stream() for _ in range(3)]
H = B // 2


def one():
    plans[0].forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, torch.cuda.current_stream().cuda_stream)


def parts(nparts, streams):
    n = B // nparts
    cur = torch.cuda.current_stream()                # the capture stream under torch.cuda.graph
    streams = [cur if s is main else s for s in streams]
    for i in range(nparts):
        s = streams[i]
        if s is not cur:
            s.wait_stream(cur)
        o = out[i * n:(i + 1) * n]; t = tan[i * n:(i + 1) * n]
        plans[i].forward(x[i * n:(i + 1) * n].data_ptr(), n, lam, o.data_ptr(), t.data_ptr(), True, 1e-10, s.cuda_stream)
    for i in range(nparts):
        if streams[i] is not cur:
            cur.wait_stream(streams[i])


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / n)
    return best


def graphed(fn):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(4):
            fn()
    return lambda: g.replay()


print("one launch (eager train)", round(timeit(one), 2), "us")
print("two launches, same stream", round(timeit(lambda: parts(2, [main, main])), 2), "us")
print("two launches, two streams (eager)", round(timeit(lambda: parts(2, [main, side[0]])), 2), "us")
print("four launches, four streams (eager)", round(timeit(lambda: parts(4, [main] + side)), 2), "us")
for label, fn in (("one launch", one), ("two launches / two streams", lambda: parts(2, [main, side[0]])),
                  ("four launches / four streams", lambda: parts(4, [main] + side)), ("two launches / one stream", lambda: parts(2, [main, main]))):
    try:
        g = graphed(fn)
        print("graph x4:", label, round(timeit(g, 100) / 4, 2), "us per forward")
    except Exception as e:
        print("graph:", label, "failed:", type(e).__name__, str(e)[:200])
