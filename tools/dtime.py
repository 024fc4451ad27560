#!/usr/bin/env python3
"""Kernel time of dmel_dot_kernel (lambd.grad = <grad_out, tangent>) from trains of launches, alone and behind the forward that
writes the tangent (the cache state it sees inside a step).  usage: DMEL_LIB=<lib.so> python tools/dtime.py [c2]"""
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
g = torch.from_numpy(synth.cotangent((B, 1, M, T), seed=1)).cuda()
out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out); dl = torch.zeros(1, device="cuda")
plan = capi.Plan(L, hop, M, sr, max_batch=B)
s = torch.cuda.current_stream().cuda_stream


def train(fn, n=200):
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


fwd = lambda: plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, s)
dot = lambda: plan.backward(g.data_ptr(), tan.data_ptr(), g.numel(), dl.data_ptr(), s)
fwd(); dot(); torch.cuda.synchronize()
ref = float((g.double() * tan.double()).sum())
t_dot, t_fwd = train(dot), train(fwd)
t_both = train(lambda: (fwd(), dot()))
print(os.path.basename(os.environ.get("DMEL_LIB", "libdmel_hip.so")), name, f"dot alone {t_dot:.2f} us, forward alone {t_fwd:.2f}, forward + dot {t_both:.2f} "
      f"(dot inside a step {t_both - t_fwd:.2f}); d lambd {float(dl):.6f} vs fp64 {ref:.6f}")
