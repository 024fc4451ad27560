#!/usr/bin/env python3
"""Forward + backward time of the long-transform path (n_fft 8192 / 16384) on ESC-50-shaped input."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmel_amd
from dmel_amd import capi
res = {}
for lam in (256.0, 700.0, 1400.0):
    B, L, hop, M, sr = 32, 220500, 441, 128, 44100
    T = L // hop + 1
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    x = 0.1 * torch.randn(B, L, device="cuda:0")
    out = torch.empty(B, 1, M, T, device="cuda:0"); tan = torch.empty_like(out)
    st = torch.cuda.current_stream().cuda_stream
    for train in (True, False):
        for _ in range(3):
            plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr() if train else None, True, 1e-10, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr() if train else None, True, 1e-10, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        res[f"n_fft_{capi.n_fft(lam)}_{'train' if train else 'infer'}"] = dict(us=round(us, 1), mframes_per_s=round(B * T / us, 1))
print(json.dumps(res))
