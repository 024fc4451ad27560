#!/usr/bin/env python3
"""Forward-kernel time per frame against batch size (config-2 geometry): how much of the time is the two-round tail."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmel_amd
from dmel_amd import capi
L, hop, M, sr, lam = 16000, 512, 128, 16000, 128.0
T = L // hop + 1
res = {}
for B in (64, 128, 256, 512, 1024, 2048, 4096):
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    x = 0.1 * torch.randn(B, L, device="cuda:0")
    out = torch.empty(B, 1, M, T, device="cuda:0"); tan = torch.empty_like(out)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(10):
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    e0.record()
    for _ in range(n):
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), None, True, 1e-10, st)       # inference: no tangent, two frames per FFT
    e1.record(); torch.cuda.synchronize()
    us_inf = e0.elapsed_time(e1) * 1000 / n
    res[B] = dict(us=round(us, 2), ns_per_frame=round(us * 1000 / (B * T), 3), rounds=B * T / 8 / 512,
                  inference_us=round(us_inf, 2), inference_ns_per_frame=round(us_inf * 1000 / (B * T), 3))
print(json.dumps(res))
