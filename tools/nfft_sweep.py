#!/usr/bin/env python3
"""Training-forward time per frame for every n_fft of the fused kernel at a batch that fills the chip several times."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmel_amd
from dmel_amd import capi
res = {}
for N in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    lam = N / 6.0 * 0.9
    hop = max(1, N // 2)
    L = hop * 63
    M = 64 if N < 1024 else 128
    sr = 16000
    T = L // hop + 1
    B = max(1, (1 << 17) // T)          # ~131k frames
    assert capi.n_fft(lam) == N, (N, capi.n_fft(lam))
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    x = 0.1 * torch.randn(B, L, device="cuda:0")
    out = torch.empty(B, 1, M, T, device="cuda:0"); tan = torch.empty_like(out)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    frames = B * T
    info = plan.info()
    res[N] = dict(us=round(us, 1), ns_per_frame=round(us * 1000 / frames, 3), ns_per_frame_per_nlog2n=round(us * 1e3 / frames / (N * max(1, N.bit_length() - 1)) * 1e3, 3),
                  frames=frames, lds=info["lds_bytes"], fpt=info["frames_per_tile"])
print(json.dumps(res))
