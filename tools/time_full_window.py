#!/usr/bin/env python3
"""Training step (forward + dot, C ABI) of the layer's default branch optimized=False (window = whole clip, n_fft = 2 n_points,
time_frequency.py:41,51) at the clip lengths of the reference's datasets: Audio-MNIST 8000 samples (n_fft 16000: chirp-z),
ESC-50 40000 samples (n_fft 80000), and a power-of-two clip for comparison (n_fft 16384: the fused kernel).  The first run of a
process on a fresh box is several times slower (code objects paging in): one untimed pass over all shapes comes first."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import capi, synth

SHAPES = ((64, 8000, 400.0), (32, 40000, 400.0), (64, 8192, 400.0))
res = {}
for timed in (False, True):
    for (B, L, lam) in SHAPES:
        hop, M, sr = 80, 64, 8000
        T = L // hop + 1
        x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
        out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out); g = torch.randn_like(out)
        dl = torch.zeros(1, device="cuda")
        plan = capi.Plan(L, hop, M, sr, max_batch=B)
        s = torch.cuda.current_stream().cuda_stream

        def step():
            plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, s, extra_flags=capi.DMEL_FLAG_FULL_WINDOW)
            plan.backward(g.data_ptr(), tan.data_ptr(), out.numel(), dl.data_ptr(), s)
        step(); torch.cuda.synchronize()
        if not timed:
            continue
        n = 3
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[f"{B}x{L}"] = dict(n_fft=2 * L, kernel_path=plan.info()["kernel_path"], ms_per_step=round(dt * 1e3, 2),
                               frames_per_s=round(B * T / dt))
print(json.dumps(res))
