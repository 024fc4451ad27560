#!/usr/bin/env python3
"""Training forward (+ dot) time at the shapes of the reference's own experiments (search_spaces.py:4-33 ESC-50: batch 32 x 40000
samples @ 8 kHz, hop 80, 64 mels; :36-66 Audio-MNIST: batch 64 x 8000; init_lambd = 8000 x / 6 for x in 0.01, 0.035, 0.3, i.e.
n_fft 128, 512, 4096), plus lambd values the run may drift to.  One JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import capi, synth

SHAPES = {"esc50": (32, 40000), "audio_mnist": (64, 8000)}
LAMBDS = [8000 * v / 6 for v in (0.01, 0.035, 0.3)] + [200.0, 700.0]
res = {}
for name, (B, L) in SHAPES.items():
    hop, M, sr = 80, 64, 8000
    T = L // hop + 1
    x = torch.from_numpy(synth.waveforms(B, L, seed=1)).cuda()
    out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out); g = torch.randn_like(out)
    dl = torch.zeros(1, device="cuda")
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    s = torch.cuda.current_stream().cuda_stream
    for lam in LAMBDS:
        def step():
            plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, s)
            plan.backward(g.data_ptr(), tan.data_ptr(), out.numel(), dl.data_ptr(), s)
        for _ in range(5): step()
        torch.cuda.synchronize()
        best = 1e9
        n = 50
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): step()
            e1.record(); torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / n)
        info = plan.info()
        res[f"{name}_lambd{lam:.1f}"] = dict(n_fft=info["n_fft"], kernel_path=info["kernel_path"], frames=B * T, step_us=round(best, 1),
                                             frames_per_s=round(B * T / best * 1e6), ns_per_frame=round(best * 1e3 / (B * T), 2))
print(json.dumps(res))
