#!/usr/bin/env python3
"""Timing ablations of the fused forward kernel (GPU box): full kernel vs FFT phase only vs MFMA phase only.
Uses the debug bits of dmel_forward's flags (0x100 skip the mel contraction, 0x200 skip the FFT phase)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("DMEL_LIB", os.path.join(ROOT, "differentiable-mel-spectrogram_amd", "build", "libdmel_hip_ablate.so"))
import dmel_amd
from dmel_amd import capi, synth
sys.path.insert(0, ROOT)
from bench import CONFIGS

names = sys.argv[1:] or ["c2"]
for name in names:
    B, L, sr, lam, hop, M = CONFIGS[name]
    T = L // hop + 1
    x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
    out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out)
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    s = torch.cuda.current_stream().cuda_stream
    for label, fl, tg in (("full train", 0, True), ("fft only (train)", 0x100, True), ("gemm+epilogue only", 0x200, True),
                          ("gemm only, no epilogue", 0x600, True), ("epilogue only, no mfma", 0xA00, True),
                          ("neither", 0x300, True), ("no clip sum (mean = 0)", 0x1000, True), ("fft only, no clip sum", 0x1100, True),
                          ("full infer", 0, False), ("fft only (infer)", 0x100, False)):
        for _ in range(5):
            plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr() if tg else None, True, 1e-10, s, extra_flags=fl)
        torch.cuda.synchronize()
        plan.set_profiling(True)
        for _ in range(50):
            plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr() if tg else None, True, 1e-10, s, extra_flags=fl)
        torch.cuda.synchronize()
        pr = plan.get_profile(); plan.set_profiling(False)
        print(f"{name} {label:22s} fwd {1e3*pr['fwd_ms']/pr['fwd_launches']:8.2f} us   prep {1e3*pr['prep_ms']/max(1,pr['prep_launches']):6.2f} us   info {plan.info()}")
