#!/usr/bin/env python3
"""Kernel time of the fused forward (C ABI, lambd by value) from a train of launches between two HIP events.
usage: DMEL_LIB=<lib.so> python tools/ktime.py [c2] [train|infer] [launches]   -> one line: config mode us-per-launch"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import capi, synth
from bench import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
# besides bench.py's configs: BASELINE config 4's global batch on one GPU, and the reference's ESC-50 shape (search_spaces.py:4-33) at its
# largest starting lambd (n_fft 4096) and one lambd a run drifts to (n_fft 8192)
EXTRA = {"c4": (2048, 16000, 16000, 128.0, 512, 128), "esc_n4096": (32, 40000, 8000, 400.0, 80, 64), "esc_n8192": (32, 40000, 8000, 700.0, 80, 64),
         "esc_n512": (32, 40000, 8000, 8000 * 0.035 / 6, 80, 64), "esc_n128": (32, 40000, 8000, 8000 * 0.01 / 6, 80, 64),
         "c1x64": (256, 16000, 16000, 64.0, 256, 64)}       # (BASELINE config 1's layer on 256 clips: n_fft 512 on short clips, no prep kernel)
B, L, sr, lam, hop, M = CONFIGS[name] if name in CONFIGS else EXTRA[name]
T = L // hop + 1
x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out)
plan = capi.Plan(L, hop, M, sr, max_batch=B)
s = torch.cuda.current_stream().cuda_stream
tp = tan.data_ptr() if mode == "train" else None
for _ in range(20):
    plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tp, True, 1e-10, s)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tp, True, 1e-10, s)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, 1e3 * e0.elapsed_time(e1) / n)
print(os.path.basename(os.environ.get("DMEL_LIB", "libdmel_hip.so")), name, mode, round(best, 2), "us/launch (incl. prep for long clips)", plan.info()["lds_bytes"])
