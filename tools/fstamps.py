#!/usr/bin/env python3
"""Phase timeline of dmel_fbgrad_lds_kernel (GPU box; diagnostic build: python tools/stamps.py build first).  python tools/fstamps.py [c2|c3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DMEL_LIB"] = os.path.join(ROOT, "differentiable-mel-spectrogram_amd", "build", "libdmel_hip_stamps.so")
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import dmel_amd
from dmel_amd import capi

CONFIGS = {"c2": (256, 16000, 512, 128, 16000, 128.0), "c3": (32, 160000, 512, 128, 16000, 256.0)}
NAMES = ["start -> first requests issued", "first block arrives, parked (barrier)", "K loop", "partial tile stores issued"]
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
B, L, hop, M, sr, lam = CONFIGS[name]
plan = capi.Plan(L, hop, M, sr)
x = 0.1 * torch.randn(B, L, device="cuda:0")
T = L // hop + 1
g = torch.randn(B, 1, M, T, device="cuda:0")
y = torch.empty_like(g)
st = torch.cuda.current_stream().cuda_stream
plan.forward(x.data_ptr(), B, lam, y.data_ptr(), None, True, 1e-10, st)
n = capi.n_fft(lam)
gfb = torch.empty(n // 2 + 1, M, device="cuda:0")
for _ in range(3):
    plan.backward_fb(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gfb.data_ptr(), False, st)
torch.cuda.synchronize()
buf = np.zeros(1024 * 8 * 8, dtype=np.uint64)
lib = capi.load()
lib.dmel_debug_read_fstamps.argtypes = [C.c_void_p, C.c_int]
assert lib.dmel_debug_read_fstamps(buf.ctypes.data, buf.size) == 0
full = buf.reshape(1024, 8, 8).astype(np.int64)
nwg = int((full[:, 0, 0] != 0).sum())
s = full[:nwg, :, :5]
tot = s[:, :, 4] - s[:, :, 0]
print(f"{name}: n_fft {n}, {nwg} workgroups x 8 waves; per-wave lifetime median {np.median(tot):.0f} max {tot.max()}")
d = np.diff(s, axis=2)
for i, nm in enumerate(NAMES):
    v = d[:, :, i]
    print(f"  {nm:40s} median {np.median(v):8.0f}   p90 {np.percentile(v, 90):8.0f}   max {v.max():8d}   share {100 * np.median(v) / np.median(tot):5.1f} %")
