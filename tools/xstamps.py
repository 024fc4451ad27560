#!/usr/bin/env python3
"""Phase timeline of dmel_xgrad_wave_kernel (GPU box; diagnostic build: python tools/stamps.py build first).
  python tools/xstamps.py [c2|c3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "differentiable-mel-spectrogram_amd", "build", "libdmel_hip_stamps.so")
os.environ["DMEL_LIB"] = LIB
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import dmel_amd
from dmel_amd import capi

CONFIGS = {"c2": (256, 16000, 512, 128, 16000, 128.0), "c3": (32, 160000, 512, 128, 16000, 256.0)}
NAMES = ["start -> prologue issued", "prologue loads arrive (barrier)", "first transform", "bin pass", "second transform", "barrier",
         "overlap-add + segment store", "tile sum"]
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
B, L, hop, M, sr, lam = CONFIGS[name]
plan = capi.Plan(L, hop, M, sr)
x = 0.1 * torch.randn(B, L, device="cuda:0")
T = L // hop + 1
g = torch.randn(B, 1, M, T, device="cuda:0")
y = torch.empty_like(g)
st = torch.cuda.current_stream().cuda_stream
plan.forward(x.data_ptr(), B, lam, y.data_ptr(), None, True, 1e-10, st)
gx = torch.empty_like(x)
for _ in range(3):
    plan.backward_x(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gx.data_ptr(), True, st)
torch.cuda.synchronize()
n = capi.n_fft(lam)
waves = 8 if n == 1024 else 4
SL = 16
buf = np.zeros(4096 * 8 * SL, dtype=np.uint64)
lib = capi.load()
lib.dmel_debug_read_xstamps.argtypes = [C.c_void_p, C.c_int]
assert lib.dmel_debug_read_xstamps(buf.ctypes.data, buf.size) == 0
full = buf.reshape(4096, 8, SL).astype(np.int64)
nwg = int((full[:, 0, 0] != 0).sum())
s = full[:nwg, :waves, :9]
t_first = s[:, :, 0].min()
tot = s[:, :, 8] - s[:, :, 0]
print(f"{name}: n_fft {n}, {nwg} workgroups (of the first 4096) x {waves} waves; span first start -> last end {s[:, :, 8].max() - t_first} cycles")
print(f"per-wave lifetime: median {np.median(tot):.0f}  max {tot.max()}")
d = np.diff(s, axis=2)
for i, nm in enumerate(NAMES):
    v = d[:, :, i]
    print(f"  {nm:36s} median {np.median(v):8.0f}   p90 {np.percentile(v, 90):8.0f}   max {v.max():8d}   share {100 * np.median(v) / np.median(tot):5.1f} %")
start = s[:, 0, 0] - t_first
print("workgroup start times: p10 %d  median %d  p90 %d  max %d" % (np.percentile(start, 10), np.median(start), np.percentile(start, 90), start.max()))
