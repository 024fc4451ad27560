#!/usr/bin/env python3
"""Debug aid: power spectrogram of the fused kernel (kSpec: two frames per FFT; kSpecTrain: frame + tangent) against a numpy fp64
STFT, bin by bin.  usage: tools/dbg_spec.py <lambd> [L hop B]   (prints which bins / frames are off)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dmel_amd
from dmel_amd import capi

lam = float(sys.argv[1]) if len(sys.argv) > 1 else 256.0
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
hop = int(sys.argv[3]) if len(sys.argv) > 3 else 512
B = int(sys.argv[4]) if len(sys.argv) > 4 else 3
N = capi.n_fft(lam)
F, T = N // 2 + 1, L // hop + 1
rng = np.random.default_rng(5)
x_np = (rng.standard_normal((B, L)) + 0.3).astype(np.float32)
x = torch.from_numpy(x_np).cuda()
plan = capi.Plan(L, hop, 64, 16000, 0.0, None, False)
s = torch.cuda.current_stream().cuda_stream

n = np.arange(N, dtype=np.float64)
d = n - N / 2
den = abs(lam) + 1e-15
w = np.exp(-0.5 * (d / den) ** 2)
dw = w * d * d / den ** 3 * np.sign(lam)
xm = x_np.astype(np.float64) - x_np.astype(np.float64).mean(axis=1, keepdims=True)
xp = np.pad(xm, ((0, 0), (N // 2, N // 2)))
ref = np.empty((B, F, T)); dref = np.empty((B, F, T))
for t in range(T):
    fr = xp[:, t * hop:t * hop + N]
    S = np.fft.rfft(fr * w, axis=1); D = np.fft.rfft(fr * dw, axis=1)
    ref[:, :, t] = np.abs(S) ** 2
    dref[:, :, t] = 2 * (S.real * D.real + S.imag * D.imag)

def report(name, got, want):
    sc = np.abs(want).max()
    err = np.abs(got - want) / sc
    bad = np.argwhere(err > 1e-4)
    print(f"{name}: max err / max {err.max():.3e}; bad entries {len(bad)} of {err.size}")
    if len(bad):
        ks = sorted(set(int(k) for _, k, _ in bad)); ts = sorted(set(int(t) for _, _, t in bad))
        print("   bins:", ks[:40], "..." if len(ks) > 40 else "", f"({len(ks)} distinct)")
        print("   frames:", ts[:40], f"({len(ts)} distinct)")

spec = torch.empty(B, F, T, device="cuda")
plan.spectrogram_ex(x.data_ptr(), B, lam, N, spec.data_ptr(), None, s, remove_dc=True)
torch.cuda.synchronize()
report(f"kSpec n_fft {N}", spec.cpu().numpy().astype(np.float64), ref)
tan = torch.empty_like(spec)
plan.spectrogram_ex(x.data_ptr(), B, lam, N, spec.data_ptr(), tan.data_ptr(), s, remove_dc=True)
torch.cuda.synchronize()
report(f"kSpecTrain n_fft {N} spec", spec.cpu().numpy().astype(np.float64), ref)
report(f"kSpecTrain n_fft {N} tangent", tan.cpu().numpy().astype(np.float64), dref)

if os.environ.get("DBG_PERM"):
    g = spec.cpu().numpy().astype(np.float64)[0, :, 5]; w_ = ref[0, :, 5]
    for k in list(range(0, 40)) + list(range(F - 8, F)):
        j = int(np.argmin(np.abs(w_ - g[k]) / (np.abs(w_) + 1e-30)))
        print(k, f"got {g[k]:.5e} want {w_[k]:.5e} nearest ref bin {j} ({w_[j]:.5e})")

if os.environ.get("DBG_PD"):
    # library built with -DDMEL_DBG_PD=1 (PD := own Z) or =2 (PD := partner's Z): kSpec output = 0.25 * (re, im) in frames (2s, 2s+1)
    plan.spectrogram_ex(x.data_ptr(), B, lam, N, spec.data_ptr(), None, s, remove_dc=True)
    torch.cuda.synchronize()
    g = spec.cpu().numpy().astype(np.float64)[0]
    t = 4
    z = np.fft.fft((xp[0, t * hop:t * hop + N] + 1j * xp[0, (t + 1) * hop:(t + 1) * hop + N]) * w)
    gz = 4 * (g[:, t] + 1j * g[:, t + 1])
    for k in list(range(0, 36)) + [255, 256, 257, 511, 512, N // 2 - 1, N // 2]:
        if k >= F: continue
        j = int(np.argmin(np.abs(z - gz[k]))); jc = int(np.argmin(np.abs(np.conj(z) - gz[k])))
        print(k, f"got {gz[k]:.4f}  Z[k] {z[k]:.4f}  Z[N-k] {z[(N - k) % N]:.4f}  nearest Z[{j}] err {abs(z[j]-gz[k]):.2e}  nearest conj Z[{jc}] err {abs(np.conj(z[jc])-gz[k]):.2e}")
