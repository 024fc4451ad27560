#!/usr/bin/env python3
"""Where does a workgroup of the fused forward kernel spend its cycles?  (GPU box; diagnostic build.)

Builds libdmel_hip_stamps.so with -DDMEL_STAMPS (s_memtime stamps at the phase boundaries, written to a
buffer nothing else reads), runs config 2 once and prints, per phase, the median / max wave time in
shader cycles (100 MHz s_memtime ticks are NOT used: s_memtime counts shader clocks on gfx950).
Read the SHARES, not the total: the stamps fence overlaps the real kernel has.

  python tools/stamps.py build      (dev container: cross-compiles)
  python tools/stamps.py run [c2]   (GPU box)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "differentiable-mel-spectrogram_amd")
LIB = os.path.join(PKG, "build", "libdmel_hip_stamps.so")
NAMES = ["start->loads issued", "window table + clip mean", "samples arrive, window", "radix-R #1", "twiddle + LDS transposition",
         "radix-R #2", "twiddle + radix-C + Z store", "barrier wait", "MFMA loops", "half-tile exchange", "epilogue"]

if sys.argv[1] == "build":
    srcs = [os.path.join(PKG, "csrc", f) for f in ("dmel_fwd.hip", "dmel_aux.hip", "dmel_big.hip", "dmel_xgrad.hip", "dmel_api.cpp", "dmel_comm.cpp")]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DDMEL_STAMPS", "-shared", "-o", LIB] + sys.argv[2:]   # e.g. -DDMEL_ONLY_NFFT=1024
    for s in srcs:
        cmd += ["-x", "hip", s]
    cmd += ["-ldl"]
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(cmd)
    print(LIB)
    sys.exit(0)

os.environ["DMEL_LIB"] = LIB
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import dmel_amd
from dmel_amd import capi, synth
from bench import CONFIGS

name = sys.argv[2] if len(sys.argv) > 2 else "c2"
# besides bench.py's configs: the reference's ESC-50 shape (search_spaces.py:4-33) at its three starting lambd and one drift case
EXTRA = {"esc_n128": (32, 40000, 8000, 8000 * 0.01 / 6, 80, 64), "esc_n512": (32, 40000, 8000, 8000 * 0.035 / 6, 80, 64),
         "esc_n4096": (32, 40000, 8000, 400.0, 80, 64), "esc_n8192": (32, 40000, 8000, 700.0, 80, 64)}
B, L, sr, lam, hop, M = CONFIGS[name] if name in CONFIGS else EXTRA[name]
T = L // hop + 1
x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out)
plan = capi.Plan(L, hop, M, sr, max_batch=B)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, s)
torch.cuda.synchronize()
info = plan.info()
nwg = min(info["grid_fwd"], 4096)
waves = 8 if info["n_fft"] >= 1024 else 4
SL = 32
buf = np.zeros(4096 * 8 * SL, dtype=np.uint64)
L_ = capi.load()
L_.dmel_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
rc = L_.dmel_debug_read_stamps(buf.ctypes.data, buf.size)
assert rc == 0, rc
full = buf.reshape(4096, 8, SL)[:nwg, :waves, :].astype(np.int64)
st = full[:, :, :12]
two_tiles = bool((full[:, :, 16 + 11] != 0).any())
t_first = st[:, :, 0].min()
print(f"{name}: n_fft {info['n_fft']}, {nwg} workgroups x {waves} waves; kernel span (first start -> last end) "
      f"{(st[:, :, 11].max() - t_first)} cycles")
d = np.diff(st, axis=2)          # (wg, wave, 11)
tot = (st[:, :, 11] - st[:, :, 0])
print(f"per-wave lifetime: median {np.median(tot):.0f}  max {tot.max()} cycles")
for i, nm in enumerate(NAMES):
    v = d[:, :, i]
    print(f"  {nm:32s} median {np.median(v):8.0f}   p90 {np.percentile(v, 90):8.0f}   max {v.max():8d}   share {100 * np.median(v) / np.median(tot):5.1f} %")
start = st[:, 0, 0] - t_first
print("workgroup start times (cycles after the first): p10 %d  median %d  p90 %d  max %d" %
      (np.percentile(start, 10), np.median(start), np.percentile(start, 90), start.max()))
# first round (the workgroups the dispatcher places at once) against the later ones
cut = np.sort(start)[min(len(start) - 1, 2 * 256 - 1)] if len(start) > 512 else start.max()
for label, sel in (("round 1 (placed at launch)", start <= cut), ("later rounds", start > cut)):
    if sel.sum() == 0:
        continue
    tt = tot[sel]
    print(f"{label}: {sel.sum()} workgroups, wave lifetime median {np.median(tt):.0f} max {tt.max()}")
    for i, nm in enumerate(NAMES):
        v = d[sel][:, :, i]
        print(f"    {nm:32s} median {np.median(v):8.0f}   p90 {np.percentile(v, 90):8.0f}")
# placement: HW_ID bits (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (3 bits; gfx950 uses more SEs: take 16:13); XCC_ID 3:0
raw = buf.reshape(4096, 8, SL)[:nwg, 0, :]
hw, xcc = raw[:, 12].astype(np.int64), raw[:, 13].astype(np.int64) & 0xF
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0xF
place = xcc * 10000 + se * 100 + sh * 50 + cu
uniq = len(set(place.tolist()))
print(f"placement: {uniq} distinct (xcc, se, sh, cu) over {nwg} workgroups; XCC of the first 16 workgroups: {xcc[:16].tolist()}")
t0w = st[:, 0, 0]
for x in range(int(xcc.max()) + 1):
    sel = np.flatnonzero(xcc == x)
    base = t0w[sel].min()
    order = sel[np.argsort(t0w[sel])]
    firsts = [(int(i), int(t0w[i] - base), int(place[i] % 10000)) for i in order[:10]]
    print(f"  xcc {x}: {len(sel)} workgroups; first ten (blockIdx, start-cycles, se*100+sh*50+cu): {firsts}")
    # per CU: start times of its workgroups
    cus = {}
    for i in sel:
        cus.setdefault(int(place[i]), []).append((int(t0w[i] - base), int(i)))
    ex = sorted(cus.items())[:3]
    for c, lst in ex:
        print(f"     cu {c % 10000}: {sorted(lst)}")
if two_tiles:
    # second tile of two-tile workgroups: slots 16 + (2 .. 11); slot 18 = the tile begins (tile 0's epilogue stores are issued)
    t2 = np.concatenate([full[:, :, 11:12], full[:, :, 18:28]], axis=2)
    ok = (full[:, :, 27] != 0)
    d2 = np.diff(t2, axis=2)
    life2 = (full[:, :, 27] - full[:, :, 0])[ok]
    print(f"two tiles per workgroup: wave lifetime over both tiles median {np.median(life2):.0f} max {life2.max()} cycles")
    labels = ["tile 0 epilogue issued -> tile 1 begins", "barriers between the tiles, window rewrite, samples arrive, window"] + NAMES[3:]
    for i, nm in enumerate(labels):
        v = d2[:, :, i][ok]
        print(f"    tile 1: {nm:68s} median {np.median(v):8.0f}   p90 {np.percentile(v, 90):8.0f}")
end = st[:, :, 11].max(axis=1) - t_first
print("workgroup end times: p10 %d  median %d  p90 %d  max %d" % (np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
