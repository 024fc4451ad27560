#!/usr/bin/env python3
"""A/B timing of flag-selected variants of ONE library on ONE box, alternating (trains of launches between two HIP events).
usage: DMEL_LIB=<lib.so> python tools/xtime.py <config> <hex-flags>[,<hex-flags>...] [launches] [rounds]
Flags are the debug bits of a -DDMEL_ABLATE build (csrc/dmel_fwd.hip); 0 is the plain kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import capi, synth
from bench import CONFIGS

name = sys.argv[1]
flags = [int(f, 16) for f in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
train = os.environ.get("XTIME_MODE", "train") == "train"
EXTRA = {"c4": (2048, 16000, 16000, 128.0, 512, 128), "esc_n4096": (32, 40000, 8000, 400.0, 80, 64)}
B, L, sr, lam, hop, M = CONFIGS[name] if name in CONFIGS else EXTRA[name]
T = L // hop + 1
x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out)
plan = capi.Plan(L, hop, M, sr, max_batch=B)
s = torch.cuda.current_stream().cuda_stream
best = {f: 1e9 for f in flags}
for f in flags:
    for _ in range(20):
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr() if train else None, True, 1e-10, s, extra_flags=f)
torch.cuda.synchronize()
for rep in range(rounds):
    for f in flags:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr() if train else None, True, 1e-10, s, extra_flags=f)
        e1.record()
        torch.cuda.synchronize()
        best[f] = min(best[f], 1e3 * e0.elapsed_time(e1) / n)
print(os.path.basename(os.environ.get("DMEL_LIB", "libdmel_hip.so")), name, " ".join(f"{f:#x}:{best[f]:.2f}" for f in flags), "us/launch")
