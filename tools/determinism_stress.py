#!/usr/bin/env python3
"""Run-to-run determinism of the fused forward under load: the same launch repeated, every output compared bit for bit with the
first.  usage: python tools/determinism_stress.py [reps]  (config-2-like shapes at n_fft 512 / 1024 / 2048, HTK and dense banks)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import capi, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
s = torch.cuda.current_stream().cuda_stream
bad_total = 0
for (B, L, sr, lam, hop, M, dense) in ((8, 16000, 16000, 128.0, 512, 128, True), (8, 16000, 16000, 128.0, 512, 128, False),
                                       (256, 16000, 16000, 128.0, 512, 128, True), (256, 16000, 16000, 128.0, 512, 128, False),
                                       (8, 16000, 16000, 64.0, 256, 64, True), (4, 40000, 16000, 256.0, 512, 128, True)):
    T = L // hop + 1
    x = torch.from_numpy(synth.waveforms(B, L, seed=3)).cuda()
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    n = capi.n_fft(lam)
    if dense:
        fb = torch.rand((n // 2 + 1, M), device="cuda") + 0.01
        plan.set_filterbank_dev(n, fb.data_ptr(), s)
    for train in (True, False):
        ref_o = torch.empty((B, 1, M, T), device="cuda"); ref_t = torch.empty_like(ref_o)
        plan.forward(x.data_ptr(), B, lam, ref_o.data_ptr(), ref_t.data_ptr() if train else None, True, 1e-10, s)
        torch.cuda.synchronize()
        bad = 0
        outs = [(torch.empty_like(ref_o), torch.empty_like(ref_o)) for _ in range(8)]
        for r in range(reps):
            o, t = outs[r % 8]
            plan.forward(x.data_ptr(), B, lam, o.data_ptr(), t.data_ptr() if train else None, True, 1e-10, s)
            if r % 8 == 7:
                torch.cuda.synchronize()
                for (oo, tt) in outs:
                    if not torch.equal(oo, ref_o) or (train and not torch.equal(tt, ref_t)):
                        bad += 1
        print(f"B {B} L {L} n_fft {n} dense {dense} train {train}: {bad} of {reps // 8 * 8} launches differ from the first", flush=True)
        bad_total += bad
print("TOTAL", bad_total)
