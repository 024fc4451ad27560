#!/usr/bin/env python3
"""dmel_backward_fb alone at BASELINE config 2 / 3 (for rocprofv3 --kernel-trace --stats / --pmc, or event timing)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmel_amd  # noqa: E402
from dmel_amd import capi  # noqa: E402

CONFIGS = {"c2": (256, 16000, 512, 128, 16000, 128.0), "c3": (32, 160000, 512, 128, 16000, 256.0)}


def main():
    names = [a for a in sys.argv[1:] if a in CONFIGS] or ["c2"]
    logs = [True] if "log" in sys.argv else ([False] if "lin" in sys.argv else [False, True])
    reps = 50
    res = {}
    for name in names:
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
        for log in logs:
          ref = None
          for mode, fl in (("fp32", 0), ("bf16x3", capi.DMEL_FLAG_MFMA_BF16X3)):
            for _ in range(5):
                plan.backward_fb(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gfb.data_ptr(), log, st, extra_flags=fl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                plan.backward_fb(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gfb.data_ptr(), log, st, extra_flags=fl)
            e1.record()
            torch.cuda.synchronize()
            res[f"{name}_{'log' if log else 'lin'}_{mode}"] = round(e0.elapsed_time(e1) * 1000 / reps, 2)
            # the same on a SAVED spectrogram (dmel_backward_fb_saved: what a layer with save_spec=True runs): no recompute
            spec = torch.empty(B, n // 2 + 1, T, device="cuda:0")
            plan.spectrogram(x.data_ptr(), B, lam, spec.data_ptr(), st, remove_dc=True)
            for _ in range(5):
                plan.backward_fb_saved(spec.data_ptr(), B, n, g.data_ptr(), y.data_ptr(), gfb.data_ptr(), log, st, extra_flags=fl)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                plan.backward_fb_saved(spec.data_ptr(), B, n, g.data_ptr(), y.data_ptr(), gfb.data_ptr(), log, st, extra_flags=fl)
            e1.record()
            torch.cuda.synchronize()
            res[f"{name}_{'log' if log else 'lin'}_{mode}_saved_spec"] = round(e0.elapsed_time(e1) * 1000 / reps, 2)
            plan.backward_fb(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gfb.data_ptr(), log, st, extra_flags=fl)
            torch.cuda.synchronize()
            if ref is None:
                ref = gfb.clone()
            else:
                res[f"{name}_{'log' if log else 'lin'}_{mode}_max_err_over_max"] = float((gfb - ref).abs().max() / ref.abs().max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
