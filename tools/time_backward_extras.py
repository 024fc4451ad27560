#!/usr/bin/env python3
"""Times the optional backward outputs at BASELINE config 2 / 3: dmel_backward_fb (spectrogram recompute + grad_fb GEMM +
slice reduction) and dmel_backward_x (frame gradients + ordered overlap-add)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmel_amd  # noqa: E402
from dmel_amd import capi  # noqa: E402

CONFIGS = {"c2": (256, 16000, 512, 128, 16000, 128.0), "c3": (32, 160000, 512, 128, 16000, 256.0)}


def main():
    res = {}
    for name, (B, L, hop, M, sr, lam) in CONFIGS.items():
        plan = capi.Plan(L, hop, M, sr)
        x = 0.1 * torch.randn(B, L, device="cuda:0")
        T = L // hop + 1
        g = torch.randn(B, 1, M, T, device="cuda:0")
        y = torch.empty_like(g)
        st = torch.cuda.current_stream().cuda_stream
        plan.forward(x.data_ptr(), B, lam, y.data_ptr(), None, True, 1e-10, st)
        n = capi.n_fft(lam)
        gfb = torch.empty(n // 2 + 1, M, device="cuda:0")
        for log in (False, True):
            for _ in range(5):
                plan.backward_fb(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gfb.data_ptr(), log, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                plan.backward_fb(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gfb.data_ptr(), log, st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 50
            flops = 2.0 * B * T * (n // 2 + 1) * M
            plan.set_profiling(True)
            for _ in range(20):
                plan.backward_fb(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gfb.data_ptr(), log, st)
            torch.cuda.synchronize()
            pr = plan.get_profile()
            plan.set_profiling(False)
            spec_us = 1e3 * pr["fwd_ms"] / max(1, pr["fwd_launches"])
            gemm_us = 1e3 * pr["bwd_ms"] / max(1, pr["bwd_launches"])
            res[f"{name}_{'log' if log else 'lin'}"] = dict(us=round(us, 2), spectrogram_us=round(spec_us, 2), gemm_and_reduce_us=round(gemm_us, 2),
                                                            gemm_tflops=round(flops / gemm_us / 1e6, 2))
        gx = torch.empty_like(x)
        for log in (False, True):
            for _ in range(3):
                plan.backward_x(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gx.data_ptr(), log, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                plan.backward_x(x.data_ptr(), B, lam, g.data_ptr(), y.data_ptr(), gx.data_ptr(), log, st)
            e1.record()
            torch.cuda.synchronize()
            res[f"{name}_xgrad_{'log' if log else 'lin'}"] = dict(us=round(e0.elapsed_time(e1) * 1000 / 20, 2))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
