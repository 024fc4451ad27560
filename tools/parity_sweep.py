#!/usr/bin/env python3
"""A larger seeded parity sweep than the test suite runs (tests/test_hip_random_shapes.py: 72 configurations): N random configurations
from fresh seeds -- clip length, hop, lambd (every kernel path), n_mels, sample rate, band limits, window normalisation, batch --
through the nn.Module (training forward + d lambd, linear and log) against the fp64 CPU oracle.  One JSON report:
  python tools/parity_sweep.py [N=400] > gpurun_out/r04_parity_sweep.json        (GPU box; the oracle takes ~a second per configuration)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch
import cases as C
from oracle import dmel_oracle as O
import test_hip_random_shapes as R
from test_hip_parity import _layer, parity_stats

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cases = []
for seed in (31337, 4242, 99, 123456):
    cases += [dict(c, name=f"s{seed}_" + c["name"]) for c in R._random_cases(n // 4, seed=seed)]
worst = {"mel_plain_rel": (0.0, None), "logmel_plain_rel": (0.0, None), "dlam_rel_well_conditioned": (0.0, None), "dlam_over_cancellation_sum": (0.0, None)}
hist = {}
fails = []
for case in cases:
    x_np = C.make_input(case).astype(np.float32)
    g_np = C.make_cotangent(case)
    x, g = torch.from_numpy(x_np).to("cuda:0"), torch.from_numpy(g_np).to("cuda:0")
    nf = O.n_fft(case["lambd"])
    hist[nf] = hist.get(nf, 0) + 1
    for log in (False, True):
        layer = _layer(case, log=log)
        y = layer(x)
        (y * g).sum().backward()
        o_ref, t_ref = O.forward(x_np, case["lambd"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"], case["normalize_window"], apply_log=log)
        o = y.detach().cpu().numpy()
        st = parity_stats(np.exp(o.astype(np.float64)), np.exp(o_ref.astype(np.float64))) if log else parity_stats(o, o_ref)
        key = "logmel_plain_rel" if log else "mel_plain_rel"
        if st["plain_max_rel"] > worst[key][0]:
            worst[key] = (st["plain_max_rel"], case["name"])
        if st["plain_max_rel"] > 1e-4 or st["max_abs_where_exp_is_zero"] > 1e-30:
            fails.append((case["name"], log, st))
        exp_d = O.backward(g_np, t_ref)
        got_d = float(layer.lambd.grad)
        cancel = float(np.abs(g_np.astype(np.float64) * t_ref.astype(np.float64)).sum())
        if abs(exp_d) > 1e-3 * cancel:
            r = abs(got_d - exp_d) / abs(exp_d)
            if r > worst["dlam_rel_well_conditioned"][0]:
                worst["dlam_rel_well_conditioned"] = (r, case["name"])
            if r > 1e-4:
                fails.append((case["name"], log, "dlam", got_d, exp_d))
        r2 = abs(got_d - exp_d) / (cancel + 1e-30)
        if r2 > worst["dlam_over_cancellation_sum"][0]:
            worst["dlam_over_cancellation_sum"] = (r2, case["name"])
print(json.dumps({"_how": "tools/parity_sweep.py: seeded random configurations (generator of tests/test_hip_random_shapes.py, four fresh seeds) through the nn.Module "
                          "against the fp64 oracle; plain relative errors on EVERY element (no floor); d lambd relative where |d lambd| > 1e-3 sum|g t|, and over that sum everywhere",
                  "configurations": len(cases), "n_fft_histogram": {str(k): v for k, v in sorted(hist.items())},
                  "worst": {k: {"value": v[0], "case": v[1]} for k, v in worst.items()}, "above_the_bar": fails}, indent=1))
