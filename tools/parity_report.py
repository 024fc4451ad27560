#!/usr/bin/env python3
"""Prints, for every golden case, the HIP path's error against the reference fixture and the oracle
(GPU box only).  Diagnostic companion of tests/test_hip_parity.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch
import cases as C
from oracle import dmel_oracle as O
from dmel_amd import capi


def rel(got, exp):
    scale = np.maximum(np.abs(exp), 1e-6 * np.abs(exp).max() + 1e-30)
    return float((np.abs(got.astype(np.float64) - exp.astype(np.float64)) / scale).max())


print(f"{'case':18s} {'nfft':>5s} {'mel_rel(gold)':>13s} {'mel_rel(orc)':>13s} {'tan_rel(orc)':>13s} {'dlin_rel':>10s} {'dlog_rel':>10s} {'cond_lin':>9s}")
for case in C.CASES:
    if not case["optimized"]:
        continue
    gold = C.load(case)
    x_np = C.make_input(case).astype(np.float32)
    g_np = C.make_cotangent(case)
    x = torch.from_numpy(x_np).cuda(); g = torch.from_numpy(g_np).cuda()
    plan = capi.Plan(case["L"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"], case["normalize_window"])
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for log in (False, True):
        out = torch.empty(C.out_shape(case), dtype=torch.float32, device="cuda"); tan = torch.empty_like(out)
        dl = torch.zeros(1, device="cuda")
        plan.forward(x.data_ptr(), case["B"], case["lambd"], out.data_ptr(), tan.data_ptr(), log, 1e-10, s)
        plan.backward(g.data_ptr(), tan.data_ptr(), out.numel(), dl.data_ptr(), s)
        torch.cuda.synchronize()
        res[log] = (out.cpu().numpy(), tan.cpu().numpy(), float(dl))
    o_ref, t_ref = O.forward(x_np, case["lambd"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"], case["normalize_window"])
    idx = C.sample_index(case)
    exp = gold["mel"].reshape(-1) if idx is None else gold["mel_sampled"]
    got = res[False][0].reshape(-1) if idx is None else res[False][0].reshape(-1)[idx]
    tscale = np.abs(t_ref).max() + 1e-30
    dlin, dlog = res[False][2], res[True][2]
    el = abs(dlin - float(gold["dlam_lin"])) / (abs(float(gold["dlam_lin"])) + 1e-30)
    eg = abs(dlog - float(gold["dlam_log"])) / (abs(float(gold["dlam_log"])) + 1e-30)
    cond = float(np.abs(g_np.astype(np.float64) * t_ref).sum() / (abs(float(gold["dlam_lin"])) + 1e-30))
    print(f"{case['name']:18s} {int(gold['n_fft']):5d} {rel(got, exp):13.3e} {rel(res[False][0], o_ref):13.3e} "
          f"{float(np.abs(res[False][1]-t_ref).max())/tscale:13.3e} {el:10.2e} {eg:10.2e} {cond:9.2e}")
