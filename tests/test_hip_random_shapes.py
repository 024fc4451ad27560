"""GPU parity sweep over seeded random configurations: clip length, hop, lambd (every kernel path: direct DFT, the fused kernel
with part of a wave / one wave / several waves per frame), n_mels, sample rate, band limits, window normalisation, batch.
Each configuration: training forward (+ tangent through d lambd), inference forward, log and linear outputs, against the CPU
oracle (oracle/, pinned to the reference's goldens by tests/test_oracle_golden.py).  Tolerances as in test_hip_parity.py."""
import numpy as np
import pytest
import torch

import cases as C
from oracle import dmel_oracle as O
from test_hip_parity import TOL, _dlam_tol, _layer, _log_err, _rel_err, assert_parity

pytestmark = pytest.mark.gpu


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    # log-uniform lambd so that every n_fft from 4 to 16384 is drawn; clip lengths from shorter than a frame to a few frames
    while len(out) < n:
        n_fft_target = int(2 ** rng.integers(2, 15))                       # 4 ... 16384
        lam = float(rng.uniform(0.55, 0.99) * n_fft_target / 6.0) * (1.0 if rng.random() < 0.8 else -1.0)
        nf = O.n_fft(lam)
        L = int(rng.integers(max(8, nf // 4), max(64, min(6 * nf, 30000))))
        hop = int(rng.integers(1, max(2, L // 3))) if rng.random() < 0.3 else int(rng.integers(max(1, nf // 8), max(2, nf)))
        T = L // hop + 1
        if T > 400:
            hop = max(1, L // 300)
            T = L // hop + 1
        sr = int(rng.choice([8000, 16000, 22050, 44100]))
        n_mels = int(rng.choice([1, 7, 16, 40, 64, 80, 128, 130]))
        f_min = float(rng.choice([0.0, 0.0, 50.0, 300.0]))
        f_max = None if rng.random() < 0.6 else float(rng.uniform(0.3, 0.5) * sr)
        if f_max is not None and f_max <= f_min + 100.0:
            f_max = None
        B = int(rng.integers(1, 5))
        if B * T * nf > 6_000_000:                                         # keep the oracle in seconds
            continue
        out.append(dict(C.BY_NAME["g1_c1"], name=f"r{len(out)}_n{nf}_L{L}_h{hop}_m{n_mels}", B=B, L=L, lambd=lam, hop=hop, n_mels=n_mels,
                        sr=sr, f_min=f_min, f_max=f_max, normalize_window=bool(rng.random() < 0.4), seed=1000 + len(out)))
    return out


def _assert_dlam(got_d, exp_d, g_np, t_ref, name):
    """rel 1e-4 with the fp32 floor of an ill-conditioned sum (_dlam_tol) always; and north_star's plain relative 1e-4, without
    any floor, wherever the sum is not dominated by cancellation (|d lambd| more than 1e-3 of sum |g t|) -- as
    test_matches_reference_golden does for the fixtures"""
    assert abs(got_d - exp_d) <= _dlam_tol(exp_d, g_np, t_ref), (name, got_d, exp_d)
    cancel = float(np.abs(g_np.astype(np.float64) * t_ref.astype(np.float64)).sum())
    if abs(exp_d) > 1e-3 * cancel:
        assert abs(got_d - exp_d) <= TOL * abs(exp_d), (name, got_d, exp_d, abs(got_d - exp_d) / abs(exp_d))



RANDOM_CASES = _random_cases(36, seed=20240607) + [dict(c, name="s2_" + c["name"]) for c in _random_cases(36, seed=977)]


@pytest.mark.parametrize("case", RANDOM_CASES, ids=[c["name"] for c in RANDOM_CASES])
def test_random_configuration_matches_oracle(case):
    x_np = C.make_input(case).astype(np.float32)
    g_np = C.make_cotangent(case)
    x = torch.from_numpy(x_np).to("cuda:0")
    g = torch.from_numpy(g_np).to("cuda:0")
    for log in (False, True):
        layer = _layer(case, log=log)
        y = layer(x)
        assert y.shape == C.out_shape(case)
        (y * g).sum().backward()
        o_ref, t_ref = O.forward(x_np, case["lambd"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"],
                                 case["normalize_window"], apply_log=log)
        o = y.detach().cpu().numpy()
        assert (_log_err(o, o_ref) if log else _rel_err(o, o_ref)) <= TOL
        # the same comparison with nothing hidden: the PLAIN relative error of every element (no floor)
        if log:
            assert_parity("random/" + case["name"] + "/exp_logmel", np.exp(o.astype(np.float64)), np.exp(o_ref.astype(np.float64)), allow_floor=False)
        else:
            assert_parity("random/" + case["name"] + "/mel", o, o_ref, allow_floor=False)
        exp_d = O.backward(g_np, t_ref)
        _assert_dlam(float(layer.lambd.grad), exp_d, g_np, t_ref, case["name"])
        with torch.no_grad():
            yi = _layer(case, log=log, trainable=False)(x)
        oi = yi.cpu().numpy()
        assert (_log_err(oi, o_ref) if log else _rel_err(oi, o_ref)) <= TOL


def _random_full_window_cases(n, seed):
    """optimized=False (time_frequency.py:41,51): window = whole clip, n_fft = 2 L.  Power-of-two L take the fused kernel (n_fft up
    to 16384), every other L the chirp-z kernel: sequence in LDS, split into two half transforms (16384 < M <= 32768), or in
    global memory."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        kind = rng.integers(0, 4)
        if kind == 0:
            L = int(2 ** rng.integers(4, 14))                                  # 16 ... 8192: fused kernel
        elif kind == 1:
            L = int(rng.integers(20, 4000))                                    # chirp-z in LDS
        elif kind == 2:
            L = int(rng.integers(4097, 8190))                                  # chirp-z, split mode (M = 32768 -> 2 x 16384)
        else:
            L = int(rng.choice([8000, 8193, 12000, 16384]))                     # split boundary, global-memory FFT, n_fft 32768
        hop = int(rng.integers(max(1, L // 12), max(2, L // 2)))
        lam = float(rng.uniform(0.05, 0.4) * L) * (1.0 if rng.random() < 0.8 else -1.0)
        n_mels = int(rng.choice([8, 40, 64, 128]))
        sr = int(rng.choice([8000, 16000]))
        B = int(rng.integers(1, 4))
        out.append(dict(C.BY_NAME["g7_mel_nonopt_1024n"], name=f"fw{len(out)}_L{L}_h{hop}_m{n_mels}", B=B, L=L, lambd=lam, hop=hop, n_mels=n_mels,
                        sr=sr, f_min=0.0, f_max=None, normalize_window=bool(rng.random() < 0.5), seed=3000 + len(out)))
    return out


FULL_WINDOW_CASES = _random_full_window_cases(20, seed=4242)


@pytest.mark.parametrize("case", FULL_WINDOW_CASES, ids=[c["name"] for c in FULL_WINDOW_CASES])
def test_random_full_window_configuration_matches_oracle(case):
    x_np = C.make_input(case).astype(np.float32)
    g_np = C.make_cotangent(case)
    x = torch.from_numpy(x_np).to("cuda:0")
    g = torch.from_numpy(g_np).to("cuda:0")
    for log in (False, True):
        layer = _layer(case, log=log)
        y = layer(x)
        assert y.shape == C.out_shape(case)
        (y * g).sum().backward()
        o_ref, t_ref = O.forward(x_np, case["lambd"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"],
                                 case["normalize_window"], apply_log=log, optimized=False)
        o = y.detach().cpu().numpy()
        assert (_log_err(o, o_ref) if log else _rel_err(o, o_ref)) <= TOL
        exp_d = O.backward(g_np, t_ref)
        _assert_dlam(float(layer.lambd.grad), exp_d, g_np, t_ref, case["name"])
        with torch.no_grad():
            yi = _layer(case, log=log, trainable=False)(x)
        oi = yi.cpu().numpy()
        assert (_log_err(oi, o_ref) if log else _rel_err(oi, o_ref)) <= TOL


# ---- the optional gradients on the same kind of sweep: dL/dx (wave-FFT kernels up to n_fft 2048, LDS transforms beyond) and dL/dfb ----
GRAD_CASES = [c for c in _random_cases(60, seed=4711) if c["B"] * (c["L"] // c["hop"] + 1) * O.n_fft(c["lambd"]) <= 1_500_000][:28]


@pytest.mark.parametrize("case", GRAD_CASES, ids=[c["name"] for c in GRAD_CASES])
def test_random_configuration_optional_gradients_match_oracle(case):
    from dmel_amd import capi
    from test_hip_parity import _gfb_err, _gx_err
    x_np = C.make_input(case).astype(np.float32)
    g_np = C.make_cotangent(case)
    g = torch.from_numpy(g_np).to("cuda:0")
    n = capi.n_fft(case["lambd"])
    for log in (False, True):
        x = torch.from_numpy(x_np).to("cuda:0").requires_grad_(True)
        layer = _layer(case, log=log)
        y = layer(x)
        (y * g).sum().backward()
        y_np = y.detach().cpu().numpy()
        ref_x = O.backward_x(x_np, case["lambd"], case["hop"], case["sr"], g_np, y_np if log else None, case["f_min"], case["f_max"],
                             case["normalize_window"])
        assert _gx_err(x.grad.cpu().numpy(), ref_x) <= TOL, case["name"]
        plan = capi.Plan(case["L"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"], case["normalize_window"])
        gfb = torch.empty((n // 2 + 1, case["n_mels"]), dtype=torch.float32, device="cuda:0")
        plan.backward_fb(x.detach().data_ptr(), case["B"], case["lambd"], g.data_ptr(), y.detach().data_ptr(), gfb.data_ptr(), log,
                         torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ref_fb = O.backward_fb(x_np, case["lambd"], case["hop"], g_np, y_np if log else None, case["normalize_window"])
        assert _gfb_err(gfb.cpu().numpy(), ref_fb) <= TOL, case["name"]
