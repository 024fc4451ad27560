"""Pin the CPU oracle (oracle/dmel_oracle.c) against the vectors captured from the reference.

Every case of tests/golden/cases.py: outputs of the reference's own MelSpectrogramLayer
(models.py:33-56), log (models.py:73) and autograd d/dlambd.  Tolerances are the gate of
SURVEY.md 8(c): rel <= 1e-4 on mel, abs <= 1e-4 on log-mel, rel <= 1e-4 on d lambd.
"""
import numpy as np
import pytest

import cases as C
from oracle import dmel_oracle as O

TOL = 1e-4


def _expected(case, gold):
    idx = C.sample_index(case)
    if idx is None:
        return gold["mel"].reshape(-1), None
    return gold["mel_sampled"], idx


def _mel_close(got, exp, case):
    # rel 1e-4 on mel; elements far below the frame's scale are fp32 noise in the reference
    # itself (its own floor is 1e-5 of the largest bins), so they get the matching absolute slack.
    scale = np.maximum(np.abs(exp), 1e-6 * np.abs(exp).max() + 1e-30)
    err = np.abs(got.astype(np.float64) - exp.astype(np.float64)) / scale
    assert err.max() <= TOL, f"{case['name']}: max rel err {err.max():.3e}"


@pytest.mark.parametrize("case", C.CASES, ids=[c["name"] for c in C.CASES])
def test_oracle_matches_reference(case):
    gold = C.load(case)
    x = C.make_input(case).astype(np.float32)
    g = C.make_cotangent(case)
    if case["optimized"]:
        assert O.n_fft(case["lambd"]) == int(gold["n_fft"])

    mel, dmel = O.forward(x, case["lambd"], case["hop"], case["n_mels"], case["sr"], f_min=case["f_min"],
                          f_max=case["f_max"], normalize_window=case["normalize_window"], apply_log=False, optimized=case["optimized"])
    y, dy = O.forward(x, case["lambd"], case["hop"], case["n_mels"], case["sr"], f_min=case["f_min"],
                      f_max=case["f_max"], normalize_window=case["normalize_window"], apply_log=True, optimized=case["optimized"])
    assert mel.shape == C.out_shape(case)
    exp, idx = _expected(case, gold)
    got = mel.reshape(-1) if idx is None else mel.reshape(-1)[idx]
    _mel_close(got, exp, case)
    goty = y.reshape(-1) if idx is None else y.reshape(-1)[idx]
    expy = np.log(exp.astype(np.float32) + np.float32(1e-10))
    assert np.abs(goty - expy).max() <= TOL

    # per-example checksums cover the elements a sampled fixture does not store
    np.testing.assert_allclose(mel.astype(np.float64).reshape(case["B"], -1).sum(1), gold["mel_sum"], rtol=TOL, atol=1e-12)
    np.testing.assert_allclose(y.astype(np.float64).reshape(case["B"], -1).sum(1), gold["y_sum"], rtol=TOL, atol=1e-2 * TOL)

    for got_d, exp_d, what in ((O.backward(g, dmel), float(gold["dlam_lin"]), "lin"),
                               (O.backward(g, dy), float(gold["dlam_log"]), "log")):
        if case["kind"] == "zero":
            assert got_d == 0.0 and exp_d == 0.0 and np.isfinite(got_d)
        else:
            assert abs(got_d - exp_d) <= TOL * abs(exp_d) + 1e-7, f"{case['name']} dlam_{what}: {got_d} vs {exp_d}"


def dc_reference_input(case, gold):
    """what the rest of the reference's path saw behind models.py:38 for a DC-dominated fixture: (clips, the means to subtract).
    fp32 clips: the clips as they are and the fp32 mean torch.mean produced (stored in the fixture); fp64 clips: the subtraction
    is done here in fp64, as the reference did, and nothing is left to subtract."""
    x = C.make_input(case)
    if case["dtype"] == "float64":
        return (x - gold["mean_ref"][:, None]).astype(np.float32), np.zeros(case["B"], np.float32)
    return x, gold["mean_ref"].astype(np.float32)


@pytest.mark.parametrize("case", C.DC_CASES, ids=[c["name"] for c in C.DC_CASES])
def test_oracle_matches_reference_dc_dominated(case):
    """G13: clips whose offset is 10 ... 500 x their signal.  The rounding of the clip mean (models.py:38, torch.mean in fp32) then shows
    in the lowest mel bands: one ulp of the mean moves them by up to 3e-2 (measured), and torch's own sum is one ulp off the
    correctly rounded mean in 3 of the 9 fp32 clips here.  So: (1) with the mean the reference itself subtracted (stored in the
    fixture) the oracle meets the plain 1e-4 bar on every element and on d lambd; (2) the oracle's own (correctly rounded) mean
    is within one ulp of the reference's."""
    gold = C.load(case)
    xin, mean = dc_reference_input(case, gold)
    g = C.make_cotangent(case)
    exp = gold["mel"]
    for log in (False, True):
        out, tan = O.forward(xin, case["lambd"], case["hop"], case["n_mels"], case["sr"], apply_log=log, mean=mean)
        if log:
            assert np.abs(out - np.log(exp + np.float32(1e-10))).max() <= TOL
        else:
            assert (np.abs(out.astype(np.float64) - exp) / np.abs(exp)).max() <= TOL, case["name"]
        exp_d = float(gold["dlam_log" if log else "dlam_lin"])
        cancel = float(np.abs(g.astype(np.float64) * tan.astype(np.float64)).sum())
        assert abs(O.backward(g, tan) - exp_d) <= TOL * abs(exp_d) + 2e-8 * cancel + 1e-7, (case["name"], log)
    if case["dtype"] == "float32":
        ulp = np.spacing(np.abs(gold["mean_ref"].astype(np.float32))).astype(np.float64)      # per clip: 0.49999... is in the binade below 0.5
        own = np.float32(C.make_input(case).astype(np.float64).mean(1)).astype(np.float64)
        assert (np.abs(own - gold["mean_ref"]) <= 1.0001 * ulp).all()


def test_zero_clip_is_log_eps_not_nan():
    case = C.BY_NAME["g6_zero"]
    y, dy = O.forward(C.make_input(case), case["lambd"], case["hop"], case["n_mels"], case["sr"], apply_log=True)
    assert np.allclose(y, np.log(1e-10)) and np.all(dy == 0)


def test_n_fft_truncation_rule():
    # time_frequency.py:60-65: int() truncates, then 1 << (x-1).bit_length()
    assert O.n_fft(512.0 / 6.0) in (512, 1024)   # fp32 product decides; pinned by the fixture above
    assert O.n_fft(64.0) == 512 and O.n_fft(128.0) == 1024 and O.n_fft(256.0) == 2048
    assert O.n_fft(-64.0) == 512
    assert O.n_fft(0.0) == 2 and O.n_fft(0.2) == 1 and O.n_fft(0.4) == 2 and O.n_fft(0.6) == 4
    assert O.n_fft(8000 * 0.3 / 6) == 4096


def test_window_centre_is_n_over_2():
    # time_frequency.py:24: centred at N/2 (not (N-1)/2): w[N/2] == 1, w[0] != w[N-1]
    w, dw = O.window(64.0, 512)
    assert w[256] == 1.0 and w[0] != w[511] and w[1] == w[511]
    wn, _ = O.window(64.0, 512, normalize=True)
    assert abs(float((wn.astype(np.float64) ** 2).sum()) - 1.0) < 1e-6
    # derivative by central differences in fp64
    h = 1e-3
    wp, _ = O.window(64.0 + h, 512)
    wm, _ = O.window(64.0 - h, 512)
    num = (wp.astype(np.float64) - wm.astype(np.float64)) / (2 * h)
    assert np.abs(num - dw).max() < 5e-5


@pytest.mark.parametrize("L,fixture", [(128, "g7_dspec.npz"), (100, "g7_dspec_100.npz")])
def test_dspec_oracle_matches_reference(L, fixture):
    """G7: the reference's non-optimized SpectrogramLayer (models.py:171-200), hop = 1, lambd = 6.38, at L = 128 (the reference's
    own use) and L = 100 (n_fft = 200: not a power of two, the oracle's chirp-z transform)."""
    import os
    from dmel_amd import synth
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    x = synth.waveforms(2, L, seed=77, scale=1.0)
    spec, tan = O.dspec(x, 6.38, hop=1)
    assert spec.shape == gold["spec"].shape == (2, 1, L + 1, L + 1)
    scale = np.maximum(np.abs(gold["spec"]), 1e-6 * gold["spec"].max())
    assert float((np.abs(spec - gold["spec"]) / scale).max()) <= TOL
    g = synth.cotangent(spec.shape, seed=78)
    d = O.backward(g, tan)
    assert abs(d - float(gold["dlam_lin"])) <= TOL * abs(float(gold["dlam_lin"]))


FBGRAD_CASES = ("g1_c1", "g2_c2", "g5_n128", "g6_n256_ragged")


def _gfb_err(got, exp):
    # one scale for the whole matrix: rows far above the band edge are sums of tiny, cancelling terms
    return float(np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30))


@pytest.mark.parametrize("name", FBGRAD_CASES)
def test_oracle_fbgrad_matches_reference(name):
    """d loss / d mel_fb (the adjoint of models.py:53) against torch autograd through the reference's own forward
    with the filterbank made a leaf (tests/golden/make_golden.py: run_fbgrad)."""
    import os
    case = C.BY_NAME[name]
    gold = np.load(os.path.join(os.path.dirname(C.__file__), f"g8_fbgrad_{name}.npz"))
    x = C.make_input(case).astype(np.float32)
    g = C.make_cotangent(case)
    got_lin = O.backward_fb(x, case["lambd"], case["hop"], g, None, case["normalize_window"])
    assert got_lin.shape == gold["gfb_lin"].shape
    assert _gfb_err(got_lin, gold["gfb_lin"]) <= 1e-4
    y, _ = O.forward(x, case["lambd"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"],
                     case["normalize_window"], apply_log=True, want_tangent=False)
    got_log = O.backward_fb(x, case["lambd"], case["hop"], g, y, case["normalize_window"])
    assert _gfb_err(got_log, gold["gfb_log"]) <= 1e-4


XGRAD_CASES = ("g1_c1", "g5_n128", "g6_n256_ragged", "g6_tone_dc", "g6_n32")


def _gx_err(got, exp):
    return float(np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30))


@pytest.mark.parametrize("name", XGRAD_CASES)
def test_oracle_xgrad_matches_reference(name):
    """d loss / d x (adjoint of models.py:38-53) against torch autograd through the reference's own forward
    (tests/golden/make_golden.py: run_xgrad; first two clips stored)."""
    import os
    case = C.BY_NAME[name]
    gold = np.load(os.path.join(os.path.dirname(C.__file__), f"g10_xgrad_{name}.npz"))
    x = C.make_input(case).astype(np.float32)
    g = C.make_cotangent(case)
    got_lin = O.backward_x(x, case["lambd"], case["hop"], case["sr"], g, None, case["f_min"], case["f_max"], case["normalize_window"])
    k = gold["gx_lin"].shape[0]
    assert _gx_err(got_lin[:k], gold["gx_lin"]) <= 1e-4
    y, _ = O.forward(x, case["lambd"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"],
                     case["normalize_window"], apply_log=True, want_tangent=False)
    got_log = O.backward_x(x, case["lambd"], case["hop"], case["sr"], g, y, case["f_min"], case["f_max"], case["normalize_window"])
    assert _gx_err(got_log[:k], gold["gx_log"]) <= 1e-4
    # DC removal: the gradient of every clip sums to zero
    assert np.abs(got_lin.sum(1)).max() <= 1e-9 * np.abs(got_lin).sum(1).max()
