"""every way the layers are used without the transform length hanging on lambd's HOST value is sync-free -- the step
(forward, backward to lambd / the waveform / the filterbank, Adam) is captured ONCE into a HIP graph (a host read inside would fail the
capture) and its replays reproduce, bit for bit, the eagerly issued steps of a layer that reads lambd to the host at every forward
(lambd_sync=True: the by-value entry points).  models.py:171-200 (DSPEC), models.py:33-56 with optimized=False, x.requires_grad."""
import numpy as np
import pytest
import torch

import cases as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(make, x0, g, steps, graphed, wants_x):
    """`steps` Adam steps on a fresh layer; returns (parameters..., grad_x of the last step)"""
    layer = make(not graphed)                                   # eager reference: lambd_sync=True
    params = [p for p in layer.parameters()]
    # (a filterbank entry moves by ~lr per Adam step whatever its gradient: small steps keep mel = P @ fb positive under the log)
    opt = torch.optim.Adam([{"params": [p], "lr": 0.02 if p.dim() == 0 else 1e-5} for p in params], capturable=True)
    x = x0.clone().requires_grad_(wants_x)
    gx = torch.zeros_like(x0)

    def step():
        opt.zero_grad(set_to_none=False)
        if x.grad is not None:
            x.grad.zero_()
        layer(x).backward(g)
        if wants_x:
            gx.copy_(x.grad)
        opt.step()

    if not graphed:
        for _ in range(steps):
            step()
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()                                          # workspaces, optimizer state
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()                                              # any host read of lambd in here fails the capture
        for _ in range(steps - 2):
            graph.replay()
    torch.cuda.synchronize()
    return [p.detach().clone() for p in params] + [gx.clone()]


def _same(a, b):
    for u, v in zip(a, b):
        assert torch.equal(u, v), float((u - v).abs().max())


@pytest.mark.parametrize("wants_x", [False, True])
@pytest.mark.parametrize("n_points,hop", [(128, 1), (100, 1), (256, 4)])
def test_dspec_layer_step_is_sync_free_and_graph_capturable(n_points, hop, wants_x):
    """SpectrogramLayer in the reference's own configuration (optimized=False, search_spaces.py:71-91): powers of two and the chirp-z
    length 200"""
    from dmel_amd import SpectrogramLayer
    gen = torch.Generator().manual_seed(3)
    x0 = torch.randn(6, n_points, generator=gen).to(DEV)
    g = torch.randn(6, 1, n_points + 1, n_points // hop + 1, generator=gen).to(DEV)

    def make(sync):
        return SpectrogramLayer(torch.tensor(6.38), device=DEV, optimized=False, hop_length=hop, lambd_sync=sync).to(DEV)

    ref = _run(make, x0, g, 6, False, wants_x)
    got = _run(make, x0, g, 6, True, wants_x)
    assert abs(float(ref[0]) - 6.38) > 1e-3
    _same(ref, got)


@pytest.mark.parametrize("wants_x,learnable", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("log", [False, True])
def test_mel_layer_full_window_branch_is_sync_free_and_graph_capturable(wants_x, learnable, log):
    """optimized=False (the constructor's default, time_frequency.py:41,51): n_fft = 2 n_points whatever lambd is; with the waveform's
    gradient, with a trainable filterbank, with both"""
    from dmel_amd import MelSpectrogramLayer
    L, hop, M, sr = 512, 64, 24, 8000
    gen = torch.Generator().manual_seed(4)
    x0 = torch.randn(5, L, generator=gen).to(DEV)
    g = torch.randn(5, 1, M, L // hop + 1, generator=gen).to(DEV)

    def make(sync):
        return MelSpectrogramLayer(torch.tensor(40.0), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=DEV, optimized=False,
                                   log=log, learnable_fb=learnable, lambd_sync=sync).to(DEV)

    ref = _run(make, x0, g, 6, False, wants_x)
    got = _run(make, x0, g, 6, True, wants_x)
    _same(ref, got)


@pytest.mark.parametrize("log", [False, True])
def test_waveform_gradient_with_a_trainable_filterbank_is_sync_free(log):
    """optimized=True: the matrix fixes n_fft, so x.requires_grad needs no host value of lambd either (dmel_backward_x_dev)"""
    from dmel_amd import MelSpectrogramLayer
    case = C.BY_NAME["g1_c1"]
    x0 = torch.from_numpy(C.make_input(case).astype(np.float32)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)

    def make(sync):
        return MelSpectrogramLayer(torch.tensor(float(case["lambd"])), n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
                                   hop_length=case["hop"], device=DEV, optimized=True, log=log, learnable_fb=True, lambd_sync=sync,
                                   save_spec=False).to(DEV)       # (the recomputed spectrogram on both sides: the same bits)

    ref = _run(make, x0, g, 5, False, True)
    got = _run(make, x0, g, 5, True, True)
    _same(ref, got)


@pytest.mark.parametrize("log", [False, True])
def test_waveform_gradient_of_the_default_layer_is_sync_free(log):
    """HTK bank, optimized=True, x.requires_grad: the transform length follows lambd, so the forward is the tracked one (one launch per
    candidate n_fft) and the waveform gradient is issued the same way (DMEL_FLAG_CHECK_NFFT over a NaN-filled grad_x)"""
    from dmel_amd import MelSpectrogramLayer
    case = C.BY_NAME["g1_c1"]
    x0 = torch.from_numpy(C.make_input(case).astype(np.float32)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)

    def make(sync):
        return MelSpectrogramLayer(torch.tensor(float(case["lambd"])), n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
                                   hop_length=case["hop"], device=DEV, optimized=True, log=log, lambd_sync=sync).to(DEV)

    ref = _run(make, x0, g, 6, False, True)
    got = _run(make, x0, g, 6, True, True)
    assert torch.isfinite(got[-1]).all() and float(got[-1].abs().max()) > 0
    _same(ref, got)


@pytest.mark.parametrize("lam0", [21.42, 21.58])
def test_waveform_gradient_follows_lambd_across_an_n_fft_boundary_inside_a_graph(lam0):
    """6 lambd = 129 is the boundary between n_fft 128 and 256 (time_frequency.py:39,60-65).  Adam moves lambd by 0.02 per step from
    just below / just above it; the captured step holds the launches for n_fft, 2 n_fft and n_fft / 2 and the device value picks --
    forward AND waveform gradient -- so the replays keep matching the layer that reads lambd at every forward.  One of the two
    starting points crosses, whichever way the gradient points."""
    from dmel_amd import MelSpectrogramLayer, capi
    L, hop, M, sr, B = 2000, 100, 20, 8000, 3
    gen = torch.Generator().manual_seed(11)
    x0 = torch.randn(B, L, generator=gen).to(DEV)
    g = torch.randn(B, 1, M, L // hop + 1, generator=gen).to(DEV)

    def make(sync):
        return MelSpectrogramLayer(torch.tensor(lam0), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=DEV, optimized=True,
                                   log=True, lambd_sync=sync).to(DEV)

    ref = _run(make, x0, g, 10, False, True)
    got = _run(make, x0, g, 10, True, True)
    assert torch.isfinite(got[-1]).all()
    _same(ref, got)
    _CROSSED.append(capi.n_fft(float(got[0])) != capi.n_fft(lam0))
    if len(_CROSSED) == 2:
        assert any(_CROSSED), "neither starting point crossed the boundary: the test did not exercise the guards"


_CROSSED = []


@pytest.mark.parametrize("log", [False, True])
@pytest.mark.parametrize("mfma", ["fp32", "bf16x3"])
def test_saved_spectrogram_and_split_bf16_filterbank_gradient(log, mfma):
    """save_spec=True (the training forward writes the power spectrogram the contraction consumed; dmel_backward_fb_saved skips the
    recompute) and mfma='bf16x3' (DMEL_FLAG_MFMA_BF16X3: the gradient's GEMM as three split-bf16 products on the bf16 matrix pipe)
    against the exact, recomputing path: d mel_fb within 1e-4 of its largest entry (the bar of the reference fixtures), d lambd the same
    bits (the forward is the same kernel), and against the fp64 oracle"""
    from dmel_amd import MelSpectrogramLayer
    from oracle import dmel_oracle as O
    for name in ("g1_c1", "g2_c2"):
        case = C.BY_NAME[name]
        x_np = C.make_input(case).astype(np.float32)
        g_np = C.make_cotangent(case)
        x, g = torch.from_numpy(x_np).to(DEV), torch.from_numpy(g_np).to(DEV)

        def run(**kw):
            lay = MelSpectrogramLayer(torch.tensor(float(case["lambd"])), n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
                                      hop_length=case["hop"], device=DEV, optimized=True, log=log, learnable_fb=True, **kw).to(DEV)
            y = lay(x)
            y.backward(g)
            torch.cuda.synchronize()
            return y.detach(), lay.lambd.grad.clone(), lay.mel_fb.grad.clone()

        y0, dl0, gfb0 = run(save_spec=False)
        y1, dl1, gfb1 = run(save_spec=True, mfma=mfma)
        if mfma == "fp32":
            assert torch.equal(dl0, dl1) and torch.equal(y0, y1)              # the forward is the same kernel
        else:
            # the forward's contraction ran as three split-bf16 products too (kTrainH): the output within the 1e-4 bar of the exact one
            if log:
                assert float((y1 - y0).abs().max()) <= 1e-4
            else:
                assert float(((y1 - y0).abs() / y0.abs().clamp_min(1e-6 * float(y0.abs().max()))).max()) <= 1e-4
            assert abs(float(dl1) - float(dl0)) <= 1e-4 * abs(float(dl0)) + 2e-8 * float((g.abs()).sum())
        scale = float(gfb0.abs().max())
        assert float((gfb1 - gfb0).abs().max()) <= 1e-4 * scale
        ref = O.backward_fb(x_np, case["lambd"], case["hop"], g_np, y0.cpu().numpy() if log else None, case["normalize_window"])
        assert float(np.abs(gfb1.cpu().numpy().astype(np.float64) - ref).max()) <= 1e-4 * float(np.abs(ref).max())


@pytest.mark.parametrize("B,L,hop,M,n", [(3, 1000, 100, 7, 64), (2, 2000, 100, 20, 128), (16, 16000, 512, 128, 1024), (256, 16000, 512, 128, 1024),
                                         (5, 4000, 50, 130, 256)])
@pytest.mark.parametrize("log", [False, True])
def test_d_lambd_riding_in_the_filterbank_gradient_launch(B, L, hop, M, n, log):
    """dmel_backward_fb_saved_dl = dmel_backward_fb_saved + dmel_backward_scratch in one launch: d lambd has the BITS of the stand-alone dot
    kernel (same partition, same order of additions) whatever the element count (tails that are not multiples of four included) and
    whether or not the launch has room for the extra workgroups (130 mels: two column tiles, the kernels run one after the other);
    the filterbank gradient agrees to rounding (the slice count differs by the rows the dot takes)"""
    from dmel_amd import capi
    gen = torch.Generator().manual_seed(B * 1000 + M)
    T, F = L // hop + 1, n // 2 + 1
    plan = capi.Plan(L, hop, M, 8000, max_batch=B)
    st = torch.cuda.current_stream().cuda_stream
    spec = torch.rand(B, F, T, generator=gen).to(DEV)
    g = torch.randn(B, 1, M, T, generator=gen).to(DEV)
    tan = torch.randn(B, 1, M, T, generator=gen).to(DEV)
    out = torch.randn(B, 1, M, T, generator=gen).to(DEV)
    scratch = torch.zeros(plan.scratch_bytes(B), dtype=torch.uint8, device=DEV)
    gfb_a, gfb_b = torch.empty(F, M, device=DEV), torch.empty(F, M, device=DEV)
    dl_a, dl_b = torch.full((1,), 7.0, device=DEV), torch.full((1,), -3.0, device=DEV)
    for flags in (0, capi.DMEL_FLAG_MFMA_BF16X3):
        plan.backward_scratch(g.data_ptr(), tan.data_ptr(), g.numel(), dl_a.data_ptr(), st, scratch.data_ptr())
        plan.backward_fb_saved(spec.data_ptr(), B, n, g.data_ptr(), out.data_ptr(), gfb_a.data_ptr(), log, st, extra_flags=flags)
        for _ in range(3):                                              # the ticket counter is left at zero every time
            plan.backward_fb_saved_dl(spec.data_ptr(), B, n, g.data_ptr(), out.data_ptr(), tan.data_ptr(), gfb_b.data_ptr(), dl_b.data_ptr(),
                                      scratch.data_ptr(), log, st, extra_flags=flags)
        torch.cuda.synchronize()
        assert torch.equal(dl_a, dl_b), (float(dl_a), float(dl_b))
        ref = float((g.double() * tan.double()).sum())
        assert abs(float(dl_b) - ref) <= 1e-6 * float((g.double() * tan.double()).abs().sum())
        assert float((gfb_a - gfb_b).abs().max()) <= 2e-6 * float(gfb_a.abs().max())


@pytest.mark.parametrize("name", ["g1_c1", "g2_c2", "g5_n128", "g5_n4096", "g6_n2048_short", "g6_n64"])
def test_dense_bank_forward_on_the_bf16_matrix_pipe(name):
    """DMEL_FLAG_MFMA_BF16X3 through a caller-supplied DENSE filterbank (what a trained matrix is): output and tangent against the exact
    fp32-MFMA path of the same kernel family and against the fp64 oracle evaluated with the same matrix; the HTK bank ignores the flag"""
    from dmel_amd import capi
    from oracle import dmel_oracle as O
    case = C.BY_NAME[name]
    x_np = C.make_input(case).astype(np.float32)
    x = torch.from_numpy(x_np).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    n = capi.n_fft(case["lambd"])
    plan = capi.Plan(case["L"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"], case["normalize_window"])
    shape = C.out_shape(case)

    def fwd(flags, log):
        out = torch.empty(shape, dtype=torch.float32, device=DEV)
        tan = torch.empty_like(out)
        plan.forward(x.data_ptr(), case["B"], case["lambd"], out.data_ptr(), tan.data_ptr(), log, 1e-10, st, extra_flags=flags)
        torch.cuda.synchronize()
        return out, tan

    # the built-in (banded) bank: the flag changes nothing, bit for bit
    o0, t0 = fwd(0, True)
    o1, t1 = fwd(capi.DMEL_FLAG_MFMA_BF16X3, True)
    assert torch.equal(o0, o1) and torch.equal(t0, t1)
    gen = torch.Generator().manual_seed(12)
    fb = (torch.rand((n // 2 + 1, case["n_mels"]), generator=gen) + 0.01).to(DEV)
    plan.set_filterbank_dev(n, fb.data_ptr(), st)
    for log in (False, True):
        oe, te = fwd(0, log)
        oh, th = fwd(capi.DMEL_FLAG_MFMA_BF16X3, log)
        assert not torch.equal(oe, oh) or n < 64                                # (n_fft 32: no split path, the exact one runs)
        if log:
            assert float((oh - oe).abs().max()) <= 1e-4
        else:
            assert float(((oh - oe).abs() / oe.abs()).max()) <= 1e-4               # a dense positive bank: no element near zero
        tscale = float(te.abs().max())
        assert float((th - te).abs().max()) <= 1e-4 * tscale


def test_a_captured_step_may_hold_a_neighbour_n_fft_that_no_eager_step_has_run():
    """GraphedStep captures with the launches lambd can REACH (dmel_plan_force_launch), which near a boundary adds the
    2 n_fft candidate although the eager warm-up issued the primary only -- and the tracked waveform gradient runs
    dmel_backward_x_dev once per candidate, whose workspace cannot grow under capture (n_fft 1024 -> 2048 at hop 512 needs more).
    The eager calls now size it for the neighbours too."""
    from dmel_amd import MelSpectrogramLayer
    L, hop, M, sr, B = 16000, 512, 128, 16000, 4
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, generator=gen).to(DEV).requires_grad_(True)
    g = torch.randn(B, 1, M, L // hop + 1, generator=gen).to(DEV)
    layer = MelSpectrogramLayer(torch.tensor(128.0), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=DEV, optimized=True,
                                log=True).to(DEV)
    plan = layer._plan_for(torch.device(DEV))
    plan.force_launch(1024, 0)                                  # eager: the primary launch alone

    def step():
        if x.grad is not None:
            x.grad.zero_()
        layer.zero_grad(set_to_none=False)
        layer(x).backward(g)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(); step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref_gx, ref_dl = x.grad.clone(), layer.lambd.grad.clone()
    plan.force_launch(1024, 3)                                  # the graph holds both neighbours
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    plan.force_launch(0, 0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(x.grad, ref_gx) and torch.equal(layer.lambd.grad, ref_dl)
    assert layer.lambd_status()["error"] == 0
