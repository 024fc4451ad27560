"""GPU soak tests: run-to-run determinism under load (round 3's tools/determinism_stress.py, now this test) and the hand-off of the dot kernel.

The dot kernel's last-ticket hand-off (csrc/dmel_aux.hip: relaxed agent-scope store of the partial, `s_waitcnt vmcnt(0)`, relaxed
ticket) follows the guide's recipe but sits outside HIP's formal memory model: it is covered here by
thousands of repetitions whose results must be bit-identical -- a partial read before it landed would show as a different sum."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_dot_kernel_handoff_is_bitwise_repeatable_under_load():
    from dmel_amd import capi, synth
    s = torch.cuda.current_stream().cuda_stream
    for (B, L, hop, M) in ((256, 16000, 512, 128), (32, 160000, 512, 128), (3, 16000, 256, 64)):
        T = L // hop + 1
        plan = capi.Plan(L, hop, M, 16000, max_batch=B)
        gen = torch.Generator(device="cpu").manual_seed(11)
        g = torch.randn((B, 1, M, T), generator=gen).to(DEV)
        t = torch.randn((B, 1, M, T), generator=gen).to(DEV)
        # another stream keeps the memory system busy while the dot kernels run (the hand-off must not depend on a quiet chip)
        noise_a, noise_b = torch.empty(1 << 24, device=DEV), torch.empty(1 << 24, device=DEV)
        side = torch.cuda.Stream()
        outs = torch.zeros(4096, device=DEV)
        for r in range(outs.numel()):
            if r % 64 == 0:
                with torch.cuda.stream(side):
                    noise_b.copy_(noise_a, non_blocking=True)
            plan.backward(g.data_ptr(), t.data_ptr(), g.numel(), outs[r:].data_ptr(), s)
        torch.cuda.synchronize()
        ref = (g.double() * t.double()).sum()
        assert torch.all(outs == outs[0]), (B, L, int((outs != outs[0]).sum()))
        assert abs(float(outs[0]) - float(ref)) <= 1e-5 * float((g.double() * t.double()).abs().sum())


@pytest.mark.parametrize("shape", [(8, 16000, 16000, 128.0, 512, 128, True), (256, 16000, 16000, 128.0, 512, 128, False),
                                   (256, 16000, 16000, 128.0, 512, 128, True), (8, 16000, 16000, 64.0, 256, 64, True),
                                   (4, 40000, 16000, 256.0, 512, 128, False)])
def test_forward_is_bitwise_repeatable_under_load(shape):
    """the same launch repeated 160 times, every output compared bit for bit with the first (HTK and dense banks, both modes)"""
    from dmel_amd import capi, synth
    B, L, sr, lam, hop, M, dense = shape
    s = torch.cuda.current_stream().cuda_stream
    T = L // hop + 1
    x = torch.from_numpy(synth.waveforms(B, L, seed=3)).to(DEV)
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    n = capi.n_fft(lam)
    if dense:
        fb = torch.rand((n // 2 + 1, M), device=DEV) + 0.01
        plan.set_filterbank_dev(n, fb.data_ptr(), s)
    for train in (True, False):
        ref_o = torch.empty((B, 1, M, T), device=DEV)
        ref_t = torch.empty_like(ref_o)
        plan.forward(x.data_ptr(), B, lam, ref_o.data_ptr(), ref_t.data_ptr() if train else None, True, 1e-10, s)
        torch.cuda.synchronize()
        outs = [(torch.empty_like(ref_o), torch.empty_like(ref_o)) for _ in range(8)]
        bad = 0
        for r in range(160):
            o, t = outs[r % 8]
            plan.forward(x.data_ptr(), B, lam, o.data_ptr(), t.data_ptr() if train else None, True, 1e-10, s)
            if r % 8 == 7:
                torch.cuda.synchronize()
                bad += sum(1 for (oo, tt) in outs if not torch.equal(oo, ref_o) or (train and not torch.equal(tt, ref_t)))
        assert bad == 0, (shape, train, bad)


def test_torch_ops_refuse_a_handle_that_is_not_a_live_plan():
    """the plan travels through torch.ops.dmel as an integer; anything that is not a live plan of this process is
    refused by a registry look-up instead of being dereferenced"""
    import dmel_amd
    from dmel_amd import MelSpectrogramLayer, capi
    lay = MelSpectrogramLayer(torch.tensor(64.0), n_mels=64, n_points=16000, sample_rate=16000, hop_length=256, device=DEV, optimized=True).to(DEV)
    x = torch.randn(2, 16000, device=DEV)
    y = lay(x).detach()                       # (the autograd node of a live result would hold a plan reference of its own)
    plan = lay._plan_for(torch.device(DEV))
    h = plan.handle
    assert plan.is_live()
    lam = lay.lambd.detach()
    for bogus in (12345, h + 64):
        with pytest.raises(RuntimeError, match="not the handle of a live plan"):
            torch.ops.dmel.forward(x, lam, bogus, 0, 1e-10, False, False, False)
    extra = plan.retain()                     # a second reference: the plan survives its owner
    assert extra == h
    plan.close()
    assert capi.load().dmel_plan_is_live(extra) == 1
    o, _ = torch.ops.dmel.forward(x, lam, extra, 0, 1e-10, True, False, False)       # with the tangent: the kernel the layer's training forward ran
    torch.cuda.synchronize()
    assert torch.equal(o, y)
    capi.release_handle(extra)
    assert capi.load().dmel_plan_is_live(extra) == 0
    with pytest.raises(RuntimeError, match="not the handle of a live plan"):
        torch.ops.dmel.forward(x, lam, extra, 0, 1e-10, False, False, False)


def test_graphed_step_keeps_its_plans_alive_for_the_life_of_the_graph():
    """a captured HIP graph launches kernels that read the plan's tables: GraphedStep takes a reference at every capture, so dropping the
    creator's reference while the graph is still replayed is safe, and close() gives the reference back"""
    from dmel_amd import GraphedStep, MelSpectrogramLayer, capi
    lay = MelSpectrogramLayer(torch.tensor(64.0), n_mels=64, n_points=16000, sample_rate=16000, hop_length=256, device=DEV,
                              optimized=True, log=True).to(DEV)
    opt = torch.optim.Adam([lay.lambd], lr=1e-3, capturable=True)
    x = torch.randn(4, 16000, device=DEV)
    g = torch.randn(4, 1, 64, 63, device=DEV)

    def step():
        opt.zero_grad(set_to_none=True)
        lay(x).backward(g)
        opt.step()

    for _ in range(3):
        step()
    gs = GraphedStep(step, [lay], max_ahead=4, steps_per_replay=2)
    for _ in range(8):
        gs()
    torch.cuda.synchronize()
    plan = lay._plan_for(torch.device(DEV))
    h = plan.handle
    assert gs._held_plans == [h]
    lib = capi.load()
    before = float(lay.lambd.detach())
    plan.close()                                  # the creator's reference goes; the graph's keeps the plan
    assert lib.dmel_plan_is_live(h) == 1
    gs.graph.replay()                             # the captured launches still find live tables
    torch.cuda.synchronize()
    assert float(lay.lambd.detach()) != before and torch.isfinite(lay.lambd.detach())
    gs.close()
    assert lib.dmel_plan_is_live(h) == 0


def test_dense_bank_forward_does_not_read_what_earlier_kernels_left_in_lds():
    """Round 4 bug: phase 2 pads runs of k-steps with zero filterbank blocks and reads the A operands of the padding; with a DENSE bank at
    n_fft 1024 four of those bins lay in slot padding nothing writes, and a NaN pattern left there by an earlier kernel times a zero
    coefficient poisoned the output (a trainable-filterbank run went NaN after a few hundred steps).  Here every CU's LDS is filled with
    NaNs first (forwards of other transform sizes on an all-NaN waveform), then the dense forward must still be finite and repeatable."""
    from dmel_amd import capi
    s = torch.cuda.current_stream().cuda_stream
    B, L, hop, M, sr = 256, 16000, 512, 128, 16000
    T = L // hop + 1
    x = torch.randn(B, L, device=DEV) * 0.1
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    fb = torch.rand((513, M), device=DEV) + 0.01
    plan.set_filterbank_dev(1024, fb.data_ptr(), s)
    out, tan = torch.empty((B, 1, M, T), device=DEV), torch.empty((B, 1, M, T), device=DEV)
    plan.forward(x.data_ptr(), B, 128.0, out.data_ptr(), tan.data_ptr(), True, 1e-10, s)
    torch.cuda.synchronize()
    ref_o, ref_t = out.clone(), tan.clone()
    # the built-in HTK bank too: its last mel tile's padded k-steps reach bin 527, whose tangent entry is the padding column of the
    # transposition plane's last row
    htk = capi.Plan(L, hop, M, sr, max_batch=B)
    ho, ht = torch.empty_like(out), torch.empty_like(out)
    htk.forward(x.data_ptr(), B, 128.0, ho.data_ptr(), ht.data_ptr(), True, 1e-10, s)
    torch.cuda.synchronize()
    ref_ho, ref_ht = ho.clone(), ht.clone()
    poison = capi.Plan(L, hop, M, sr, max_batch=B)
    xn = torch.full((B, L), float("nan"), device=DEV)
    po, pt = torch.empty_like(out), torch.empty_like(out)
    for rep in range(4):
        for lam in (256.0, 512.0, 64.0, 32.0):                        # n_fft 2048, 4096, 512, 256: other LDS maps, all NaN
            poison.forward(xn.data_ptr(), B, lam, po.data_ptr(), pt.data_ptr(), True, 1e-10, s)
        out.zero_(); tan.zero_()
        plan.forward(x.data_ptr(), B, 128.0, out.data_ptr(), tan.data_ptr(), True, 1e-10, s)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all() and torch.isfinite(tan).all(), rep
        assert torch.equal(out, ref_o) and torch.equal(tan, ref_t), rep
        for lam in (256.0, 512.0):
            poison.forward(xn.data_ptr(), B, lam, po.data_ptr(), pt.data_ptr(), True, 1e-10, s)
        ho.zero_(); ht.zero_()
        htk.forward(x.data_ptr(), B, 128.0, ho.data_ptr(), ht.data_ptr(), True, 1e-10, s)
        torch.cuda.synchronize()
        assert torch.equal(ho, ref_ho) and torch.equal(ht, ref_ht), rep


def _poison_lds(reps=2):
    """leave NaN bit patterns in every CU's LDS: forwards of several transform sizes (different LDS maps) on an all-NaN waveform"""
    from dmel_amd import capi
    s = torch.cuda.current_stream().cuda_stream
    B, L, hop, M, sr = 256, 16000, 512, 128, 16000
    T = L // hop + 1
    if not hasattr(_poison_lds, "state"):
        _poison_lds.state = (capi.Plan(L, hop, M, sr, max_batch=B), torch.full((B, L), float("nan"), device=DEV),
                             torch.empty((B, 1, M, T), device=DEV), torch.empty((B, 1, M, T), device=DEV))
    plan, xn, po, pt = _poison_lds.state
    for _ in range(reps):
        for lam in (128.0, 256.0, 512.0, 64.0, 32.0, 16.0):            # n_fft 1024, 2048, 4096, 512, 256, 128
            plan.forward(xn.data_ptr(), B, lam, po.data_ptr(), pt.data_ptr(), True, 1e-10, s)
    torch.cuda.synchronize()


@pytest.mark.parametrize("lam,L,hop,M", [(16.0, 8000, 80, 64), (40.0, 8000, 80, 64), (64.0, 16000, 256, 64), (128.0, 16000, 512, 128),
                                         (256.0, 32000, 512, 128), (400.0, 40000, 400, 64), (700.0, 40000, 800, 40)])
def test_no_path_depends_on_what_earlier_kernels_left_in_lds(lam, L, hop, M):
    """every kernel family at every plan -- training and inference forward (HTK and dense bank, exact and split-bf16), spectrogram, d/d
    filterbank, d/d waveform -- must give the same bits whether LDS held zeros or NaNs when it started: nothing may read LDS it (or an
    earlier phase of the same kernel) has not written, not even behind a zero coefficient"""
    from dmel_amd import capi
    s = torch.cuda.current_stream().cuda_stream
    B, sr = 6, 16000
    T = L // hop + 1
    n = capi.n_fft(lam)
    gen = torch.Generator().manual_seed(int(lam))
    x = (torch.randn(B, L, generator=gen) * 0.1).to(DEV)
    g = torch.randn(B, 1, M, T, generator=gen).to(DEV)
    htk, dense = capi.Plan(L, hop, M, sr, max_batch=B), capi.Plan(L, hop, M, sr, max_batch=B)
    fb = (torch.rand((n // 2 + 1, M), generator=gen) + 0.01).to(DEV)
    dense.set_filterbank_dev(n, fb.data_ptr(), s)

    def run_all():
        res = []
        for plan in (htk, dense):
            for flags in (0, capi.DMEL_FLAG_MFMA_BF16X3):
                out, tan = torch.empty((B, 1, M, T), device=DEV), torch.empty((B, 1, M, T), device=DEV)
                plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, s, extra_flags=flags)
                res += [out, tan]
                oi = torch.empty((B, 1, M, T), device=DEV)
                plan.forward(x.data_ptr(), B, lam, oi.data_ptr(), None, True, 1e-10, s, extra_flags=flags)      # inference: two frames per FFT
                res.append(oi)
                gfb = torch.empty((n // 2 + 1, M), device=DEV)
                plan.backward_fb(x.data_ptr(), B, lam, g.data_ptr(), out.data_ptr(), gfb.data_ptr(), True, s, extra_flags=flags)
                res.append(gfb)
            gx = torch.empty_like(x)
            plan.backward_x(x.data_ptr(), B, lam, g.data_ptr(), res[-4].data_ptr(), gx.data_ptr(), True, s)
            res.append(gx)
        spec = torch.empty((B, n // 2 + 1, T), device=DEV)
        htk.spectrogram(x.data_ptr(), B, lam, spec.data_ptr(), s, remove_dc=True)
        res.append(spec)
        torch.cuda.synchronize()
        return res

    ref = run_all()
    assert all(torch.isfinite(r).all() for r in ref)
    for rep in range(2):
        _poison_lds()
        got = run_all()
        for i, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), (rep, i, float((a - b).abs().max()))


def test_results_do_not_depend_on_what_freed_device_memory_held():
    """outputs, tangents, per-call scratch and the autograd functions' temporaries come from torch's caching allocator uninitialised: the
    same steps must give the same bits whether the freed blocks they land in held zeros or NaNs (the per-call scratch "needs no
    initialisation", include/dmel.h)"""
    from dmel_amd import MelSpectrogramLayer, SpectrogramLayer
    gen = torch.Generator().manual_seed(21)
    x = (torch.randn(8, 16000, generator=gen) * 0.1).to(DEV)

    def steps():
        res = []
        for kw in (dict(optimized=True), dict(optimized=True, learnable_fb=True), dict(optimized=True, learnable_fb=True, mfma="bf16x3"),
                   dict(optimized=False)):
            torch.manual_seed(0)
            lay = MelSpectrogramLayer(torch.tensor(100.0), n_mels=64, n_points=16000 if kw["optimized"] else 2048, sample_rate=16000,
                                      hop_length=256, device=DEV, log=True, **kw).to(DEV)
            xx = (x if kw["optimized"] else x[:, :2048].contiguous()).clone().requires_grad_(True)
            y = lay(xx)
            g = torch.ones_like(y) * 0.5
            y.backward(g)
            res += [y.detach().clone(), lay.lambd.grad.clone(), xx.grad.clone()] + ([lay.mel_fb.grad.clone()] if kw.get("learnable_fb") else [])
        sp = SpectrogramLayer(torch.tensor(6.38), device=DEV, optimized=False, hop_length=1).to(DEV)
        xs = x[:, :128].contiguous().clone().requires_grad_(True)
        ys = sp(xs)
        ys.sum().backward()
        res += [ys.detach().clone(), sp.lambd.grad.clone(), xs.grad.clone()]
        torch.cuda.synchronize()
        return res

    def fill_cache(value):
        blocks = [torch.full((n,), value, device=DEV) for n in (1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10) for _ in range(3)]
        torch.cuda.synchronize()
        del blocks

    torch.cuda.empty_cache()
    fill_cache(0.0)
    ref = steps()
    assert all(torch.isfinite(r).all() for r in ref)
    for value in (float("nan"), 1e30):
        fill_cache(value)
        got = steps()
        for i, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), (value, i, float((a - b).abs().max()))


def test_two_tiles_per_workgroup_give_the_same_bits(tmp_path):
    """The TPW = 2 instantiations (n_fft 256 / 512, and 1024 for the modes that pack two frames per transform) run when a launch is large
    enough for them (forward_tiles_per_wg); the suite's shapes are small, so they are forced here (DMEL_TILES_PER_WG=2, read once per
    process: subprocesses) and must reproduce the one-tile kernels bit for bit -- the second tile rewrites the window table over the
    exchange region, with 64 mels run 1 carries a piece of another wave's tile (round 4)."""
    import os
    import subprocess
    import sys
    import numpy as np
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import dmel_amd
from dmel_amd import capi
out = {}
st = torch.cuda.current_stream().cuda_stream
for (L, hop, M, lam, train) in [(4000, 50, 64, 40.0, True), (4000, 50, 64, 80.0, True), (4000, 50, 128, 80.0, True), (6000, 100, 40, 35.0, True),
                                (4000, 50, 130, 80.0, True), (16000, 512, 128, 128.0, False), (4000, 50, 64, 80.0, False)]:
    B = 3
    T = L // hop + 1
    x = torch.from_numpy(np.random.default_rng(L + M).standard_normal((B, L)).astype(np.float32)).cuda()
    plan = capi.Plan(L, hop, M, 8000, max_batch=B)
    o = torch.empty((B, 1, M, T), device="cuda"); t = torch.empty_like(o)
    plan.forward(x.data_ptr(), B, lam, o.data_ptr(), t.data_ptr() if train else None, True, 1e-10, st)
    torch.cuda.synchronize()
    key = f"{L}_{hop}_{M}_{lam}_{train}"
    out["o_" + key] = o.cpu().numpy()
    out["g_" + key] = np.array(plan.info()["grid_fwd"])
    if train:
        out["t_" + key] = t.cpu().numpy()
np.savez(sys.argv[2], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, force in (("one", "1"), ("two", "2")):
        env = dict(os.environ, DMEL_TILES_PER_WG=force)
        path = str(tmp_path / f"{tag}.npz")
        p = subprocess.run([sys.executable, "-c", code, root, path], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[tag] = np.load(path)
    assert set(res["one"].files) == set(res["two"].files) and len(res["one"].files) == 19
    for k in res["one"].files:
        a, b = res["one"][k], res["two"][k]
        if k.startswith("g_"):
            assert int(b) < int(a), (k, int(a), int(b))                 # the forced launches really used two-tile workgroups
        else:
            assert np.isfinite(a).all() and np.array_equal(a, b), (k, float(np.abs(a - b).max()))


def test_dot_ticket_tree_at_every_group_boundary():
    """Round 6: the dot kernel's hand-off is a tree of tickets (groups of 16 workgroups of 4096 elements; csrc/dmel_kernels.h: kDotGroup).  Element
    counts around every boundary of that scheme -- one workgroup, a partly filled group, exactly one group, one workgroup more, many groups with a
    short last one, the cap of 512 workgroups and beyond it -- each launched several times on the same scratch (the counters re-arm themselves),
    with and without accumulation, against an fp64 sum."""
    import torch
    from dmel_amd import capi
    dev = "cuda:0"
    plan = capi.Plan(4000, 128, 32, 8000)
    s = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev).manual_seed(5)
    per = 4096
    counts = [1, 3, per - 1, per, per + 1, 15 * per + 7, 16 * per, 16 * per + 1, 17 * per, 33 * per + 5, 255 * per + 9, 256 * per, 511 * per + 1, 512 * per,
              512 * per + 3, 3 * 512 * per + 11]
    dl = torch.zeros(1, device=dev)
    for n in counts:
        g = torch.randn(n, generator=gen, device=dev)
        t = torch.randn(n, generator=gen, device=dev)
        ref = float((g.double() * t.double()).sum())
        scale = float((g.double() * t.double()).abs().sum())
        for rep in range(3):
            plan.backward(g.data_ptr(), t.data_ptr(), n, dl.data_ptr(), s)
            torch.cuda.synchronize()
            assert abs(float(dl) - ref) <= 2e-7 * scale + 1e-6, (n, rep, float(dl), ref)
        first = float(dl)
        plan.backward_scratch(g.data_ptr(), t.data_ptr(), n, dl.data_ptr(), s, None, accumulate=True)
        torch.cuda.synchronize()
        assert abs(float(dl) - 2.0 * ref) <= 4e-7 * scale + 2e-6, (n, float(dl), 2 * ref, first)
