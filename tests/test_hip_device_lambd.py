"""GPU tests of the sync-free boundary (include/dmel.h: dmel_forward_dev / dmel_backward_scratch, torch.ops.dmel.*):
lambd stays on the device, the kernels check the n_fft they were launched for, guard launches cover boundary crossings,
uncovered ones fail loudly, a whole training step is capturable into a HIP graph, and steps on different streams are
independent.  The by-value path (the host reads lambd, as time_frequency.py:39 does) is the comparison throughout."""
import copy
import pickle

import numpy as np
import pytest
import torch

import cases as C
from oracle import dmel_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mk(case, lam=None, sync=False, log=True, **kw):
    from dmel_amd import MelSpectrogramLayer
    lam = case["lambd"] if lam is None else lam
    return MelSpectrogramLayer(torch.tensor(float(lam), dtype=torch.float32), n_mels=case["n_mels"], n_points=case["L"],
                               sample_rate=case["sr"], f_min=case["f_min"], f_max=case["f_max"], hop_length=case["hop"], device=DEV,
                               optimized=True, normalize_window=case["normalize_window"], log=log, lambd_sync=sync, **kw).to(DEV)


@pytest.mark.parametrize("name", ["g1_c1", "g2_c2", "g3_c3", "g5_n128", "g5_n4096", "g6_n32", "g6_neglambd", "g6_normwin", "g6_zero"])
def test_device_lambd_equals_host_lambd_bitwise(name):
    """Both entry points run the same kernels on the same value: outputs, tangents and d lambd are identical bit for bit."""
    case = C.BY_NAME[name]
    x = torch.from_numpy(C.make_input(case).astype(np.float32)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
    res = []
    for sync in (False, True):
        layer = _mk(case, sync=sync)
        y = layer(x)
        y.backward(g)
        res.append((y.detach().clone(), layer.lambd.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    with torch.no_grad():                       # inference kernels (two frames per FFT)
        a, b = _mk(case, sync=False)(x), _mk(case, sync=True)(x)
    assert torch.equal(a, b)


def test_c_abi_forward_dev_and_scratch():
    """The C ABI itself: dmel_forward_dev + dmel_backward_scratch with caller scratch against dmel_forward + dmel_backward."""
    from dmel_amd import capi
    case = C.BY_NAME["g2_c2"]
    B, M, T = case["B"], case["n_mels"], case["L"] // case["hop"] + 1
    x = torch.from_numpy(C.make_input(case).astype(np.float32)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
    lam = torch.tensor([case["lambd"]], dtype=torch.float32, device=DEV)
    plan = capi.Plan(case["L"], case["hop"], M, case["sr"])
    s = torch.cuda.current_stream().cuda_stream
    out0, tan0, d0 = torch.empty((B, 1, M, T), device=DEV), torch.empty((B, 1, M, T), device=DEV), torch.zeros(1, device=DEV)
    plan.forward(x.data_ptr(), B, case["lambd"], out0.data_ptr(), tan0.data_ptr(), True, 1e-10, s)
    plan.backward(g.data_ptr(), tan0.data_ptr(), g.numel(), d0.data_ptr(), s)
    out1, tan1, d1 = torch.empty_like(out0), torch.empty_like(tan0), torch.zeros(1, device=DEV)
    scratch = torch.full((plan.scratch_bytes(B),), 0xA5, dtype=torch.uint8, device=DEV)       # garbage on purpose: no initialisation needed
    plan.forward_dev(x.data_ptr(), B, lam.data_ptr(), out1.data_ptr(), tan1.data_ptr(), True, 1e-10, s, scratch_ptr=scratch.data_ptr())
    plan.backward_scratch(g.data_ptr(), tan1.data_ptr(), g.numel(), d1.data_ptr(), s, scratch.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out0, out1) and torch.equal(tan0, tan1) and torch.equal(d0, d1)
    st = plan.lambd_status()
    assert st["known"] == 1 and st["lambd_seen"] == case["lambd"] and st["n_fft_seen"] == 1024 and st["error"] == 0
    assert st["seq_issued"] == 1 and st["seq_seen"] == 1


def test_boundary_crossing_is_covered_by_guard_launches():
    """lambd moves across 6*lambd = 512 (n_fft 512 -> 1024), on up to n_fft 16384 (several waves per frame) and back down to the
    direct-DFT kernel between forwards without the host being told: the guard launch of the right n_fft does the work, and the
    host's picture follows from the kernels' report."""
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    layer = _mk(case, lam=85.0)
    layer.set_tracking(8, 1)                                  # always guard both neighbours
    for lam, n in ((85.0, 512), (85.6, 1024), (170.0, 1024), (171.0, 2048), (342.0, 4096), (683.0, 8192), (1366.0, 16384), (683.0, 8192),
                   (342.0, 4096), (171.0, 2048), (85.6, 1024), (60.0, 512), (42.0, 256), (21.0, 128), (10.6, 64), (5.3, 32), (2.6, 16), (5.3, 32)):
        layer.lambd.data.fill_(lam)
        y = layer(x)
        ref = _mk(case, lam=lam, sync=True)(x)
        assert torch.equal(y, ref), lam
        torch.cuda.synchronize()
        st = layer.lambd_status()
        assert st["n_fft_seen"] == n and st["error"] == 0 and st["guards"] == 3


def test_uncovered_jump_fails_loudly_then_recovers():
    """With the guards switched off, a jump of lambd to another n_fft is covered by no launch: the outputs are NaN (never
    stale memory), the next forward raises, and after that the layer works again."""
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    layer = _mk(case, lam=64.0)
    layer.set_tracking(8, 2)                                  # never guard
    y0 = layer(x)
    assert torch.isfinite(y0).all()
    layer.lambd.data.fill_(200.0)                             # n_fft 512 -> 2048 behind the host's back
    y1 = layer(x)
    y1.backward(torch.ones_like(y1))
    torch.cuda.synchronize()
    assert torch.isnan(y1).all() and torch.isnan(layer.lambd.grad).all()
    assert layer.lambd_status()["error"] == 1
    with pytest.raises(RuntimeError, match="faster than the sync-free forward"):
        layer(x)
    y2 = layer(x)                                             # tracking was reset: one blocking read, then fine
    assert torch.equal(y2, _mk(case, lam=200.0, sync=True)(x))
    # resync() after a manual rewrite avoids the episode altogether
    layer.lambd.data.fill_(20.0)
    layer.resync()
    assert torch.equal(layer(x), _mk(case, lam=20.0, sync=True)(x))


def _train(layer, x, g, steps, lr, opt_name="adam"):
    opt = (torch.optim.Adam if opt_name == "adam" else torch.optim.SGD)([layer.lambd], lr=lr)
    traj = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        layer(x).backward(g)
        opt.step()
        traj.append(layer.lambd.detach().clone())
    return torch.stack(traj).cpu().numpy()


@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_sync_free_training_follows_the_host_read_path_across_boundaries(sign):
    """Adam with lr_tf = 1.0 (search_spaces.py:20) moves lambd about one sample per step; started next to a power-of-two
    boundary the run crosses it within a few steps.  The sync-free loop (no host read, the host runs ahead) must produce the
    same lambd trajectory, bit for bit, as the loop that reads lambd at every forward."""
    case = dict(C.BY_NAME["g1_c1"])
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    g = sign * torch.from_numpy(C.make_cotangent(case)).to(DEV)
    lam0 = 85.45 if sign > 0 else 85.55                      # 6 * 85.5 = 513: the 512 / 1024 boundary
    a = _train(_mk(case, lam=lam0, sync=True), x, g, 40, 1.0)
    free = _mk(case, lam=lam0, sync=False)
    b = _train(free, x, g, 40, 1.0)
    torch.cuda.synchronize()
    assert np.array_equal(a, b)
    from dmel_amd import capi
    ns = {capi.n_fft(float(v)) for v in a} | {capi.n_fft(lam0)}
    assert len(ns) >= 2, "the run did not cross a boundary: the test would prove nothing"
    assert free.lambd_status()["error"] == 0


def test_whole_step_is_graph_capturable():
    """forward + backward + optimizer.step() on lambd captured ONCE into a HIP graph and replayed: no host code runs per
    step, lambd lives on the device, and the replayed trajectory equals the eager one (crossing a boundary on the way,
    which the guard launches inside the graph cover)."""
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
    lam0, steps = 85.45, 24
    eager = _train(_mk(case, lam=lam0, sync=True), x, g, steps, 0.25)

    layer = _mk(case, lam=lam0, sync=False)
    opt = torch.optim.Adam([layer.lambd], lr=0.25, capturable=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                             # warm-up on the side stream (allocator, optimizer state, tracking)
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            layer(x).backward(g)
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # restart from lam0 with fresh optimizer state so that the comparison is exact
    layer.lambd.data.fill_(lam0)
    for st in opt.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        layer(x).backward(g)
        opt.step()
    traj = []
    layer.lambd.data.fill_(lam0)
    for st in opt.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    for _ in range(steps):
        graph.replay()
        traj.append(layer.lambd.detach().clone())
    torch.cuda.synchronize()
    got = torch.stack(traj).cpu().numpy()
    # capturable Adam keeps its step counter on the device and evaluates the bias corrections there: the same formula in
    # fp32 instead of Python floats, so the trajectories agree to rounding, not to the bit
    np.testing.assert_allclose(got, eager, rtol=1e-5)
    from dmel_amd import capi
    assert len({capi.n_fft(float(v)) for v in got} | {capi.n_fft(lam0)}) >= 2
    assert layer.lambd_status()["error"] == 0


def test_two_streams_through_one_layer():
    """Steps of one layer on two streams at once (an evaluation stream beside the training stream): every call carries its
    own scratch, so the results equal the serial ones."""
    case = C.BY_NAME["g2_c2"]
    x = torch.from_numpy(C.make_input(case).astype(np.float32)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
    xs = [x * (1.0 + 0.1 * i) for i in range(8)]
    layer = _mk(case)
    serial = []
    for xi in xs:
        layer.lambd.grad = None
        y = layer(xi)
        y.backward(g)
        serial.append((y.detach().clone(), layer.lambd.grad.clone()))
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    lam = layer.lambd
    outs = [None] * len(xs)
    for rep in range(20):
        for i, xi in enumerate(xs):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                y = layer(xi)
                (dl,) = torch.autograd.grad(y, lam, g)
                outs[i] = (y.detach(), dl)
        torch.cuda.synchronize()
        for i in range(len(xs)):
            assert torch.equal(outs[i][0], serial[i][0]) and torch.equal(outs[i][1].reshape(()), serial[i][1].reshape(())), (rep, i)


def test_plan_owned_scratch_is_ordered_across_streams():
    """The entry points without caller scratch share the plan's: calls from different streams are ordered by the library."""
    from dmel_amd import capi
    case = C.BY_NAME["g3_c3"]                                 # long clips: the partial-sum kernel writes the shared psum
    B, M, T = case["B"], case["n_mels"], case["L"] // case["hop"] + 1
    x = torch.from_numpy(C.make_input(case).astype(np.float32)).to(DEV)
    xs = [x * (1.0 + 0.25 * i) + 0.01 * i for i in range(4)]
    plan = capi.Plan(case["L"], case["hop"], M, case["sr"])
    ref = []
    s0 = torch.cuda.current_stream().cuda_stream
    for xi in xs:
        o = torch.empty((B, 1, M, T), device=DEV)
        plan.forward(xi.data_ptr(), B, case["lambd"], o.data_ptr(), None, True, 1e-10, s0)
        ref.append(o)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in xs]
    for rep in range(10):
        outs = [torch.empty((B, 1, M, T), device=DEV) for _ in xs]
        for xi, o, st in zip(xs, outs, streams):
            plan.forward(xi.data_ptr(), B, case["lambd"], o.data_ptr(), None, True, 1e-10, st.cuda_stream)
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            assert torch.equal(o, r)


def test_layer_survives_deepcopy_pickle_and_state_dict():
    """The reference module is plain torch: copy.deepcopy (best-model snapshots, EMA) and pickling work; ours must too, after
    plans (device table caches) exist."""
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    layer = _mk(case)
    y = layer(x)
    twin = copy.deepcopy(layer)
    assert torch.equal(twin(x), y)
    blob = pickle.dumps(layer)
    assert torch.equal(pickle.loads(blob)(x), y)
    other = _mk(case, lam=200.0)
    other(x)                                                  # tracking now knows 200.0
    other.load_state_dict(layer.state_dict())                 # ... and is reset by the load
    assert torch.equal(other(x), y)


def test_torch_ops_are_registered():
    from dmel_amd import capi
    ops = capi.torch_ops()
    for name in ("forward", "backward", "mel_spectrogram", "mel_fbanks"):
        assert hasattr(ops, name)
    fb = ops.mel_fbanks(513, 0.0, 8000.0, 128, 16000)
    assert np.array_equal(fb.numpy(), O.mel_fbanks(513, 0.0, 8000.0, 128, 16000))


@pytest.mark.parametrize("k", [1, 2])
def test_graphed_step_recaptures_when_n_fft_changes(k):
    """dmel_amd.GraphedStep: the step replays from a HIP graph; the object reads the report of the replay issued max_ahead calls
    earlier (an exact, timing-independent value: every rank of a data-parallel job reads the same) and re-captures when the
    launches that value asks for differ from what the graph holds.  Every call takes exactly k steps -- capture calls run them
    eagerly -- so the trajectory equals the eagerly issued one step for step across the 512 -> 1024 boundary."""
    from dmel_amd import GraphedStep, capi
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
    lam0, steps = 84.0, 48
    ref_layer = _mk(case, lam=lam0, sync=True)
    ref_opt = torch.optim.Adam([ref_layer.lambd], lr=0.25, capturable=True)
    ref = []
    for _ in range(steps):
        ref_opt.zero_grad(set_to_none=True)
        ref_layer(x).backward(g)
        ref_opt.step()
        ref.append(float(ref_layer.lambd.detach()))
    layer = _mk(case, lam=lam0, sync=False)
    opt = torch.optim.Adam([layer.lambd], lr=0.25, capturable=True)

    def step():
        opt.zero_grad(set_to_none=True)
        layer(x).backward(g)
        opt.step()

    gs = GraphedStep(step, [layer], max_ahead=4, steps_per_replay=k)
    got = []
    for i in range(steps // k):
        gs()
        torch.cuda.synchronize()
        got.append(float(layer.lambd.detach()))
        np.testing.assert_allclose(got[-1], ref[(i + 1) * k - 1], rtol=1e-5)      # call i has taken exactly (i + 1) k steps
    assert gs.captures >= 3, "both-guards graph, guard-free graph, and lambd crossed 85.5 (n_fft 512 -> 1024)"
    assert {capi.n_fft(v) for v in got} == {512, 1024}
    assert layer.lambd_status()["error"] == 0


def test_graphed_step_with_an_eager_forward_between_replays():
    """a validation forward through the same layer between replays used to freeze the host's picture of lambd
    (its call number ran past the one the graph reports under), so the graph was never re-captured and a boundary crossing
    produced NaN.  Execution numbers are drawn on the device now: replays and eager calls share one sequence."""
    from dmel_amd import GraphedStep, capi
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
    layer = _mk(case, lam=84.0, sync=False)
    opt = torch.optim.Adam([layer.lambd], lr=0.25, capturable=True)

    def step():
        opt.zero_grad(set_to_none=True)
        layer(x).backward(g)
        opt.step()

    gs = GraphedStep(step, [layer], max_ahead=4)
    for i in range(60):
        gs()
        if i in (3, 4, 11, 30):
            with torch.no_grad():
                y = layer(x)                                 # eager, same plan, same stream
            assert torch.isfinite(y).all()
    torch.cuda.synchronize()
    st = layer.lambd_status()
    assert st["error"] == 0 and capi.n_fft(float(layer.lambd.detach())) == 1024 and gs.captures >= 3
    with torch.no_grad():
        assert torch.isfinite(layer(x)).all()


def test_execution_numbers_and_reports():
    """every forward that executes draws the next execution number; dmel_plan_lambd_report returns the value that execution read"""
    from dmel_amd import capi
    case = C.BY_NAME["g1_c1"]
    B, M, T = case["B"], case["n_mels"], case["L"] // case["hop"] + 1
    x = torch.from_numpy(C.make_input(case).astype(np.float32)).to(DEV)
    lam = torch.tensor([64.0], dtype=torch.float32, device=DEV)
    plan = capi.Plan(case["L"], case["hop"], M, case["sr"])
    out = torch.empty((B, 1, M, T), device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    ring = capi.lambd_ring_size()                                              # 256 since round 4 (dmel_lambd_ring_size)
    n = ring + 6
    for i in range(n):
        plan.forward_dev(x.data_ptr(), B, lam.data_ptr(), out.data_ptr(), None, True, 1e-10, s)
        lam += 0.001
    torch.cuda.synchronize()
    st = plan.lambd_status()
    assert st["seq_seen"] == n and st["seq_issued"] == n and st["calls"] == n and st["error"] == 0
    assert plan.lambd_report(n) == pytest.approx(64.0 + 0.001 * (n - 1), abs=2e-3)
    assert plan.lambd_report(7) == pytest.approx(64.0 + 0.006, abs=2e-3)      # still in the ring
    assert plan.lambd_report(6) is None and plan.lambd_report(n + 1) is None and plan.lambd_report(0) is None
    # a captured forward draws a fresh number at every replay
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.forward_dev(x.data_ptr(), B, lam.data_ptr(), out.data_ptr(), None, True, 1e-10, torch.cuda.current_stream().cuda_stream)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    st = plan.lambd_status()
    assert st["seq_seen"] == n + 5 and st["calls"] == n + 1 and st["seq_issued"] == n + 5
    plan.forward_dev(x.data_ptr(), B, lam.data_ptr(), out.data_ptr(), None, True, 1e-10, s)      # eager again: number n + 6, not n + 2
    torch.cuda.synchronize()
    assert plan.lambd_status()["seq_seen"] == n + 6


def test_forced_launches_are_checked_on_the_device():
    """dmel_plan_force_launch: the caller's choice is launched as given; a value it does not cover still poisons and reports"""
    from dmel_amd import capi
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    layer = _mk(case, lam=64.0, sync=False)
    with torch.no_grad():
        y0 = layer(x)
        plan = next(iter(layer._plans.values()))
        plan.force_launch(512, 0)
        assert torch.equal(layer(x), y0) and layer.lambd_status()["guards"] == 0
        plan.force_launch(256, 2)                            # 256 does not match, its guard 512 does
        assert torch.equal(layer(x), y0) and layer.lambd_status()["guards"] == 2
        plan.force_launch(256, 1)                            # nothing covers 512
        y = layer(x)
        torch.cuda.synchronize()
        assert torch.isnan(y).all() and layer.lambd_status()["error"] == 1
        plan.force_launch(0, 0)
        with pytest.raises(RuntimeError):
            layer(x)
        assert torch.equal(layer(x), y0)


def test_backward_after_the_layer_is_gone():
    """the autograd node of torch.ops.dmel.mel_spectrogram keeps a reference on the plan, so a layer that
    is collected before backward runs (a temporary, or `del net; loss.backward()`) is no use-after-free."""
    import gc
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
    keep = _mk(case)
    keep(x).backward(g)
    want = keep.lambd.grad.clone()
    lam = torch.nn.Parameter(torch.tensor(float(case["lambd"]), device=DEV))

    def temp_layer_output():
        lay = _mk(case)
        lay.lambd = lam                                     # the parameter outlives the module
        return lay(x)

    y = temp_layer_output()
    gc.collect()
    torch.cuda.synchronize()
    junk = [torch.empty(1 << 20, device=DEV) for _ in range(8)]          # churn the allocators
    y.backward(g)
    torch.cuda.synchronize()
    assert torch.equal(lam.grad, want)
    from dmel_amd import nets
    net = nets.MelLinearNet(10, torch.tensor(float(case["lambd"])), DEV, case["n_mels"], case["sr"], case["L"], hop_length=case["hop"],
                            optimized=True).to(DEV)
    params = list(net.parameters())
    logits, _ = net(x)
    loss = logits.square().mean()
    del net, logits
    gc.collect()
    loss.backward()
    torch.cuda.synchronize()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    del junk
