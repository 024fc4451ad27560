"""GraphedStep(inputs=..., steps_per_replay=K): K static input slots per set, the device-side copies of the next K batches issued on a
side stream while the previous replay runs, one replay per K batches (the loop of train.py:25-49 sees a new batch every
step).  The steps must be the steps: bit for bit what K eager steps on the same batches give."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batches(n, B, L, M, T):
    gen = torch.Generator().manual_seed(31)
    return [(torch.randn(B, L, generator=gen).to(DEV), torch.randn(B, 1, M, T, generator=gen).to(DEV)) for _ in range(n)]


def _layer_opt(lam0, B, L, hop, M, sr):
    from dmel_amd import MelSpectrogramLayer
    layer = MelSpectrogramLayer(torch.tensor(lam0), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=DEV, optimized=True,
                                log=True).to(DEV)
    return layer, torch.optim.Adam([layer.lambd], lr=0.05, capturable=True)


@pytest.mark.parametrize("by_address", [False, True], ids=["copied", "by_address"])
@pytest.mark.parametrize("K,n", [(4, 14), (1, 5), (10, 30)])
def test_fed_replays_equal_eager_steps_on_the_same_batches(K, n, by_address):
    from dmel_amd import GraphedStep
    B, L, hop, M, sr, lam0 = 3, 4000, 128, 32, 8000, 40.0
    T = L // hop + 1
    data = _batches(n, B, L, M, T)

    # reference: plain eager steps
    layer, opt = _layer_opt(lam0, B, L, hop, M, sr)
    ref = []
    for x, g in data:
        opt.zero_grad(set_to_none=False)
        layer(x).backward(g)
        opt.step()
        ref.append(layer.lambd.detach().clone())
    torch.cuda.synchronize()

    layer2, opt2 = _layer_opt(lam0, B, L, hop, M, sr)
    hist = torch.zeros(n + K, device=DEV)
    k = torch.zeros(1, dtype=torch.long, device=DEV)

    def step(x, g):
        opt2.zero_grad(set_to_none=False)
        layer2(x).backward(g)
        opt2.step()
        hist.index_copy_(0, k, layer2.lambd.detach().view(1))
        k.add_(1)

    gs = GraphedStep(step, [layer2], steps_per_replay=K, inputs=[data[0][0], data[0][1]], zero_copy=[True, False] if by_address else None)
    issued = 0
    for x, g in data:
        issued += K if gs.feed(x, g) else 0
    issued += gs.flush()
    torch.cuda.synchronize()
    assert issued == n
    assert gs.captures >= 1 and layer2.lambd_status()["error"] == 0
    assert torch.equal(hist[:n], torch.stack(ref)), (hist[:n] - torch.stack(ref)).abs().max().item()
    # and it keeps going after a flush: the next call takes an exact picture of the plans and replays again
    for x, g in data[:K]:
        gs.feed(x, g)
    torch.cuda.synchronize()
    assert layer2.lambd_status()["error"] == 0


@pytest.mark.parametrize("lam0,n_fft,L", [(4.0, 32, 16000), (20.0, 128, 16000), (40.0, 256, 16000), (80.0, 512, 16000), (150.0, 1024, 16000),
                                         (300.0, 2048, 16000), (600.0, 4096, 16000), (1200.0, 8192, 16000), (2500.0, 16384, 16000),
                                         (46.7, 512, 40000), (400.0, 4096, 40000), (300.0, 2048, 160000)])
@pytest.mark.parametrize("grad", [False, True], ids=["infer", "train"])
def test_a_batch_passed_by_address_gives_the_bits_of_the_batch_passed_directly(lam0, n_fft, L, grad):
    """DMEL_FLAG_X_INDIRECT: the kernels read the batch's address from a cell -- every fused transform size, with and without the tangent, short
    clips (the fused kernel adds the clip up itself) and long ones (dmel_prep_kernel's partial sums: the ESC-50 clip, BASELINE config 3's)"""
    from dmel_amd import MelSpectrogramLayer, SlotInput
    B, hop, M, sr = 5, 512, 64, 16000
    gen = torch.Generator().manual_seed(7)
    xs = [torch.randn(B, L, generator=gen).to(DEV) for _ in range(2)]
    layer = MelSpectrogramLayer(torch.tensor(lam0), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=DEV, optimized=True, log=True).to(DEV)
    layer.lambd.requires_grad_(grad)
    cell = torch.zeros(1, dtype=torch.int64, device=DEV)
    slot = SlotInput(cell, (B, L))
    g = torch.randn(B, 1, M, L // hop + 1, generator=gen).to(DEV)
    for x in xs:                                       # the same cell, two batches: only the 8 bytes change
        cell.fill_(x.data_ptr())
        want = layer(x)
        got = layer(slot)
        assert layer.plan_info()["n_fft"] == n_fft
        assert torch.equal(got, want)
        if grad:
            layer.lambd.grad = None
            want.backward(g)
            d0 = layer.lambd.grad.clone()
            layer.lambd.grad = None
            got.backward(g)
            assert torch.equal(layer.lambd.grad, d0)


def test_by_address_outside_the_fused_kernel():
    """Round 6: the direct-DFT kernel (n_fft < 32) and the global-memory transform (n_fft > 16384) read the pointer cell too --
    a lambd that leaves the fused kernel's range inside a by-address loop is served, bit for bit as with the batch passed directly."""
    from dmel_amd import MelSpectrogramLayer, SlotInput
    B, L, hop, M, sr = 2, 16000, 512, 64, 16000
    cell = torch.zeros(1, dtype=torch.int64, device=DEV)
    x = torch.randn(B, L, device=DEV)
    cell.fill_(x.data_ptr())
    for lam in (5000.0, 2.0):                              # n_fft 32768 (several launches read x) and 16
        a = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=DEV, optimized=True, log=True).to(DEV)
        b = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=DEV, optimized=True, log=True).to(DEV)
        ya, yb = a(SlotInput(cell, (B, L))), b(x)
        ya.sum().backward(); yb.sum().backward()
        torch.cuda.synchronize()
        assert a.plan_info()["kernel_path"] in (1, 3) and a.lambd_status()["error"] == 0
        assert torch.equal(ya, yb) and torch.equal(a.lambd.grad, b.lambd.grad)
    # optimized=False / a host-read lambd / a trainable filterbank: not the hot path
    for kw in (dict(optimized=False), dict(lambd_sync=True)):
        lay = MelSpectrogramLayer(torch.tensor(100.0), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=DEV, log=True,
                                  **{"optimized": True, **kw}).to(DEV)
        with pytest.raises(RuntimeError, match="SlotInput"):
            lay(SlotInput(cell, (B, L)))
    with pytest.raises(ValueError):
        SlotInput(torch.zeros(2, dtype=torch.int64, device=DEV), (B, L))


def test_the_reference_loop_on_a_net_fed_by_address():
    """train.py:25-49 on MelConvNet (models.py:105-136; no dropout): waveforms by address, labels copied, the two learning rates of
    main.py:36-53 -- against eager steps on the same batches.  (The heads run on MIOpen / rocBLAS whose backward is not promised to be
    bit-stable between an eager and a captured call: lambd and the weights are compared to 1e-5.)"""
    from dmel_amd import GraphedStep
    from dmel_amd.nets import MelConvNet
    B, L, hop, M, sr, K, n = 4, 4000, 200, 16, 8000, 3, 9
    gen = torch.Generator().manual_seed(5)
    data = [(torch.randn(B, L, generator=gen).to(DEV), torch.randint(0, 5, (B,), generator=gen).to(DEV)) for _ in range(n)]

    def make():
        torch.manual_seed(3)
        net = MelConvNet(5, 30.0, DEV, M, sr, L, hop_length=hop, optimized=True, energy_normalize=True).to(DEV)
        groups = [{"params": [p], "lr": (0.05 if name == "spectrogram_layer.lambd" else 1e-3)} for name, p in net.named_parameters()]
        return net, torch.optim.Adam(groups, capturable=True)

    crit = torch.nn.CrossEntropyLoss()
    net, opt = make()
    for x, y in data:
        opt.zero_grad(set_to_none=False)
        crit(net(x)[0], y).backward()
        opt.step()
    torch.cuda.synchronize()

    net2, opt2 = make()

    def step(x, y):
        opt2.zero_grad(set_to_none=False)
        crit(net2(x)[0], y).backward()
        opt2.step()

    gs = GraphedStep(step, [net2.spectrogram_layer], steps_per_replay=K, inputs=[data[0][0], data[0][1]], zero_copy=[True, False])
    for x, y in data:
        gs.feed(x, y)
    assert gs.flush() == 0
    torch.cuda.synchronize()
    assert net2.spectrogram_layer.lambd_status()["error"] == 0 and gs.captures >= 1
    lam, lam2 = float(net.spectrogram_layer.lambd), float(net2.spectrogram_layer.lambd)
    assert abs(lam - 30.0) > 1e-3 and abs(lam - lam2) <= 1e-5 * abs(lam), (lam, lam2)
    for (k1, p1), (k2, p2) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert k1 == k2 and torch.allclose(p1, p2, rtol=1e-4, atol=1e-6), k1


def test_by_address_batches_may_be_dropped_by_the_caller_right_after_feed():
    """feed() keeps a by-address batch alive until the replay that reads it is over: the loop of train.py:25-49 rebinds `inputs` every iteration,
    and the caching allocator hands the freed block to the next batch at once -- without the reference the replay would read the NEXT batch."""
    from dmel_amd import GraphedStep
    B, L, hop, M, sr, lam0, K, n = 8, 16000, 512, 64, 16000, 128.0, 4, 22
    T = L // hop + 1
    g = torch.randn(B, 1, M, T, generator=torch.Generator().manual_seed(2)).to(DEV)

    def batches():
        gen = torch.Generator(device=DEV).manual_seed(77)
        for _ in range(n):
            yield torch.randn(B, L, generator=gen, device=DEV)        # a fresh tensor; nobody else holds it

    layer, opt = _layer_opt(lam0, B, L, hop, M, sr)
    ref = []
    for x in batches():
        opt.zero_grad(set_to_none=False)
        layer(x).backward(g)
        opt.step()
        ref.append(layer.lambd.detach().clone())
        del x
    torch.cuda.synchronize()

    layer2, opt2 = _layer_opt(lam0, B, L, hop, M, sr)
    hist = torch.zeros(n + K, device=DEV)
    k = torch.zeros(1, dtype=torch.long, device=DEV)

    def step(x):
        opt2.zero_grad(set_to_none=False)
        layer2(x).backward(g)
        opt2.step()
        hist.index_copy_(0, k, layer2.lambd.detach().view(1))
        k.add_(1)

    gs = GraphedStep(step, [layer2], steps_per_replay=K, inputs=[torch.empty(B, L, device=DEV)], zero_copy=[True])
    ptrs = set()
    for x in batches():
        ptrs.add(x.data_ptr())
        gs.feed(x)
        del x
        junk = torch.full((B, L), float("nan"), device=DEV)          # what the allocator would hand out next if the batch had been freed
        del junk
    gs.flush()
    torch.cuda.synchronize()
    assert layer2.lambd_status()["error"] == 0
    assert len(ptrs) > 1                                              # the batches really lived at different addresses
    assert torch.equal(hist[:n], torch.stack(ref)), (hist[:n] - torch.stack(ref)).abs().max().item()


def test_copied_batches_may_be_dropped_by_the_caller_right_after_feed():
    """The same for COPIED inputs: feed() issues `slot.copy_(batch)` on its side stream, behind an event of an earlier replay;
    a device batch that the caller drops at once went back to the caching allocator of the CURRENT stream and was handed to the next
    allocation while the copy was still queued -- the slot then received the next tensor's contents.  copy_into() now records the side
    stream on the source."""
    from dmel_amd import GraphedStep
    B, L, hop, M, sr, lam0, K, n = 8, 16000, 512, 64, 16000, 128.0, 4, 22
    T = L // hop + 1
    g = torch.randn(B, 1, M, T, generator=torch.Generator().manual_seed(2)).to(DEV)

    def batches():
        gen = torch.Generator(device=DEV).manual_seed(78)
        for _ in range(n):
            yield torch.randn(B, L, generator=gen, device=DEV)

    layer, opt = _layer_opt(lam0, B, L, hop, M, sr)
    ref = []
    for x in batches():
        opt.zero_grad(set_to_none=False)
        layer(x).backward(g)
        opt.step()
        ref.append(layer.lambd.detach().clone())
        del x
    torch.cuda.synchronize()

    layer2, opt2 = _layer_opt(lam0, B, L, hop, M, sr)
    hist = torch.zeros(n + K, device=DEV)
    k = torch.zeros(1, dtype=torch.long, device=DEV)

    def step(x):
        opt2.zero_grad(set_to_none=False)
        layer2(x).backward(g)
        opt2.step()
        hist.index_copy_(0, k, layer2.lambd.detach().view(1))
        k.add_(1)

    gs = GraphedStep(step, [layer2], steps_per_replay=K, inputs=[torch.empty(B, L, device=DEV)])
    for x in batches():
        torch.cuda.current_stream().synchronize()                    # the batch is ready (feed() waits for nothing on the current stream)
        gs.feed(x)
        del x
        junk = torch.full((B, L), float("nan"), device=DEV)          # what the allocator hands out next if the batch's block is free
        del junk
    gs.flush()
    torch.cuda.synchronize()
    assert layer2.lambd_status()["error"] == 0
    assert torch.equal(hist[:n], torch.stack(ref)), (hist[:n] - torch.stack(ref)).abs().max().item()
