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


@pytest.mark.parametrize("K,n", [(4, 14), (1, 5), (10, 30)])
def test_fed_replays_equal_eager_steps_on_the_same_batches(K, n):
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

    gs = GraphedStep(step, [layer2], steps_per_replay=K, inputs=[data[0][0], data[0][1]])
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
