"""the dynamic n_fft switch pinned against the REFERENCE.  tests/golden/g12_traj_*.npz hold 48 Adam steps of the
reference's own MelSpectrogramLayer + log (models.py:14-56,73; torch.optim.Adam as in main.py:52) whose lambd crosses a power-of-two
boundary of next_power_of_2(int(6 lambd)) (time_frequency.py:39,60-65) twice -- down and back up.  The same steps here, (i) through
GraphedStep (lambd on the device, guard launches, re-captures) and (ii) with lambd_sync=True (a host read per forward, as the
reference does), must visit the same n_fft at every step and keep lambd within 1e-4 relative of the reference's."""
import os

import numpy as np
import pytest
import torch

import cases as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["g12_traj_512_1024", "g12_traj_1024_2048"]


def _run(gold, graphed):
    from dmel_amd import GraphedStep, MelSpectrogramLayer, capi
    cfg = {k: float(gold[k]) for k in ("lam0", "B", "L", "hop", "M", "sr", "lr", "steps", "flip")}
    steps, flip = int(cfg["steps"]), int(cfg["flip"])
    x_np, g_np = C.traj_inputs(cfg)
    x, g = torch.from_numpy(x_np).to(DEV), torch.from_numpy(g_np).to(DEV)
    layer = MelSpectrogramLayer(torch.tensor(cfg["lam0"], dtype=torch.float32), n_mels=int(cfg["M"]), n_points=int(cfg["L"]),
                                sample_rate=int(cfg["sr"]), hop_length=int(cfg["hop"]), device=DEV, optimized=True, log=True,
                                lambd_sync=not graphed).to(DEV)
    opt = torch.optim.Adam([layer.lambd], lr=cfg["lr"], capturable=True)
    # the history is written on the device (a captured step cannot index with a host counter)
    hist = torch.zeros(steps + 1, 2, dtype=torch.float64, device=DEV)          # (lambd before the step, its gradient)
    k = torch.zeros(1, dtype=torch.long, device=DEV)
    sign = torch.ones(1, device=DEV)

    def step(xb=x):
        opt.zero_grad(set_to_none=False)
        (layer(xb) * (g * sign)).sum().backward()
        hist.index_copy_(0, k, torch.stack([layer.lambd.detach().double(), layer.lambd.grad.double()]).view(1, 2))
        opt.step()
        k.add_(1)
        sign.copy_(torch.where(k < flip, 1.0, -1.0).float())

    if graphed == "by_address":
        # the batch handed to the captured step by address (DMEL_FLAG_X_INDIRECT): re-captures keep the pointer cells
        run = GraphedStep(step, [layer], steps_per_replay=2, inputs=[x], zero_copy=[True])
        for _ in range(steps):
            run.feed(x)
        run.flush()
    else:
        run = GraphedStep(step, [layer], steps_per_replay=1) if graphed else step
        for _ in range(steps):
            run()
    torch.cuda.synchronize()
    assert layer.lambd_status()["error"] == 0
    h = hist.cpu().numpy()
    lam = np.concatenate([h[:steps, 0], [float(layer.lambd.detach())]])
    return lam, h[:steps, 1], np.asarray([capi.n_fft(float(v)) for v in lam[:steps]]), (run.captures if graphed else 0)


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("graphed", [True, False, "by_address"], ids=["graphed_device_lambd", "lambd_sync", "graphed_batch_by_address"])
def test_training_trajectory_across_an_n_fft_boundary_matches_the_reference(name, graphed):
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    lam, dlam, nfft, captures = _run(gold, graphed)
    assert int(gold["crossings"]) >= 2
    assert np.array_equal(nfft, gold["n_fft"]), (nfft.tolist(), gold["n_fft"].tolist())
    rel = np.abs(lam - gold["lam"]) / np.abs(gold["lam"])
    assert rel.max() <= 1e-4, (int(rel.argmax()), float(rel.max()))
    # the gradient of every step too (1e-4 of its own magnitude, or of the largest one where a step's gradient is small)
    scale = np.maximum(np.abs(gold["dlam"]), 1e-2 * np.abs(gold["dlam"]).max())
    assert (np.abs(dlam - gold["dlam"]) / scale).max() <= 2e-4, float((np.abs(dlam - gold["dlam"]) / scale).max())
    if graphed:
        assert captures >= 2, captures              # the graph was captured again when lambd moved to another n_fft
