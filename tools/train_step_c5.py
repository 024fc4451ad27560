#!/usr/bin/env python3
"""BASELINE config 5: the DMEL front end inside a real training step.

ESC-50-shaped clips (5 s @ 44.1 kHz = 220500 samples, batch 32), lambd 256 (n_fft 2048), hop 441 (the
reference's 10 ms convention), 128 mels -> MelConvNet (models.py:105-136) -> CrossEntropy, Adam with the two
learning-rate groups of main.py:36-53.  Times the whole step with torch events and isolates the front end
(fused forward kernel + dot kernel) with the library's HIP-event profiling.  Prints one JSON line.
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import nets, synth

B, L, sr, lam, hop, M, ncls = 32, 220500, 44100, 256.0, 441, 128, 50
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
torch.manual_seed(0)
net = nets.MelConvNet(ncls, torch.tensor(lam), dev, M, sr, L, hop_length=hop, optimized=True, energy_normalize=True).to(dev)
opt = nets.make_optimizer(net, lr_model=1e-4, lr_tf=1.0)
loss_fn = torch.nn.CrossEntropyLoss()
x = torch.from_numpy(synth.waveforms(B, L, seed=0)).to(dev)
y = (torch.arange(B, device=dev) * 7) % ncls
T = L // hop + 1

def step():
    opt.zero_grad(set_to_none=True)
    logits, s = net(x)
    loss = loss_fn(logits, y)
    loss.backward()
    opt.step()
    return loss

for _ in range(3):
    step()
torch.cuda.synchronize()
plan = net.spectrogram_layer._plan_for(torch.device(dev))
plan.set_profiling(True)
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
pr = plan.get_profile()
plan.set_profiling(False)
front_ms = (pr["prep_ms"] + pr["fwd_ms"] + pr["bwd_ms"]) / steps
print(json.dumps({
    "config": f"config 5: batch {B} x {L} @ {sr} Hz, lambd {lam} (n_fft {net.spectrogram_layer.n_fft()}), hop {hop}, {M} mels, MelConvNet, Adam 2 LR groups",
    "frames_per_step": B * T, "step_ms": round(1e3 * dt, 3),
    "frontend_ms": round(front_ms, 4), "frontend_share": round(front_ms / (1e3 * dt), 4),
    "frontend_frames_per_s": round(B * T / (front_ms * 1e-3), 1),
    "frontend_kernels_ms": {"prep": round(pr["prep_ms"] / steps, 4), "fused_forward": round(pr["fwd_ms"] / steps, 4), "dot_backward": round(pr["bwd_ms"] / steps, 4)},
    "lambd_after": float(net.spectrogram_layer.lambd.detach()), "loss": float(loss), "info": plan.info()}))
