#!/usr/bin/env python3
"""Host time of the pieces of one training step through the nn.Module (config 2): where an eagerly issued step spends its
time on the host.  Each piece is timed alone in a loop (the GPU idles or trails; no synchronisation inside the loops)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import MelSpectrogramLayer, synth, capi

B, L, sr, lam, hop, M = 256, 16000, 16000, 128.0, 512, 128
dev = "cuda:0"
T = L // hop + 1
x = torch.from_numpy(synth.waveforms(B, L, seed=0)).to(dev)
g = torch.from_numpy(synth.cotangent((B, 1, M, T), seed=1)).to(dev)
layer = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=dev, optimized=True, log=True).to(dev)
opt = torch.optim.Adam([layer.lambd], lr=1e-3, fused=True, capturable=True)
N = 300

def t(fn, n=N):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return round(1e6 * dt, 2)

res = {}
def step():
    opt.zero_grad(set_to_none=True); layer(x).backward(g); opt.step()
res["whole_step_us"] = t(step)
with torch.no_grad():
    res["forward_no_grad_us"] = t(lambda: layer(x))
res["forward_grad_us"] = t(lambda: layer(x))
op = capi.torch_ops().mel_spectrogram.default
plan = layer._plan_for(torch.device(dev))
res["op_direct_grad_us"] = t(lambda: op(x, layer.lambd, plan.handle, 1, 1e-10, False, False))
fwd = capi.torch_ops().forward.default
res["op_forward_only_us"] = t(lambda: fwd(x, layer.lambd, plan.handle, 1, 1e-10, True, False, False))
def fb():
    layer(x).backward(g)
res["forward_backward_us"] = t(fb)
layer(x).backward(g)
res["opt_step_us"] = t(lambda: opt.step())
res["zero_grad_us"] = t(lambda: opt.zero_grad(set_to_none=True))
sgd = torch.optim.SGD([layer.lambd], lr=1e-6)
layer(x).backward(g)
res["sgd_step_us"] = t(lambda: sgd.step())
e = torch.empty(4, device=dev)
res["torch_empty_us"] = t(lambda: torch.empty((B, 1, M, T), device=dev))
res["tiny_kernel_us"] = t(lambda: e.add_(1.0))
s = torch.cuda.current_stream().cuda_stream
out = torch.empty((B, 1, M, T), device=dev); tan = torch.empty_like(out); dl = torch.zeros(1, device=dev)
res["capi_forward_us"] = t(lambda: plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, s))
res["capi_backward_us"] = t(lambda: plan.backward(g.data_ptr(), tan.data_ptr(), g.numel(), dl.data_ptr(), s))
gr = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step(); step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(gr):
    step()
res["graph_replay_host_us"] = t(gr.replay)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): gr.replay()
torch.cuda.synchronize()
res["graph_replay_step_us"] = round(1e6 * (time.perf_counter() - t0) / N, 2)
print(json.dumps(res))
