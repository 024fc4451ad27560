#!/usr/bin/env python3
"""A training step with the filterbank as a second parameter (learnable_fb=True: dense bank in the forward, dmel_backward_fb in the
backward) at BASELINE config 2: GPU time per step from a HIP-graph replay of the nn.Module step, and for comparison the
lambd-only step.  One JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import MelSpectrogramLayer, synth
from bench import CONFIGS

B, L, sr, lam, hop, M = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
dev = "cuda:0"
x = torch.from_numpy(synth.waveforms(B, L, seed=0)).to(dev)
T = L // hop + 1
g = torch.from_numpy(synth.cotangent((B, 1, M, T), seed=1)).to(dev)
res = {}
for name, lfb, native, kw in (("lambd_only", False, False, {}),
                              ("lambd_and_filterbank_recompute_fp32", True, False, dict(save_spec=False)),
                              ("lambd_and_filterbank", True, False, {}),
                              ("lambd_and_filterbank_bf16x3", True, False, dict(mfma="bf16x3")),
                              ("lambd_and_filterbank_LambdAdam", True, True, {}),
                              ("lambd_and_filterbank_bf16x3_LambdAdam", True, True, dict(mfma="bf16x3"))):
    layer = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=dev, optimized=True,
                                log=True, learnable_fb=lfb, **kw).to(dev)
    opt = dmel_amd.LambdAdam(layer.parameters(), lr=0.0) if native else torch.optim.Adam(layer.parameters(), lr=0.0, fused=True, capturable=True)

    def step():
        opt.zero_grad(set_to_none=True)
        layer(x).backward(g)
        opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    eager = 1e3 * e0.elapsed_time(e1) / n
    graphed = None
    try:
        gs = dmel_amd.GraphedStep(step, [layer])
        for _ in range(16): gs()                             # past GraphedStep's delayed look (the guard-free graph is in place)
        torch.cuda.synchronize()
        ng = 100
        e0.record()
        for _ in range(ng): gs()
        e1.record(); torch.cuda.synchronize()
        graphed = 1e3 * e0.elapsed_time(e1) / ng
    except Exception as e:                                   # noqa: BLE001
        graphed = f"{type(e).__name__}: {e}"[:200]
    res[name] = {"eager_us": round(eager, 1), "graph_us": graphed if isinstance(graphed, str) else round(graphed, 1)}
print(json.dumps(res))
