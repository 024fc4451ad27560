#!/usr/bin/env python3
"""Device-side cost of forward + backward steps of the C ABI when replayed from a HIP graph (K steps per replay), next to eager issue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import capi, synth

SHAPES = {"esc_n128": (32, 40000, 8000, 8000 * 0.01 / 6, 80, 64), "esc_n512": (32, 40000, 8000, 8000 * 0.035 / 6, 80, 64),
          "c3": (32, 160000, 16000, 256.0, 512, 128), "c5": (32, 220500, 44100, 256.0, 441, 128), "c2": (256, 16000, 16000, 128.0, 512, 128)}
K = 10
for name in sys.argv[1:] or list(SHAPES):
    B, L, sr, lam, hop, M = SHAPES[name]
    T = L // hop + 1
    x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
    g = torch.from_numpy(synth.cotangent((B, 1, M, T), seed=1)).cuda()
    out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out); dl = torch.zeros(1, device="cuda")
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    def step(s):
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, s)
        plan.backward(g.data_ptr(), tan.data_ptr(), out.numel(), dl.data_ptr(), s)
    cur = torch.cuda.current_stream()
    for _ in range(5): step(cur.cuda_stream)
    torch.cuda.synchronize()
    ref = float(dl.item())
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(K): step(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    assert float(dl.item()) == ref, (float(dl.item()), ref)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / (20 * K))
    beste = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): step(cur.cuda_stream)
        e1.record(); torch.cuda.synchronize()
        beste = min(beste, 1e3 * e0.elapsed_time(e1) / 100)
    print(f"{name}: graph x{K} {best:.2f} us/step, eager {beste:.2f} us/step")
