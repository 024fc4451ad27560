#!/usr/bin/env python3
"""Host time against device time of GraphedStep.feed() at BASELINE config 2 (K batches per replay from a pool > 256 MiB).
usage: python tools/time_feed.py [K] [addr|static|static_addr] [config]      addr: the batch by address (GraphedStep(zero_copy=[True]));
config: c2 (default), c3, c5, esc128, esc512, esc4096"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dmel_amd
from dmel_amd import GraphedStep, MelSpectrogramLayer, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ADDR = len(sys.argv) > 2 and sys.argv[2] == "addr"
SHAPES = {"c2": (256, 16000, 16000, 128.0, 512, 128), "c3": (32, 160000, 16000, 256.0, 512, 128), "c5": (32, 220500, 44100, 256.0, 441, 128),
          "esc128": (32, 40000, 8000, 13.3, 80, 64), "esc512": (32, 40000, 8000, 46.7, 80, 64), "esc4096": (32, 40000, 8000, 400.0, 80, 64)}
CFG = sys.argv[3] if len(sys.argv) > 3 else "c2"
B, L, sr, lam, hop, M = SHAPES[CFG]
dev = "cuda:0"
POOL = max(24, int(400e6 / (B * L * 4)) + 1)
pool = [torch.from_numpy(synth.waveforms(B, L, seed=1000 + i)).to(dev) for i in range(POOL)]
g = torch.from_numpy(synth.cotangent((B, 1, M, L // hop + 1), seed=1)).to(dev)
layer = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=dev, optimized=True, log=True).to(dev)
opt = torch.optim.Adam([layer.lambd], lr=1e-9, fused=True, capturable=True)
def step(xb):
    opt.zero_grad(set_to_none=True); layer(xb).backward(g); opt.step()
if len(sys.argv) > 2 and sys.argv[2] in ("static", "static_addr"):
    # no feed: K steps per replay on K fixed batches of the pool, handed directly / through fixed address cells
    from dmel_amd import SlotInput
    cells = torch.tensor([pool[j % POOL].data_ptr() for j in range(K)], dtype=torch.int64, device=dev)
    args = [SlotInput(cells[j:j + 1], (B, L)) if sys.argv[2] == "static_addr" else pool[j % POOL] for j in range(K)]
    ctr = [0]
    def step0():
        step(args[ctr[0] % K]); ctr[0] += 1
    gs0 = GraphedStep(step0, [layer], steps_per_replay=K)
    for _ in range(12): gs0()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(24): gs0()
        torch.cuda.synchronize()
        print(f"K={K} {sys.argv[2]}: wall {1e6*(time.perf_counter()-t0)/(24*K):.1f} us per step")
    sys.exit(0)
gs = GraphedStep(step, [layer], steps_per_replay=K, inputs=[pool[0]], zero_copy=[True] if ADDR else None)
it = [0]
def fed():
    gs.feed(pool[it[0] % POOL]); it[0] += 1
for _ in range(12 * K): fed()
torch.cuda.synchronize()
n = 24 * K
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(n): fed()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{CFG} K={K}{' by address' if ADDR else ''}: host {1e6*(t1-t0)/n:.1f} us per feed, wall {1e6*(t2-t0)/n:.1f} us per step")
