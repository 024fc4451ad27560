#!/usr/bin/env python3
"""Host time to ISSUE one step (forward + dot [+ all-reduce hand-off]) through ctypes, GPU not waited for: is the eager
loop host-bound?  Uses a tiny batch so that the GPU never becomes the bottleneck."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmel_amd
from dmel_amd import capi
from dmel_amd import dist as ddist
B, L, hop, M, sr, lam = 2, 16000, 512, 128, 16000, 128.0
T = L // hop + 1
plan = capi.Plan(L, hop, M, sr, max_batch=B)
x = torch.randn(B, L, device="cuda:0"); g = torch.randn(B, 1, M, T, device="cuda:0")
out = torch.empty_like(g); tan = torch.empty_like(g); dl = [torch.zeros(1, device="cuda:0") for _ in range(32)]
st = torch.cuda.current_stream().cuda_stream
def step(k):
    plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, st)
    plan.backward(g.data_ptr(), tan.data_ptr(), out.numel(), dl[k % 32].data_ptr(), st)
for k in range(50): step(k)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for k in range(n): step(k)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("issue us/step (2 launches):", round((t1 - t0) / n * 1e6, 2), " drained after", round((t2 - t1) * 1e6, 1), "us")
import ctypes as C
Lb = capi.load()
a_f = (plan._h, x.data_ptr(), B, C.c_float(lam), 1, C.c_double(1e-10), out.data_ptr(), tan.data_ptr(), st)
a_b = [(plan._h, g.data_ptr(), tan.data_ptr(), out.numel(), 0, dl[i].data_ptr(), st) for i in range(32)]
t0 = time.perf_counter()
for k in range(n):
    Lb.dmel_forward(*a_f); Lb.dmel_backward(*a_b[k % 32])
t1 = time.perf_counter(); torch.cuda.synchronize()
print("issue us/step, raw ctypes with pre-built arguments:", round((t1 - t0) / n * 1e6, 2))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
sar = ddist.ScalarAllReduce()
print("native:", sar.native)
for k in range(50): step(k); sar.reduce_async(dl[k % 32], st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(n):
    step(k); sar.reduce_async(dl[k % 32], st)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("issue us/step with the all-reduce hand-off:", round((t1 - t0) / n * 1e6, 2))
