"""world_size-2 gloo test of the multi-GPU recipe: shard the batch, compute locally, all-reduce d lambd.

The local compute here is the CPU oracle (this container has no GPU); the thing under test is the
sharding arithmetic and the collective, which are the same objects bench.py and users call on GPUs.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
import numpy as np, torch, torch.distributed as dist
import dmel_amd
from dmel_amd import dist as ddist, synth
from oracle import dmel_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
B, L, sr, lam, hop, M = 7, 4000, 16000, 40.0, 160, 32
x = synth.waveforms(B, L, seed=3); g = synth.cotangent((B, 1, M, L // hop + 1), seed=4)
lo, hi = ddist.shard_bounds(B, rank, world)
O.set_threads(1)
_, t = O.forward(x[lo:hi], lam, hop, M, sr, apply_log=True)
local = torch.tensor([O.backward(g[lo:hi], t)], dtype=torch.float32)
work = ddist.allreduce_grad_(local, async_op=True)
work.wait()
_, tf = O.forward(x, lam, hop, M, sr, apply_log=True)
full = O.backward(g, tf)
assert abs(float(local) - full) <= 1e-5 * abs(full) + 1e-6, (float(local), full)
class Layer(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.lambd = torch.nn.Parameter(torch.tensor(lam))
lay = Layer(); lay.lambd.grad = torch.tensor(float(rank + 1))
ddist.allreduce_lambd_grad(lay, average=True)
assert abs(float(lay.lambd.grad) - sum(range(1, world + 1)) / world) < 1e-6
# the low-overhead reducer bench.py uses: without a GPU the native RCCL communicator cannot exist, every rank
# must agree on the torch.distributed fallback, and the result must be the same sum
sar = ddist.ScalarAllReduce()
assert sar.native is False and sar.why
v = torch.tensor([float(rank + 1), 10.0 * (rank + 1)])
t = sar.reduce_async(v, 0)
sar.wait(t, 0)
assert torch.allclose(v, torch.tensor([3.0, 30.0])), v
w = torch.tensor([float(rank + 1)])
sar.reduce(w, 0)                       # the in-stream form a training step uses before optimizer.step()
assert float(w) == 3.0
sar.close()
# a failure on ONE rank only (here: rank 1 cannot load RCCL) must make every rank fall back, not hang (the ranks run
# the same collectives in the same order whatever happens locally)
from dmel_amd import capi
if rank == 1:
    def boom():
        raise RuntimeError("injected: RCCL not loadable on this rank")
    capi.Comm.unique_id = staticmethod(boom)
sar2 = ddist.ScalarAllReduce()
assert sar2.native is False and sar2.why
z = torch.tensor([2.0 * (rank + 1)])
sar2.reduce(z, 0)
assert float(z) == 6.0
sar2.close()
dist.destroy_process_group()
print("rank", rank, "ok", lo, hi)
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_shard_bounds_cover_batch():
    from dmel_amd import dist as ddist
    for B in (0, 1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [ddist.shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        ddist.shard_bounds(4, 2, 2)


def test_allreduce_of_dlambd_world2_gloo(tmp_path):
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate()
        outs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
