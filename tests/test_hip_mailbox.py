"""GPU tests of the peer-to-peer mailbox all-reduce (include/dmel.h: dmel_mailbox_*, dmel_plan_attach_mailbox): the exchange of
lambd.grad folded into the tail of the backward's dot kernel.  One GPU is all this pool offers, so the cross-process path runs as
two ranks that SHARE device 0: inboxes exported / opened through HIP IPC, system-scope stores into the other process's memory,
polls of the own inbox -- everything a second GPU over xGMI would exercise except the link itself."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import cases as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mk(case, lam=None):
    from dmel_amd import MelSpectrogramLayer
    lam = case["lambd"] if lam is None else lam
    return MelSpectrogramLayer(torch.tensor(float(lam), dtype=torch.float32), n_mels=case["n_mels"], n_points=case["L"],
                               sample_rate=case["sr"], f_min=case["f_min"], f_max=case["f_max"], hop_length=case["hop"], device=DEV,
                               optimized=True, normalize_window=case["normalize_window"], log=True).to(DEV)


def test_one_rank_mailbox_is_the_identity():
    """world = 1: the granule goes to the rank's own inbox and comes back; results equal the plain backward bit for bit, over
    many steps (both parity slots), also replayed from a HIP graph"""
    from dmel_amd import capi
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
    plain, boxed = _mk(case), _mk(case)
    mb = capi.Mailbox(0, 1)
    mb.connect([mb.handle])
    boxed._plan_for(torch.device(DEV)).attach_mailbox(mb)
    for _ in range(5):
        for lay in (plain, boxed):
            lay.lambd.grad = None
            lay(x).backward(g)
        assert torch.equal(plain.lambd.grad, boxed.lambd.grad)
    buf = torch.tensor([3.25], device=DEV)
    mb.allreduce(buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert float(buf) == 3.25 and mb.error() is None
    # captured: the exchange is part of the dot kernel, nothing else to capture
    opt = torch.optim.Adam([boxed.lambd], lr=1e-3, capturable=True)
    ref_opt = torch.optim.Adam([plain.lambd], lr=1e-3, capturable=True)

    def step(lay, o):
        o.zero_grad(set_to_none=True)
        lay(x).backward(g)
        o.step()

    step(boxed, opt); step(plain, ref_opt)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step(boxed, opt)
    for _ in range(6):
        gr.replay()
        step(plain, ref_opt)
    torch.cuda.synchronize()
    assert torch.equal(plain.lambd.detach(), boxed.lambd.detach()) and mb.error() is None
    boxed._plan_for(torch.device(DEV)).attach_mailbox(None)


WORKER = r'''
import os, sys, json, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
import numpy as np, torch, torch.distributed as dist
import cases as C
import dmel_amd
from dmel_amd import MelSpectrogramLayer, capi, dist as ddist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
DEV = "cuda:0"                      # every rank on the one GPU of the box
case = C.BY_NAME["g2_c2"]
x = torch.from_numpy(C.make_input(case).astype(np.float32)).to(DEV)
g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
B = x.shape[0]
lo, hi = ddist.shard_bounds(B, rank, world)

def mk():
    return MelSpectrogramLayer(torch.tensor(float(case["lambd"])), n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
                               hop_length=case["hop"], device=DEV, optimized=True, log=True).to(DEV)

full, mine = mk(), mk()
mar = ddist.MailboxAllReduce()
mar.attach(mine, DEV)                                      # default bound: 120 s of wall clock, no poll count
opt_full = torch.optim.Adam([full.lambd], lr=0.05)
opt_mine = torch.optim.Adam([mine.lambd], lr=0.05)
for step in range(7):                                      # odd and even exchanges: both parity slots of every inbox
    opt_full.zero_grad(set_to_none=True); opt_mine.zero_grad(set_to_none=True)
    full(x).backward(g)                                   # the whole batch on one rank: the reference
    if step == 3 and rank == world - 1:
        time.sleep(1.5)                                   # a LATE rank (a checkpoint, a slow loader): the others wait, nobody gets NaN
    mine(x[lo:hi]).backward(g[lo:hi])                     # this rank's shard; the backward's kernel exchanges the partial sums
    torch.cuda.synchronize()
    mar.check()
    a, b = float(full.lambd.grad), float(mine.lambd.grad)
    assert abs(a - b) <= 2e-5 * abs(a) + 1e-6, (step, a, b)
    allg = [None] * world
    dist.all_gather_object(allg, b)
    assert all(v == allg[0] for v in allg), allg           # every rank holds the same bits
    opt_full.step(); opt_mine.step()
# a value already in memory: the tiny-launch form
v = torch.tensor([float(rank + 1)], device=DEV)
mar.reduce(v, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); mar.check()
assert float(v) == world * (world + 1) / 2
# a DEAD rank (the last one never joins this exchange): bounded wait, NaN, and the error reaches the caller BY ITSELF at the next
# call through the attached layer -- not a hung device, not a silent NaN in lambd
dist.barrier()
mar.set_timeout(0.4)
if rank != world - 1:
    t0 = time.time()
    mine.lambd.grad = None
    mine(x[lo:hi]).backward(g[lo:hi])
    torch.cuda.synchronize()
    assert 0.3 < time.time() - t0 < 30.0
    assert torch.isnan(mine.lambd.grad).all()
    try:
        mine(x[lo:hi]); raise SystemExit("the timeout went unnoticed")
    except RuntimeError as e:                             # DMEL_ERR_MAILBOX_TIMEOUT through torch.ops.dmel (a RuntimeError) or ctypes (DmelError)
        assert "never arrived" in str(e), e
    try:
        mar.check(); raise SystemExit("check() did not report the timeout")
    except RuntimeError as e:
        assert "never arrived" in str(e)
    mine(x[lo:hi])                                        # the error word has been read: the layer works again
dist.barrier()
# closing while the layer still holds the mailbox: the plan is detached first (ADVICE r03: a freed mailbox was dereferenced)
mar.close()
mine.lambd.grad = None
mine(x[lo:hi]).backward(g[lo:hi])                          # local again
torch.cuda.synchronize()
assert torch.isfinite(mine.lambd.grad).all()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_exchange_through_ipc_mapped_inboxes(tmp_path, world):
    """two, four and eight ranks (config 4's world size) sharing the one GPU: both parity slots, a late rank (waited for), a dead rank (NaN + DMEL_ERR_MAILBOX_TIMEOUT
    at the next call), close() with a plan still attached"""
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate()
        outs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_mailbox_destroy_refuses_while_a_plan_holds_it():
    from dmel_amd import capi
    case = C.BY_NAME["g1_c1"]
    lay = _mk(case)
    plan = lay._plan_for(torch.device(DEV))
    mb = capi.Mailbox(0, 1)
    mb.connect([mb.handle])
    plan.attach_mailbox(mb)
    with pytest.raises(capi.DmelError, match="still attached"):
        mb.close()
    plan.attach_mailbox(None)
    mb.close()
