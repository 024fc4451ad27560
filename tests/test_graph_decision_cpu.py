"""GraphedStep's re-capture decision on CPU: two gloo ranks, a stand-in for the device, observations that arrive at different
times on the two ranks.

A re-capture runs the step eagerly and the step contains the all-reduce of lambd.grad: ranks that re-captured at different
calls would issue different numbers of collectives and hang.  The real GraphedStep object runs here
against a fake plan / backend that mimic what the library guarantees -- execution numbers, the report ring, exact reports
through ``lambd_report`` -- and what it does not: ``lambd_status()`` returns the picture of a host that looks LATE by a
rank-dependent number of executions.  The decisions must coincide anyway, every forward must be covered by the launches of
the graph that ran it, and every call must take exactly ``steps_per_replay`` steps.
"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tools"))
import torch, torch.distributed as dist
import dmel_amd
from dmel_amd import GraphedStep, capi
import graph_rehearsal as GR
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
K = int(os.environ.get("K", "1"))


def allreduce(v):
    t = torch.tensor([v], dtype=torch.float64)
    dist.all_reduce(t)                                   # the collective a mismatched re-capture would strand
    return float(t)


ncalls = 400
MA = int(os.environ.get("MAX_AHEAD", "4"))
caps, steps, colls, uncovered = GR.rehearse(GraphedStep, capi.n_fft, capi.decide_launch, allreduce, rank, world, calls=ncalls, k=K, max_ahead=MA)
assert steps == ncalls * K, (steps, ncalls * K)          # every call = exactly K steps, capture calls included
assert not uncovered, uncovered[:3]
assert len(caps) >= 3, caps                              # both-guards graph, guard-free graph, boundary crossing(s)
mine = torch.tensor(caps + [-1] * (64 - len(caps)))
both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert all(torch.equal(b, both[0]) for b in both), [b.tolist() for b in both]
print(json.dumps(dict(rank=rank, captures=caps, collectives=colls)))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("k,max_ahead", [(1, 4), (4, 4), (20, 8)])
def test_graphed_step_recaptures_at_the_same_call_on_every_rank(tmp_path, k, max_ahead):
    """(20, 8) is bench.py's graph x20: nine replays of twenty forwards would overrun the stand-in's 64-deep report ring; GraphedStep
    bounds its run-ahead by the ring and the decisions still coincide"""
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", K=str(k), MAX_AHEAD=str(max_ahead))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate()
        outs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_launch_tracker_is_a_pure_function_of_the_reports():
    from dmel_amd.graph import LaunchTracker
    from dmel_amd import capi
    a, b = LaunchTracker(capi.decide_launch), LaunchTracker(capi.decide_launch)
    seqs = [(10, 84.0), (11, 84.05), (15, 84.25), (19, 84.45)]
    for s, v in seqs:
        a.observe(s, v)
    for s, v in seqs:
        b.observe(s, v); b.observe(s, v)                     # looking twice at the same execution changes nothing
    assert a.want(10.0) == b.want(10.0) and a.rate == b.rate and a.n_obs == b.n_obs == 4
    one = LaunchTracker(capi.decide_launch)
    one.observe(3, 128.0)
    assert one.want(5.0) == (1024, 3)                        # drift unknown: both neighbours
    assert a.want(10.0) == (512, 0) and a.want(12.0) == (512, 2)    # 84.45 + 2 * 0.05 * 12 = 85.65 passes 85.5 (6 lambd >= 513): guard 1024
