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
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import dmel_amd
from dmel_amd import GraphedStep, capi
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
K = int(os.environ.get("K", "1"))
LAG = (0, 5)[rank]              # this rank's host sees the pinned "latest" word LAG executions late


class FakePlan:
    """what a dmel_plan shows a host: execution numbers, a 64-deep report ring, a status picture that depends on timing"""
    def __init__(self, world_state):
        self.w = world_state
        self.execs, self.calls, self.ring, self.forced, self.error = 0, 0, {{}}, (0, 0), 0
        self.idle = True            # the fake device has finished everything (after backend.synchronize())
        self.uncovered = []
    def execute(self, launches):
        lam = self.w["lam"]
        self.execs += 1
        self.ring[self.execs % 64] = (self.execs, lam)
        n = capi.n_fft(lam)
        cover = [launches[0]] + ([launches[0] * 2] if launches[1] & 2 else []) + ([launches[0] // 2] if launches[1] & 1 else [])
        if n not in cover:
            self.error = 1
            self.uncovered.append((self.execs, lam, launches))
        self.idle = False
    def automatic(self):
        # the library's own choice for an eager call (from its timing-dependent picture): always safe here
        return (capi.n_fft(self.w["lam"]), 3)
    def lambd_status(self):
        seen = self.execs if self.idle else max(1, self.execs - LAG)
        seq, lam = self.ring.get(seen % 64, (0, 0.0))
        n, g = capi.decide_launch(lam, 0.0, 3.0) if seq else (0, 0)
        return dict(known=int(seq > 0), lambd_seen=lam, n_fft_seen=n, seq_issued=self.execs, seq_seen=seq, rate=0.0, guards=0,
                    error=self.error, error_seq=0, error_lambd=0.0, next_n_fft=n, next_guards=g, calls=self.calls)
    def lambd_report(self, number):
        ent = self.ring.get(number % 64)
        return ent[1] if ent and ent[0] == number else None
    def force_launch(self, n, g):
        self.forced = (n, g)


class FakeLayer:
    lambd_sync = False
    def __init__(self, plan):
        self._plans = {{0: plan}}
    def set_tracking(self, *a):
        pass


class FakeGraph:
    def __init__(self, be, launches, k):
        self.be, self.launches, self.k = be, launches, k
    def replay(self):
        for _ in range(self.k):
            self.be.real_step(self.launches)


class FakeEvent:
    def record(self): pass
    def synchronize(self): pass


class FakeBackend:
    def __init__(self, plan, world_state):
        self.plan, self.w, self.capturing, self.steps, self.collectives = plan, world_state, False, 0, 0
    def real_step(self, launches):
        self.plan.execute(launches)
        g = torch.tensor([self.w["grad"](self.w["lam"]) * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(g)                                   # the collective a mismatched re-capture would strand
        self.collectives += 1
        self.w["lam"] += self.w["lr"] * float(g) / sum(range(1, world + 1))
        self.steps += 1
    def synchronize(self):
        self.plan.idle = True
    def run_eager(self, fn, n):
        for _ in range(n):
            fn()
    def capture(self, fn, k):
        self.capturing = True
        try:
            for _ in range(k):
                fn()
        finally:
            self.capturing = False
        return FakeGraph(self, self.plan.forced, k)
    def event(self):
        return FakeEvent()


state = dict(lam=84.0, lr=0.05, grad=lambda lam: 1.0 if lam < 90.0 else -1.0)      # up through 85.33 (n_fft 512 -> 1024), back and forth around 90
plan = FakePlan(state)
be = FakeBackend(plan, state)


def step():
    plan.calls += 1
    if be.capturing:
        return                                              # a captured forward executes when its graph is replayed
    be.real_step(plan.forced if plan.forced[0] else plan.automatic())


gs = GraphedStep(step, [FakeLayer(plan)], max_ahead=4, steps_per_replay=K, backend=be)
ncalls = 400
for i in range(ncalls):
    gs()
    if i == 150:                                            # an eager validation pass through the same layer, on every rank
        plan.calls += 1
        plan.execute(plan.automatic())
assert be.steps == ncalls * K, (be.steps, ncalls * K)        # every call = exactly K steps, capture calls included
assert not plan.uncovered, plan.uncovered[:3]
assert gs.captures >= 3, gs.captures                         # both-guards graph, guard-free graph, boundary crossing(s)
mine = torch.tensor(gs.capture_calls + [-1] * (64 - len(gs.capture_calls)))
both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert all(torch.equal(b, both[0]) for b in both), [b.tolist() for b in both]
print(json.dumps(dict(rank=rank, captures=gs.capture_calls, lam=state["lam"], collectives=be.collectives)))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("k", [1, 4])
def test_graphed_step_recaptures_at_the_same_call_on_every_rank(tmp_path, k):
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", K=str(k))
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
