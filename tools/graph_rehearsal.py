"""CPU stand-in for the device side of dmel_amd.GraphedStep: what a dmel_plan shows a host (execution numbers, the 64-deep report
ring, exact reports through lambd_report, a "latest" picture that depends on WHEN the host looks) and a backend whose graphs
replay by calling the step.  Used by tests/test_graph_decision_cpu.py and by `bench.py --dry-run` to rehearse, with real
collectives between real ranks, that every rank re-captures at the same call."""
from __future__ import annotations


class FakePlan:
    ring_size = 64                            # what dmel_lambd_ring_size() says for this stand-in (GraphedStep bounds its run-ahead by it)

    def __init__(self, world_state, n_fft_of, decide, lag=0):
        self.w, self.n_fft_of, self.decide, self.lag = world_state, n_fft_of, decide, int(lag)
        self.execs, self.calls, self.ring, self.forced, self.error = 0, 0, {}, (0, 0), 0
        self.idle = True                      # the device has finished everything (after backend.synchronize())
        self.uncovered = []

    def execute(self, launches):
        lam = self.w["lam"]
        self.execs += 1
        self.ring[self.execs % 64] = (self.execs, lam)
        n = self.n_fft_of(lam)
        cover = [launches[0]] + ([launches[0] * 2] if launches[1] & 2 else []) + ([launches[0] // 2] if launches[1] & 1 else [])
        if n not in cover:
            self.error = 1
            self.uncovered.append((self.execs, lam, launches))
        self.idle = False

    def automatic(self):
        return (self.n_fft_of(self.w["lam"]), 3)       # the library's own choice for an eager call: always safe here

    def lambd_status(self):
        seen = self.execs if self.idle else max(1, self.execs - self.lag)      # a host that looks `lag` executions late
        seq, lam = self.ring.get(seen % 64, (0, 0.0))
        n, g = self.decide(lam, 0.0, 3.0) if seq else (0, 0)
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
        self._plans = {0: plan}

    def set_tracking(self, *a):
        pass


class _Graph:
    def __init__(self, be, launches, k):
        self.be, self.launches, self.k = be, launches, k

    def replay(self):
        for _ in range(self.k):
            self.be.real_step(self.launches)


class _Event:
    def record(self):
        pass

    def synchronize(self):
        pass


class FakeBackend:
    """``allreduce(value) -> sum over ranks`` is the collective a mismatched re-capture would strand"""

    def __init__(self, plan, world_state, allreduce, weight=1.0, norm=1.0):
        self.plan, self.w, self.allreduce, self.weight, self.norm = plan, world_state, allreduce, weight, norm
        self.capturing, self.steps, self.collectives = False, 0, 0

    def real_step(self, launches):
        self.plan.execute(launches)
        g = self.allreduce(self.w["grad"](self.w["lam"]) * self.weight)
        self.collectives += 1
        self.w["lam"] += self.w["lr"] * g / self.norm
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
        return _Graph(self, self.plan.forced, k)

    def event(self):
        return _Event()


def rehearse(GraphedStep, n_fft_of, decide, allreduce, rank, world, calls=400, k=1, max_ahead=4, lag=None, eval_at=(150,)):
    """Runs the real GraphedStep against the stand-in: lambd climbs through 85.33 (n_fft 512 -> 1024), then oscillates around 90.
    Returns (capture_calls, steps, collectives, uncovered)."""
    state = dict(lam=84.0, lr=0.05, grad=lambda lam: 1.0 if lam < 90.0 else -1.0)
    plan = FakePlan(state, n_fft_of, decide, lag=(0, 5, 2, 7)[rank % 4] if lag is None else lag)
    be = FakeBackend(plan, state, allreduce, weight=float(rank + 1), norm=float(sum(range(1, world + 1))))

    def step():
        plan.calls += 1
        if be.capturing:
            return                                       # a captured forward executes when its graph is replayed
        be.real_step(plan.forced if plan.forced[0] else plan.automatic())

    gs = GraphedStep(step, [FakeLayer(plan)], max_ahead=max_ahead, steps_per_replay=k, backend=be, decide=decide)
    for i in range(calls):
        gs()
        if i in eval_at:                                 # an eager validation pass through the same layer, on every rank
            plan.calls += 1
            plan.execute(plan.automatic())
    return list(gs.capture_calls), be.steps, be.collectives, list(plan.uncovered)
