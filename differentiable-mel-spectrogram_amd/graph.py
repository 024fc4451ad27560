"""A training step replayed from a HIP graph, kept valid while ``lambd`` moves -- with the same decisions on every rank.

HIP streams and graphs are this package's answer to per-launch host overhead: at BASELINE config 2 the kernels of one
step (fused forward, dot, optimizer update) take ~35 us while issuing them from Python takes ~100 us.  A step captured
once replays with one launch.  What a captured DMEL forward cannot do by itself is change its n_fft: the graph holds the
launch for one n_fft (plus guard launches for the neighbours within reach, include/dmel.h).  ``GraphedStep`` closes that
gap the way the reference's per-forward host read does (time_frequency.py:39), without the read: every forward that
executes reports ``(execution number, lambd)`` into a pinned ring, and ``__call__`` re-captures when the launches that
value asks for differ from what the graph holds.

*Which* report it looks at is what makes this safe with several ranks.  A re-capture runs the step eagerly, and the step
of a data-parallel job contains the all-reduce of ``lambd.grad``: ranks that re-captured at different calls would issue
different numbers of collectives and hang.  The pinned "latest report" depends on when a host happens to look; the report
of ONE PARTICULAR execution does not.  Call ``i`` therefore waits (an event, not a device synchronisation) for the replay
issued ``max_ahead`` calls earlier and reads exactly that replay's report (``dmel_plan_lambd_report``): every rank issues
the same forwards in the same order, ``lambd`` is identical on all of them after each all-reduced update, so every rank
reads the same number and -- the decision being a pure function of the numbers read so far (``dmel_decide_launch``) --
re-captures at the same call.  The launches are chosen for everything ``lambd`` can reach until that delayed look comes
round: ``2 * rate * ((max_ahead + 1) * forwards per replay + 2)``.
"""
from __future__ import annotations


class LaunchTracker:
    """What one plan's captured launches must cover, from the reports of particular executions only (no timing enters).

    ``decide(lambd, rate, stale_forwards) -> (n_fft, guards)`` is ``capi.decide_launch``; tests pass their own."""

    BOTH = 1.0e30        # a rate that puts both neighbours within reach: the drift is unknown until two reports have been seen

    def __init__(self, decide):
        self.decide = decide
        self.lam, self.seq, self.rate, self.n_obs = None, None, 0.0, 0
        self.seq0, self.per_replay, self.held = 0, 1, None

    def observe(self, seq: int, lam: float) -> None:
        if self.lam is not None and seq == self.seq:
            return                                       # the same execution again
        if self.lam is not None:
            self.rate = max(0.98 * self.rate, abs(lam - self.lam) / max(1, (seq - self.seq) & 0xFFFFFFFF))
        self.lam, self.seq = float(lam), int(seq)
        self.n_obs += 1

    def want(self, horizon: float):
        return tuple(self.decide(self.lam, self.rate if self.n_obs >= 2 else self.BOTH, horizon))

    def rebase(self, seq: int, lam: float, per_replay: int) -> None:
        """an exact picture (the device was idle): execution numbers of the replays that follow count from here"""
        self.observe(seq, lam)
        self.seq0, self.per_replay = int(seq), max(1, int(per_replay))


class _CudaBackend:
    """torch / HIP plumbing of GraphedStep (tests substitute a CPU stand-in to drive the decision logic)."""

    def __init__(self):
        import torch
        self.torch = torch
        self.dev = torch.cuda.current_device()

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)

    def run_eager(self, fn, n):
        t = self.torch
        side = t.cuda.Stream(self.dev)               # allocator pools, optimizer state, table builds: off the capturing stream
        side.wait_stream(t.cuda.current_stream(self.dev))
        with t.cuda.stream(side):
            for _ in range(n):
                fn()
        t.cuda.current_stream(self.dev).wait_stream(side)

    def capture(self, fn, k):
        g = self.torch.cuda.CUDAGraph()
        with self.torch.cuda.graph(g):
            for _ in range(k):
                fn()
        return g

    def event(self):
        return self.torch.cuda.Event()

    # -- input slots (GraphedStep(inputs=...)) ---------------------------------------------------------------------------------
    def empty_like(self, t):
        return self.torch.empty_like(t)

    def copy_stream(self):
        return self.torch.cuda.Stream(self.dev)

    def copy_into(self, stream, dsts, srcs, after=None):
        """dst.copy_(src) for every pair on `stream` (after the event `after`, if any); returns nothing: record_on() marks the end"""
        t = self.torch
        if after is not None:
            stream.wait_event(after)
        # (torch's own copy: a copy kernel of this package that fits into the registers the resident forward workgroups leave free
        # -- 16 per lane, 512 workgroups -- measured SLOWER, 49.8 against 42.5 us per step at BASELINE config 2, round 5)
        with t.cuda.stream(stream):
            for d, s_ in zip(dsts, srcs):
                d.copy_(s_, non_blocking=True)
                # a device batch was allocated on ANOTHER stream: without this the caching allocator may hand its memory to the next
                # allocation as soon as the caller drops the tensor (`gs.feed(inputs, labels.to(device))`), while this copy is still
                # queued behind the event above -- a silent race on the training batch
                if s_.is_cuda:
                    s_.record_stream(stream)

    def pointer_cells(self, n):
        """(n int64 cells on the device, their pinned host staging): GraphedStep(zero_copy=...) publishes batch ADDRESSES through them"""
        t = self.torch
        host = t.zeros(n, dtype=t.int64).pin_memory()
        return t.zeros(n, dtype=t.int64, device=f"cuda:{self.dev}"), host, host.numpy()      # (numpy: an item write costs 0.1 us, not 10)

    def slot_input(self, cell, like):
        from .layer import SlotInput
        t = self.torch
        if like.dtype != t.float32 or like.dim() != 2 or not like.is_cuda:
            raise ValueError("GraphedStep(zero_copy=...): only the (batch, n_points) float32 device input of the layer can be passed by address")
        return SlotInput(cell, like.shape)

    def publish_cells(self, dev, host):
        """pinned host -> device copy of the address cells on the CURRENT stream: run eagerly, or captured as the first node of a set's graph
        (a replay then reads the staging when it executes: feed() does not touch a set's staging until the replay that read it is over)"""
        dev.copy_(host, non_blocking=True)

    def check_by_address(self, t_, like):
        if t_.dtype is not like.dtype or t_.shape != like.shape or t_.device != like.device or not t_.is_contiguous():
            raise ValueError(f"feed(): a zero-copy input must be a contiguous float32 tensor of shape {tuple(like.shape)} on {like.device}")
        return t_.data_ptr()

    def record_on(self, stream=None):
        ev = self.torch.cuda.Event()
        ev.record(stream if stream is not None else self.torch.cuda.current_stream(self.dev))
        return ev

    def wait_on_current(self, ev):
        self.torch.cuda.current_stream(self.dev).wait_event(ev)


class GraphedStep:
    """``gs = GraphedStep(step_fn, layers=[net.spectrogram_layer]); for batch in loader: x_static.copy_(batch); gs()``

    ``step_fn()`` is one whole training step written as usual (zero_grad, forward, backward, [all-reduce], optimizer.step)
    on static tensors; optimizers must be capturable.  ``layers``: the MelSpectrogramLayers used inside (``lambd_sync=False``).
    ``steps_per_replay`` unrolls that many steps into one graph (the ~8 us between two graph launches are paid once per
    replay).  Every call performs exactly ``steps_per_replay`` steps: replayed from the graph, or -- on the first call and
    whenever the graph has to be captured again -- run eagerly, after which the graph is captured for the calls that follow
    (capturing executes nothing).  ``warmup`` extra eager steps before the first capture are real steps too (default 0).
    With several ranks every rank must make the same calls in the same order (as any SPMD loop does); eager forwards through
    the same layers between calls (a validation pass) are fine as long as every rank makes them.

    **Input slots** (round 5; the loop of train.py:25-49 gets a NEW batch every step): ``GraphedStep(step_fn, layers,
    steps_per_replay=K, inputs=[x_like, y_like])`` makes ``step_fn(x, y)`` take its batch as arguments and keeps two sets of K
    static slots per input.  ``gs.feed(x, y)`` copies one batch into the next slot ON A SIDE STREAM -- while the replay of the
    previous K batches runs -- and after every K-th batch joins that stream (an event) and replays the graph that reads the filled
    set, so one graph launch serves K batches and the copies cost the step nothing (one replay per step plus a serial
    ``x_static.copy_`` measured 52.5 us per step at BASELINE config 2, the slots 42.2, against 33 for a resident batch: the copies overlap the replay only in part -- the forward fills the register file of every CU, so a copy kernel runs between the forward launches).  The batches handed to
    ``feed`` must be ready when it is called (tensors already resident, pinned host memory, or produced on ``gs.copy_stream``):
    nothing on the current stream is waited for, or the copy would queue behind the running replay.  ``feed`` returns True when
    it issued the K steps.  ``flush()`` runs a partly filled set eagerly (the end of an epoch).

    **By address** (``zero_copy=[True, False]``, one flag per input): a flagged input -- the layer's ``(batch, n_points)`` float32
    waveforms, already on the device -- is not copied at all: its slot is a ``dmel_amd.SlotInput`` (an 8-byte cell holding the
    batch's address, which the fused forward reads when it runs: ``DMEL_FLAG_X_INDIRECT``), ``feed`` notes the tensor's address and
    the K addresses travel in ONE 8K-byte copy that is the first node of the set's graph.  ``step_fn`` receives the ``SlotInput`` where it received
    the slot tensor and must hand it to the layer (``net(x)`` of the reference's nets does, models.py:70) and to nothing else.
    ``feed`` keeps the K tensors alive until the replay that reads them has finished; they must not be written meanwhile.  (A by-address
    batch is read by the replay itself, on the current stream: work queued on that stream before ``feed`` -- the ``x.to(device)`` that made
    the tensor -- is ordered before it as usual; only copied inputs go through the side stream.)"""

    def __init__(self, step_fn, layers, max_ahead: int = 8, steps_per_replay: int = 1, warmup: int = 0, backend=None, decide=None,
                 inputs=None, zero_copy=None):
        self.step_fn, self.layers = step_fn, list(layers)
        self.inputs = None if inputs is None else list(inputs)
        if zero_copy is not None and self.inputs is None:
            raise ValueError("GraphedStep(zero_copy=...) needs inputs=[...]")
        self._by_addr = [False] * len(self.inputs or []) if zero_copy is None else [bool(z) for z in zero_copy]
        if len(self._by_addr) != len(self.inputs or []):
            raise ValueError("GraphedStep: zero_copy takes one flag per input")
        self._cells, self._staging, self._alive, self._last_event = None, None, [[], []], None
        self._slots, self._set, self._fill, self._copy_stream, self._set_free = None, 0, 0, None, [None, None]
        self.max_ahead, self.k, self.warmup = max(1, int(max_ahead)), int(steps_per_replay), int(warmup)
        self.backend = backend
        if decide is None:
            from . import capi
            decide = capi.decide_launch
        self._decide = decide
        if self.inputs is not None and self.warmup:
            raise ValueError("GraphedStep(inputs=...): warm-up steps would train on whatever the slots hold; run them yourself")
        self.graph = None
        self._graphs = [None, None]                      # with input slots: one graph per slot set
        self._held_plans = []                            # raw handles retained for the life of the captured graph (include/dmel.h)
        self._ring_size = None
        self.captures, self.capture_calls, self.calls = 0, [], 0
        self._trackers = {}                              # id(plan) -> LaunchTracker
        self._ring, self._replays, self._expected_calls = [], 0, {}
        for lay in self.layers:
            if getattr(lay, "lambd_sync", False):
                raise ValueError("GraphedStep needs lambd_sync=False layers (a host read cannot be captured)")
            lay.set_tracking(self.max_ahead, 3)          # eager steps: guard near boundaries only; this object watches the rest

    # -- helpers ------------------------------------------------------------------------------------------------------------
    def _plans(self):
        return [p for lay in self.layers for p in lay._plans.values()]

    def _tracker(self, plan) -> LaunchTracker:
        tr = self._trackers.get(id(plan))
        if tr is None:
            tr = self._trackers[id(plan)] = LaunchTracker(self._decide)
        return tr

    def _horizon(self, tr: LaunchTracker) -> float:
        return float((self.max_ahead + 1) * tr.per_replay + 2)

    @staticmethod
    def _status(plan) -> dict:
        st = plan.lambd_status()
        if st["error"]:
            raise RuntimeError(f"DMEL layer: a forward was not covered by the launches issued for it (lambd {st['error_lambd']} at execution "
                               f"{st['error_seq']}): its outputs were NaN.  lambd moved faster than the graph's guards allow")
        return st

    def _rebase(self, per_replay=None) -> None:
        """the device is idle: take the exact picture of every plan"""
        for plan in self._plans():
            st = self._status(plan)
            tr = self._tracker(plan)
            tr.rebase(st["seq_seen"], st["lambd_seen"], tr.per_replay if per_replay is None else per_replay[id(plan)])
            self._expected_calls[id(plan)] = st["calls"]
        self._replays = 0

    def _one_step(self, slot_set):
        """a callable that runs the next step: step_fn() -- or, with input slots, step_fn(*slot j of `slot_set`), j = 0, 1, ... per call"""
        if self.inputs is None:
            return self.step_fn
        ctr = [0]

        def one():
            if self._cells is not None and ctr[0] % self.k == 0:
                self.backend.publish_cells(self._cells[slot_set], self._staging[slot_set])      # the K addresses of this set's by-address inputs
            self.step_fn(*self._slots[slot_set][ctr[0] % self.k])
            ctr[0] += 1
        return one

    def _make_slots(self) -> None:
        b = self.backend
        n_addr = sum(self._by_addr)
        if n_addr:
            pairs = [b.pointer_cells(self.k * n_addr) for _ in range(2)]
            self._cells, self._staging, self._staging_np = [p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs]
        self._n_addr = n_addr
        self._addr_idx = [i for i, z in enumerate(self._by_addr) if z]
        self._copy_idx = [i for i, z in enumerate(self._by_addr) if not z]
        self._slots = []
        for s in range(2):
            rows = []
            for j in range(self.k):
                row, a = [], 0
                for t, by_addr in zip(self.inputs, self._by_addr):
                    if by_addr:
                        row.append(b.slot_input(self._cells[s][j * n_addr + a:j * n_addr + a + 1], t))
                        a += 1
                    else:
                        row.append(b.empty_like(t))
                rows.append(row)
            self._slots.append(rows)
        self._copy_stream = b.copy_stream()

    def _note_addresses(self, s, j, batch) -> None:
        """slot j of set s: the addresses of the by-address inputs into the host staging; the tensors stay referenced"""
        b = self.backend
        if j == 0:
            # the replay that last read this set (two replays ago) reads its staging when it EXECUTES and its batches until it ends: wait
            # for it (the other set's replay is what the device is busy with meanwhile), then the tensors it read may go
            if self._set_free[s] is not None:
                self._set_free[s].synchronize()
            self._alive[s] = []
        base, stage, alive = j * self._n_addr, self._staging_np[s], self._alive[s]
        for a, i in enumerate(self._addr_idx):
            stage[base + a] = b.check_by_address(batch[i], self.inputs[i])
            alive.append(batch[i])

    @property
    def copy_stream(self):
        """the side stream feed() copies on (produce batches on it, or hand feed() tensors that are ready)"""
        if self.backend is None:
            self.backend = _CudaBackend()
        if self.inputs is not None and self._slots is None:
            self._make_slots()
        return self._copy_stream

    def feed(self, *batch) -> bool:
        """one batch into the next slot (side stream); after the K-th: join, then the K steps (one replay).  True when they were issued.

        Host waits: with by-address inputs the first ``feed`` of a set BLOCKS the host until the replay that last read that set's
        address staging has finished (an event synchronise: the pinned staging is rewritten right after) -- the host therefore runs at most one
        replay (K steps) ahead of the device whatever ``max_ahead`` says; copied inputs wait on the device (a stream wait), not on the host."""
        if self.inputs is None:
            raise ValueError("feed() needs GraphedStep(inputs=[...])")
        if len(batch) != len(self.inputs):
            raise ValueError(f"feed() takes {len(self.inputs)} tensors")
        b = self.backend
        if b is None:
            b = self.backend = _CudaBackend()
        if self._slots is None:
            self._make_slots()
        s, j = self._set, self._fill
        # the first copy into a set waits until the replay that last read that set has finished; the others follow it in stream order
        if self._cells is not None:
            self._note_addresses(s, j, batch)
        after = self._set_free[s] if j == 0 else None
        if self._copy_idx:
            row = self._slots[s][j]
            b.copy_into(self._copy_stream, [row[i] for i in self._copy_idx], [batch[i] for i in self._copy_idx], after=after)
        self._fill = j + 1
        if self._fill < self.k:
            return False
        if self._copy_idx:
            b.wait_on_current(b.record_on(self._copy_stream))
        self._fill = 0
        self._last_event = None
        self()                                           # the K steps on set s (replayed, or eager + capture)
        # (a replay leaves the event it recorded for the run-ahead bound; a capture ran its steps eagerly and synchronised)
        self._set_free[s] = self._last_event
        self._set = s ^ 1
        return True

    def flush(self) -> int:
        """runs the batches of a partly filled set eagerly (no graph holds fewer than K steps); returns how many.

        Ends with a device-wide synchronise (the eager forwards are not the graph's: the next call needs an exact picture of lambd).  With several
        ranks every rank must call ``flush()`` at the same step with the same fill: a rank that flushes while the others replay holds their mailbox
        / RCCL exchange until it arrives there too (``feed`` counts are a pure function of the batch count, so an SPMD loop does this by itself)."""
        n, self._fill = self._fill, 0
        if n == 0 or self.inputs is None:
            return 0
        b = self.backend
        if self._copy_idx:
            b.wait_on_current(b.record_on(self._copy_stream))
        if self._cells is not None:
            b.publish_cells(self._cells[self._set], self._staging[self._set])
        for j in range(n):
            self.step_fn(*self._slots[self._set][j])
        b.synchronize()                                  # eager forwards the graph did not issue: the next call takes an exact picture
        self._set_free[self._set] = None
        self._alive[self._set] = []
        return n

    def _capture(self) -> None:
        b = self.backend
        if b is None:
            b = self.backend = _CudaBackend()
        if self.inputs is not None and self._slots is None:
            self._make_slots()
        b.synchronize()
        first = self.graph is None
        before = {id(p): p.lambd_status() for p in self._plans()}
        n_eager = self.k + (self.warmup if first else 0)
        b.run_eager(self._one_step(self._set), n_eager)  # this call's steps (and what capture needs warm)
        b.synchronize()
        plans = self._plans()                            # plans are created by the first forward
        per = {}
        for p in plans:
            st0, st1 = before.get(id(p)), self._status(p)
            per[id(p)] = max(1, (st1["calls"] - (st0["calls"] if st0 else 0)) * self.k // n_eager)
            # the eager forwards just run reported too: exact values, the same on every rank (the drift per forward is known
            # from the first capture on when this call ran two or more)
            lo = st0["seq_seen"] if (st0 and st0["known"]) else st1["seq_seen"] - min(st1["seq_seen"], 32)
            for seq in range(lo + 1, st1["seq_seen"]):
                lam = p.lambd_report(seq & 0xFFFFFFFF)
                if lam is not None:
                    self._tracker(p).observe(seq & 0xFFFFFFFF, lam)
        self._rebase(per)
        # The report of replay j is looked up (max_ahead + 1) replays later at the most: it must still be in the plan's ring by then
        # (dmel_lambd_ring_size entries, one per executed forward), or the look-up falls back to a device synchronisation whose
        # picture depends on timing -- and ranks could re-capture at different calls.  Bound the run-ahead.
        if self._ring_size is None:
            rs = getattr(plans[0], "ring_size", None) if plans else None
            if rs is None:
                try:
                    from . import capi
                    rs = capi.lambd_ring_size()
                except Exception:                        # noqa: BLE001 -- CPU stand-ins of the tests carry no library
                    rs = 1 << 30
            self._ring_size = int(rs() if callable(rs) else rs)
        worst = max([self._tracker(p).per_replay for p in plans] or [1])
        fit = (self._ring_size - 1) // max(1, worst) - 1
        if fit < 1:
            raise ValueError(f"GraphedStep: {worst} forwards per replay do not fit the report ring of {self._ring_size}: lower steps_per_replay")
        if self.max_ahead > fit:
            self.max_ahead = fit
        for plan in plans:
            tr = self._tracker(plan)
            tr.held = tr.want(self._horizon(tr))
            plan.force_launch(*tr.held)                  # the graph holds exactly these launches, on every rank
        try:
            if self.inputs is None:
                self.graph = b.capture(self.step_fn, self.k)
            else:                                        # one graph per slot set: they differ in the addresses their steps read
                self._graphs = [b.capture(self._one_step(0), self.k), b.capture(self._one_step(1), self.k)]
                self.graph = self._graphs[0]
        finally:
            for plan in plans:
                plan.force_launch(0, 0)
        for plan in plans:
            self._expected_calls[id(plan)] = plan.lambd_status()["calls"]
        # the graph now launches kernels that read these plans' tables: it keeps a reference of its own until it is replaced or dropped
        # (a layer deleted while its graph is still replayed was a use-after-free: VERDICT r03)
        held = []
        for plan in plans:
            ret = getattr(plan, "retain", None)
            if ret is not None:
                held.append(ret())
        self._release_held()
        self._held_plans = held
        self.captures += 1
        self.capture_calls.append(self.calls)
        self._ring = [b.event() for _ in range(self.max_ahead + 1)]
        self._replays = 0

    # -- one call = steps_per_replay steps ----------------------------------------------------------------------------------
    def __call__(self):
        self.calls += 1
        if self.graph is None:
            return self._capture()
        plans = self._plans()
        # forwards through these layers that this object did not issue (an eager validation pass) shift the execution numbers:
        # take an exact picture again.  Host-side call counts: the same on every rank.
        if any(self._status(p)["calls"] != self._expected_calls.get(id(p)) for p in plans):
            self.backend.synchronize()
            self._rebase()
            if any(self._tracker(p).want(self._horizon(self._tracker(p))) != self._tracker(p).held for p in plans):
                return self._capture()
        j = self._replays - self.max_ahead               # the replay (1-based, since the last exact picture) whose report is due
        if j >= 1:
            self._ring[(j - 1) % len(self._ring)].synchronize()     # bounds the queue; not a device synchronisation
            stale = False
            for plan in plans:
                tr = self._tracker(plan)
                seq = (tr.seq0 + j * tr.per_replay) & 0xFFFFFFFF
                lam = plan.lambd_report(seq)
                if lam is None:                          # overwritten (more than 64 executions ahead): fall back to an exact picture
                    stale = True
                    break
                tr.observe(seq, lam)
            if stale:
                self.backend.synchronize()
                self._rebase()
            if any(self._tracker(p).want(self._horizon(self._tracker(p))) != self._tracker(p).held for p in plans):
                return self._capture()
        (self.graph if self.inputs is None else self._graphs[self._set]).replay()
        self._last_event = self._ring[self._replays % len(self._ring)]
        self._last_event.record()
        self._replays += 1

    def steps_done_per_call(self) -> int:
        return self.k

    def _release_held(self) -> None:
        held, self._held_plans = self._held_plans, []
        if held:
            from . import capi
            for h in held:
                capi.release_handle(h)

    def close(self) -> None:
        """drops the captured graph and the plan references it held"""
        if self.graph is not None and self.backend is not None:
            self.backend.synchronize()
        self.graph = None
        self._graphs = [None, None]
        self._alive = [[], []]                           # (the device is idle: nothing reads the fed batches any more)
        self._release_held()

    def __del__(self):
        try:
            self.close()
        except Exception:                                # noqa: BLE001
            pass
