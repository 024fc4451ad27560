"""A training step replayed from a HIP graph, kept valid while ``lambd`` moves.

HIP streams and graphs are this package's answer to per-launch host overhead: at BASELINE config 2 the kernels of one
step (fused forward, dot, optimizer update) take ~40 us while issuing them from Python takes ~100 us.  A step captured
once replays with one launch.  What a captured DMEL forward cannot do by itself is change its n_fft: the graph holds the
launch for the n_fft the host saw at capture time (plus guard launches for the neighbours that were within reach then, see
include/dmel.h).  ``GraphedStep`` closes that gap the way the reference's per-forward host read does, without the read: the
kernels report ``lambd`` into pinned memory at every replay, ``__call__`` looks at that word (no synchronisation) and
re-captures when the launch the library would choose now differs from what the graph holds.  Replays are allowed to queue
``max_ahead`` deep (an event ring, not a device synchronisation), so the picture is never older than that.
"""
from __future__ import annotations

import torch


class GraphedStep:
    """``gs = GraphedStep(step_fn, layers=[net.spectrogram_layer]); for _ in range(n): gs()``

    ``step_fn()`` is one whole training step written as usual (zero_grad, forward, backward, optimizer.step) on static
    tensors; optimizers must be capturable.  ``layers``: the MelSpectrogramLayers used inside (with ``lambd_sync=False``).
    ``steps_per_replay`` unrolls that many steps into one graph (the ~8 us between two graph launches are paid once per
    replay).  The first call runs eagerly (warm-up) and captures."""

    def __init__(self, step_fn, layers, max_ahead: int = 8, steps_per_replay: int = 1, warmup: int = 3):
        self.step_fn, self.layers = step_fn, list(layers)
        self.max_ahead, self.k, self.warmup = int(max_ahead), int(steps_per_replay), int(warmup)
        self.graph, self.held = None, None
        self.captures = 0
        self._ring, self._i = [], 0
        for lay in self.layers:
            if getattr(lay, "lambd_sync", False):
                raise ValueError("GraphedStep needs lambd_sync=False layers (a host read cannot be captured)")
            lay.set_tracking(self.max_ahead, 3)          # guard near boundaries only, also under capture: this object watches

    def _decision(self):
        out = []
        for lay in self.layers:
            for plan in lay._plans.values():
                st = plan.lambd_status()
                if st["error"]:
                    raise RuntimeError(f"DMEL layer: a replay was not covered by the graph's launches (lambd {st['error_lambd']} at call "
                                       f"{st['error_seq']}): outputs were NaN.  lambd moved faster than max_ahead={self.max_ahead} replays allow")
                out.append((st["next_n_fft"], st["next_guards"]))
        return tuple(out)

    def _capture(self):
        dev = torch.cuda.current_device()
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                    # eager steps: allocator, optimizer state, lambd tracking, table builds
            for _ in range(self.warmup if self.graph is None else 1):
                self.step_fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.held = self._decision()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(self.k):
                self.step_fn()
        self.graph = g
        self.captures += 1
        self._ring = [torch.cuda.Event() for _ in range(max(1, self.max_ahead))]
        self._i = 0

    def __call__(self):
        if self.graph is None or self._decision() != self.held:
            self._capture()
        self.graph.replay()
        ev = self._ring[self._i % len(self._ring)]
        if self._i >= len(self._ring):
            ev.synchronize()                             # the replay issued max_ahead calls ago: bounds the queue, not a device sync
        ev.record()
        self._i += 1

    def steps_done_per_call(self) -> int:
        return self.k
