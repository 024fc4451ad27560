"""Multi-GPU use of the layer: one process per GPU, batch sharded, one scalar all-reduce.

Every (clip, frame) is independent in the forward; the only cross-GPU datum of this path is
dL/dlambd, one fp32 scalar.  ``torch.distributed`` with backend "nccl" is RCCL on
ROCm; "gloo" is used by the CPU tests.  No collective touches the forward path.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [begin, end) of the clips owned by ``rank``; the first ``global_batch % world`` ranks get one extra."""
    if world < 1 or not (0 <= rank < world) or global_batch < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(global_batch, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def allreduce_grad_(grad: torch.Tensor, async_op: bool = False, average: bool = False, group=None):
    """In-place SUM (or mean) all-reduce of a gradient tensor.  With ``async_op`` returns the work handle;
    ``handle.wait()`` orders the caller's current stream after the collective without blocking the host."""
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    if average:
        grad.div_(dist.get_world_size(group))
    return dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def allreduce_lambd_grad(layer, average: bool = True, group=None) -> None:
    """Call after backward(): reduces ``layer.lambd.grad`` over the data-parallel group.  ``average=True``
    matches mean-reduced losses (the reference's losses are, main.py:60,63) as DDP would."""
    g = layer.lambd.grad
    if g is None:
        raise RuntimeError("lambd.grad is None: run backward() first (is the layer trainable?)")
    allreduce_grad_(g, average=average, group=group)


class ScalarAllReduce:
    """Low-overhead all-reduce of small fp32 gradient buffers for step times of tens of microseconds.

    ``torch.distributed.all_reduce(async_op=True)`` costs ~45 us of host time per call on this stack, more
    than a whole forward+backward of the layer at BASELINE config 2; this class issues the same
    ``ncclAllReduce`` through the C ABI (``dmel_comm_*``: RCCL's own stream, event-ordered, ~5 us host).
    The 128-byte RCCL id travels over the already initialised ``torch.distributed`` group.  If the native
    communicator cannot be created the object falls back to ``torch.distributed`` (``self.native`` is False).
    """

    def __init__(self, group=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.native, self.why, self._comm = False, "", None
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend(group) == "nccl" else "cpu"

        def agree(ok: bool) -> bool:        # every rank calls this the same number of times, whatever happened locally
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1

        # 1. can every rank load RCCL through the C ABI?  (a local probe: nothing collective can hang on a rank that cannot)
        capi = None
        try:
            from . import capi as _capi
            _capi.Comm.unique_id()
            capi, ok = _capi, True
        except Exception as e:      # noqa: BLE001 -- any failure means: use the torch path
            ok, self.why = False, f"{type(e).__name__}: {e}"
        if not agree(ok):
            self.why = self.why or "another rank cannot load RCCL through libdmel_hip.so"
            return
        # 2. rank 0's id to everyone (unconditional broadcast), 3. the collective communicator init, 4. agree on the outcome
        ids = [capi.Comm.unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(ids, src=0, group=group)
        try:
            self._comm = capi.Comm(ids[0], self.rank, self.world)
            ok = True
        except Exception as e:      # noqa: BLE001
            ok, self.why = False, f"{type(e).__name__}: {e}"
        if agree(ok):
            self.native = True
        else:
            if self._comm is not None:
                self._comm.close()
                self._comm = None
            self.why = self.why or "another rank could not create the native communicator"

    def reduce(self, grad: torch.Tensor, stream: int) -> None:
        """SUM all-reduce of ``grad`` in place, issued IN ``stream`` like a kernel launch: what a training step needs when
        the optimizer update that follows consumes the reduced gradient.  Capturable into a HIP graph with the step."""
        if self.native:
            self._comm.allreduce(grad.data_ptr(), grad.numel(), stream)
        else:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group)

    def reduce_async(self, grad: torch.Tensor, stream: int):
        """SUM all-reduce of ``grad`` in place after the work already queued on ``stream``; returns a ticket."""
        if self.native:
            return self._comm.allreduce_async(grad.data_ptr(), grad.numel(), stream)
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group)      # stream-ordered, no overlap
        return -1

    def wait(self, ticket, stream: int) -> None:
        """Order ``stream`` after the all-reduce ``ticket`` (no host blocking)."""
        if self.native and ticket is not None and ticket >= 0:
            self._comm.wait(ticket, stream)

    def close(self):
        if self._comm is not None:
            self._comm.close()
            self._comm = None


class MailboxAllReduce:
    """The scalar all-reduce without RCCL and without a launch of its own: a peer-to-peer mailbox folded into the tail of the
    backward's dot kernel (include/dmel.h, dmel_mailbox_*).  Opt-in; ``ScalarAllReduce`` (RCCL) stays the default.

        mar = MailboxAllReduce()                  # collective: every rank creates an inbox, the IPC handles are all-gathered
        mar.attach(layer, x.device)               # from now on layer's backward leaves the SUM over ranks in lambd.grad
        ... loss.backward(); optimizer.step()     # no reduce call in the step

    Every rank must run the same number of backwards through the attached layer (as with any collective).  Ranks are other
    processes on GPUs of one node reachable through hipIpcOpenMemHandle (xGMI peers, or the same GPU).  If any rank fails to set
    the mailbox up, every rank raises (the set-up collectives are symmetric)."""

    def __init__(self, group=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        from . import capi
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._plans = []
        err, self.mailbox = "", None
        try:
            self.mailbox = capi.Mailbox(self.rank, self.world)
        except Exception as e:      # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        info = [None] * self.world
        dist.all_gather_object(info, (self.mailbox.handle if self.mailbox else None, err), group=group)
        bad = [f"rank {r}: {e}" for r, (h, e) in enumerate(info) if h is None]
        if not bad:
            try:
                self.mailbox.connect([h for h, _ in info])
            except Exception as e:      # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
        oks = [None] * self.world
        dist.all_gather_object(oks, err if not bad else "create failed elsewhere", group=group)
        bad += [f"rank {r}: {e}" for r, e in enumerate(oks) if e and not bad]
        if bad:
            self.close()
            raise RuntimeError("mailbox all-reduce could not be set up (" + "; ".join(bad) + "): use ScalarAllReduce")

    def attach(self, layer, device) -> None:
        """``layer``: a MelSpectrogramLayer (its plan for ``device`` is created if needed)."""
        import torch as _t
        plan = layer._plan_for(_t.device(device))
        plan.attach_mailbox(self.mailbox)
        if not any(p is plan for p in self._plans):
            self._plans.append(plan)

    def detach(self, layer, device) -> None:
        import torch as _t
        plan = layer._plan_for(_t.device(device))
        plan.attach_mailbox(None)
        self._plans = [p for p in self._plans if p is not plan]

    def reduce(self, grad: torch.Tensor, stream: int) -> None:
        """For gradients that did not come out of an attached layer: grad[0] = sum over ranks (one tiny launch in ``stream``)."""
        self.mailbox.allreduce(grad.data_ptr(), stream)

    def check(self) -> None:
        """Raises if an exchange gave up on a rank since the last call (and clears the error).  Not mandatory: the next forward /
        backward through an attached layer raises ``DmelError`` (DMEL_ERR_MAILBOX_TIMEOUT) by itself until this has been called."""
        e = self.mailbox.error() if self.mailbox is not None else None
        if e is not None:
            raise RuntimeError(f"mailbox all-reduce: rank {e[1]} never arrived at exchange {e[0]} (the result was NaN)")

    def set_timeout(self, seconds: float) -> None:
        """wall-clock bound of one exchange (default 120 s; 0 = wait for ever, as RCCL does)"""
        self.mailbox.set_timeout_ms(int(round(1e3 * float(seconds))))

    def close(self):
        """detaches every plan this object attached (their next backward is local again), then frees the mailbox"""
        if self.mailbox is not None:
            for plan in self._plans:
                try:
                    plan.attach_mailbox(None)
                except Exception:      # noqa: BLE001 -- a plan that is already gone has detached itself
                    pass
            self._plans = []
            self.mailbox.close()
            self.mailbox = None
