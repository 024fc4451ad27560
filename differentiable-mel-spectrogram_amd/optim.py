"""Opt-in optimizer for the layer's window parameter.

The reference trains ``lambd`` with ``torch.optim.Adam`` (main.py:40-53: one parameter group per learning rate, ``lr_tf`` for the
spectrogram layer).  On a step of a few tens of microseconds the two kernels torch launches for that ONE scalar are a quarter of
the step, and the trainable filterbank's 65 664 entries land on two workgroups (42 us); ``LambdAdam`` applies the same update -- torch's fused / capturable arithmetic: fp32, step count kept on the
device, no host synchronisation -- as one launch per parameter through ``dmel_adam_step`` (include/dmel.h).  The rest of the model
keeps its own optimizer; ``torch.optim.Adam`` on ``lambd`` keeps working and is what bench.py's headline uses.

    opt_model = torch.optim.Adam(net.model_parameters(), lr=lr_model)
    opt_tf = dmel_amd.LambdAdam([net.spectrogram_layer.lambd], lr=lr_tf)

``LambdAdam([layer.lambd], lr=lr_tf, fused_into_backward=layer)`` (round 5) goes one step further: the update of ``lambd`` is applied by
the workgroup that finishes the backward's dot product (``dmel_plan_attach_adam``: the same arithmetic on the gradient it has just
written), and ``step()`` launches nothing for that parameter -- the step of train.py:47-49 is then forward kernel + dot kernel.  For
steps with ONE backward per update whose only gradient of the layer is ``lambd.grad`` (no trainable filterbank, no waveform gradient,
no gradient accumulation, no all-reduce of ``lambd.grad`` outside the layer's mailbox).  The plans carry the pointers to ``lambd`` and its
Adam state only BETWEEN ``zero_grad()`` (which attaches them, and keeps the tensors referenced from the plan) and ``step()`` (which takes
them off again): a backward outside that bracket -- an evaluation pass, a gradient check, another optimizer's step -- is a plain backward, and
an optimizer that is dropped takes its attachment with it (a finalizer).  A HIP graph captured around the whole step holds the update
in its dot-kernel node (and the step function, hence the optimizer and its state, through ``GraphedStep``).

``lr`` is passed by value at every ``step()``: a HIP graph captured around the step holds the value of the capture (as with
torch's capturable Adam and a Python-float lr), so re-capture when a scheduler changes it.
"""
from __future__ import annotations

import weakref

import torch

from . import capi


def _detach_plans(plans: list) -> None:
    """takes the fused update off every plan in `plans` (a list the optimizer shares with its finalizer) and empties it"""
    while plans:
        plan = plans.pop()
        try:
            plan.attach_adam(0)
        except Exception:                                  # noqa: BLE001 -- interpreter shutdown: the library may be gone already
            pass


class LambdAdam(torch.optim.Optimizer):
    """torch.optim.Adam (no amsgrad) for the layer's fp32 CUDA parameters (lambd, the trainable filterbank), one launch per parameter."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, maximize: bool = False,
                 fused_into_backward=None):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0) or weight_decay < 0.0:
            raise ValueError("LambdAdam: lr / betas / eps / weight_decay out of range")
        self._fused_layer = fused_into_backward
        self._fused_plans = []                     # plans that carry the update right now (shared with the finalizer below)
        self._finalizer = weakref.finalize(self, _detach_plans, self._fused_plans)
        # capturable=True: torch's Optimizer.load_state_dict then moves state["step"] to the parameter's device as fp32 (its
        # _process_value_according_to_param_policy); without the key a checkpoint loaded with map_location="cpu" left the step count
        # on the host and step() handed a host pointer to the kernel
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, maximize=maximize, capturable=True))
        self._tickets = {}        # id(param) -> int32 device word re-armed by every launch: private, neither saved nor cast by load_state_dict
        for group in self.param_groups:
            for p in group["params"]:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise ValueError("LambdAdam takes contiguous fp32 CUDA parameters (the layer's lambd and mel_fb); use torch.optim for the rest")

        if self._fused_layer is not None:
            lay = self._fused_layer
            if getattr(lay, "learnable_fb", False) or getattr(lay, "lambd_sync", False):
                raise ValueError("LambdAdam(fused_into_backward=layer): the layer's only parameter must be lambd, kept on the device (lambd_sync=False)")
            ps = [p for g in self.param_groups for p in g["params"]]
            if len(ps) != 1 or ps[0] is not lay.lambd or ps[0].numel() != 1:
                raise ValueError("LambdAdam(fused_into_backward=layer) takes exactly [layer.lambd]")

    def _attach(self):
        """(re-)attaches the update to the layer's plans with the current hyper-parameters: plans appear with the first forward on a device"""
        lay, group = self._fused_layer, self.param_groups[0]
        p = group["params"][0]
        st = self._checked_state(p)
        b1, b2 = group["betas"]
        _detach_plans(self._fused_plans)                        # (plans of an earlier bracket that never reached step())
        for plan in lay._plans.values():
            plan.attach_adam(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), group["lr"], b1, b2,
                             group["eps"], group["weight_decay"], group["maximize"], keep=(p, st["exp_avg"], st["exp_avg_sq"], st["step"]))
            self._fused_plans.append(plan)

    def detach(self):
        """stops the fused update for good (the plans go back to writing the gradient only; step() launches the update again)"""
        _detach_plans(self._fused_plans)
        self._fused_layer = None

    def zero_grad(self, set_to_none: bool = True):
        # the natural place before every backward: hyper-parameters changed by a scheduler, plans created since, state loaded from a checkpoint
        if self._fused_layer is not None:
            self._attach()
        return super().zero_grad(set_to_none=set_to_none)

    def _ticket(self, p):
        t = self._tickets.get(id(p))
        if t is None or t.device != p.device:
            t = self._tickets[id(p)] = torch.zeros((), dtype=torch.int32, device=p.device)
        return t

    def _checked_state(self, p):
        """the parameter's state, created on first use; whatever load_state_dict left is validated (and repaired where that is exact)
        before a pointer of it reaches the kernel"""
        st = self.state[p]
        st.pop("ticket", None)                           # checkpoints written by round 3 carried the ticket word: drop it
        if "step" not in st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            return st
        step = st["step"]
        if not torch.is_tensor(step):
            step = torch.tensor(float(step), dtype=torch.float32)
        if step.device != p.device or step.dtype != torch.float32 or step.dim() != 0:
            st["step"] = step.detach().to(device=p.device, dtype=torch.float32).reshape(())
        for key in ("exp_avg", "exp_avg_sq"):
            m = st.get(key)
            if m is None or m.shape != p.shape:
                raise RuntimeError(f"LambdAdam: state['{key}'] is missing or does not have the parameter's shape")
            if m.device != p.device or m.dtype != torch.float32 or not m.is_contiguous():
                st[key] = m.detach().to(device=p.device, dtype=torch.float32).contiguous()
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._fused_layer is not None:
            if not self._fused_plans:
                raise RuntimeError("LambdAdam(fused_into_backward=layer): call zero_grad() before the backward (it attaches the update to the layer's plans)")
            _detach_plans(self._fused_plans)             # the backward's dot kernel has applied the update; later backwards are plain ones
            return loss
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                    raise RuntimeError("LambdAdam: the gradient must be a contiguous fp32 tensor on the parameter's device")
                st = self._checked_state(p)
                with torch.cuda.device(p.device):
                    capi.adam_step(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(),
                                   self._ticket(p).data_ptr(), p.numel(), group["lr"], b1, b2, group["eps"], group["weight_decay"], group["maximize"],
                                   torch.cuda.current_stream(p.device).cuda_stream)
        return loss
