"""Host-side mirror of the reference's layer interface, running on libdmel_hip.so.

``MelSpectrogramLayer`` keeps the constructor, attributes, parameter name (``lambd``), output
shape/dtype and error behaviour of the reference's ``models.MelSpectrogramLayer``
(models.py:14-56) so that the wrapping nets (models.py:58-166), the two-LR-group optimizer keyed
on ``"spectrogram_layer.lambd"`` (main.py:36-48) and ``load_state_dict(strict=True)``
(utils.py:270) work unchanged.  The arithmetic runs in the HIP kernels behind the C ABI
(``include/dmel.h``), reached in two ways: the hot path goes through the torch-registered ops
``torch.ops.dmel.*`` (csrc/dmel_torch.cpp, a C++ autograd function: forward + backward to
``lambd.grad``), the optional gradients (waveform, learnable filterbank) and the DSPEC layer through
``ctypes`` (``capi.py``).  torch is used for device memory, streams and autograd plumbing only.
There is no CPU fallback.

``lambd`` normally never leaves the device (``lambd_sync=False``): the kernels read it themselves and
check the n_fft they were launched for (include/dmel.h, dmel_forward_dev), so a training step queues
without the host waiting and can be captured into a HIP graph.  ``lambd_sync=True`` reads it to the
host at every forward, as the reference does (time_frequency.py:39).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import capi


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """``torch.cuda.device(dev)`` only when ``dev`` is not already current: the context manager costs ~8 us of host time per
    use, which is a third of the forward kernel at BASELINE config 2."""

    __slots__ = ("ctx",)

    def __init__(self, dev):
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.ctx = None if idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class _DmelFunction(torch.autograd.Function):
    """forward: dmel_forward (carries d out / d lambd); backward: dmel_backward (one dot product) and, when a
    filterbank tensor that requires grad was passed, dmel_backward_fb (adjoint of models.py:53)."""

    @staticmethod
    def forward(ctx, x, lambd, plan, lam_host, log, eps, full_window=False, fb=None, out_dtype=torch.float32):
        B = x.shape[0]
        want_tangent = ctx.needs_input_grad[1]
        want_fb = fb is not None and ctx.needs_input_grad[7]
        want_x = ctx.needs_input_grad[0]
        # dL/dx and dL/dfb rebuild 1 / (mel + eps) from the saved log output: that needs the fp32 values, so a bf16 output
        # is produced by rounding an fp32 one here instead of in the kernel (bit-identical either way)
        round_later = out_dtype == torch.bfloat16 and log and (want_x or want_fb)
        kdtype = torch.float32 if round_later else out_dtype
        out = torch.empty((B, 1, plan.n_mels, plan.n_time), dtype=kdtype, device=x.device)     # models.py:36 (fp32 there)
        tangent = torch.empty(out.shape, dtype=torch.float32, device=x.device) if want_tangent else None
        flags = capi.DMEL_FLAG_FULL_WINDOW if full_window else 0
        with _on_device(x.device):
            plan.forward(x.data_ptr(), B, lam_host, out.data_ptr(), tangent.data_ptr() if want_tangent else None,
                         log, eps, _stream_ptr(x.device),
                         extra_flags=flags | (capi.DMEL_FLAG_OUT_BF16 if kdtype == torch.bfloat16 else 0))
        ctx.plan = plan
        ctx.lambd_shape = lambd.shape
        ctx.lambd_dtype = lambd.dtype
        ctx.want_tangent, ctx.want_fb, ctx.want_x = want_tangent, want_fb, want_x
        ctx.fb_args = (lam_host, bool(log), flags, None if fb is None else (tuple(fb.shape), fb.dtype))
        saved = []
        if want_tangent:
            saved.append(tangent)
        if want_fb or want_x:
            saved.append(x)
            if log:
                saved.append(out)
        ctx.save_for_backward(*saved)
        return out.to(torch.bfloat16) if round_later else out

    @staticmethod
    def backward(ctx, grad_out):
        saved = list(ctx.saved_tensors)
        bf16 = grad_out.dtype == torch.bfloat16          # gradient of a bf16 output: read as it is, widened in the kernel
        g = grad_out
        if not bf16 and g.dtype != torch.float32:
            g = g.to(torch.float32)
        if not g.is_contiguous():
            g = g.contiguous()
        dl = gfb = gx = None
        with _on_device(g.device):
            if ctx.want_tangent:
                tangent = saved.pop(0)
                dl = torch.empty((1,), dtype=torch.float32, device=g.device)
                ctx.plan.backward(g.data_ptr(), tangent.data_ptr(), g.numel(), dl.data_ptr(), _stream_ptr(g.device), grad_bf16=bf16)
                if ctx.lambd_shape != dl.shape:
                    dl = dl.reshape(ctx.lambd_shape)
                if ctx.lambd_dtype != torch.float32:
                    dl = dl.to(ctx.lambd_dtype)
            if ctx.want_fb or ctx.want_x:
                lam_host, log, flags, fb_meta = ctx.fb_args
                x = saved.pop(0)
                out = saved.pop(0).to(torch.float32) if log else None
                g = g.to(torch.float32)
            if ctx.want_x:
                gx = torch.empty_like(x)
                ctx.plan.backward_x(x.data_ptr(), x.shape[0], lam_host, g.data_ptr(), out.data_ptr() if log else None,
                                    gx.data_ptr(), log, _stream_ptr(g.device), extra_flags=flags)
            if ctx.want_fb:
                fb_shape, fb_dtype = fb_meta
                gfb = torch.empty(fb_shape, dtype=torch.float32, device=g.device)
                ctx.plan.backward_fb(x.data_ptr(), x.shape[0], lam_host, g.data_ptr(), out.data_ptr() if log else None,
                                     gfb.data_ptr(), log, _stream_ptr(g.device), extra_flags=flags)
                gfb = gfb.to(fb_dtype)
        return gx, dl, None, None, None, None, None, gfb, None


class _DmelFbDevFunction(torch.autograd.Function):
    """The layer with lambd left on the device wherever the transform length does not hang on lambd's host value: a trainable
    filterbank (its row count fixes n_fft: dmel_forward_dev_fixed checks lambd against it on the device) and the optimized=False
    branch (n_fft = 2 n_points whatever lambd is), with or without the gradients w.r.t. the filterbank (dmel_backward_fb_dev) and the
    waveform (dmel_backward_x_dev).  No host read anywhere: the step queues without waiting and can be captured into a HIP graph.

    ``n_fft == 0`` (HTK bank, optimized=True, x.requires_grad): the tracked forward dmel_forward_dev -- one launch per candidate n_fft,
    the device value of lambd picks the one that works -- and a waveform gradient issued the same way (DMEL_FLAG_CHECK_NFFT: one
    dmel_backward_x_dev per candidate over a NaN-filled grad_x; a lambd no launch covered leaves the NaN, as that forward's output is)."""

    @staticmethod
    def forward(ctx, x, lambd, plan, n_fft, log, eps, fb, out_dtype, full_window=False, mfma_flags=0, save_spec=False):
        B = x.shape[0]
        want_tangent = ctx.needs_input_grad[1]
        want_fb = fb is not None and ctx.needs_input_grad[6]
        want_x = ctx.needs_input_grad[0]
        # the spectrogram the contraction consumes, kept for the filterbank gradient (fused training kernel only)
        tracked = n_fft == 0
        if tracked and (fb is not None or full_window):
            raise ValueError("n_fft = 0 (tracked forward) is the HTK bank with optimized=True")
        keep_spec = bool(save_spec and want_fb and want_tangent and not full_window and 32 <= n_fft <= 16384 and (n_fft & (n_fft - 1)) == 0)
        round_later = out_dtype == torch.bfloat16 and log and (want_fb or want_x)       # see _DmelFunction.forward
        kdtype = torch.float32 if round_later else out_dtype
        out = torch.empty((B, 1, plan.n_mels, plan.n_time), dtype=kdtype, device=x.device)
        tangent = torch.empty(out.shape, dtype=torch.float32, device=x.device) if want_tangent else None
        scratch = torch.empty((plan.scratch_bytes(B),), dtype=torch.uint8, device=x.device)
        lam = lambd.detach()
        if lam.dtype != torch.float32:
            lam = lam.to(torch.float32)
        flags = (capi.DMEL_FLAG_FULL_WINDOW if full_window else 0) | int(mfma_flags)
        spec = torch.empty((B, n_fft // 2 + 1, plan.n_time), dtype=torch.float32, device=x.device) if keep_spec else None
        ctx.cands = None
        with _on_device(x.device):
            if tracked:
                plan.forward_dev(x.data_ptr(), B, lam.data_ptr(), out.data_ptr(), tangent.data_ptr() if want_tangent else None, log, eps,
                                 _stream_ptr(x.device), scratch.data_ptr(),
                                 extra_flags=capi.DMEL_FLAG_OUT_BF16 if kdtype == torch.bfloat16 else 0)
                # what that call launched for (host-side bookkeeping of the plan: no device read)
                n0, guards = plan.info()["n_fft"], plan.lambd_status()["guards"]
                ctx.cands = [n0] + ([2 * n0] if guards & 2 else []) + ([n0 // 2] if (guards & 1) and n0 >= 2 else [])
            elif keep_spec:
                plan.forward_dev_fixed_spec(x.data_ptr(), B, lam.data_ptr(), n_fft, out.data_ptr(), tangent.data_ptr(), spec.data_ptr(),
                                            log, eps, _stream_ptr(x.device), scratch.data_ptr(),
                                            extra_flags=flags | (capi.DMEL_FLAG_OUT_BF16 if kdtype == torch.bfloat16 else 0))
            else:
                plan.forward_dev_fixed(x.data_ptr(), B, lam.data_ptr(), n_fft, out.data_ptr(), tangent.data_ptr() if want_tangent else None,
                                       log, eps, _stream_ptr(x.device), scratch.data_ptr(),
                                       extra_flags=flags | (capi.DMEL_FLAG_OUT_BF16 if kdtype == torch.bfloat16 else 0))
        ctx.plan, ctx.n_fft, ctx.log, ctx.flags, ctx.keep_spec = plan, n_fft, bool(log), flags, keep_spec
        ctx.lambd_shape, ctx.lambd_dtype = lambd.shape, lambd.dtype
        ctx.want_tangent, ctx.want_fb, ctx.want_x = want_tangent, want_fb, want_x
        ctx.fb_meta = None if fb is None else (tuple(fb.shape), fb.dtype)
        saved = [scratch]
        if want_tangent:
            saved.append(tangent)
        if want_fb or want_x:
            saved += [x, lam]
            if log:
                saved.append(out)
        if keep_spec:
            saved.append(spec)
        ctx.save_for_backward(*saved)
        return out.to(torch.bfloat16) if round_later else out

    @staticmethod
    def backward(ctx, grad_out):
        saved = list(ctx.saved_tensors)
        spec = saved.pop() if ctx.keep_spec else None
        scratch = saved.pop(0)
        bf16 = grad_out.dtype == torch.bfloat16
        g = grad_out
        if not bf16 and g.dtype != torch.float32:
            g = g.to(torch.float32)
        if not g.is_contiguous():
            g = g.contiguous()
        dl = gfb = gx = None
        # d lambd rides in the launch of the filterbank gradient (dmel_backward_fb_saved_dl: one kernel less per step, the same bits)
        ride = ctx.want_tangent and ctx.want_fb and spec is not None and not bf16 and g.numel() > 0
        with _on_device(g.device):
            if ctx.want_tangent:
                tangent = saved.pop(0)
                dl = torch.empty(tuple(ctx.lambd_shape), dtype=torch.float32, device=g.device)
                if not ride:
                    ctx.plan.backward_scratch(g.data_ptr(), tangent.data_ptr(), g.numel(), dl.data_ptr(), _stream_ptr(g.device),
                                              scratch.data_ptr(), grad_bf16=bf16)
            if ctx.want_fb or ctx.want_x:
                x, lam = saved.pop(0), saved.pop(0)
                out = saved.pop(0).to(torch.float32) if ctx.log else None
                g32 = g.to(torch.float32)
            if ctx.want_x and ctx.cands is not None:
                if max(ctx.cands) > 16384:
                    # transforms beyond the fused kernels have no checked backward: the one case that reads lambd (n_fft >= 16384)
                    gx = torch.empty_like(x)
                    ctx.plan.backward_x(x.data_ptr(), x.shape[0], float(lam), g32.data_ptr(), out.data_ptr() if ctx.log else None,
                                        gx.data_ptr(), ctx.log, _stream_ptr(g.device))
                else:
                    gx = torch.full_like(x, float("nan"))
                    for n in ctx.cands:
                        ctx.plan.backward_x_dev(x.data_ptr(), x.shape[0], lam.data_ptr(), n, g32.data_ptr(), out.data_ptr() if ctx.log else None,
                                                gx.data_ptr(), ctx.log, _stream_ptr(g.device), extra_flags=capi.DMEL_FLAG_CHECK_NFFT)
            elif ctx.want_x:
                gx = torch.empty_like(x)
                ctx.plan.backward_x_dev(x.data_ptr(), x.shape[0], lam.data_ptr(), ctx.n_fft, g32.data_ptr(), out.data_ptr() if ctx.log else None,
                                        gx.data_ptr(), ctx.log, _stream_ptr(g.device), extra_flags=ctx.flags)
            if ctx.want_fb:
                fb_shape, fb_dtype = ctx.fb_meta
                gfb = torch.empty(fb_shape, dtype=torch.float32, device=g.device)
                if ride:
                    ctx.plan.backward_fb_saved_dl(spec.data_ptr(), x.shape[0], ctx.n_fft, g32.data_ptr(), out.data_ptr() if ctx.log else None,
                                                  tangent.data_ptr(), gfb.data_ptr(), dl.data_ptr(), scratch.data_ptr(), ctx.log,
                                                  _stream_ptr(g.device), extra_flags=ctx.flags)
                elif spec is not None:
                    ctx.plan.backward_fb_saved(spec.data_ptr(), x.shape[0], ctx.n_fft, g32.data_ptr(), out.data_ptr() if ctx.log else None,
                                               gfb.data_ptr(), ctx.log, _stream_ptr(g.device), extra_flags=ctx.flags)
                else:
                    ctx.plan.backward_fb_dev(x.data_ptr(), x.shape[0], lam.data_ptr(), ctx.n_fft, g32.data_ptr(),
                                             out.data_ptr() if ctx.log else None, gfb.data_ptr(), ctx.log, _stream_ptr(g.device), extra_flags=ctx.flags)
                gfb = gfb.to(fb_dtype)
            if dl is not None and ctx.lambd_dtype != torch.float32:
                dl = dl.to(ctx.lambd_dtype)
        return gx, dl, None, None, None, None, gfb, None, None, None, None


class SlotInput:
    """A batch handed to the layer BY ADDRESS (round 5): ``cell`` is a one-element int64 device tensor holding the address of a contiguous
    fp32 ``(batch, n_points)`` tensor on the same device; the fused forward reads that address when it RUNS (``DMEL_FLAG_X_INDIRECT``).
    A step captured into a HIP graph reads static addresses -- with a ``SlotInput`` the static address is the cell's, and giving the
    replayed step a new batch is an 8-byte write instead of a copy of the batch (``GraphedStep(..., zero_copy=[True, ...]).feed``).
    Only the layer understands it: pass it where the step passes ``x`` to ``net(x)`` / ``layer(x)`` (the reference's nets hand ``x``
    to the layer and to nothing else, models.py:70,93,122,154)."""

    __slots__ = ("cell", "shape", "device")

    def __init__(self, cell, shape):
        if cell.dtype != torch.int64 or cell.numel() != 1 or not cell.is_cuda:
            raise ValueError("SlotInput: cell must be a one-element int64 device tensor")
        self.cell, self.shape, self.device = cell, tuple(int(v) for v in shape), cell.device

    def dim(self):
        return len(self.shape)

    def view(self):
        """the (batch, n_points) fp32 view with zero strides whose data pointer is the cell's (what torch.ops.dmel.* is handed)"""
        return torch.as_strided(self.cell.view(torch.float32), self.shape, (0,) * len(self.shape))


_MEL_OP = None


def _mel_op():
    """torch.ops.dmel.mel_spectrogram.default, resolved once (the packet lookup costs microseconds per call)."""
    global _MEL_OP
    if _MEL_OP is None:
        _MEL_OP = capi.torch_ops().mel_spectrogram.default
    return _MEL_OP


def _to_f32(x: torch.Tensor) -> torch.Tensor:
    """The kernels compute in fp32.  The reference removes the clip mean in the INPUT dtype (models.py:38: ``x[idx] - torch.mean(x[idx])``):
    for fp64 clips (GaussPulse, datasets.py:33) that subtraction happens here, in fp64, BEFORE the cast -- a DC offset far above the
    signal would otherwise cost the signal its low bits in the rounding to fp32 (tests/golden g13_dc_*_fp64: 1e-2 on the lowest mel
    band).  The kernels' own DC removal then finds a mean of rounding size.  Narrower dtypes are widened as they are."""
    if x.dtype == torch.float64:
        x = x - x.mean(dim=1, keepdim=True)
    return x.to(torch.float32)


class MelSpectrogramLayer(nn.Module):
    """Differentiable (log-)Mel spectrogram with a trainable Gaussian window width.

    Signature-compatible with the reference (models.py:15):
        MelSpectrogramLayer(init_lambd, n_mels, n_points, sample_rate, f_min=0, f_max=None,
                            hop_length=1, device='cpu', optimized=False, normalize_window=False)
    Extra keyword-only options: ``log=True`` fuses ``torch.log(s + eps)`` (models.py:73) into the
    kernel epilogue (default False = linear mel power, exactly what the reference layer returns).
    ``learnable_fb=True`` registers the (n_fft/2+1, n_mels) filterbank of models.py:42-48 as a second
    parameter ``mel_fb`` (initialised to the HTK bank) and returns its gradient; the matrix is tied to the
    n_fft it was built for, so the forward raises once ``lambd`` has moved to another power of two.  Off by
    default: the reference has no such parameter and its checkpoints have no such key.
    ``out_dtype=torch.bfloat16`` stores the output as bf16 (the fp32 result rounded to nearest even; BASELINE config 2
    "bf16 activations"); the arithmetic, the saved tangent and ``lambd.grad`` stay fp32.
    ``lambd_sync=False`` (default) keeps ``lambd`` on the device: no host read per forward, the step is HIP-graph
    capturable; a change of ``lambd`` that crosses a power-of-two n_fft boundary is followed by guard launches (see
    include/dmel.h), and one the guards do not cover (e.g. ``lambd`` rewritten by hand to a far value in the middle of a
    run: call ``resync()`` after that) yields NaN outputs and a RuntimeError at the next forward.  ``lambd_sync=True``
    reads ``lambd`` to the host at every forward like the reference (time_frequency.py:39).
    ``mfma="bf16x3"`` (with ``learnable_fb``): the dense contractions -- the forward through the trained matrix, the filterbank
    gradient's GEMM -- run on the bf16 matrix pipe as three split-bf16 products per fp32 product (~2e-5 relative, inside the 1e-4
    bar; default "fp32": exact).  ``save_spec=True`` (with ``learnable_fb``, default): the training forward keeps the power
    spectrogram for the filterbank gradient instead of recomputing it in the backward.

    forward(x: (B, n_points)) -> (B, 1, n_mels, n_points // hop_length + 1) float32.
    """

    def __init__(self, init_lambd, n_mels, n_points, sample_rate, f_min=0, f_max=None, hop_length=1,
                 device="cpu", optimized=False, normalize_window=False, *, log=False, eps=1e-10, learnable_fb=False,
                 out_dtype=torch.float32, lambd_sync=False, mfma="fp32", save_spec=True):
        super().__init__()
        if not torch.is_tensor(init_lambd):
            init_lambd = torch.tensor(float(init_lambd), dtype=torch.float32)
        if mfma not in ("fp32", "bf16x3"):
            raise ValueError("mfma must be 'fp32' (exact fp32 MFMA, the default) or 'bf16x3' (DMEL_FLAG_MFMA_BF16X3: three split-bf16 "
                             "products per fp32 product on the bf16 matrix pipe, for the DENSE contractions of a trainable filterbank)")
        self.mfma = mfma
        self.save_spec = bool(save_spec)      # trainable filterbank: the training forward also writes the (B, F, T) power spectrogram, so
                                              # that the filterbank gradient skips its recompute (16.8 MB per step at BASELINE config 2)
        self.hop_length = hop_length
        self.lambd = nn.Parameter(init_lambd)                        # models.py:19
        self.device = device
        self.optimized = optimized
        self.normalize_window = normalize_window
        self.f_min = f_min
        self.f_max = f_max if f_max is not None else sample_rate // 2   # models.py:25
        self.n_mels = n_mels
        self.sample_rate = sample_rate
        self.n_freq = n_mels                                          # models.py:29
        self.n_time = n_points // hop_length + 1                      # models.py:30
        self.n_points = n_points
        self.log = bool(log)
        self.eps = float(eps)
        if out_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("out_dtype must be torch.float32 (the reference's, models.py:36) or torch.bfloat16")
        self.out_dtype = out_dtype
        self.lambd_sync = bool(lambd_sync)
        self._plans = {}                                              # device index -> capi.Plan (not state)
        self._fb_synced = {}                                          # device index -> (version, data_ptr) last sent to the plan
        if learnable_fb:
            n0 = capi.n_fft(float(init_lambd)) if optimized else 2 * n_points
            fb0 = capi.mel_fbanks_host(n0 // 2 + 1, float(self.f_min), float(self.f_max), n_mels, sample_rate)   # models.py:42-48
            self.mel_fb = nn.Parameter(torch.from_numpy(fb0))
        else:
            self.mel_fb = None

    # -- plumbing -----------------------------------------------------------------------------
    def _plan_for(self, dev: torch.device) -> capi.Plan:
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        plan = self._plans.get(idx)
        if plan is None:
            with torch.cuda.device(idx):
                plan = capi.Plan(self.n_points, self.hop_length, self.n_mels, self.sample_rate, float(self.f_min),
                                 float(self.f_max), bool(self.normalize_window))
            if getattr(self, "_tracking", None) is not None:
                plan.set_tracking(*self._tracking)
            self._plans[idx] = plan
        return plan

    def plan_info(self, device=None) -> dict:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return self._plan_for(dev).info()

    def _lambd_host(self) -> float:
        """Host value of lambd: a device->host read (the reference does one per sample, time_frequency.py:39).  Never
        cached: writes through ``lambd.data`` leave no trace torch could be asked about."""
        return float(self.lambd.detach())

    def n_fft(self) -> int:
        """n_fft the next forward will use: next_pow2(int(6*|lambd|)) (time_frequency.py:39,60-65), or 2*n_points
        in the optimized=False branch (time_frequency.py:51).  Reads lambd to the host."""
        return capi.n_fft(self._lambd_host()) if self.optimized else 2 * self.n_points

    def resync(self):
        """Forget what the sync-free path knows about lambd (call after rewriting it from outside the optimizer, e.g.
        ``layer.lambd.data.fill_(v)``); the next forward reads it once.  ``load_state_dict`` does this by itself."""
        for plan in self._plans.values():
            plan.lambd_reset()

    def lambd_status(self, device=None) -> dict:
        """What the kernels last reported (no synchronisation): see dmel_lambd_status in include/dmel.h."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return self._plan_for(dev).lambd_status()

    def set_tracking(self, max_ahead: int = 8, guard_mode: int = 0):
        """Run-ahead bound and guard policy of the sync-free path (dmel_plan_set_tracking); applies to plans made later too."""
        self._tracking = (int(max_ahead), int(guard_mode))
        for plan in self._plans.values():
            plan.set_tracking(*self._tracking)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.resync()

    # plans are caches of device tables: copies and pickles of the layer start without them
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_plans"] = {}
        state["_fb_synced"] = {}
        return state

    # -- forward ------------------------------------------------------------------------------
    def forward(self, x):
        if x.dim() != 2:
            raise ValueError(f"expected x of shape (batch, n_points), got {tuple(x.shape)}")
        batch_size, n_points = x.shape
        if n_points != self.n_points:
            # the reference fails here too (RuntimeError from the slice-assign at models.py:54)
            raise RuntimeError(f"input has {n_points} points, the layer was built for n_points={self.n_points}")
        if isinstance(x, SlotInput):
            # the batch by address: the hot path only (HTK bank, optimized=True, lambd on the device)
            if self.mel_fb is not None or not self.optimized or self.lambd_sync:
                raise RuntimeError("a SlotInput needs the default layer: HTK bank, optimized=True, lambd_sync=False")
            if self.lambd.device != x.device:
                raise RuntimeError(f"lambd is on {self.lambd.device} but the slot is on {x.device}; call layer.to(device)")
            lam = self.lambd if self.lambd.dtype == torch.float32 else self.lambd.to(torch.float32)
            flags = (capi.DMEL_FLAG_LOG if self.log else 0) | capi.DMEL_FLAG_X_INDIRECT
            return _mel_op()(x.view(), lam, self._plan_for(x.device).handle, flags, self.eps, False, self.out_dtype == torch.bfloat16)
        if not x.is_cuda:
            raise RuntimeError("dmel_amd runs on MI355X only: x must be a CUDA/HIP tensor (no CPU fallback)")
        if self.lambd.device != x.device:
            raise RuntimeError(f"lambd is on {self.lambd.device} but x is on {x.device}; call layer.to(x.device)")
        # dtype / layout conversions only when needed (each no-op torch call still costs ~2 us of host time); when x requires
        # grad its gradient (dmel_backward_x) flows back through these torch ops
        xf = x
        if xf.dtype != torch.float32:
            xf = _to_f32(xf)
        if not xf.is_contiguous():
            xf = xf.contiguous()
        plan = self._plan_for(x.device)
        fb = self.mel_fb
        if fb is None and not x.requires_grad:
            # the hot path: torch-registered op, C++ autograd node, lambd read on the device unless lambd_sync
            flags = (capi.DMEL_FLAG_LOG if self.log else 0) | (0 if self.optimized else capi.DMEL_FLAG_FULL_WINDOW)
            lam = self.lambd
            if lam.dtype != torch.float32:
                lam = lam.to(torch.float32)
            return _mel_op()(xf, lam, plan.handle, flags, self.eps, self.lambd_sync, self.out_dtype == torch.bfloat16)
        if not self.lambd_sync:
            # sync-free: a trainable filterbank fixes n_fft (its row count; lambd is read and checked on the device: one that has left
            # that n_fft gives NaN now and a RuntimeError at the next forward, as models.py:53 fails on the shape), the optimized=False
            # branch runs n_fft = 2 n_points whatever lambd is, and the HTK bank with optimized=True and x.requires_grad (n = 0) runs the
            # tracked forward and a backward that issues the waveform gradient for every n_fft that forward launched for.
            # (What reads lambd to the host: lambd_sync=True, and the waveform gradient at n_fft >= 16384.)
            full = not self.optimized
            n = 2 * self.n_points if full else (2 * (fb.shape[0] - 1) if fb is not None else 0)
            if fb is not None:
                if fb.shape[0] != n // 2 + 1:
                    raise RuntimeError(f"mel_fb was built for n_fft={2 * (fb.shape[0] - 1)} but this layer runs n_fft={n}; "
                                       "a learnable filterbank is tied to one n_fft")
                if fb.device != x.device:
                    raise RuntimeError(f"mel_fb is on {fb.device} but x is on {x.device}; call layer.to(x.device)")
                fbd = fb.detach()
                if fbd.dtype != torch.float32 or not fbd.is_contiguous():
                    fbd = fbd.to(torch.float32).contiguous()
                with _on_device(x.device):
                    plan.set_filterbank_dev(n, fbd.data_ptr(), _stream_ptr(x.device))
            return _DmelFbDevFunction.apply(xf, self.lambd, plan, n, self.log, self.eps, fb, self.out_dtype, full,
                                            capi.DMEL_FLAG_MFMA_BF16X3 if self.mfma == "bf16x3" else 0, self.save_spec)
        lam_host = self._lambd_host()
        if fb is not None:
            n = capi.n_fft(lam_host) if self.optimized else 2 * self.n_points
            if fb.shape[0] != n // 2 + 1:
                raise RuntimeError(f"mel_fb was built for n_fft={2 * (fb.shape[0] - 1)} but lambd={lam_host} now gives n_fft={n}; "
                                   "a learnable filterbank is tied to one n_fft")
            if fb.device != x.device:
                raise RuntimeError(f"mel_fb is on {fb.device} but x is on {x.device}; call layer.to(x.device)")
            # the plan's tables are refreshed from the parameter's storage by one small kernel on the current stream at every
            # forward (no host copy, no synchronisation: the matrix changes at every optimizer step, and writes through
            # .data leave no trace to ask torch about)
            fbd = fb.detach()
            if fbd.dtype != torch.float32 or not fbd.is_contiguous():
                fbd = fbd.to(torch.float32).contiguous()
            with _on_device(x.device):
                plan.set_filterbank_dev(n, fbd.data_ptr(), _stream_ptr(x.device))
        return _DmelFunction.apply(xf, self.lambd, plan, lam_host, self.log, self.eps, not self.optimized, fb, self.out_dtype)

    def extra_repr(self):
        return (f"n_mels={self.n_mels}, n_points={self.n_points}, sample_rate={self.sample_rate}, "
                f"hop_length={self.hop_length}, f_min={self.f_min}, f_max={self.f_max}, "
                f"normalize_window={self.normalize_window}, log={self.log}")


class _DspecFunction(torch.autograd.Function):
    """forward: dmel_spectrogram_ex (carries d spec / d lambd); backward: dmel_backward and, for a waveform that requires grad,
    dmel_backward_x_spec."""

    @staticmethod
    def forward(ctx, x, lambd, plan, lam_host, n_fft, half_window):
        """lam_host None: lambd is read by the kernels from the parameter's storage (no host read: capturable)"""
        B = x.shape[0]
        out = torch.empty((B, 1, n_fft // 2 + 1, plan.n_time), dtype=torch.float32, device=x.device)
        want_tangent = ctx.needs_input_grad[1]
        tangent = torch.empty_like(out) if want_tangent else None
        lam = None
        if lam_host is None:
            lam = lambd.detach()
            if lam.dtype != torch.float32:
                lam = lam.to(torch.float32)
        with _on_device(x.device):
            if lam is None:
                plan.spectrogram_ex(x.data_ptr(), B, lam_host, n_fft, out.data_ptr(), tangent.data_ptr() if want_tangent else None,
                                    _stream_ptr(x.device), remove_dc=True, half_window=half_window)
            else:
                plan.spectrogram_ex_dev(x.data_ptr(), B, lam.data_ptr(), n_fft, out.data_ptr(), tangent.data_ptr() if want_tangent else None,
                                        _stream_ptr(x.device), remove_dc=True, half_window=half_window)
        ctx.plan, ctx.lambd_shape, ctx.lambd_dtype = plan, lambd.shape, lambd.dtype
        ctx.want_tangent, ctx.want_x = want_tangent, ctx.needs_input_grad[0]
        ctx.args = (lam_host, n_fft, half_window)
        saved = ([tangent] if want_tangent else []) + ([x] if ctx.want_x else []) + ([lam] if (ctx.want_x and lam is not None) else [])
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        saved = list(ctx.saved_tensors)
        g = grad_out.to(torch.float32).contiguous()
        dl = gx = None
        with _on_device(g.device):
            if ctx.want_tangent:
                tangent = saved.pop(0)
                dl = torch.empty((1,), dtype=torch.float32, device=g.device)
                ctx.plan.backward(g.data_ptr(), tangent.data_ptr(), g.numel(), dl.data_ptr(), _stream_ptr(g.device))
                dl = dl.reshape(ctx.lambd_shape).to(ctx.lambd_dtype)
            if ctx.want_x:
                x = saved.pop(0)
                lam_host, n_fft, half_window = ctx.args
                gx = torch.empty_like(x)
                if lam_host is None:
                    lam = saved.pop(0)
                    ctx.plan.backward_x_spec_dev(x.data_ptr(), x.shape[0], lam.data_ptr(), n_fft, g.data_ptr(), gx.data_ptr(),
                                                 _stream_ptr(g.device), half_window=half_window)
                else:
                    ctx.plan.backward_x_spec(x.data_ptr(), x.shape[0], lam_host, n_fft, g.data_ptr(), gx.data_ptr(), _stream_ptr(g.device),
                                             half_window=half_window)
        return gx, dl, None, None, None, None


class SpectrogramLayer(nn.Module):
    """DSPEC: differentiable spectrogram with a trainable Gaussian window width.

    Signature-compatible with the reference (models.py:171-200):
        SpectrogramLayer(init_lambd, device='cpu', optimized=False, size=(512, 1024), hop_length=1, normalize_window=False)
    ``optimized=False``: window = whole signal, n_fft = 2*n_points (time_frequency.py:41,51), output
    ``(B, 1, n_points + 1, n_points // hop_length + 1)``; any n_points (a length that is not a power of two takes the
    chirp-z path of csrc/dmel_big.hip).
    ``optimized=True``: n_fft = next_pow2(int(6*|lambd|)) and the output must have the shape ``size``.
    """

    def __init__(self, init_lambd, device="cpu", optimized=False, size=(512, 1024), hop_length=1, normalize_window=False, *, lambd_sync=False):
        super().__init__()
        self.lambd_sync = bool(lambd_sync)          # True: read lambd to the host at every forward, as the reference does
        if not torch.is_tensor(init_lambd):
            init_lambd = torch.tensor(float(init_lambd), dtype=torch.float32)
        self.hop_length = hop_length
        self.lambd = nn.Parameter(init_lambd)                        # models.py:176
        self.device = device
        self.size = size
        self.optimized = optimized
        self.normalize_window = normalize_window
        self._plans = {}

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_plans"] = {}
        return state

    def forward(self, x):
        if x.dim() != 2:
            raise ValueError(f"expected x of shape (batch, n_points), got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("dmel_amd runs on MI355X only: x must be a CUDA/HIP tensor (no CPU fallback)")
        batch_size, n_points = x.shape
        if self.lambd.device != x.device:
            raise RuntimeError(f"lambd is on {self.lambd.device} but x is on {x.device}; call layer.to(x.device)")
        # optimized=False (the reference's only DSPEC configuration, search_spaces.py:71-91): n_fft = 2 n_points whatever lambd is, so
        # lambd stays on the device -- no host read, the step is HIP-graph capturable.  optimized=True derives n_fft from lambd on the
        # host like the reference (time_frequency.py:39): the output SHAPE depends on it.
        lam_host = float(self.lambd.detach()) if (self.optimized or self.lambd_sync) else None
        if self.optimized:
            n_fft, half = capi.n_fft(lam_host), False
            expect = (n_fft // 2 + 1, n_points // self.hop_length + 1)
            if tuple(self.size) != expect:     # the reference's slice-assign at models.py:198 fails the same way
                raise RuntimeError(f"size={tuple(self.size)} but the spectrogram is {expect}")
        else:
            n_fft, half = 2 * n_points, True      # any clip length: powers of two on the FFT kernels, the rest through Bluestein
        key = (x.device.index, n_points)
        plan = self._plans.get(key)
        if plan is None:
            with torch.cuda.device(x.device):
                plan = capi.Plan(n_points, self.hop_length, 1, 2, 0.0, 1.0, bool(self.normalize_window))
            self._plans[key] = plan
        xf = x if x.dtype == torch.float32 else _to_f32(x)
        return _DspecFunction.apply(xf.contiguous(), self.lambd, plan, lam_host, n_fft, half)


# BASELINE.json's north_star calls the layer by this name; the reference has no such symbol.
DifferentiableMelSpectrogram = MelSpectrogramLayer


def dmel_log_mel(x, lambd, n_mels, sample_rate, hop_length, f_min=0.0, f_max=None, normalize_window=False,
                 log=True, eps=1e-10, _plan_cache={}):
    """Functional form: log-mel (or mel) of x with window width ``lambd`` (a tensor that may require grad)."""
    key = (x.device.index, x.shape[1], hop_length, n_mels, sample_rate, float(f_min), f_max, bool(normalize_window))
    plan = _plan_cache.get(key)
    if plan is None:
        with torch.cuda.device(x.device):
            plan = capi.Plan(x.shape[1], hop_length, n_mels, sample_rate, float(f_min),
                             None if f_max is None else float(f_max), bool(normalize_window))
        _plan_cache[key] = plan
    return _DmelFunction.apply(_to_f32(x.detach()).contiguous(), lambd, plan, float(lambd.detach()), log, eps)
