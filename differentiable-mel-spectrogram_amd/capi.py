"""ctypes binding of libdmel_hip.so (include/dmel.h).

This is the only way the Python layer reaches the kernels: raw device pointers, sizes and a
stream handle go through the C ABI.  There is no CPU fallback: if the shared object is missing or
no gfx950 device is visible, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMEL_LIB") or os.path.join(_PKG_DIR, "libdmel_hip.so")   # DMEL_LIB: diagnostic builds (tools/stamps.py)

DMEL_OK = 0
DMEL_ERR_INVALID_ARGUMENT = 1
DMEL_ERR_UNSUPPORTED = 2
DMEL_ERR_HIP = 3
DMEL_ERR_NO_DEVICE = 4
DMEL_ERR_OUT_OF_MEMORY = 5
DMEL_ERR_LAMBD_TRACKING = 6
DMEL_ERR_MAILBOX_TIMEOUT = 7
DMEL_FLAG_MFMA_BF16X3 = 8
DMEL_FLAG_CHECK_NFFT = 16
DMEL_FLAG_X_INDIRECT = 32
DMEL_FLAG_LOG = 1
DMEL_FLAG_FULL_WINDOW = 2
DMEL_FLAG_OUT_BF16 = 4
DMEL_DTYPE_F32, DMEL_DTYPE_BF16 = 0, 1
MAX_NFFT = 16384          # largest transform of the HIP kernels (kMaxNfft in csrc/dmel_kernels.h)

# every symbol include/dmel.h declares (tests check the library exports exactly these)
SYMBOLS = (
    "dmel_abi_version", "dmel_n_fft", "dmel_window_host", "dmel_mel_fbanks_host", "dmel_contraction_partition_host", "dmel_last_error",
    "dmel_device_count", "dmel_plan_create", "dmel_plan_destroy", "dmel_plan_set_filterbank",
    "dmel_forward", "dmel_backward", "dmel_backward_ex", "dmel_backward_fb", "dmel_backward_x", "dmel_spectrogram", "dmel_plan_get_info",
    "dmel_plan_set_profiling", "dmel_plan_get_profile", "dmel_spectrogram_ex",
    "dmel_comm_unique_id", "dmel_comm_create", "dmel_comm_destroy", "dmel_comm_allreduce_async", "dmel_comm_wait",
    "dmel_comm_allreduce", "dmel_scratch_bytes", "dmel_plan_set_filterbank_dev", "dmel_forward_scratch", "dmel_forward_dev", "dmel_forward_dev_fixed", "dmel_backward_fb_dev", "dmel_backward_scratch", "dmel_plan_get_config",
    "dmel_plan_lambd_status", "dmel_plan_set_tracking", "dmel_plan_lambd_reset",
    "dmel_plan_retain", "dmel_plan_release", "dmel_plan_lambd_report", "dmel_decide_launch", "dmel_plan_force_launch",
    "dmel_adam_step", "dmel_mailbox_create", "dmel_mailbox_connect", "dmel_mailbox_destroy", "dmel_mailbox_allreduce", "dmel_mailbox_error",
    "dmel_mailbox_set_spin_limit", "dmel_mailbox_set_timeout_ms", "dmel_plan_is_live", "dmel_lambd_ring_size", "dmel_spectrogram_ex_dev", "dmel_forward_dev_fixed_spec", "dmel_backward_fb_saved", "dmel_backward_fb_saved_dl", "dmel_backward_x_dev", "dmel_backward_x_spec_dev", "dmel_plan_attach_mailbox", "dmel_backward_x_spec", "dmel_plan_attach_adam",
)
TORCH_LIB_PATH = os.path.join(_PKG_DIR, "libdmel_torch.so")


class DmelConfig(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("hop_length", C.c_int32), ("n_mels", C.c_int32),
                ("sample_rate", C.c_int32), ("f_min", C.c_double), ("f_max", C.c_double),
                ("normalize_window", C.c_int32), ("max_batch", C.c_int32)]


class DmelPlanInfo(C.Structure):
    _fields_ = [("n_fft", C.c_int32), ("n_freqs", C.c_int32), ("n_time", C.c_int32),
                ("frames_per_tile", C.c_int32), ("grid_fwd", C.c_int32), ("fb_blocks", C.c_int32),
                ("fb_blocks_dense", C.c_int32), ("lds_bytes", C.c_int32), ("kernel_path", C.c_int32),
                ("contraction", C.c_int32), ("wl_steps", C.c_int32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class DmelLambdStatus(C.Structure):
    _fields_ = [("known", C.c_int32), ("lambd_seen", C.c_float), ("n_fft_seen", C.c_int32), ("seq_issued", C.c_uint32),
                ("seq_seen", C.c_uint32), ("rate", C.c_float), ("guards", C.c_int32), ("error", C.c_int32),
                ("error_seq", C.c_uint32), ("error_lambd", C.c_float), ("next_n_fft", C.c_int32), ("next_guards", C.c_int32),
                ("calls", C.c_uint32)]


class DmelProfile(C.Structure):
    _fields_ = [("prep_ms", C.c_double), ("fwd_ms", C.c_double), ("bwd_ms", C.c_double),
                ("prep_launches", C.c_int32), ("fwd_launches", C.c_int32), ("bwd_launches", C.c_int32)]


class DmelError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libdmel_hip: status {status}: {message}")
        self.status = status


_lib = None


def load():
    """dlopen libdmel_hip.so.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()' or python differentiable-mel-spectrogram_amd/build.py). "
            "There is no CPU fallback for the DMEL layer.")
    L = C.CDLL(LIB_PATH)
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    L.dmel_abi_version.restype = C.c_int32
    L.dmel_n_fft.argtypes = [C.c_float]
    L.dmel_n_fft.restype = C.c_int32
    L.dmel_window_host.argtypes = [C.c_float, C.c_int32, C.c_int32, fp, fp]
    L.dmel_window_host.restype = C.c_int
    L.dmel_mel_fbanks_host.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32, fp]
    L.dmel_mel_fbanks_host.restype = C.c_int
    L.dmel_last_error.restype = C.c_char_p
    L.dmel_device_count.restype = C.c_int32
    L.dmel_plan_create.argtypes = [C.POINTER(DmelConfig), C.POINTER(vp)]
    L.dmel_plan_create.restype = C.c_int
    L.dmel_plan_destroy.argtypes = [vp]
    L.dmel_plan_destroy.restype = C.c_int
    L.dmel_plan_set_filterbank.argtypes = [vp, C.c_int32, fp]
    L.dmel_plan_set_filterbank.restype = C.c_int
    L.dmel_forward.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_uint32, C.c_double, vp, vp, vp]
    L.dmel_forward.restype = C.c_int
    L.dmel_backward.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, vp, vp]
    L.dmel_backward.restype = C.c_int
    L.dmel_backward_ex.argtypes = [vp, vp, C.c_int32, vp, C.c_int64, C.c_int32, vp, vp]
    L.dmel_backward_ex.restype = C.c_int
    L.dmel_backward_fb.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_uint32, vp, vp, vp, vp]
    L.dmel_backward_fb.restype = C.c_int
    L.dmel_backward_x.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_uint32, vp, vp, vp, vp]
    L.dmel_backward_x.restype = C.c_int
    L.dmel_backward_x_spec.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_int32, C.c_uint32, vp, vp, vp]
    L.dmel_backward_x_spec.restype = C.c_int
    L.dmel_spectrogram.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_int32, vp, vp]
    L.dmel_spectrogram.restype = C.c_int
    L.dmel_spectrogram_ex.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_int32, C.c_uint32, vp, vp, vp]
    L.dmel_spectrogram_ex.restype = C.c_int
    L.dmel_comm_unique_id.argtypes = [C.c_char_p]
    L.dmel_comm_unique_id.restype = C.c_int
    L.dmel_comm_create.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.dmel_comm_create.restype = C.c_int
    L.dmel_comm_destroy.argtypes = [vp]
    L.dmel_comm_destroy.restype = C.c_int
    L.dmel_comm_allreduce_async.argtypes = [vp, vp, C.c_int32, vp, C.POINTER(C.c_int32)]
    L.dmel_comm_allreduce_async.restype = C.c_int
    L.dmel_comm_allreduce.argtypes = [vp, vp, C.c_int32, vp]
    L.dmel_comm_allreduce.restype = C.c_int
    L.dmel_comm_wait.argtypes = [vp, C.c_int32, vp]
    L.dmel_comm_wait.restype = C.c_int
    L.dmel_plan_get_info.argtypes = [vp, C.POINTER(DmelPlanInfo)]
    L.dmel_plan_get_info.restype = C.c_int
    L.dmel_plan_set_profiling.argtypes = [vp, C.c_int32]
    L.dmel_plan_set_profiling.restype = C.c_int
    L.dmel_plan_get_profile.argtypes = [vp, C.POINTER(DmelProfile)]
    L.dmel_plan_get_profile.restype = C.c_int
    L.dmel_plan_set_filterbank_dev.argtypes = [vp, C.c_int32, vp, vp]
    L.dmel_plan_set_filterbank_dev.restype = C.c_int
    L.dmel_scratch_bytes.argtypes = [vp, C.c_int32]
    L.dmel_scratch_bytes.restype = C.c_size_t
    L.dmel_forward_scratch.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_uint32, C.c_double, vp, vp, vp, vp]
    L.dmel_forward_scratch.restype = C.c_int
    L.dmel_forward_dev.argtypes = [vp, vp, C.c_int32, vp, C.c_uint32, C.c_double, vp, vp, vp, vp]
    L.dmel_forward_dev.restype = C.c_int
    L.dmel_forward_dev_fixed.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.c_uint32, C.c_double, vp, vp, vp, vp]
    L.dmel_forward_dev_fixed.restype = C.c_int
    L.dmel_backward_fb_dev.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.c_uint32, vp, vp, vp, vp]
    L.dmel_backward_fb_dev.restype = C.c_int
    L.dmel_backward_scratch.argtypes = [vp, vp, C.c_int32, vp, C.c_int64, C.c_int32, vp, vp, vp]
    L.dmel_backward_scratch.restype = C.c_int
    L.dmel_plan_get_config.argtypes = [vp, C.POINTER(DmelConfig)]
    L.dmel_plan_get_config.restype = C.c_int
    L.dmel_plan_lambd_status.argtypes = [vp, C.POINTER(DmelLambdStatus)]
    L.dmel_plan_lambd_status.restype = C.c_int
    L.dmel_plan_set_tracking.argtypes = [vp, C.c_int32, C.c_int32]
    L.dmel_plan_set_tracking.restype = C.c_int
    L.dmel_plan_lambd_reset.argtypes = [vp]
    L.dmel_plan_lambd_reset.restype = C.c_int
    L.dmel_plan_retain.argtypes = [vp]
    L.dmel_plan_retain.restype = C.c_int
    L.dmel_plan_release.argtypes = [vp]
    L.dmel_plan_release.restype = C.c_int
    L.dmel_plan_lambd_report.argtypes = [vp, C.c_uint32, fp, C.POINTER(C.c_int32)]
    L.dmel_plan_lambd_report.restype = C.c_int
    L.dmel_decide_launch.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.dmel_decide_launch.restype = C.c_int
    L.dmel_plan_force_launch.argtypes = [vp, C.c_int32, C.c_int32]
    L.dmel_plan_force_launch.restype = C.c_int
    L.dmel_mailbox_create.argtypes = [C.c_int32, C.c_int32, C.POINTER(vp), C.c_char_p]
    L.dmel_mailbox_create.restype = C.c_int
    L.dmel_mailbox_connect.argtypes = [vp, C.c_char_p]
    L.dmel_mailbox_connect.restype = C.c_int
    L.dmel_mailbox_destroy.argtypes = [vp]
    L.dmel_mailbox_destroy.restype = C.c_int
    L.dmel_adam_step.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, vp]
    L.dmel_adam_step.restype = C.c_int
    L.dmel_plan_attach_adam.argtypes = [vp, vp, vp, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32]
    L.dmel_plan_attach_adam.restype = C.c_int
    L.dmel_mailbox_allreduce.argtypes = [vp, vp, vp]
    L.dmel_mailbox_allreduce.restype = C.c_int
    L.dmel_mailbox_error.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    L.dmel_mailbox_error.restype = C.c_int
    L.dmel_mailbox_set_spin_limit.argtypes = [vp, C.c_uint32]
    L.dmel_mailbox_set_spin_limit.restype = C.c_int
    L.dmel_lambd_ring_size.restype = C.c_int32
    L.dmel_forward_dev_fixed_spec.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.c_uint32, C.c_double, vp, vp, vp, vp, vp]
    L.dmel_forward_dev_fixed_spec.restype = C.c_int
    L.dmel_backward_fb_saved.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_uint32, vp, vp, vp, vp]
    L.dmel_backward_fb_saved.restype = C.c_int
    L.dmel_backward_fb_saved_dl.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
    L.dmel_backward_fb_saved_dl.restype = C.c_int
    L.dmel_spectrogram_ex_dev.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.c_uint32, vp, vp, vp]
    L.dmel_spectrogram_ex_dev.restype = C.c_int
    L.dmel_backward_x_dev.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.c_uint32, vp, vp, vp, vp]
    L.dmel_backward_x_dev.restype = C.c_int
    L.dmel_backward_x_spec_dev.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.c_uint32, vp, vp, vp]
    L.dmel_backward_x_spec_dev.restype = C.c_int
    L.dmel_plan_is_live.argtypes = [vp]
    L.dmel_plan_is_live.restype = C.c_int32
    L.dmel_mailbox_set_timeout_ms.argtypes = [vp, C.c_uint64]
    L.dmel_mailbox_set_timeout_ms.restype = C.c_int
    L.dmel_plan_attach_mailbox.argtypes = [vp, vp]
    L.dmel_plan_attach_mailbox.restype = C.c_int
    _lib = L
    return L


_torch_ops = None


def torch_ops():
    """The torch-registered ops (TORCH_LIBRARY(dmel, ...), csrc/dmel_torch.cpp): ``torch.ops.dmel``.  Loads libdmel_torch.so
    (and through it libdmel_hip.so) on first use; raises if it has not been built."""
    global _torch_ops
    if _torch_ops is None:
        import torch
        load()
        if not os.path.exists(TORCH_LIB_PATH):
            raise RuntimeError(f"{TORCH_LIB_PATH} is missing: build it first (python differentiable-mel-spectrogram_amd/build.py). "
                               "There is no CPU fallback for the DMEL layer.")
        torch.ops.load_library(TORCH_LIB_PATH)
        _torch_ops = torch.ops.dmel
    return _torch_ops


def _check(status: int):
    if status != DMEL_OK:
        raise DmelError(status, (load().dmel_last_error() or b"").decode("utf-8", "replace"))


def release_handle(handle: int) -> None:
    """dmel_plan_release of a handle obtained from ``Plan.retain``"""
    if handle:
        load().dmel_plan_release(C.c_void_p(int(handle)))


def lambd_ring_size() -> int:
    return int(load().dmel_lambd_ring_size())


def n_fft(lambd: float) -> int:
    return int(load().dmel_n_fft(C.c_float(float(lambd))))


def decide_launch(lambd: float, rate: float, stale_forwards: float) -> tuple[int, int]:
    """(n_fft, guards) a forward must be launched for when lambd may move by ``rate`` per forward for ``stale_forwards`` forwards
    unobserved (dmel_decide_launch: a pure function, no device)."""
    n, g = C.c_int32(0), C.c_int32(0)
    _check(load().dmel_decide_launch(C.c_float(float(lambd)), C.c_float(float(rate)), C.c_float(float(stale_forwards)), C.byref(n), C.byref(g)))
    return int(n.value), int(g.value)


def adam_step(param_ptr: int, grad_ptr: int, exp_avg_ptr: int, exp_avg_sq_ptr: int, step_ptr: int, ticket_ptr: int, n: int, lr: float,
              beta1: float, beta2: float, eps: float, weight_decay: float, maximize: bool, stream: int) -> None:
    """dmel_adam_step: torch.optim.Adam's update of an fp32 parameter as one launch on ``stream`` (device pointers; ``ticket_ptr``
    = one zeroed device word, may be 0 up to 1024 elements)."""
    _check(load().dmel_adam_step(param_ptr, grad_ptr, exp_avg_ptr, exp_avg_sq_ptr, step_ptr, ticket_ptr or None, int(n), C.c_double(float(lr)),
                                 C.c_double(float(beta1)), C.c_double(float(beta2)), C.c_double(float(eps)), C.c_double(float(weight_decay)),
                                 1 if maximize else 0, stream))


def device_count() -> int:
    return int(load().dmel_device_count())


def window_host(lambd: float, n: int, normalize: bool = False):
    import numpy as np
    w = np.empty(n, np.float32)
    dw = np.empty(n, np.float32)
    _check(load().dmel_window_host(float(lambd), n, int(normalize), w.ctypes.data_as(C.POINTER(C.c_float)),
                                   dw.ctypes.data_as(C.POINTER(C.c_float))))
    return w, dw


def contraction_partition(units, waves: int = 8):
    """dmel_contraction_partition_host: (own[waves], [(wave, tile, first, units), ...]) for one group of mel tiles"""
    L = load()
    i32 = C.c_int32
    n = len(units)
    u = (i32 * 8)(*([int(v) for v in units] + [0] * (8 - n)))
    own, npc = (i32 * 8)(), i32(0)
    pw, pt, pf, pu = (i32 * 8)(), (i32 * 8)(), (i32 * 8)(), (i32 * 8)()
    L.dmel_contraction_partition_host.restype = C.c_int
    _check(L.dmel_contraction_partition_host(u, n, int(waves), own, C.byref(npc), pw, pt, pf, pu))
    return list(own)[:waves], [(pw[i], pt[i], pf[i], pu[i]) for i in range(npc.value)]


def mel_fbanks_host(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int):
    import numpy as np
    fb = np.empty((n_freqs, n_mels), np.float32)
    _check(load().dmel_mel_fbanks_host(n_freqs, float(f_min), float(f_max), n_mels, sample_rate,
                                       fb.ctypes.data_as(C.POINTER(C.c_float))))
    return fb


class Plan:
    """Owner of a dmel_plan handle (MelSpectrogramLayer.__init__ state, models.py:15-30)."""

    def __init__(self, n_points: int, hop_length: int, n_mels: int, sample_rate: int, f_min: float = 0.0,
                 f_max: float | None = None, normalize_window: bool = False, max_batch: int = 0):
        self._h = C.c_void_p()
        cfg = DmelConfig(int(n_points), int(hop_length), int(n_mels), int(sample_rate), float(f_min),
                         -1.0 if f_max is None else float(f_max), int(bool(normalize_window)), int(max_batch))
        _check(load().dmel_plan_create(C.byref(cfg), C.byref(self._h)))
        self.n_time = int(n_points) // int(hop_length) + 1
        self.n_mels = int(n_mels)
        self.n_points = int(n_points)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().dmel_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def retain(self) -> int:
        """dmel_plan_retain: one more reference; returns the raw handle to give to ``release_handle`` later (it stays valid after
        this wrapper is gone: what a captured HIP graph needs)"""
        _check(load().dmel_plan_retain(self._h))
        return int(self._h.value)

    def is_live(self) -> bool:
        return bool(self._h.value) and bool(load().dmel_plan_is_live(self._h))

    def __deepcopy__(self, memo):
        raise TypeError("a dmel plan is a cache of device tables bound to one GPU: copy the layer, not the plan")

    def __reduce__(self):
        raise TypeError("a dmel plan is a cache of device tables bound to one GPU and cannot be pickled; the layers drop "
                        "their plans when copied or pickled and rebuild them on first use")

    def forward(self, x_ptr: int, batch: int, lambd: float, out_ptr: int, tangent_ptr: int | None,
                log: bool, eps: float, stream: int, extra_flags: int = 0):
        _check(load().dmel_forward(self._h, x_ptr, batch, C.c_float(float(lambd)),
                                   (DMEL_FLAG_LOG if log else 0) | int(extra_flags), float(eps), out_ptr, tangent_ptr, stream))

    @property
    def handle(self) -> int:
        """The dmel_plan* as an integer (what the torch ops take)."""
        return int(self._h.value or 0)

    def scratch_bytes(self, batch: int) -> int:
        return int(load().dmel_scratch_bytes(self._h, int(batch)))

    def forward_dev(self, x_ptr: int, batch: int, lambd_ptr: int, out_ptr: int, tangent_ptr: int | None, log: bool, eps: float,
                    stream: int, scratch_ptr: int | None = None, extra_flags: int = 0):
        """dmel_forward_dev: lambd stays on the device (no host read)."""
        _check(load().dmel_forward_dev(self._h, x_ptr, batch, lambd_ptr, (DMEL_FLAG_LOG if log else 0) | int(extra_flags), float(eps),
                                       out_ptr, tangent_ptr, scratch_ptr, stream))

    def forward_dev_fixed(self, x_ptr: int, batch: int, lambd_ptr: int, n_fft_: int, out_ptr: int, tangent_ptr: int | None, log: bool,
                          eps: float, stream: int, scratch_ptr: int | None = None, extra_flags: int = 0):
        """dmel_forward_dev_fixed: one launch for ``n_fft_``, lambd read and checked on the device (trainable filterbank)."""
        _check(load().dmel_forward_dev_fixed(self._h, x_ptr, batch, lambd_ptr, int(n_fft_), (DMEL_FLAG_LOG if log else 0) | int(extra_flags),
                                             float(eps), out_ptr, tangent_ptr, scratch_ptr, stream))

    def forward_dev_fixed_spec(self, x_ptr: int, batch: int, lambd_ptr: int, n_fft_: int, out_ptr: int, tangent_ptr: int, spec_ptr: int,
                               log: bool, eps: float, stream: int, scratch_ptr: int | None = None, extra_flags: int = 0):
        """dmel_forward_dev_fixed_spec: the training forward that also writes the (B, F, T) power spectrogram for backward_fb_saved."""
        _check(load().dmel_forward_dev_fixed_spec(self._h, x_ptr, batch, lambd_ptr, int(n_fft_), (DMEL_FLAG_LOG if log else 0) | int(extra_flags),
                                                  float(eps), out_ptr, tangent_ptr, spec_ptr, scratch_ptr, stream))

    def backward_fb_saved(self, spec_ptr: int, batch: int, n_fft_: int, grad_ptr: int, out_ptr: int | None, grad_fb_ptr: int, log: bool,
                          stream: int, extra_flags: int = 0):
        _check(load().dmel_backward_fb_saved(self._h, spec_ptr, batch, int(n_fft_), (DMEL_FLAG_LOG if log else 0) | int(extra_flags), grad_ptr,
                                             out_ptr if log else None, grad_fb_ptr, stream))

    def backward_fb_saved_dl(self, spec_ptr: int, batch: int, n_fft_: int, grad_ptr: int, out_ptr: int | None, tangent_ptr: int, grad_fb_ptr: int,
                             dlambd_ptr: int, scratch_ptr: int | None, log: bool, stream: int, extra_flags: int = 0):
        """dmel_backward_fb_saved_dl: the filterbank gradient on the saved spectrogram and d lambd in one launch (fp32 grad_out)."""
        _check(load().dmel_backward_fb_saved_dl(self._h, spec_ptr, batch, int(n_fft_), (DMEL_FLAG_LOG if log else 0) | int(extra_flags), grad_ptr,
                                                out_ptr if log else None, tangent_ptr, grad_fb_ptr, dlambd_ptr, scratch_ptr, stream))

    def backward_fb_dev(self, x_ptr: int, batch: int, lambd_ptr: int, n_fft_: int, grad_ptr: int, out_ptr: int | None, grad_fb_ptr: int,
                        log: bool, stream: int, extra_flags: int = 0):
        """dmel_backward_fb_dev: dmel_backward_fb with lambd read on the device."""
        _check(load().dmel_backward_fb_dev(self._h, x_ptr, batch, lambd_ptr, int(n_fft_), (DMEL_FLAG_LOG if log else 0) | int(extra_flags), grad_ptr,
                                           out_ptr if log else None, grad_fb_ptr, stream))

    def backward_x_dev(self, x_ptr: int, batch: int, lambd_ptr: int, n_fft_: int, grad_ptr: int, out_ptr: int | None, grad_x_ptr: int,
                       log: bool, stream: int, extra_flags: int = 0):
        """dmel_backward_x_dev: dmel_backward_x with lambd read on the device (n_fft_: what this step's forward ran)."""
        _check(load().dmel_backward_x_dev(self._h, x_ptr, batch, lambd_ptr, int(n_fft_), (DMEL_FLAG_LOG if log else 0) | int(extra_flags), grad_ptr,
                                          out_ptr if log else None, grad_x_ptr, stream))

    def backward_x_spec_dev(self, x_ptr: int, batch: int, lambd_ptr: int, n_fft_: int, grad_spec_ptr: int, grad_x_ptr: int, stream: int,
                            half_window: bool = False):
        _check(load().dmel_backward_x_spec_dev(self._h, x_ptr, batch, lambd_ptr, int(n_fft_), 1 | (2 if half_window else 0),
                                               grad_spec_ptr, grad_x_ptr, stream))

    def spectrogram_ex_dev(self, x_ptr: int, batch: int, lambd_ptr: int, n_fft_: int, spec_ptr: int, tangent_ptr, stream: int,
                           remove_dc: bool = True, half_window: bool = False):
        flags = (1 if remove_dc else 0) | (2 if half_window else 0)
        _check(load().dmel_spectrogram_ex_dev(self._h, x_ptr, batch, lambd_ptr, int(n_fft_), flags, spec_ptr, tangent_ptr, stream))

    def backward_scratch(self, grad_ptr: int, tangent_ptr: int, count: int, dlambd_ptr: int, stream: int, scratch_ptr: int | None,
                         accumulate: bool = False, grad_bf16: bool = False):
        _check(load().dmel_backward_scratch(self._h, grad_ptr, DMEL_DTYPE_BF16 if grad_bf16 else DMEL_DTYPE_F32, tangent_ptr,
                                            int(count), int(accumulate), dlambd_ptr, scratch_ptr, stream))

    def lambd_status(self) -> dict:
        st = DmelLambdStatus()
        _check(load().dmel_plan_lambd_status(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def set_tracking(self, max_ahead: int = 8, guard_mode: int = 0):
        _check(load().dmel_plan_set_tracking(self._h, int(max_ahead), int(guard_mode)))

    def lambd_reset(self):
        _check(load().dmel_plan_lambd_reset(self._h))

    def lambd_report(self, number: int):
        """lambd as read by execution ``number`` (None if its report has left the ring): timing-independent, see include/dmel.h."""
        lam, found = C.c_float(0.0), C.c_int32(0)
        _check(load().dmel_plan_lambd_report(self._h, C.c_uint32(int(number) & 0xFFFFFFFF), C.cast(C.byref(lam), C.POINTER(C.c_float)), C.byref(found)))
        return float(lam.value) if found.value else None

    def attach_mailbox(self, mailbox):
        """dmel_plan_attach_mailbox: backward() on this plan returns the sum over the mailbox's ranks (None detaches)."""
        _check(load().dmel_plan_attach_mailbox(self._h, mailbox._h if mailbox is not None else None))
        self._mailbox = mailbox                       # keep it alive as long as the plan points at it

    def attach_adam(self, param_ptr, exp_avg_ptr=0, exp_avg_sq_ptr=0, step_ptr=0, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                    weight_decay=0.0, maximize=False, keep=None):
        """dmel_plan_attach_adam: every backward() on this plan ends with Adam's update of the fp32 device scalar at ``param_ptr`` by the
        gradient it has just written (``param_ptr`` = 0 / None detaches).  ``keep``: the objects that own the four addresses -- referenced
        from the plan for as long as it carries them (the caching allocator must not hand that memory to anyone else meanwhile)."""
        self._adam_keep = keep if param_ptr else None
        _check(load().dmel_plan_attach_adam(self._h, param_ptr or None, exp_avg_ptr or None, exp_avg_sq_ptr or None, step_ptr or None,
                                            C.c_double(float(lr)), C.c_double(float(beta1)), C.c_double(float(beta2)), C.c_double(float(eps)),
                                            C.c_double(float(weight_decay)), 1 if maximize else 0))

    def force_launch(self, n_fft_: int = 0, guards: int = 0):
        """dmel_plan_force_launch: the caller chooses the launches of dmel_forward_dev (n_fft_ = 0: automatic again)."""
        _check(load().dmel_plan_force_launch(self._h, int(n_fft_), int(guards)))

    def backward(self, grad_ptr: int, tangent_ptr: int, count: int, dlambd_ptr: int, stream: int, accumulate: bool = False,
                 grad_bf16: bool = False):
        if grad_bf16:
            _check(load().dmel_backward_ex(self._h, grad_ptr, DMEL_DTYPE_BF16, tangent_ptr, int(count), int(accumulate), dlambd_ptr, stream))
        else:
            _check(load().dmel_backward(self._h, grad_ptr, tangent_ptr, int(count), int(accumulate), dlambd_ptr, stream))

    def backward_fb(self, x_ptr: int, batch: int, lambd: float, grad_ptr: int, out_ptr: int | None, grad_fb_ptr: int,
                    log: bool, stream: int, extra_flags: int = 0):
        """grad of the loss w.r.t. the (n_fft/2+1, n_mels) filterbank (adjoint of models.py:53)."""
        _check(load().dmel_backward_fb(self._h, x_ptr, batch, C.c_float(float(lambd)),
                                       (DMEL_FLAG_LOG if log else 0) | int(extra_flags), grad_ptr, out_ptr if log else None,
                                       grad_fb_ptr, stream))

    def backward_x(self, x_ptr: int, batch: int, lambd: float, grad_ptr: int, out_ptr: int | None, grad_x_ptr: int,
                   log: bool, stream: int, extra_flags: int = 0):
        """grad of the loss w.r.t. the waveform (adjoint of models.py:38-53)."""
        _check(load().dmel_backward_x(self._h, x_ptr, batch, C.c_float(float(lambd)), (DMEL_FLAG_LOG if log else 0) | int(extra_flags), grad_ptr,
                                      out_ptr if log else None, grad_x_ptr, stream))

    def backward_x_spec(self, x_ptr: int, batch: int, lambd: float, n_fft_: int, grad_spec_ptr: int, grad_x_ptr: int, stream: int,
                        half_window: bool = False):
        """grad of the loss w.r.t. the waveform through the spectrogram layer (models.py:171-200)."""
        _check(load().dmel_backward_x_spec(self._h, x_ptr, batch, C.c_float(float(lambd)), int(n_fft_), 1 | (2 if half_window else 0),
                                           grad_spec_ptr, grad_x_ptr, stream))

    def spectrogram(self, x_ptr: int, batch: int, lambd: float, spec_ptr: int, stream: int, remove_dc: bool = False):
        _check(load().dmel_spectrogram(self._h, x_ptr, batch, C.c_float(float(lambd)), int(remove_dc), spec_ptr, stream))

    def spectrogram_ex(self, x_ptr: int, batch: int, lambd: float, n_fft_: int, spec_ptr: int, tangent_ptr, stream: int,
                       remove_dc: bool = True, half_window: bool = False):
        flags = (1 if remove_dc else 0) | (2 if half_window else 0)
        _check(load().dmel_spectrogram_ex(self._h, x_ptr, batch, C.c_float(float(lambd)), int(n_fft_), flags, spec_ptr,
                                          tangent_ptr, stream))

    def set_filterbank(self, n_fft_: int, fb):
        import numpy as np
        if fb is None:
            _check(load().dmel_plan_set_filterbank(self._h, int(n_fft_), None))
            return
        fb = np.ascontiguousarray(fb, dtype=np.float32)
        if fb.shape != (n_fft_ // 2 + 1, self.n_mels):
            raise ValueError(f"filterbank must be ({n_fft_ // 2 + 1}, {self.n_mels}), got {fb.shape}")
        _check(load().dmel_plan_set_filterbank(self._h, int(n_fft_), fb.ctypes.data_as(C.POINTER(C.c_float))))

    def set_filterbank_dev(self, n_fft_: int, fb_ptr: int, stream: int):
        """(n_fft/2+1, n_mels) fp32 row-major DEVICE matrix -> the plan's tables, on ``stream`` (no host copy, no sync)."""
        _check(load().dmel_plan_set_filterbank_dev(self._h, int(n_fft_), fb_ptr, stream))

    def set_profiling(self, enable: bool):
        _check(load().dmel_plan_set_profiling(self._h, int(bool(enable))))

    def get_profile(self) -> dict:
        pr = DmelProfile()
        _check(load().dmel_plan_get_profile(self._h, C.byref(pr)))
        return {k: getattr(pr, k) for k, _ in pr._fields_}

    def info(self) -> dict:
        inf = DmelPlanInfo()
        _check(load().dmel_plan_get_info(self._h, C.byref(inf)))
        return inf.as_dict()


class Comm:
    """Owner of a dmel_comm handle: native RCCL all-reduce of the scalar gradient (include/dmel.h)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.ID_BYTES)
        _check(load().dmel_comm_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == Comm.ID_BYTES
        self._h = C.c_void_p()
        _check(load().dmel_comm_create(C.create_string_buffer(unique_id, Comm.ID_BYTES), int(rank), int(world), C.byref(self._h)))

    def allreduce_async(self, buf_ptr: int, count: int, stream: int) -> int:
        t = C.c_int32(-1)
        _check(load().dmel_comm_allreduce_async(self._h, buf_ptr, int(count), stream, C.byref(t)))
        return int(t.value)

    def allreduce(self, buf_ptr: int, count: int, stream: int) -> None:
        """In-stream SUM all-reduce (ordered like a kernel launch on ``stream``)."""
        _check(load().dmel_comm_allreduce(self._h, buf_ptr, int(count), stream))

    def wait(self, ticket: int, stream: int) -> None:
        _check(load().dmel_comm_wait(self._h, int(ticket), stream))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().dmel_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mailbox:
    """Owner of a dmel_mailbox handle: the peer-to-peer all-reduce of lambd.grad folded into the backward's kernel (include/dmel.h)."""

    HANDLE_BYTES = 64

    def __init__(self, rank: int, world: int):
        self._h = C.c_void_p()
        buf = C.create_string_buffer(Mailbox.HANDLE_BYTES)
        _check(load().dmel_mailbox_create(int(rank), int(world), C.byref(self._h), buf))
        self.rank, self.world, self.handle = int(rank), int(world), buf.raw

    def connect(self, handles):
        """``handles``: the 64-byte handles of all ranks in rank order (this rank's own entry is not opened)."""
        blob = b"".join(handles)
        assert len(blob) == self.world * Mailbox.HANDLE_BYTES
        _check(load().dmel_mailbox_connect(self._h, C.create_string_buffer(blob, len(blob))))

    def allreduce(self, buf_ptr: int, stream: int) -> None:
        """buf[0] = sum over ranks of buf[0], one tiny launch on ``stream``."""
        _check(load().dmel_mailbox_allreduce(self._h, buf_ptr, stream))

    def error(self):
        """None, or (step, missing_rank) of the first exchange that timed out since the last call."""
        failed, step, miss = C.c_int32(0), C.c_uint32(0), C.c_int32(0)
        _check(load().dmel_mailbox_error(self._h, C.byref(failed), C.byref(step), C.byref(miss)))
        return (int(step.value), int(miss.value)) if failed.value else None

    def set_spin_limit(self, polls: int) -> None:
        """polls per source rank before an exchange gives up; 0 = no limit on the count (the wall-clock bound holds)"""
        _check(load().dmel_mailbox_set_spin_limit(self._h, int(polls)))

    def set_timeout_ms(self, ms: int) -> None:
        """wall-clock bound of one exchange (default 120 000 ms; 0 = wait for ever)"""
        _check(load().dmel_mailbox_set_timeout_ms(self._h, C.c_uint64(int(ms))))

    def close(self):
        """raises while a plan still holds the mailbox (detach or release the plans first: MailboxAllReduce.close does)"""
        if getattr(self, "_h", None) is not None and self._h.value:
            _check(load().dmel_mailbox_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
