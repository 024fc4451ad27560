"""ctypes binding of oracle/dmel_oracle.c.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (dmel_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdmel_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "dmel_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libdmel_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        L.dmel_oracle_n_fft.argtypes = [C.c_float]
        L.dmel_oracle_n_fft.restype = C.c_int
        L.dmel_oracle_threads.restype = C.c_int
        L.dmel_oracle_set_threads.argtypes = [C.c_int]
        L.dmel_oracle_window.argtypes = [C.c_float, C.c_int, C.c_int, fp, dp]
        L.dmel_oracle_window.restype = C.c_int
        L.dmel_oracle_mel_fbanks.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, fp]
        L.dmel_oracle_mel_fbanks.restype = C.c_int
        L.dmel_oracle_forward.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, fp, fp]
        L.dmel_oracle_forward.restype = C.c_int
        L.dmel_oracle_forward_ex.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                             C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, fp, fp]
        L.dmel_oracle_forward_ex.restype = C.c_int
        L.dmel_oracle_forward_mean.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                               C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, fp, fp, fp]
        L.dmel_oracle_forward_mean.restype = C.c_int
        L.dmel_oracle_backward.argtypes = [fp, fp, C.c_longlong]
        L.dmel_oracle_backward.restype = C.c_double
        L.dmel_oracle_spectrogram.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, fp]
        L.dmel_oracle_spectrogram.restype = C.c_int
        L.dmel_oracle_fbgrad.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, dp, C.c_int, dp]
        L.dmel_oracle_fbgrad.restype = C.c_int
        L.dmel_oracle_xgrad.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, dp, C.c_int, fp, dp]
        L.dmel_oracle_xgrad.restype = C.c_int
        L.dmel_oracle_dspec.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, fp, fp]
        L.dmel_oracle_dspec.restype = C.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def n_fft(lambd: float) -> int:
    return int(lib().dmel_oracle_n_fft(np.float32(lambd)))


def threads() -> int:
    return int(lib().dmel_oracle_threads())


def set_threads(n: int) -> None:
    lib().dmel_oracle_set_threads(int(n))


def window(lambd: float, n: int, normalize: bool = False):
    w = np.empty(n, np.float32)
    dw = np.empty(n, np.float64)
    rc = lib().dmel_oracle_window(np.float32(lambd), n, int(normalize), _fp(w), dw.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return w, dw


def mel_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> np.ndarray:
    fb = np.empty((n_freqs, n_mels), np.float32)
    rc = lib().dmel_oracle_mel_fbanks(n_freqs, float(f_min), float(f_max), n_mels, sample_rate, _fp(fb))
    assert rc == 0
    return fb


def forward(x: np.ndarray, lambd: float, hop: int, n_mels: int, sample_rate: int, f_min: float = 0.0,
            f_max: float | None = None, normalize_window: bool = False, apply_log: bool = False,
            eps: float = 1e-10, want_tangent: bool = True, optimized: bool = True, mean: np.ndarray | None = None):
    """Returns (out, tangent) with shape (B,1,n_mels,L//hop+1); tangent = d out / d lambd.  ``mean``: the clip means that
    models.py:38 subtracts, given (B fp32 values) instead of computed (the DC-dominated fixtures: see dmel_oracle_forward_mean)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, L = x.shape
    T = L // hop + 1
    out = np.empty((B, 1, n_mels, T), np.float32)
    tan = np.empty_like(out) if want_tangent else None
    mean_arr = None if mean is None else np.ascontiguousarray(mean, dtype=np.float32).reshape(B)
    rc = lib().dmel_oracle_forward_mean(_fp(x), B, L, np.float32(lambd), hop, n_mels, sample_rate, float(f_min),
                                        -1.0 if f_max is None else float(f_max), int(normalize_window),
                                        int(apply_log), float(eps), int(optimized), _fp(out), _fp(tan) if want_tangent else None,
                                        None if mean_arr is None else _fp(mean_arr))
    if rc != 0:
        raise RuntimeError(f"dmel_oracle_forward failed rc={rc}")
    return out, tan


def backward(grad_out: np.ndarray, tangent: np.ndarray) -> float:
    g = np.ascontiguousarray(grad_out, dtype=np.float32)
    t = np.ascontiguousarray(tangent, dtype=np.float32)
    assert g.shape == t.shape
    return float(lib().dmel_oracle_backward(_fp(g), _fp(t), g.size))


def spectrogram(x: np.ndarray, lambd: float, hop: int, normalize_window: bool = False, remove_dc: bool = False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, L = x.shape
    N = n_fft(lambd)
    spec = np.empty((B, N // 2 + 1, L // hop + 1), np.float32)
    rc = lib().dmel_oracle_spectrogram(_fp(x), B, L, np.float32(lambd), hop, int(normalize_window), int(remove_dc), _fp(spec))
    assert rc == 0
    return spec


def backward_fb(x: np.ndarray, lambd: float, hop: int, grad_out: np.ndarray, out: np.ndarray | None = None,
                normalize_window: bool = False) -> np.ndarray:
    """d loss / d mel_fb, (n_fft/2+1, n_mels) fp64: adjoint of models.py:53.  ``out`` = the LOG output (models.py:73)
    when the loss was taken on log(mel + eps); None for the linear layer."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, L = x.shape
    g = np.asarray(grad_out, dtype=np.float64).reshape(B, -1, L // hop + 1)
    if out is not None:
        g = g * np.exp(-np.asarray(out, dtype=np.float64).reshape(g.shape))      # d log(s + eps) = ds / (s + eps)
    g = np.ascontiguousarray(g)
    M = g.shape[1]
    N = n_fft(lambd)
    gfb = np.empty((N // 2 + 1, M), np.float64)
    dp = C.POINTER(C.c_double)
    rc = lib().dmel_oracle_fbgrad(_fp(x), B, L, np.float32(lambd), hop, int(normalize_window), g.ctypes.data_as(dp), M,
                                  gfb.ctypes.data_as(dp))
    if rc != 0:
        raise RuntimeError(f"dmel_oracle_fbgrad failed rc={rc}")
    return gfb


def backward_x(x: np.ndarray, lambd: float, hop: int, sample_rate: int, grad_out: np.ndarray, out: np.ndarray | None = None,
               f_min: float = 0.0, f_max: float | None = None, normalize_window: bool = False, fb: np.ndarray | None = None) -> np.ndarray:
    """d loss / d x, (B, L) fp64: adjoint of models.py:38-53.  ``out`` = the LOG output when the loss was taken on
    log(mel + eps) (models.py:73), None for the linear layer; ``fb`` overrides the HTK bank."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, L = x.shape
    g = np.asarray(grad_out, dtype=np.float64).reshape(B, -1, L // hop + 1)
    if out is not None:
        g = g * np.exp(-np.asarray(out, dtype=np.float64).reshape(g.shape))
    g = np.ascontiguousarray(g)
    M = g.shape[1]
    N = n_fft(lambd)
    if fb is None:
        fb = mel_fbanks(N // 2 + 1, f_min, float(sample_rate // 2) if f_max is None else f_max, M, sample_rate)
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    gx = np.empty((B, L), np.float64)
    dp = C.POINTER(C.c_double)
    rc = lib().dmel_oracle_xgrad(_fp(x), B, L, np.float32(lambd), hop, int(normalize_window), g.ctypes.data_as(dp), M, _fp(fb),
                                 gx.ctypes.data_as(dp))
    if rc != 0:
        raise RuntimeError(f"dmel_oracle_xgrad failed rc={rc}")
    return gx


def dspec(x: np.ndarray, lambd: float, hop: int = 1, normalize_window: bool = False):
    """Non-optimized SpectrogramLayer (models.py:171-200): returns (spec, tangent), each (B, 1, L+1, L//hop+1)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, L = x.shape
    spec = np.empty((B, 1, L + 1, L // hop + 1), np.float32)
    tan = np.empty_like(spec)
    rc = lib().dmel_oracle_dspec(_fp(x), B, L, np.float32(lambd), hop, int(normalize_window), _fp(spec), _fp(tan))
    assert rc == 0
    return spec, tan
