"""Deterministic synthetic waveforms / cotangents for tests, fixtures and bench.

The reference fixes no seeds (README.md:63 of the reference) and ships no data, so
every input used for parity is generated here from a counter-based generator
(splitmix64 -> Box-Muller in float64 -> float32).  It depends on nothing but numpy
integer arithmetic and libm, so the container that captured ``tests/golden`` and
the GPU box regenerate bit-identical inputs without sharing an RNG implementation.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(counter: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (counter + np.uint64(1)) * _GOLDEN + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(n: int, seed: int, offset: int = 0) -> np.ndarray:
    """n float64 samples in (0, 1)."""
    c = np.arange(offset, offset + n, dtype=np.uint64)
    z = _splitmix64(c, seed)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(shape, seed: int, scale: float = 1.0, dtype=np.float32) -> np.ndarray:
    """Standard normal * scale, shape ``shape``; element i uses counters 2i, 2i+1."""
    n = int(np.prod(shape))
    u = uniform01(2 * n, seed)
    r = np.sqrt(-2.0 * np.log(u[0::2]))
    g = r * np.cos(2.0 * np.pi * u[1::2])
    return (scale * g).reshape(shape).astype(dtype)


def waveforms(batch: int, n_points: int, seed: int = 0, scale: float = 0.1) -> np.ndarray:
    """Synthetic clips x ~ scale*N(0,1), fp32, (batch, n_points)."""
    return normal((batch, n_points), seed=seed, scale=scale)


def cotangent(shape, seed: int = 1) -> np.ndarray:
    """Upstream gradient G_Y ~ N(0,1), fp32."""
    return normal(shape, seed=seed, scale=1.0)


def tone_mix(batch: int, n_points: int, sample_rate: int, seed: int = 7) -> np.ndarray:
    """A few sinusoids + a DC offset + weak noise: exercises DC removal and band edges."""
    t = np.arange(n_points, dtype=np.float64) / float(sample_rate)
    u = uniform01(batch * 8, seed).reshape(batch, 8)
    x = np.zeros((batch, n_points), dtype=np.float64)
    for b in range(batch):
        for j in range(3):
            f = 50.0 + u[b, j] * (0.45 * sample_rate - 50.0)
            a = 0.05 + 0.3 * u[b, 3 + j]
            x[b] += a * np.sin(2.0 * np.pi * f * t + 6.283 * u[b, 6])
        x[b] += 0.25 * (u[b, 7] - 0.5)
    x += normal((batch, n_points), seed=seed + 1, scale=0.01, dtype=np.float64)
    return x.astype(np.float32)
