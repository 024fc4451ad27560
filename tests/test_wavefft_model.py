"""The lane-level model of the wave-cooperative FFT (tools/wavefft_sim.py) that the HIP kernel transcribes:
index maps, twiddles, padded LDS addressing -- checked against numpy.fft for every supported n_fft."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import wavefft_sim as W


@pytest.mark.parametrize("N", sorted(W.PLAN))
def test_wave_fft_model(N):
    R, C = W.PLAN[N]
    fpw = 64 // (N // R)
    rng = np.random.default_rng(N)
    x = rng.standard_normal((fpw, N)) + 1j * rng.standard_normal((fpw, N))
    assert np.abs(W.wave_fft(x, N) - np.fft.fft(x, axis=1)).max() < 1e-11 * N
    conf = W.bank_conflicts_exchange(N)
    assert max(conf.values()) <= 2
    assert W.slot_stride_bytes(N) % 128 == 48
    assert W.a_operand_conflicts(N) == 1 and W.a_operand_conflicts(N, W.slot_stride_bytes(N) // 256 * 256 + 32) == 2


@pytest.mark.parametrize("N,R,C", [(1024, 32, 1), (2048, 32, 2), (4096, 64, 1), (1024, 16, 4), (256, 16, 1), (512, 16, 2)])
def test_bpermute_pairing_model(N, R, C):
    """every PD[k], k <= N/2, of every frame of the wave is produced exactly once, from a true pair (Z[k], Z[N-k])"""
    cover = W.bperm_pairing(N, R, C)
    assert cover.shape[1] == N // 2 + 1 and (cover == 1).all(), np.argwhere(cover != 1)[:8]
