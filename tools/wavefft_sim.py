#!/usr/bin/env python3
"""Lane-level numpy model of the wave-cooperative FFT used by csrc/dmel_fwd.hip.

One 64-lane wave transforms FPW = 64/G frames of N = R*R*C complex points, R points per lane:
  stage 1  radix-R DFT in registers over a   (lane lg holds n = lg + G*a)
  twiddle  w_N^(lg*q)
  exchange through LDS (row q, column lg; padded rows)
  stage 2  radix-R DFT in registers over b   (lane C*q'+r holds element r + C*b of sub-problem q')
  twiddle  w_G^(r*p1)
  stage 3  radix-C butterflies across C adjacent lanes (DPP quad permutes on the GPU)
  result   Z[q' + R*p1 + R*R*p2] at lane C*q' + rho(p2), register p1
This file mirrors the kernel's index arithmetic one to one so that it can be checked against
numpy.fft without a GPU; tests/test_wavefft_model.py runs it for every supported N.
"""
from __future__ import annotations

import numpy as np

# N -> (R, C)
PLAN = {32: (4, 2), 64: (8, 1), 128: (8, 2), 256: (16, 1), 512: (16, 2), 1024: (32, 1), 2048: (32, 2), 4096: (64, 1)}
WAVE = 64


def bitrev(i: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def fft_inreg(v: np.ndarray) -> np.ndarray:
    """Radix-2 DIF on axis -1 (length R), output left in bit-reversed positions, like the
    unrolled register butterflies in the kernel.  v: (..., R) complex."""
    v = v.copy()
    R = v.shape[-1]
    span = R // 2
    while span >= 1:
        for base in range(0, R, 2 * span):
            for j in range(span):
                a, b = v[..., base + j].copy(), v[..., base + j + span].copy()
                tw = np.exp(-2j * np.pi * j / (2 * span))
                v[..., base + j] = a + b
                v[..., base + j + span] = (a - b) * tw
        span //= 2
    return v


def z_index(k: int, R: int, C: int) -> int:
    """Padded position of bin k inside a slot (complex units): +4 per R*R block keeps the final
    ds_write_b64 conflict-free."""
    if C == 1:
        return k
    return k + (k // (R * R)) * 4


def wave_fft(frames: np.ndarray, N: int) -> np.ndarray:
    """frames: (FPW, N) complex -> (FPW, N) complex spectrum, computed the way one wave does."""
    R, C = PLAN[N]
    G = N // R
    FPW = WAVE // G
    assert frames.shape == (FPW, N)
    lb = int(np.log2(R))
    lane = np.arange(WAVE)
    j, lg = lane // G, lane % G

    # stage 0/1: registers z[lane, a] = x[j, lg + G*a]
    z = np.stack([frames[j, lg + G * a] for a in range(R)], axis=1)
    y = fft_inreg(z)                                  # logical q at register bitrev(q)
    # twiddle w_N^(lg*q) and exchange: S[j][q][lg]
    stride = G + (C if G >= 32 else 1)
    lds = np.zeros((FPW, R * stride), complex)
    for q in range(R):
        val = y[:, bitrev(q, lb)] * np.exp(-2j * np.pi * lg * q / N)
        lds[j, q * stride + lg] = val
    # stage 2: lane lg = C*q' + r reads element r + C*b of sub-problem q'
    qp, r = lg // C, lg % C
    u = np.stack([lds[j, qp * stride + r + C * b] for b in range(R)], axis=1)
    U = fft_inreg(u)                                  # logical p1 at register bitrev(p1)
    out = np.zeros((FPW, N), complex)
    V = np.zeros((WAVE, R), complex)
    for p1 in range(R):
        val = U[:, bitrev(p1, lb)] * np.exp(-2j * np.pi * r * p1 / G)
        # stage 3: radix-C across the C adjacent lanes
        if C == 1:
            res, p2 = val, np.zeros(WAVE, int)
        elif C == 2:
            partner = val[lane ^ 1]
            res = np.where(r == 0, val + partner, partner - val)
            p2 = r
        else:  # C == 4, DIF: xor 2 then xor 1; lane r ends with p2 = bitrev2(r)
            partner = val[lane ^ 2]
            t = np.where(r < 2, val + partner, partner - val)
            t = np.where(r == 3, t * (-1j), t)
            partner = t[lane ^ 1]
            res = np.where((r & 1) == 0, t + partner, partner - t)
            p2 = np.array([0, 2, 1, 3])[r]
        V[:, p1] = res
        k = qp + R * p1 + R * R * p2
        out[j, k] = res
    return out


def bank_conflicts_exchange(N: int) -> dict:
    """Worst-case LDS conflict degree of the exchange reads/writes (8-byte elements)."""
    R, C = PLAN[N]
    G = N // R
    FPW = WAVE // G
    stride = G + (C if G >= 32 else 1)
    slot = slot_stride_bytes(N) // 8
    lane = np.arange(WAVE)
    j, lg = lane // G, lane % G
    qp, r = lg // C, lg % C

    def degree(addr_words8, group, banks64):
        worst = 1
        for g0 in range(0, WAVE, group):
            a = addr_words8[g0:g0 + group]
            bank = (a * 2) % (64 if banks64 else 32)
            uniq = {}
            for ad, bk in zip(a, bank):
                uniq.setdefault(bk, set()).add(ad)
            worst = max(worst, max(len(s) for s in uniq.values()))
        return worst

    wr = max(degree(j * slot + q * stride + lg, 16, False) for q in range(R))
    rd = max(degree(j * slot + qp * stride + r + C * b, 32, True) for b in range(R))
    p2 = (np.array([0, 2, 1, 3])[r] if C == 4 else r) if C > 1 else np.zeros(WAVE, int)
    zw = max(degree(j * slot + np.array([z_index(int(kk), R, C) for kk in qp + R * p1 + R * R * p2]), 16, False)
             for p1 in range(R))
    return dict(exchange_write=wr, exchange_read=rd, z_write=zw)


def slot_stride_bytes(N: int) -> int:
    """Bytes between consecutive FFT slots in LDS: room for the padded exchange image and the
    padded spectrum, rounded so that stride = 48 (mod 128): the A operands are read with ds_read_b32, which banks
    modulo 128 B in 32-lane groups that touch 16 B per slot (see a_operand_conflicts)."""
    R, C = PLAN[N]
    G = N // R
    stride = G + (C if G >= 32 else 1)
    need = max(R * stride, z_index(N - 1, R, C) + 1) * 8
    s = (need + 127) // 128 * 128 + 48
    return s


def a_operand_conflicts(N: int, stride_bytes: int | None = None) -> int:
    """Worst conflict degree of the phase-2 A-operand reads (ds_read_b32 / ds_read2_b32: two 32-lane groups, banks modulo
    32 dwords): lane -> (slot8, P|D, k offset) as in dmel_fwd.hip."""
    R, C = PLAN[N]
    st = slot_stride_bytes(N) if stride_bytes is None else stride_bytes
    lane = np.arange(WAVE)
    row16 = lane & 15
    slot8 = 2 * (row16 >> 2) + (row16 & 1)
    typ = (row16 >> 1) & 1
    kofs = lane >> 4
    worst = 1
    for ksg in range(0, min(8, N // 32)):
        zk0 = np.array([z_index(int(k), R, C) for k in 16 * ksg + kofs])
        for u in range(4):
            addr = slot8 * st + typ * 4 + 8 * (zk0 + 4 * u)
            for g0 in (0, 32):
                banks = {}
                for a in addr[g0:g0 + 32]:
                    banks.setdefault((a // 4) % 32, set()).add(a // 4)
                worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def bperm_pairing(N: int, R: int, C: int) -> np.ndarray:
    """Index model of the pairing pass that takes Z[N-k] from the lane holding it (ds_bpermute_b32; kPairBperm in
    csrc/dmel_fwd.hip) for frames of G = R*C <= 64 lanes, 64/G frames per wave.  Lane fl0 + C*qp + rho(p2) holds
    Z[qp + R*p1 + R*R*p2] in register p1.  Returns cover[frame, k] = how many times PD[k] (k <= N/2) is written with the right
    pair (Z[k], Z[N-k]); every entry must be 1."""
    G = N // R
    assert G == R * C and G <= WAVE
    FPW = WAVE // G
    rho = (lambda v: ((v & 1) << 1) | (v >> 1)) if C == 4 else (lambda v: v)      # lane <-> p2 digit reversal (its own inverse)
    lane = np.arange(WAVE)
    j, lg = lane // G, lane % G
    qp, r = lg // C, lg % C
    p2 = np.array([rho(int(v)) for v in r])
    held = qp[:, None] + R * np.arange(R)[None, :] + R * R * p2[:, None]            # bin in (lane, register)
    q0 = qp == 0
    dir_a = 2 * p2 < C
    p2m = C - 1 - p2
    fl0 = j * G
    pull1 = fl0 + ((R - qp) & (R - 1)) * C + np.array([rho(int(v)) for v in p2m])
    pull0 = np.where(q0, fl0 + np.array([rho(int(v)) for v in (C - p2) % C]), pull1)
    nyq = q0 & (2 * p2 == C)
    cover = np.zeros((FPW, N // 2 + 1), int)
    for p1 in range(R // 2 + 1):
        s_a = R // 2 if p1 == R // 2 else R - 1 - p1
        s_b = R // 2 if p1 == R // 2 else (R - p1) % R
        send = np.where(q0, held[lane, s_b], held[lane, s_a])                      # what every lane offers this round
        pull = pull0 if p1 == 0 else pull1
        got = send[pull]                                                           # bin received from the partner
        mine = held[lane, p1]
        for l in range(WAVE):
            ok = (mine[l] + got[l]) % N == 0                                       # a true pair {k, N-k}
            if p1 == 0:
                if nyq[l]:
                    kk = N // 2
                elif dir_a[l] or not q0[l]:
                    kk = mine[l] if dir_a[l] else N - mine[l]
                else:
                    continue
            elif p1 < R // 2:
                kk = mine[l] if dir_a[l] else N - mine[l]
            else:
                if not (q0[l] and dir_a[l]):
                    continue
                kk = mine[l]
            assert 0 <= kk <= N // 2 and ok and (kk == mine[l] or kk == got[l] or kk == N - mine[l]), (l, p1, mine[l], got[l], kk)
            assert min(mine[l], got[l] % N) % N == kk % N or max(mine[l], got[l]) == N - kk or kk in (mine[l], got[l]), (l, p1)
            cover[j[l], kk] += 1
    return cover


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for N in sorted(PLAN):
        R, C = PLAN[N]
        FPW = WAVE // (N // R)
        x = rng.standard_normal((FPW, N)) + 1j * rng.standard_normal((FPW, N))
        got = wave_fft(x, N)
        ref = np.fft.fft(x, axis=1)
        print(N, (R, C), "FPW", FPW, "err", np.abs(got - ref).max(), bank_conflicts_exchange(N),
              "slot", slot_stride_bytes(N))
