#!/usr/bin/env python3
"""The arithmetic behind the split schedule of the wave-local contraction (csrc/dmel_fwd.hip, mode kTrainW; built in csrc/dmel_api.cpp build_tables,
NOTEBOOK R5.15 / R5.16): how many MFMA steps per wave does a schedule need in which a wide quad of mel bands is split over s = 2 or 4 blocks of
the 4x4x1 instruction (partial sums merged across lanes afterwards), against phases of 16 whole quads as long as their widest one?  An
approximation of the host code (continuous HTK band edges, no start-bin matching, no per-phase cost).  No GPU needed.
usage: python tools/wlc_packing_plan.py"""
import math
import numpy as np


def widths(n_fft, sr, n_mels):
    f = np.linspace(0, sr // 2, n_fft // 2 + 1)
    pts = 700.0 * (10.0 ** (np.linspace(0.0, 2595.0 * np.log10(1.0 + (sr // 2) / 700.0), n_mels + 2) / 2595.0) - 1.0)
    out = []
    for q in range(n_mels // 4):
        k = np.nonzero((f > pts[4 * q]) & (f < pts[4 * q + 5]))[0]
        out.append(int(k.max() - k.min() + 1) if len(k) else 0)
    return out


def plan(w, smax, pad):
    """pieces no longer than w*, s in {1, 2, 4} per quad; groups by piece length, first fit into phases of 16 blocks; `pad` = steps a piece loses
    to a start bin with the bank-conflict-free residue (host matching keeps it near 0 for whole quads; up to 7 otherwise)"""
    best = None
    for wstar in range(4, max(w) + 1):
        groups = []
        for x in w:
            s = 1
            while math.ceil(x / s) > wstar and s < smax:
                s *= 2
            groups.append((math.ceil(x / s) + (pad if s > 1 else 0), s))
        groups.sort(reverse=True)
        phases = []
        for length, s in groups:
            for ph in phases:
                if ph[0] + s <= 16:
                    ph[0] += s
                    break
            else:
                phases.append([s, length])
        total = sum(math.ceil(ph[1] / 4) * 4 for ph in phases)
        if best is None or total < best[0]:
            best = (total, len(phases), [ph[1] for ph in phases])
    return best


if __name__ == "__main__":
    for name, n_fft, sr, m in [("config 2", 1024, 16000, 128), ("config 3", 2048, 16000, 128), ("config 5", 2048, 44100, 128),
                               ("ESC-50 shape, lambd 400", 4096, 8000, 64), ("4096, 128 mel bands", 4096, 16000, 128)]:
        w = widths(n_fft, sr, m)
        print(f"{name}: {len(w)} quads, widths sum {sum(w)} max {max(w)}; steps now {plan(w, 1, 0)[0]}; "
              f"s<=2 {plan(w, 2, 2)[:2]}; s<=4 {plan(w, 4, 2)[:2]}; lower bound {math.ceil(sum(w) / 16)}")
