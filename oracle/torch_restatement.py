"""Batched torch-CPU restatement of the reference's DMEL path.  TEST INFRASTRUCTURE ONLY.

Same library calls as the reference (torch.stft with the Gaussian window, |.|^2, matmul with the HTK
filterbank, log, torch autograd for d/dlambd) but for the whole batch at once instead of the
per-sample Python loop of models.py:37-54.  It exists for two reasons:
  * an independent check of oracle/dmel_oracle.c (different FFT, reverse-mode autograd instead of
    the closed-form tangent);
  * bench.py's cpu_baseline: it is the fastest CPU form of the reference's algorithm measured in
    BASELINE.md section 2 (the reference's own loop spends 70 % of its time in a CopySlices artefact).
Only tests/ and bench.py's cpu_baseline leg may import this module.

Reference lines followed: time_frequency.py:21-30 (window), :39,:60-65 (n_fft), :48 (stft),
:53 (power); models.py:38 (DC removal, abs), :42-53 (filterbank, contraction), :73 (log).
"""
from __future__ import annotations

import math

import torch


def n_fft_of(lambd: torch.Tensor) -> int:
    x = int((torch.abs(lambd).detach().float() * 6).cpu().numpy())   # time_frequency.py:39,61
    return 1 << (x - 1).bit_length()


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio 0.13.1 functional.melscale_fbanks (htk, norm=None); PARITY-UNPINNED (see dmel_oracle.c)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def forward(x: torch.Tensor, lambd: torch.Tensor, hop: int, n_mels: int, sample_rate: int, f_min=0.0, f_max=None,
            normalize_window=False, log=False, eps=1e-10, fb=None) -> torch.Tensor:
    """x (B, L) fp32 CPU, lambd 0-dim (may require grad) -> (B, 1, n_mels, L//hop+1) fp32."""
    f_max = sample_rate // 2 if f_max is None else f_max
    n = n_fft_of(lambd)
    a = torch.abs(lambd)
    m = torch.arange(0, n).float()
    window = torch.exp(-0.5 * torch.pow((m - n / 2) / (a + 1e-15), 2))
    if normalize_window:
        window = window / torch.sqrt(torch.sum(torch.pow(window, 2)))
    xc = x - x.mean(dim=1, keepdim=True)
    s = torch.stft(xc, n_fft=n, hop_length=hop, win_length=n, window=window, return_complex=True, pad_mode="constant")
    p = s.real * s.real + s.imag * s.imag                     # (B, F, T)
    if fb is None:
        fb = melscale_fbanks(n // 2 + 1, f_min, f_max, n_mels, sample_rate)
    mel = torch.matmul(p.transpose(-1, -2), fb).transpose(-1, -2).unsqueeze(1)
    return torch.log(mel + eps) if log else mel


def step(x: torch.Tensor, g: torch.Tensor, lambd_value: float, hop: int, n_mels: int, sample_rate: int, log=True, fb=None):
    """One forward + backward to lambd.grad; returns (out, dlambd)."""
    lam = torch.tensor(float(lambd_value), requires_grad=True)
    out = forward(x, lam, hop, n_mels, sample_rate, log=log, fb=fb)
    (dl,) = torch.autograd.grad((out * g).sum(), lam)
    return out.detach(), float(dl)
