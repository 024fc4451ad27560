#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE itself (dev container only).

Runs ``/root/reference``'s own ``models.MelSpectrogramLayer.forward`` (models.py:33-56),
``time_frequency.differentiable_spectrogram`` (time_frequency.py:32-58) and torch autograd on
CPU for every case in ``cases.py`` and writes ``tests/golden/<name>.npz``.  The reference is
imported read-only from where it lies; nothing of it is copied into this repo, only the
numeric outputs.  ``/root/reference`` does not exist on the GPU box, so this script is never
run there; tests consume the committed ``.npz`` files.

One third-party piece is NOT available: ``torchaudio`` (pinned 0.13.1, requirements.txt:8) is
absent from this image and there is no network.  ``models.py:42`` calls
``torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate)`` with the
defaults ``norm=None, mel_scale="htk"``; the stand-in below restates that published algorithm
(torchaudio/functional/functional.py, ``melscale_fbanks`` + ``_create_triangular_filterbank``)
op for op in fp32 torch.  The filterbank table is therefore PARITY-UNPINNED (formula only);
everything else in these fixtures comes out of the reference's own code and torch.

Usage:  python tests/golden/make_golden.py [case-name ...]
"""
from __future__ import annotations

import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases as C  # noqa: E402

REFERENCE = "/root/reference"


def _melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    """torchaudio 0.13.1 functional.melscale_fbanks, htk / norm=None branch."""
    assert norm is None and mel_scale == "htk"
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(zero, torch.min(down_slopes, up_slopes))


def _install_torchaudio_standin():
    ta = types.ModuleType("torchaudio")
    fn = types.ModuleType("torchaudio.functional")
    tr = types.ModuleType("torchaudio.transforms")
    fn.melscale_fbanks = _melscale_fbanks

    class _Unavailable:  # models.py:300 only touches this in code we never run
        def __init__(self, *a, **k):
            raise RuntimeError("torchaudio.transforms is not available in this image")

    class _NeverCalled(torch.nn.Module):  # panns.py:141-142 constructs the masks always, calls them only when augment=True
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            raise RuntimeError("torchaudio.transforms is not available in this image")

    tr.TimeMasking = tr.FrequencyMasking = _NeverCalled
    tr.MelSpectrogram = _Unavailable
    ta.functional, ta.transforms = fn, tr
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.functional"] = fn
    sys.modules["torchaudio.transforms"] = tr


def import_reference():
    _install_torchaudio_standin()
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    sys.dont_write_bytecode = True
    import models  # the reference's models.py
    import time_frequency  # the reference's time_frequency.py
    return models, time_frequency


def run_case(models, tf, case):
    torch.set_num_threads(8)
    x_np = C.make_input(case)
    x = torch.from_numpy(x_np)
    g = torch.from_numpy(C.make_cotangent(case))
    layer = models.MelSpectrogramLayer(
        init_lambd=torch.tensor(float(case["lambd"]), dtype=torch.float32),
        n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
        f_min=case["f_min"], f_max=case["f_max"], hop_length=case["hop"], device="cpu",
        optimized=case["optimized"], normalize_window=case["normalize_window"])
    n_fft = tf.next_power_of_2((torch.abs(layer.lambd) * 6).detach().cpu().numpy()) if case["optimized"] else 2 * case["L"]

    mel = layer(x)                                   # models.py:33-56
    y = torch.log(mel + 1e-10)                       # models.py:73
    (dl_log,) = torch.autograd.grad((y * g).sum(), layer.lambd, retain_graph=True)
    (dl_lin,) = torch.autograd.grad((mel * g).sum(), layer.lambd)

    mel_np = mel.detach().numpy().astype(np.float32)
    out = dict(n_fft=np.int64(n_fft), dlam_log=np.float32(dl_log.item()), dlam_lin=np.float32(dl_lin.item()),
               lambd=np.float32(case["lambd"]),
               mel_sum=mel_np.astype(np.float64).reshape(case["B"], -1).sum(1),
               y_sum=y.detach().numpy().astype(np.float64).reshape(case["B"], -1).sum(1))
    idx = C.sample_index(case)
    if idx is None:
        out["mel"] = mel_np
    else:
        out["mel_sampled"] = mel_np.reshape(-1)[idx]
    if case["kind"] in ("dc", "tone"):
        # the mean models.py:38 subtracted, clip by clip, in the input dtype (same expression, same torch)
        out["mean_ref"] = np.asarray([torch.mean(x[i]).item() for i in range(case["B"])], dtype=np.float64)
    return out


def run_dspec(models, tf, L=128):
    """G7: DSPEC non-optimized layer (models.py:171-200), hop=1, lambd=6.38; L=128 is the reference's own use, L=100 a clip
    length that is not a power of two (n_fft = 200)."""
    from dmel_amd import synth
    x = torch.from_numpy(synth.waveforms(2, L, seed=77, scale=1.0))
    layer = models.SpectrogramLayer(torch.tensor(6.38), optimized=False, hop_length=1)
    s = layer(x)
    g = torch.from_numpy(synth.cotangent(tuple(s.shape), seed=78))
    (dl,) = torch.autograd.grad((s * g).sum(), layer.lambd)
    return dict(spec=s.detach().numpy().astype(np.float32), dlam_lin=np.float32(dl.item()))


def run_dspec_xgrad(models, tf, L=128):
    """G7b: d loss / d x through the reference's SpectrogramLayer (models.py:171-200, optimized=False, hop 1) by torch autograd"""
    from dmel_amd import synth
    x = torch.from_numpy(synth.waveforms(2, L, seed=77, scale=1.0)).requires_grad_(True)
    layer = models.SpectrogramLayer(torch.tensor(6.38), optimized=False, hop_length=1)
    s = layer(x)
    g = torch.from_numpy(synth.cotangent(tuple(s.shape), seed=78))
    (gx,) = torch.autograd.grad((s * g).sum(), x)
    return dict(gx=gx.numpy().astype(np.float32))


FBGRAD_CASES = ("g1_c1", "g2_c2", "g5_n128", "g6_n256_ragged", "g7_mel_nonopt_256", "g7_mel_nonopt_601")   # the last two: optimized=False (n_fft = 2L; 601: not a power of two)


def run_fbgrad(models, tf, case):
    """G8b: d loss / d mel_fb out of the reference's own forward + torch autograd, with the filterbank of
    models.py:42-48 made a leaf (the stand-in hands the same leaf tensor to every per-sample call, so the
    per-sample gradients accumulate exactly as they would for a shared parameter)."""
    import torchaudio.functional as taf
    x = torch.from_numpy(C.make_input(case))
    g = torch.from_numpy(C.make_cotangent(case))
    layer = models.MelSpectrogramLayer(
        init_lambd=torch.tensor(float(case["lambd"]), dtype=torch.float32),
        n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
        f_min=case["f_min"], f_max=case["f_max"], hop_length=case["hop"], device="cpu",
        optimized=case["optimized"], normalize_window=case["normalize_window"])
    leaf = {}
    orig = taf.melscale_fbanks

    def shared_leaf(*a, **k):
        if "fb" not in leaf:
            leaf["fb"] = orig(*a, **k).detach().clone().requires_grad_(True)
        return leaf["fb"]

    taf.melscale_fbanks = shared_leaf
    try:
        mel = layer(x)
        y = torch.log(mel + 1e-10)
        (g_log,) = torch.autograd.grad((y * g).sum(), leaf["fb"], retain_graph=True)
        (g_lin,) = torch.autograd.grad((mel * g).sum(), leaf["fb"])
    finally:
        taf.melscale_fbanks = orig
    return dict(gfb_lin=g_lin.numpy().astype(np.float32), gfb_log=g_log.numpy().astype(np.float32))


def run_panns(models):
    """f4: the reference's MelPANNsNet (models.py:138-166 + panns.py:135-202) in eval mode with the closed-form weights of
    cases.fill_state: clipwise outputs and the state_dict contract (keys + shapes)."""
    import json
    cfg = C.PANNS_CFG
    from dmel_amd import synth
    out = {}
    for energy_normalize in (True, False):
        net = models.MelPANNsNet(cfg["n_classes"], torch.tensor(cfg["lambd"]), "cpu", cfg["n_mels"], cfg["sr"], cfg["L"],
                                 hop_length=cfg["hop"], optimized=True, energy_normalize=energy_normalize)
        C.fill_state(net, seed=cfg["seed"])
        net.eval()
        x = torch.from_numpy(synth.waveforms(cfg["B"], cfg["L"], seed=cfg["seed"]))
        with torch.no_grad():
            y, s = net(x)
        out["clipwise_log" if energy_normalize else "clipwise_lin"] = y.numpy().astype(np.float32)
        if energy_normalize:
            out["s_log"] = s.numpy().astype(np.float32)       # the log-mel the CNN saw: pins our Cnn6 on CPU
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    json.dump(keys, open(os.path.join(HERE, "panns_state_keys.json"), "w"), indent=1)
    return out


XGRAD_CASES = ("g1_c1", "g5_n128", "g6_n256_ragged", "g6_tone_dc", "g6_n32", "g7_mel_nonopt_256", "g7_mel_nonopt_1024n", "g7_mel_nonopt_601", "g7_mel_nonopt_8000")   # the last four: optimized=False (601, 8000: n_fft = 2L is not a power of two)


def run_xgrad(models, tf, case):
    """G10: d loss / d x out of the reference's own forward + torch autograd (x.requires_grad_()); stored as fp32, full."""
    x = torch.from_numpy(C.make_input(case)).requires_grad_(True)
    g = torch.from_numpy(C.make_cotangent(case))
    layer = models.MelSpectrogramLayer(
        init_lambd=torch.tensor(float(case["lambd"]), dtype=torch.float32),
        n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
        f_min=case["f_min"], f_max=case["f_max"], hop_length=case["hop"], device="cpu",
        optimized=case["optimized"], normalize_window=case["normalize_window"])
    mel = layer(x)
    y = torch.log(mel + 1e-10)
    (g_log,) = torch.autograd.grad((y * g).sum(), x, retain_graph=True)
    (g_lin,) = torch.autograd.grad((mel * g).sum(), x)
    keep = 2       # clips are independent: the first two pin the path, the fixtures stay small
    return dict(gx_lin=g_lin.numpy().astype(np.float32)[:keep], gx_log=g_log.numpy().astype(np.float32)[:keep])


NET_CASES = (
    # (fixture key, net class, case name, energy_normalize)
    ("conv_g1_log", "MelConvNet", "g1_c1", True),
    ("conv_g1_lin", "MelConvNet", "g1_c1", False),
    ("linear_g1_log", "MelLinearNet", "g1_c1", True),
    ("linear_g4_log", "MelLinearNet", "g4_esc_hop441", True),
    # round 6 (VERDICT r05 "missing" 5): models.py:80-103, the third wrapping net (fc 32 -> relu -> dropout -> fc)
    ("mlp_g1_log", "MelMlpNet", "g1_c1", True),
    ("mlp_g1_lin", "MelMlpNet", "g1_c1", False),
)


def run_nets(models):
    """f1: logits and s of the reference's own MelConvNet / MelLinearNet (models.py:58-78, 105-136) on the G1 and G4 inputs,
    weights from cases.fill_state.  MelLinearNet applies F.dropout(p=0.2) with its default training=True on every call
    (models.py:75): for a deterministic fixture torch.nn.functional.dropout is replaced by the identity while the reference
    runs (the test does the same to our net); MelConvNet has no dropout."""
    import torch.nn.functional as F
    out = {}
    orig = F.dropout
    F.dropout = lambda x, *a, **k: x
    try:
        for key, cls, cname, en in NET_CASES:
            case = C.BY_NAME[cname]
            net = getattr(models, cls)(C.NET_CLASSES, torch.tensor(float(case["lambd"])), "cpu", case["n_mels"], case["sr"], case["L"],
                                       hop_length=case["hop"], optimized=True, energy_normalize=en)
            C.fill_state(net, seed=C.NET_SEED)
            x = torch.from_numpy(C.make_input(case))
            with torch.no_grad():
                logits, sp = net(x)
            out[key + "_logits"] = logits.numpy().astype(np.float32)
            sp_np = sp.numpy().astype(np.float32)
            idx = C.sample_index(case)
            out[key + "_s"] = sp_np if idx is None else sp_np.reshape(-1)[idx]
    finally:
        F.dropout = orig
    return out


def run_net_keys(models):
    """state_dict keys + shapes of the reference's wrapping nets (models.py:58-136): the checkpoint contract."""
    import json
    out = {}
    for name in ("MelLinearNet", "MelMlpNet", "MelConvNet"):
        net = getattr(models, name)(10, torch.tensor(8000 * 0.035 / 6), "cpu", 64, 8000, 8000, hop_length=80, optimized=True,
                                    energy_normalize=True)
        out[name] = {k: list(v.shape) for k, v in net.state_dict().items()}
    json.dump(out, open(os.path.join(HERE, "net_state_keys.json"), "w"), indent=1)
    print("net_state_keys.json:", {k: list(v) for k, v in out.items()})


# ---- G12: a TRAINING TRAJECTORY across an n_fft boundary -------------------------------------------------------------
# The reference re-derives n_fft from lambd at every forward (time_frequency.py:39,60-65); the layer here keeps lambd on the device and
# follows it with guard launches / re-captured graphs.  These fixtures pin that machinery against the reference ITSELF: its
# MelSpectrogramLayer (models.py:14-56) + log (models.py:73) under torch.optim.Adam (main.py:52 uses Adam with lr_tf) for 48 steps from a
# lambd next to a power-of-two boundary; the loss changes sign at step 8 so that lambd turns round and crosses both ways
# where the gradient allows.  Stored: lambd BEFORE each step, the n_fft that forward used, the gradient it produced.
TRAJ = {"g12_traj_512_1024": dict(lam0=85.9, B=2, L=4000, hop=128, M=32, sr=8000, lr=0.1, steps=48, flip=8),
        "g12_traj_1024_2048": dict(lam0=171.4, B=2, L=4000, hop=128, M=32, sr=8000, lr=0.2, steps=48, flip=6)}


def run_trajectory(models, tf, cfg):
    torch.set_num_threads(8)
    x_np, g_np = C.traj_inputs(cfg)
    x, g = torch.from_numpy(x_np), torch.from_numpy(g_np)
    layer = models.MelSpectrogramLayer(init_lambd=torch.tensor(float(cfg["lam0"]), dtype=torch.float32), n_mels=cfg["M"], n_points=cfg["L"],
                                       sample_rate=cfg["sr"], hop_length=cfg["hop"], device="cpu", optimized=True)
    opt = torch.optim.Adam([layer.lambd], lr=cfg["lr"])
    lam, nfft, dlam = [], [], []
    for k in range(cfg["steps"]):
        lam.append(float(layer.lambd.detach()))
        nfft.append(int(tf.next_power_of_2((torch.abs(layer.lambd) * 6).detach().cpu().numpy())))
        opt.zero_grad()
        y = torch.log(layer(x) + 1e-10)
        sign = 1.0 if k < cfg["flip"] else -1.0
        (sign * (y * g).sum()).backward()
        dlam.append(float(layer.lambd.grad))
        opt.step()
    lam.append(float(layer.lambd.detach()))
    nfft = np.asarray(nfft, dtype=np.int64)
    crossings = int((nfft[1:] != nfft[:-1]).sum())
    # distance of 6 lambd from the integers where next_power_of_2(int(.)) changes: a port that is 1e-4 off must still take the same branch
    edges = np.asarray([2.0 ** e + 1.0 for e in range(5, 14)])
    margin = float(np.min(np.abs(6.0 * np.asarray(lam[:-1])[:, None] - edges[None, :]) / (6.0 * np.asarray(lam[:-1])[:, None])))
    if crossings < 2 or margin <= 2.5e-4:
        return None                                  # never crossed both ways, or a step too close to a boundary to pin with a 1e-4 tolerance
    return dict(lam=np.asarray(lam, dtype=np.float64), n_fft=nfft, dlam=np.asarray(dlam, dtype=np.float64), crossings=np.int64(crossings),
                margin=np.float64(margin), **{k: np.float64(v) for k, v in cfg.items()})


def main(argv):
    models, tf = import_reference()
    if len(argv) == 1 or "net_keys" in argv:
        run_net_keys(models)
        argv = [a for a in argv if a != "net_keys"]
        if len(argv) == 1 and "net_keys" in sys.argv:
            return
    names = argv[1:] or [c["name"] for c in C.CASES + C.DC_CASES] + ["g7_dspec", "g7_dspec_100", "g7_dspec_xgrad", "g7_dspec_xgrad_100"] + ["g8_fbgrad_" + n for n in FBGRAD_CASES] + ["g9_panns"] + ["g10_xgrad_" + n for n in XGRAD_CASES] + ["g11_nets"] + list(TRAJ)
    for name in names:
        if name == "g7_dspec":
            out = run_dspec(models, tf)
        elif name == "g7_dspec_100":
            out = run_dspec(models, tf, L=100)
        elif name == "g7_dspec_xgrad":
            out = run_dspec_xgrad(models, tf)
        elif name == "g7_dspec_xgrad_100":
            out = run_dspec_xgrad(models, tf, L=100)
        elif name.startswith("g10_xgrad_"):
            out = run_xgrad(models, tf, C.BY_NAME[name[len("g10_xgrad_"):]])
        elif name == "g9_panns":
            out = run_panns(models)
        elif name == "g11_nets":
            out = run_nets(models)
        elif name in TRAJ:
            # the first start on a fixed grid whose trajectory crosses both ways and keeps every step 2.5e-4 (relative) clear of a boundary
            for d in range(40):
                cfg = dict(TRAJ[name]); cfg["lam0"] = round(cfg["lam0"] + 0.013 * d, 4)
                out = run_trajectory(models, tf, cfg)
                if out is not None:
                    break
            assert out is not None, "no start on the grid gives a pinnable trajectory"
        elif name.startswith("g8_fbgrad_"):
            out = run_fbgrad(models, tf, C.BY_NAME[name[len("g8_fbgrad_"):]])
        else:
            out = run_case(models, tf, C.BY_NAME[name])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        extra = {k: (v.item() if np.ndim(v) == 0 else v.shape) for k, v in out.items()}
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  {extra}")


if __name__ == "__main__":
    main(sys.argv)
