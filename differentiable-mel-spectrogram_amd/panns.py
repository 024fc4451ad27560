"""The paper's headline classifier on top of the layer: PANNs Cnn6 behind the DMEL front end.

``MelPANNsNet`` has the constructor, attribute names, ``(clipwise_output, s)`` return value and ``state_dict`` keys of the
reference's ``models.MelPANNsNet`` (models.py:138-166) and ``Cnn6`` those of ``panns.Cnn6`` (panns.py:135-202):
``spectrogram_layer.lambd``, ``spectrogram_model.bn1.*``, ``spectrogram_model.conv_block{1..4}.{conv1,bn1}.*``,
``spectrogram_model.fc1.*``, ``spectrogram_model.fc_esc50.*`` -- so the AudioSet-pretrained Cnn6 weights load through the
same key remap as utils.py:15-36 (``load_cnn6_checkpoint``).  The CNN is stock torch (MIOpen convolutions); only
``spectrogram_layer`` is ours.

SpecAugment masks: the reference uses ``torchaudio.transforms.TimeMasking(64, iid_masks=True)`` and
``FrequencyMasking(8, iid_masks=True)`` (panns.py:141-142).  torchaudio is not available in this image, so ``_IidAxisMask``
restates its published ``mask_along_axis_iid``: per example, width ~ U[0, param), start ~ U[0, size - width), both
truncated to integers, band set to 0.  Parity unpinned (random by construction; only used when ``augment=True``).
"""
from __future__ import annotations

import collections
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layer import MelSpectrogramLayer


class _IidAxisMask(nn.Module):
    """Zero one random band per example along ``axis`` of a (B, C, H, W) tensor; no parameters, no state."""

    def __init__(self, mask_param: int, axis: int):
        super().__init__()
        self.mask_param, self.axis = int(mask_param), int(axis)

    def forward(self, x):
        size = x.shape[self.axis]
        shape = [x.shape[0], x.shape[1]] + [1, 1]
        width = torch.rand(shape[:2], device=x.device, dtype=x.dtype) * self.mask_param
        start = torch.rand(shape[:2], device=x.device, dtype=x.dtype) * (size - width)
        lo = start.long().view(shape)
        hi = (start.long() + width.long()).view(shape)
        pos = torch.arange(size, device=x.device).view([1, 1, -1, 1] if self.axis == 2 else [1, 1, 1, -1])
        return x.masked_fill((pos >= lo) & (pos < hi), 0.0)


class ConvBlock5x5(nn.Module):
    """panns.py:68-102: conv 5x5 (no bias) - batch norm - relu - pooling."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=(5, 5), stride=(1, 1), padding=(2, 2), bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        nn.init.xavier_uniform_(self.conv1.weight)                    # panns.py:7-13, 82-84

    def forward(self, x, pool_size=(2, 2), pool_type="avg"):
        x = F.relu_(self.bn1(self.conv1(x)))
        if pool_type == "avg":
            return F.avg_pool2d(x, kernel_size=pool_size)
        if pool_type == "max":
            return F.max_pool2d(x, kernel_size=pool_size)
        if pool_type == "avg+max":
            return F.avg_pool2d(x, kernel_size=pool_size) + F.max_pool2d(x, kernel_size=pool_size)
        raise ValueError(f"pool_type {pool_type!r}")


class Cnn6(nn.Module):
    """panns.py:135-202.  Input (B, 1, time_steps, mel_bins); output sigmoid scores (B, classes_num)."""

    def __init__(self, classes_num, n_mels, augment=False):
        super().__init__()
        self.augment = augment
        # the tensor is (B, 1, time, mel) here, so torchaudio's "time" masking (last axis) lands on the mel axis and its
        # "frequency" masking (axis -2) on the time axis; kept as the reference has it (panns.py:141-142, 174-176)
        self.mask_time = _IidAxisMask(64, axis=3)
        self.mask_freq = _IidAxisMask(8, axis=2)
        self.bn1 = nn.BatchNorm2d(n_mels)
        self.conv_block1 = ConvBlock5x5(1, 64)
        self.conv_block2 = ConvBlock5x5(64, 128)
        self.conv_block3 = ConvBlock5x5(128, 256)
        self.conv_block4 = ConvBlock5x5(256, 512)
        self.fc1 = nn.Linear(512, 512, bias=True)
        self.fc_esc50 = nn.Linear(512, classes_num, bias=True)
        for fc in (self.fc1, self.fc_esc50):                          # panns.py:158-161
            nn.init.xavier_uniform_(fc.weight)
            nn.init.zeros_(fc.bias)

    def forward(self, x):
        x = self.bn1(x.transpose(1, 3)).transpose(1, 3)               # batch norm over mel bins (panns.py:169-172)
        if self.training and self.augment:
            x = self.mask_freq(self.mask_time(x))
        for block in (self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4):
            x = F.dropout(block(x, pool_size=(2, 2), pool_type="avg"), p=0.2, training=self.training)
        x = x.mean(dim=3)                                             # over mel
        x = x.max(dim=2).values + x.mean(dim=2)                       # over time
        x = F.dropout(x, p=0.5, training=self.training)
        x = F.relu_(self.fc1(x))
        return torch.sigmoid(self.fc_esc50(x))                        # the embedding dropout of panns.py:197 is unused there too


class MelPANNsNet(nn.Module):
    """models.py:138-166."""

    def __init__(self, n_classes, init_lambd, device, n_mels, sample_rate, n_points, hop_length=1, optimized=False,
                 energy_normalize=False, normalize_window=False, augment=False):
        super().__init__()
        self.energy_normalize = energy_normalize
        self.spectrogram_layer = MelSpectrogramLayer(
            init_lambd, n_mels=n_mels, n_points=n_points, sample_rate=sample_rate, hop_length=hop_length, device=device,
            optimized=optimized, normalize_window=normalize_window, log=bool(energy_normalize))
        self.spectrogram_model = Cnn6(n_classes, n_mels, augment=augment)

    def forward(self, x):
        s = self.spectrogram_layer(x)                                  # (B, 1, mel, time), log-compressed when energy_normalize
        return self.spectrogram_model(s.transpose(2, 3)), s


def load_cnn6_checkpoint(model: nn.Module, checkpoint_path: str, device="cpu"):
    """utils.py:15-36 without the download: prefix every key of ``checkpoint['model']`` with ``spectrogram_model.`` and
    load non-strictly (the AudioSet head ``fc_audioset`` has no counterpart; ``fc_esc50`` keeps its initialisation).
    Returns torch's (missing_keys, unexpected_keys)."""
    if not os.path.exists(checkpoint_path):
        raise FileNotFoundError(f"{checkpoint_path} not found (Cnn6_mAP=0.343.pth from zenodo record 3987831; no network here)")
    state = torch.load(checkpoint_path, map_location=device)["model"]
    remapped = collections.OrderedDict(("spectrogram_model." + k, v) for k, v in state.items())
    return model.load_state_dict(remapped, strict=False)
