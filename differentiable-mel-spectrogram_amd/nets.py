"""Callers of the layer: the small nets the reference wraps around its mel layer.

Same constructor arguments, attribute / parameter names and ``(logits, s)`` return value as the
reference's ``MelLinearNet``, ``MelMlpNet`` and ``MelConvNet`` (models.py:58-136), so checkpoints
(``state_dict`` keys ``spectrogram_layer.lambd``, ``fc.*`` / ``fc1.*`` / ``fc2.*`` / ``conv1.*``) and the
per-parameter-group optimizer of main.py:36-48 carry over.  The heads are stock torch modules (MIOpen /
rocBLAS); only ``spectrogram_layer`` is ours, and ``energy_normalize=True`` uses its fused
``log(s + 1e-10)`` epilogue instead of a separate torch op (models.py:73, 97, 126).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layer import MelSpectrogramLayer


class _MelFrontNet(nn.Module):
    def __init__(self, init_lambd, device, n_mels, sample_rate, n_points, hop_length, optimized, energy_normalize,
                 normalize_window):
        super().__init__()
        self.spectrogram_layer = MelSpectrogramLayer(
            init_lambd, n_mels=n_mels, n_points=n_points, sample_rate=sample_rate, hop_length=hop_length,
            device=device, optimized=optimized, normalize_window=normalize_window, log=bool(energy_normalize))
        self.device = device
        self.size = (n_mels, n_points // hop_length + 1)
        self.energy_normalize = energy_normalize

    def _features(self, x):
        return self.spectrogram_layer(x)          # already log-compressed when energy_normalize


class MelLinearNet(_MelFrontNet):
    """models.py:58-78: dropout(0.2) on the flattened spectrogram, one linear layer."""

    def __init__(self, n_classes, init_lambd, device, n_mels, sample_rate, n_points, hop_length=1, optimized=False,
                 energy_normalize=False, normalize_window=False):
        super().__init__(init_lambd, device, n_mels, sample_rate, n_points, hop_length, optimized, energy_normalize,
                         normalize_window)
        self.fc = nn.Linear(self.size[0] * self.size[1], n_classes)

    def forward(self, x):
        s = self._features(x)
        h = F.dropout(s.view(-1, self.size[0] * self.size[1]), p=0.2)     # always active, as in the reference
        return self.fc(h), s


class MelMlpNet(_MelFrontNet):
    """models.py:80-103: linear(32) - relu - dropout(0.2) - linear."""

    def __init__(self, n_classes, init_lambd, device, n_mels, sample_rate, n_points, hop_length=1, optimized=False,
                 energy_normalize=False, normalize_window=False):
        super().__init__(init_lambd, device, n_mels, sample_rate, n_points, hop_length, optimized, energy_normalize,
                         normalize_window)
        self.fc1 = nn.Linear(self.size[0] * self.size[1], 32)
        self.fc2 = nn.Linear(32, n_classes)

    def forward(self, x):
        s = self._features(x)
        h = F.dropout(F.relu(self.fc1(s.view(-1, self.size[0] * self.size[1]))), p=0.2)
        return self.fc2(h), s


class MelConvNet(_MelFrontNet):
    """models.py:105-136: conv5x5(32, 'same') - relu - linear(32) - relu - linear (BASELINE config 5's CNN)."""

    def __init__(self, n_classes, init_lambd, device, n_mels, sample_rate, n_points, hop_length=1, optimized=False,
                 energy_normalize=False, normalize_window=False):
        super().__init__(init_lambd, device, n_mels, sample_rate, n_points, hop_length, optimized, energy_normalize,
                         normalize_window)
        self.hidden_state = 32
        self.conv1 = nn.Conv2d(1, self.hidden_state, 5, padding="same")
        self.fc1 = nn.Linear(self.hidden_state * self.size[0] * self.size[1], self.hidden_state)
        self.fc2 = nn.Linear(self.hidden_state, n_classes)

    def forward(self, x):
        s = self._features(x)
        h = F.relu(self.conv1(s)).view(-1, self.hidden_state * self.size[0] * self.size[1])
        return self.fc2(F.relu(self.fc1(h))), s


def make_optimizer(net: nn.Module, lr_model: float, lr_tf: float, name: str = "adam"):
    """The two learning-rate groups of main.py:36-53: ``lr_tf`` for ``spectrogram_layer.lambd``, ``lr_model`` for the rest."""
    groups = [{"params": [p], "lr": (lr_tf if n == "spectrogram_layer.lambd" else lr_model)} for n, p in net.named_parameters()]
    if name == "adam":
        return torch.optim.Adam(groups)
    if name == "sgd":
        return torch.optim.SGD(groups)
    raise ValueError(f"optimizer not found: {name}")
