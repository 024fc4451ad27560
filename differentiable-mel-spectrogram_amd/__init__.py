"""dmel_amd: MI355X-native differentiable log-Mel spectrogram layer (DMEL hot path).

Drop-in for the reference's ``models.MelSpectrogramLayer`` (models.py:14-56) and the
``log(s + 1e-10)`` line of its wrapping nets (models.py:73), executed by hand-written
HIP kernels for gfx950 behind a C-ABI shared library (``include/dmel.h``).

Heavy pieces (torch, the HIP library) are imported lazily so that ``dmel_amd.synth``
stays usable from tooling that has neither.
"""
from __future__ import annotations

import importlib

__all__ = ["MelSpectrogramLayer", "DifferentiableMelSpectrogram", "SpectrogramLayer", "SlotInput", "dmel_log_mel", "GraphedStep", "LambdAdam", "capi", "synth", "dist",
           "nets", "panns", "graph", "optim"]

_LAZY = {
    "MelSpectrogramLayer": ("layer", "MelSpectrogramLayer"),
    "DifferentiableMelSpectrogram": ("layer", "DifferentiableMelSpectrogram"),
    "dmel_log_mel": ("layer", "dmel_log_mel"),
    "SpectrogramLayer": ("layer", "SpectrogramLayer"),
    "SlotInput": ("layer", "SlotInput"),
    "GraphedStep": ("graph", "GraphedStep"),
    "LambdAdam": ("optim", "LambdAdam"),
}


def __getattr__(name):
    if name in _LAZY:
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"dmel_amd.{mod}"), attr)
    if name in ("capi", "synth", "layer", "dist", "nets", "panns", "graph", "optim"):
        return importlib.import_module(f"dmel_amd.{name}")
    raise AttributeError(name)
