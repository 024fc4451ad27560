"""Import shim: ``import dmel_amd`` loads the package that lives in
``differentiable-mel-spectrogram_amd/`` (a directory name that is not a valid
Python identifier, so it cannot be imported by name)."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "differentiable-mel-spectrogram_amd")
_spec = importlib.util.spec_from_file_location(
    "dmel_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dmel_amd"] = _mod
_spec.loader.exec_module(_mod)
