import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The shared objects are build products (git-ignored): a fresh checkout has none.  Build what is missing once per
    session -- the HIP library cross-compiles without a GPU (about a minute), the oracle is a single C file -- so that the
    suite does not depend on `__graft_entry__.build()` having run first."""
    import importlib.util
    import shutil
    pkg = os.path.join(ROOT, "differentiable-mel-spectrogram_amd")
    if not os.path.exists(os.path.join(pkg, "libdmel_hip.so")) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        spec = importlib.util.spec_from_file_location("_dmel_build", os.path.join(pkg, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    from oracle import dmel_oracle
    dmel_oracle.build()
    yield


def pytest_sessionfinish(session, exitstatus):
    """the numbers behind the parity asserts (tests/test_hip_parity.py: parity_stats) go to gpurun_out/, which travels back from the
    GPU box; profiles/r06_parity_report.json is a committed copy"""
    try:
        import json
        mod = sys.modules.get("test_hip_parity")
        rep = getattr(mod, "_REPORT", None) if mod else None
        if rep:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            json.dump(rep, open(os.path.join(out, "r06_parity_report.json"), "w"), indent=1, sort_keys=True)
    except Exception:            # noqa: BLE001 -- a report, never a reason to fail the session
        pass
