"""Builds libdmel_hip.so (gfx950 only) in-tree with hipcc.  No torch headers, no JIT cache:
the shared object sits next to this file so that it travels with the source tree."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libdmel_hip.so")
OBJ_DIR = os.path.join(PKG_DIR, "build")
SOURCES = ["dmel_fwd.hip", "dmel_aux.hip", "dmel_xgrad.hip", "dmel_api.cpp", "dmel_comm.cpp"]
HEADERS = [os.path.join(CSRC, "dmel_kernels.h"), os.path.join(CSRC, "dmel_ldsfft.h"), os.path.join(os.path.dirname(PKG_DIR), "include", "dmel.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        if force or _stale(obj, [sp] + HEADERS):
            cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
