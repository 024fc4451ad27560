"""Builds the two shared objects of the package in-tree (no JIT cache: they sit next to this file and travel with the
source tree):
  libdmel_hip.so    every kernel + the C ABI of include/dmel.h; hipcc, gfx950 only, no torch headers
  libdmel_torch.so  TORCH_LIBRARY(dmel, ...): the torch-registered ops over that C ABI (csrc/dmel_torch.cpp); plain g++
                    against the installed torch's headers, no device code"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libdmel_hip.so")
TORCH_LIB_PATH = os.path.join(PKG_DIR, "libdmel_torch.so")
OBJ_DIR = os.path.join(PKG_DIR, "build")
SOURCES = ["dmel_fwd.hip", "dmel_aux.hip", "dmel_big.hip", "dmel_xgrad.hip", "dmel_api.cpp", "dmel_comm.cpp"]
HEADERS = [os.path.join(CSRC, "dmel_kernels.h"), os.path.join(CSRC, "dmel_ldsfft.h"), os.path.join(CSRC, "dmel_wavefft.h"), os.path.join(os.path.dirname(PKG_DIR), "include", "dmel.h")]
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


# csrc/dmel_fwd.hip is compiled FWD_PARTS times (-DDMEL_FWD_SPLIT -DDMEL_FWD_PART=k): its large instantiations in parallel
FWD_PARTS = 4


def _units():
    """(source path, object path, extra flags) of every translation unit"""
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        stem = os.path.splitext(src)[0]
        if src == "dmel_fwd.hip":
            for k in range(FWD_PARTS):
                yield sp, os.path.join(OBJ_DIR, f"{stem}_part{k}.o"), ["-DDMEL_FWD_SPLIT", f"-DDMEL_FWD_PART={k}"]
        else:
            yield sp, os.path.join(OBJ_DIR, stem + ".o"), []


def _reap(procs) -> None:
    """ends the compiler processes this build started and still has running (their exact handles: nothing is matched by name)"""
    for pr in procs:
        if pr.poll() is None:
            pr.kill()
    for pr in procs:
        pr.wait()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs, running = [], []
    jobs = max(1, min(int(os.environ.get("DMEL_BUILD_JOBS", "0")) or (os.cpu_count() or 2), 8))
    for sp, obj, extra in _units():
        objs.append(obj)
        if force or _stale(obj, [sp] + HEADERS):
            cmd = [hipcc] + FLAGS + extra + ["-x", "hip", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            while len(running) >= jobs:
                if running.pop(0).wait() != 0:
                    _reap(running)
                    raise RuntimeError("hipcc failed")
            running.append(subprocess.Popen(cmd))
    while running:
        if running.pop(0).wait() != 0:
            _reap(running)                               # (the other compilers would run on, detached, for minutes: ADVICE r04)
            raise RuntimeError("hipcc failed")
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_torch(force=force, verbose=verbose)
    return LIB_PATH


def build_torch(force: bool = False, verbose: bool = False) -> str:
    """libdmel_torch.so: links against libdmel_hip.so (rpath $ORIGIN) and the torch the interpreter imports."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        raise RuntimeError("torch is not importable: libdmel_torch.so cannot be built")
    tdir = list(spec.submodule_search_locations)[0]
    src = os.path.join(CSRC, "dmel_torch.cpp")
    if not (force or _stale(TORCH_LIB_PATH, [src, HEADERS[-1], LIB_PATH])):
        return TORCH_LIB_PATH
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no C++ compiler found for libdmel_torch.so")
    inc = os.path.join(tdir, "include")
    cmd = [cxx, "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=1",
           "-I" + inc, "-I" + os.path.join(inc, "torch", "csrc", "api", "include"), "-I/opt/rocm/include",
           src, "-o", TORCH_LIB_PATH, "-L" + PKG_DIR, "-ldmel_hip", "-L" + os.path.join(tdir, "lib"),
           "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TORCH_LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
