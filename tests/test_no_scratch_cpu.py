"""No shipped instantiation of the fused forward kernel up to n_fft 4096 may spill registers to scratch memory, outside an explicit allow-list
with a measured justification per entry (VERDICT r05 "next" #4).  Reads the register / spill / scratch figures the compiler recorded in the
code objects of differentiable-mel-spectrogram_amd/build/dmel_fwd_part*.o (what tools/kres.sh prints): no GPU needed.

Round 6: 13 instantiations spilled at n_fft <= 4096 (6 ... 22 registers); the R subtractions `x - mean` of the window multiply were being hoisted
above the interior / edge branch as one block (R more live registers at the kernel's tightest point), and the edge path of the pair modes
interleaved the index arithmetic of all R entries.  With both fixed (csrc/dmel_fwd.hip: "its own copy", "four entries at a time") every training
instantiation up to 4096 is clean -- <1024, kTrain> went from 128 registers + 8 spilled to 109 -- and five pair / dense-bf16 instantiations keep
7 ... 10 spilled registers: 4 + 4 (or 5 + 5) 8-byte scratch accesses per wave and tile next to ~2 000 other instructions."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "differentiable-mel-spectrogram_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"
MODES = {0: "kTrain", 1: "kInfer", 2: "kSpec", 3: "kSpecTrain", 4: "kTrainH", 5: "kTrainW"}

# (n_fft, mode): (max spilled registers, justification -- measured on MI355X, gpurun_out/s9_modes.log of round 6 unless noted)
ALLOW = {
    (2048, 1): (8, "kInfer: 8 spilled (20 in round 5).  Inference at config 3 / 5: 35.35 | 35.28 us with 20 spilled (round-5 library) against 35.29 | 35.34 us with 8, "
                   "45.66 | 45.81 against 45.49 | 45.63: removing 12 of the 20 moved nothing, the remaining 8 are 4 stores + 4 loads of 8 bytes per wave and tile"),
    (2048, 2): (8, "kSpec (filterbank-gradient recompute, SpectrogramLayer): the same 4 + 4 accesses, the same windowing code as kInfer"),
    (2048, 4): (7, "kTrainH (opt-in dense bf16x3 contraction): 7 spilled (17 in round 5)"),
    (4096, 1): (10, "kInfer at the reference's ESC-50 grid point: 10 spilled (22 in round 5): 88.16 | 88.35 us before, 87.25 | 87.35 after"),
    (4096, 2): (8, "kSpec: as (2048, kSpec)"),
}
BIG = (8192, 16384)          # frames spread over 2 / 4 waves, 256 registers per lane: 18 ... 53 spilled in the pair modes; not the hot path (n_fft 128 ... 4096)


def kernel_resources():
    objs = sorted(f for f in os.listdir(BUILD) if re.fullmatch(r"dmel_fwd_part\d\.o", f)) if os.path.isdir(BUILD) else []
    if not objs or not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        return None
    res = {}
    tmp = tempfile.mkdtemp()
    try:
        for o in objs:
            fat, co = os.path.join(tmp, o + ".fat"), os.path.join(tmp, o + ".co")
            subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", os.path.join(BUILD, o)])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   f"--input={fat}", f"--output={co}"])
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, None])[1]      # noqa: E731
                m = re.search(r"dmel_fwd_kernelILi(\d+)ELi(\d+)ELi(\d+)E", g("name") or "")
                if m:
                    res[(int(m.group(1)), int(m.group(2)), int(m.group(3)))] = dict(vgpr=int(g("vgpr_count")), spill=int(g("vgpr_spill_count")),
                                                                                   scratch=int(g("private_segment_fixed_size")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def test_no_scratch_in_shipped_forward_kernels_up_to_4096():
    res = kernel_resources()
    if res is None:
        pytest.skip("no compiled objects (python __graft_entry__.py build) or no llvm-readelf in this image")
    assert len(res) >= 55, f"only {len(res)} instantiations found: the metadata parser no longer sees the kernels"
    bad, seen_allowed = [], set()
    for (n, mode, tpw), r in sorted(res.items()):
        if n in BIG:
            continue
        if r["spill"] == 0 and r["scratch"] == 0:
            continue
        cap = ALLOW.get((n, mode))
        if tpw == 1 and cap is not None and r["spill"] <= cap[0]:
            seen_allowed.add((n, mode))
            continue
        bad.append(f"dmel_fwd_kernel<{n}, {MODES.get(mode, mode)}, {tpw}>: {r['spill']} spilled registers, {r['scratch']} B of scratch")
    assert not bad, "register spills in a shipped instantiation (free the registers, or measure and add to ALLOW):\n  " + "\n  ".join(bad)
    stale = set(ALLOW) - seen_allowed
    assert not stale, f"ALLOW lists instantiations that no longer spill: {sorted(stale)}: remove them"
    # the kernels the bench line and BASELINE configs 2 - 5 run
    for key in ((1024, 5, 1), (2048, 5, 1), (512, 0, 1), (4096, 0, 1), (128, 0, 1)):
        assert res[key]["spill"] == 0 and res[key]["vgpr"] <= (128 if key[0] <= 2048 else 256), (key, res[key])
