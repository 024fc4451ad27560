#!/usr/bin/env python3
"""Per-phase instruction budget of the fused forward (n_fft 1024, training) from hardware counters: the -DDMEL_ABLATE build is launched with
its phase-skipping debug flags, rocprofv3 --pmc counts the instructions of every dispatch, the differences are the phases.
  on the box:  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv \\
                   -d <dir> -- python3 tools/phase_budget.py run [c2|c3]  (DMEL_LIB = the ablate build)
  then:        python3 tools/phase_budget.py parse <dir> [c2|c3] > profiles/r06_phase_budget_<config>.json"""
import csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (label, flags): 0x100 skip the contraction + epilogue, 0x200 skip the transforms (prologue, FFT, pairing), 0x400 skip the epilogue,
# 0x800 skip the MFMA loops, 0x1000 no clip sum
VARIANTS = [("full", 0x0), ("no_clip_sum", 0x1000), ("transforms_only", 0x100), ("transforms_only_no_clip_sum", 0x1100), ("contraction_and_epilogue_only", 0x200),
            ("contraction_only", 0x600), ("epilogue_only", 0xA00), ("neither", 0x300)]
REPS = 6

if sys.argv[1] == "run":
    sys.path.insert(0, ROOT)
    import torch
    import dmel_amd
    from dmel_amd import capi, synth
    from bench import CONFIGS
    B, L, sr, lam, hop, M = CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "c2"]
    T = L // hop + 1
    x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
    out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out)
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    s = torch.cuda.current_stream().cuda_stream
    for _, f in VARIANTS:
        for _ in range(REPS):
            plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, s, extra_flags=f)
    torch.cuda.synchronize()
    sys.exit(0)

d = sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "c2"
NF = {"c2": 1024, "c3": 2048, "c5": 2048}[cfg]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
per = {}
for r in csv.DictReader(open(f)):
    if f"dmel_fwd_kernel<{NF}, 5" in r["Kernel_Name"] or f"dmel_fwd_kernel<{NF}, 0" in r["Kernel_Name"]:
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(per)
assert len(ids) == len(VARIANTS) * REPS, (len(ids), len(VARIANTS) * REPS)
res = {}
for vi, (label, flags) in enumerate(VARIANTS):
    grp = [per[i] for i in ids[vi * REPS:(vi + 1) * REPS]]
    med = {c: statistics.median(g[c] for g in grp) for c in grp[0]}
    w = med["SQ_WAVES"]
    res[label] = {"flags": hex(flags), "per_wave": {c.replace("SQ_INSTS_", "").lower(): round(v / w, 1) for c, v in med.items() if c.startswith("SQ_INSTS_")}}
def diff(a, b):
    return {k: round(res[a]["per_wave"][k] - res[b]["per_wave"][k], 1) for k in res[a]["per_wave"]}
phases = {"launch, indices, lambd, exit (the kernel with both halves skipped)": res["neither"]["per_wave"],
          "clip sum (models.py:38)": diff("full", "no_clip_sum"),
          "prologue without the clip sum + window multiply + both radix-32 stages + twiddles + transposition + pairing pass + PD store": diff("transforms_only_no_clip_sum", "neither"),
          "contraction: lane tables, B ring, A reads, MFMA loops (round 5: wave-local; the MFMAs count as vector instructions)": diff("contraction_only", "neither"),
          # (round 5: the no-MFMA variant of the wave-local loop puts adds in their place, so the epilogue is taken from the two variants that keep them)
          "epilogue (scale, log, tangent, staging, stores)": diff("contraction_and_epilogue_only", "contraction_only"),
          "whole kernel": res["full"]["per_wave"]}
print(json.dumps({"_how": f"tools/phase_budget.py: SQ instruction counters (rocprofv3 --pmc) of the -DDMEL_ABLATE build launched with its phase-skipping flags at BASELINE "
                          f"config {cfg[1:]} (n_fft {NF}); medians of 6 dispatches per variant, per wave (a wave = two frames at n_fft 1024, one at 2048); phases are differences of variants",
                  "variants": res, "phases_per_wave": phases}, indent=1))
