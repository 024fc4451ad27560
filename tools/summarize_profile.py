#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of one profiling session (tools/profile_session.sh -> gpurun_out/prof_<tag>/{kt,fetch,write,sq1,sq2})
into the small files committed under profiles/.   usage: python tools/summarize_profile.py gpurun_out/prof_r02 r02"""
import csv
import glob
import hashlib
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r03")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
out = os.path.join(ROOT, "profiles")
FWD = "dmel_fwd_kernel<1024, 0"          # the training-mode forward at n_fft 1024 (any tiles-per-workgroup variant)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    assert len(hits) == 1, (pattern, hits)
    return hits[0]


def short(name):
    return name.split("(")[0].replace("void ", "")


# per-kernel stats + trace summary
shutil.copy(one("kt/**/*_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats_c2.csv"))
rows = list(csv.DictReader(open(one("kt/**/*_kernel_trace.csv"))))
summ = []
for k in sorted({r["Kernel_Name"] for r in rows}):
    rs = [r for r in rows if r["Kernel_Name"] == k]
    if len(rs) < 20:
        continue
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
    r0 = rs[0]
    summ.append(dict(kernel=short(k)[:110], calls=len(d), avg_ns=round(statistics.mean(d), 1), median_ns=statistics.median(d), min_ns=min(d),
                     max_ns=max(d), vgpr=r0.get("VGPR_Count"), accum_vgpr=r0.get("Accum_VGPR_Count"), sgpr=r0.get("SGPR_Count"),
                     lds=r0.get("LDS_Block_Size"), scratch=r0.get("Scratch_Size"), wg=r0.get("Workgroup_Size", r0.get("Workgroup_Size_X")), grid=r0.get("Grid_Size", r0.get("Grid_Size_X"))))
json.dump(summ, open(os.path.join(out, f"{tag}_kernel_trace_summary_c2.json"), "w"), indent=1)

# configs 3 and 5 (n_fft 2048): per-kernel summary of the launch trains of tools/ktime.py
other = {}
for cfg in ("c3", "c5"):
    hits = glob.glob(os.path.join(src, f"kt_{cfg}/**/*_kernel_trace.csv"), recursive=True)
    if len(hits) != 1:
        continue
    rws = list(csv.DictReader(open(hits[0])))
    lst = []
    for k in sorted({r["Kernel_Name"] for r in rws if "dmel" in r["Kernel_Name"]}):
        rs = [r for r in rws if r["Kernel_Name"] == k]
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
        r0 = rs[0]
        lst.append(dict(kernel=short(k)[:110], calls=len(d), avg_ns=round(statistics.mean(d), 1), median_ns=statistics.median(d), min_ns=min(d),
                        vgpr=r0.get("VGPR_Count"), lds=r0.get("LDS_Block_Size"), scratch=r0.get("Scratch_Size"),
                        wg=r0.get("Workgroup_Size", r0.get("Workgroup_Size_X")), grid=r0.get("Grid_Size", r0.get("Grid_Size_X"))))
    other[cfg] = lst
if other:
    json.dump(other, open(os.path.join(out, f"{tag}_kernel_trace_summary_c3_c5.json"), "w"), indent=1)

# optional backward outputs and the big-transform kernel: per-kernel summaries
for sub, name in (("kt_f2", "f2_backward_extras"), ("kt_big", "big_transforms")):
    hits = glob.glob(os.path.join(src, f"{sub}/**/*_kernel_trace.csv"), recursive=True)
    if len(hits) != 1:
        continue
    rws = list(csv.DictReader(open(hits[0])))
    lst = []
    for k in sorted({r["Kernel_Name"] for r in rws if "dmel" in r["Kernel_Name"]}):
        rs = [r for r in rws if r["Kernel_Name"] == k]
        # the scripts time several shapes with one kernel name: split by grid size
        for grid in sorted({r.get("Grid_Size", r.get("Grid_Size_X")) for r in rs}, key=lambda v: int(v)):
            rg = [r for r in rs if r.get("Grid_Size", r.get("Grid_Size_X")) == grid]
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rg]
            r0 = rg[0]
            lst.append(dict(kernel=short(k)[:110], grid=grid, calls=len(d), avg_ns=round(statistics.mean(d), 1), median_ns=statistics.median(d), min_ns=min(d),
                            vgpr=r0.get("VGPR_Count"), lds=r0.get("LDS_Block_Size"), scratch=r0.get("Scratch_Size"), wg=r0.get("Workgroup_Size", r0.get("Workgroup_Size_X"))))
    json.dump(lst, open(os.path.join(out, f"{tag}_kernel_trace_summary_{name}.json"), "w"), indent=1)
    log = os.path.join(src, sub + ".log")
    if os.path.exists(log):
        lines = [ln for ln in open(log) if ln.startswith("{")]
        if lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(out, f"{tag}_{name}_timings.json"), "w"), indent=1)
red = {}
for nm in ("bench_1rank_rccl", "bench_1rank_mailbox", "bench_2ranks_one_gpu_mailbox"):
    pth = os.path.join(src, nm + ".json")
    if os.path.exists(pth):
        lines = [ln for ln in open(pth) if ln.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            red[nm] = {"value_frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "n_ranks": d["n_gpus"], "reducer": d["config"].get("reducer"),
                       "issued": d["module_step"]["issued"], "trial_ms_per_step": d["module_step"]["trial_ms_per_step"]}
if red:
    json.dump({"_how": "bench.py --steps 200 --warmup 20 with DMEL_BENCH_FORCE_DIST=1 (one rank, the reducer still runs: one-rank RCCL communicator / "
                       "mailbox addressed to itself) and DMEL_BENCH_SHARE_GPU=1 --gpus 2 --reducer mailbox (two processes on ONE GPU exchanging through "
                       "IPC-mapped inboxes: the code path of two GPUs, not their speed -- the two ranks' kernels overlap on the one device)", **red},
              open(os.path.join(out, f"{tag}_reducers.json"), "w"), indent=1)
st = os.path.join(src, "stamps_c2.txt")
if os.path.exists(st):
    shutil.copy(st, os.path.join(out, f"{tag}_stamps_c2.txt"))

rs_path = os.path.join(src, "reference_shapes.json")
if os.path.exists(rs_path):
    lines = [ln for ln in open(rs_path) if ln.startswith("{")]
    if lines:
        json.dump({"_how": "tools/time_reference_shapes.py: forward + dot per training step through the C ABI (lambd by value), trains of 50 steps "
                           "between two HIP events, best of 3; shapes of search_spaces.py:4-33 (ESC-50: 32 x 40000 @ 8 kHz, hop 80, 64 mels) and "
                           ":36-66 (Audio-MNIST: 64 x 8000); lambd 13.3 / 46.7 / 400 are its init_lambd grid, 200 and 700 values a run may drift to",
                   "shapes": json.loads(lines[-1])}, open(os.path.join(out, f"{tag}_reference_shapes.json"), "w"), indent=1)

# one steady-state step of the timed region (graph replay): kernels in issue order with gaps
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def tight(i):      # forward, dot and the optimizer's kernels back to back: a graph replay, not the eagerly issued trial
    if i + 3 >= len(rows) or FWD not in rows[i]["Kernel_Name"] or "dmel_dot" not in rows[i + 1]["Kernel_Name"]:
        return False
    if "multi_tensor_apply" not in rows[i + 2]["Kernel_Name"]:
        return False
    return int(rows[i + 2]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) < 12000
idx = [i for i in range(len(rows)) if tight(i)]
mid = idx[len(idx) * 3 // 4]          # a forward that is followed by the optimizer's kernels: inside the graph-replayed module steps
seq, prev = [], None
for r in rows[mid:mid + 12]:
    s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    seq.append(dict(kernel=short(r["Kernel_Name"])[:80], dur_us=round((e0 - s0) / 1e3, 2), gap_before_us=None if prev is None else round((s0 - prev) / 1e3, 2)))
    prev = e0
json.dump(seq, open(os.path.join(out, f"{tag}_step_timeline_c2.json"), "w"), indent=1)

# PMC passes
res, sq = {}, {}
for name in ("fetch", "write", "sq1", "sq2"):
    path = one(f"{name}/**/*_counter_collection.csv")
    keep = [r for r in csv.DictReader(open(path)) if "dmel" in r["Kernel_Name"]]
    if name in ("fetch", "write"):
        counter = "FETCH_SIZE" if name == "fetch" else "WRITE_SIZE"
        with open(os.path.join(out, f"{tag}_pmc_{name}_size_c2.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
            for r in keep:
                if r["Counter_Name"] == counter:
                    w.writerow([r["Dispatch_Id"], short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"], r["Counter_Value"]])
        for kern, key in ((FWD, "dmel_fwd_kernel"), ("dmel_dot_kernel", "dmel_dot_kernel")):
            v = [float(r["Counter_Value"]) for r in keep if kern in r["Kernel_Name"] and r["Counter_Name"] == counter]
            res[f"{key}_{counter}_KiB_raw"] = statistics.median(v)
            res[f"{key}_{counter}_dispatches"] = len(v)
    else:
        for r in keep:
            if FWD in r["Kernel_Name"]:
                sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
bench = json.loads([ln for ln in open(os.path.join(src, "bench_c2.json")) if ln.startswith("{")][-1])
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
dot_known = 2 * 4 * 256 * 128 * 32
factor = dot_known / (res["dmel_dot_kernel_FETCH_SIZE_KiB_raw"] * 1024)      # calibration on a kernel whose reads are known exactly
traffic = int(round(res["dmel_fwd_kernel_FETCH_SIZE_KiB_raw"] * 1024 * 2 + res["dmel_fwd_kernel_WRITE_SIZE_KiB_raw"] * 1024))
sha = hashlib.sha256(open(os.path.join(ROOT, "differentiable-mel-spectrogram_amd", "csrc", "dmel_fwd.hip"), "rb").read()).hexdigest()[:16]
hbm = {
    "_how": "rocprofv3 --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE around `bench.py --steps 30 --warmup 5 --mode eager` (config 2); medians over the "
            "dispatches; raw per-dispatch values: profiles/%s_pmc_fetch_size_c2.csv, %s_pmc_write_size_c2.csv.  gfx950 correction "
            "(MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of streamed bytes -> doubled.  Calibrated in the same run on dmel_dot_kernel, which "
            "reads exactly 2 x 4 194 304 B with 16-byte loads: measured factor %.3f (2.0 expected).  WRITE_SIZE is taken as it is.  The working "
            "set fits the 256 MiB Infinity Cache, so these are fabric-side bytes, not necessarily DRAM bytes.  bench.py quotes the number only "
            "while csrc/dmel_fwd.hip still hashes to kernel_source_sha16." % (tag, tag, factor),
    "kernel_source_sha16": sha,
    "c2": {
        "dmel_fwd_kernel_FETCH_SIZE_KiB_raw": res["dmel_fwd_kernel_FETCH_SIZE_KiB_raw"],
        "dmel_fwd_kernel_WRITE_SIZE_KiB": res["dmel_fwd_kernel_WRITE_SIZE_KiB_raw"],
        "dmel_fwd_kernel_bytes_per_launch": traffic,
        "algorithmic_bytes_per_launch": alg,
        "dmel_dot_kernel_FETCH_SIZE_KiB_raw": res["dmel_dot_kernel_FETCH_SIZE_KiB_raw"],
        "dmel_dot_kernel_bytes_known": dot_known,
        "fetch_calibration_factor": round(factor, 3),
    },
}
json.dump(hbm, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
med = {k: statistics.median(v) for k, v in sq.items()}
waves = med.get("SQ_WAVES", 1.0)
sqj = {"_how": "rocprofv3 --pmc, two separate passes (tools/profile_session.sh: sq1, sq2) around `bench.py --steps 30 --warmup 5 --mode eager`; medians over the "
               "dispatches of dmel_fwd_kernel<1024, train>; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles",
       "medians_per_launch": med,
       "per_wave": {k.replace("SQ_INSTS_", "insts_"): round(v / waves, 1) for k, v in med.items() if k.startswith("SQ_INSTS_")},
       "shares_of_wave_cycles": {k: round(med[k] / med["SQ_WAVE_CYCLES"], 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in med and "SQ_WAVE_CYCLES" in med},
       "lds_bank_conflict_share": round(med.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, med.get("SQ_LDS_IDX_ACTIVE", 1.0)), 4)}
json.dump(sqj, open(os.path.join(out, f"{tag}_pmc_sq_c2.json"), "w"), indent=1)
bench["roofline"]["traffic"] = traffic
json.dump(bench, open(os.path.join(out, f"{tag}_bench_c2.json"), "w"), indent=1)
print(json.dumps([s_ for s_ in summ if "dmel" in s_["kernel"]], indent=1))
print(json.dumps(other, indent=1))
print(json.dumps(hbm["c2"], indent=1))
print(json.dumps(sqj["per_wave"]), json.dumps(sqj["shares_of_wave_cycles"]))
print({k: bench[k] for k in ("value", "ms_per_step")}, bench["module_step"])
