#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of one profiling session (gpurun_out/prof/{kt,fetch,write}, see profiles/README.md)
into the small files committed under profiles/.   usage: python tools/summarize_profile.py [gpurun_out/prof] [r01]"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
out = os.path.join(ROOT, "profiles")


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
for k in sorted({r["Kernel_Name"] for r in rows if "dmel" in r["Kernel_Name"]}):
    rs = [r for r in rows if r["Kernel_Name"] == k]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
    r0 = rs[0]
    summ.append(dict(kernel=short(k), calls=len(d), avg_ns=round(statistics.mean(d), 1), median_ns=statistics.median(d), min_ns=min(d),
                     max_ns=max(d), vgpr=r0.get("VGPR_Count"), accum_vgpr=r0.get("Accum_VGPR_Count"), sgpr=r0.get("SGPR_Count"),
                     lds=r0.get("LDS_Block_Size"), scratch=r0.get("Scratch_Size"), wg=r0.get("Workgroup_Size", r0.get("Workgroup_Size_X")), grid=r0.get("Grid_Size", r0.get("Grid_Size_X"))))
json.dump(summ, open(os.path.join(out, f"{tag}_kernel_trace_summary_c2.json"), "w"), indent=1)

# PMC passes
res = {}
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    path = one(f"{name}/**/*_counter_collection.csv")
    keep = [r for r in csv.DictReader(open(path)) if "dmel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    with open(os.path.join(out, f"{tag}_pmc_{name}_size_c2.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
        for r in keep:
            w.writerow([r["Dispatch_Id"], short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"], r["Counter_Value"]])
    for kern in ("dmel_fwd_kernel", "dmel_dot_kernel"):
        v = [float(r["Counter_Value"]) for r in keep if kern in r["Kernel_Name"]]
        res[f"{kern}_{counter}_KiB_raw"] = statistics.median(v)
        res[f"{kern}_{counter}_dispatches"] = len(v)
bench = json.loads(open(os.path.join(src, "bench_c2.json")).read().strip().splitlines()[-1])
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
dot_known = 2 * 4 * 256 * 128 * 32
factor = dot_known / (res["dmel_dot_kernel_FETCH_SIZE_KiB_raw"] * 1024)      # calibration on a kernel whose reads are known exactly
traffic = int(round(res["dmel_fwd_kernel_FETCH_SIZE_KiB_raw"] * 1024 * 2 + res["dmel_fwd_kernel_WRITE_SIZE_KiB_raw"] * 1024))
hbm = {
    "_how": "rocprofv3 --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE around `bench.py --steps 30 --warmup 5` (config 2); medians over the "
            "dispatches; raw per-dispatch values: profiles/%s_pmc_fetch_size_c2.csv, %s_pmc_write_size_c2.csv.  gfx950 correction "
            "(MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of streamed bytes -> doubled.  Calibrated in the same run on dmel_dot_kernel, which "
            "reads exactly 2 x 4 194 304 B with 16-byte loads: measured factor %.3f (2.0 expected).  WRITE_SIZE is taken as it is.  The working "
            "set fits the 256 MiB Infinity Cache, so these are fabric-side bytes, not necessarily DRAM bytes." % (tag, tag, factor),
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
bench["roofline"]["traffic"] = traffic
json.dump(bench, open(os.path.join(out, f"{tag}_bench_c2.json"), "w"), indent=1)
print(json.dumps(summ, indent=1))
print(json.dumps(hbm["c2"], indent=1))
print({k: bench[k] for k in ("value", "ms_per_step")}, bench["roofline"])
