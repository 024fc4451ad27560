#!/usr/bin/env python3
"""Per-kernel average durations from a rocprofv3 --kernel-trace output directory: tools/kstats.py <dir> [min calls]"""
import csv, glob, os, statistics, sys
d = sys.argv[1]
mincalls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = {}
for r in rows:
    key = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:90], r.get("Grid_Size", r.get("Grid_Size_X")))
    by.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= mincalls:
        print(f"{statistics.mean(v)/1e3:9.2f} us avg {statistics.median(v)/1e3:9.2f} med  x{len(v):5d}  grid {g:>8s}  {k}")
