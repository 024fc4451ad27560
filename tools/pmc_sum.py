#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per kernel name: tools/pmc_sum.py <dir> [kernel substring] -> JSON
(per-dispatch averages of every counter found)."""
import csv, glob, json, os, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "dmel_fwd_kernel"
acc, n = {}, {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if sub not in row.get("Kernel_Name", ""):
            continue
        c = row["Counter_Name"]
        acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
        n[c] = n.get(c, 0) + 1
print(json.dumps({"kernel": sub, "dispatches": max(n.values()) if n else 0, "per_dispatch": {c: acc[c] / n[c] for c in sorted(acc)}}, indent=1))
