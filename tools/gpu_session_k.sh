#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r03k
python tools/time_xgrad.py c2 c3 c5 2>&1 | tail -1
for c in c2 c3; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03k/$c -- python3 tools/time_xgrad.py $c > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r03k/$c/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]: print("$c", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
