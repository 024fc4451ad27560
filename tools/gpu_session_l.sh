#!/bin/bash
export TMPDIR=/tmp
for d in 0 1 2 4 8 32 16 63 7 15; do
export DMEL_XG_DBG=$d
rm -rf /tmp/xg_$d
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xg_$d -- python3 tools/time_xgrad.py c2 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/xg_$d/**/*kernel_stats.csv",recursive=True)[0]
out=["dbg $d:"]
for r in list(csv.DictReader(open(f)))[:3]: out.append("%s %.1f" % (r["Name"].split("(")[0][-28:], float(r["AverageNs"])/1000))
print(" | ".join(out))
PY
done
