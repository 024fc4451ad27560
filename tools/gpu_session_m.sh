#!/bin/bash
export TMPDIR=/tmp
python tools/time_fbgrad.py c2 c3 2>&1 | tail -1
for c in c2 c3; do
rm -rf /tmp/fb_$c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fb_$c -- python3 tools/time_fbgrad.py $c lin > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/fb_$c/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]: print("$c", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
rm -rf /tmp/fb_pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/fb_pmc -- python3 tools/time_fbgrad.py c2 lin > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/fb_pmc/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVE_CYCLES": cnt[k]+=1
for k in acc:
    if "fbgrad" in k: print(k, cnt[k], {c: round(v/max(1,cnt[k])) for c,v in acc[k].items()})
PY
