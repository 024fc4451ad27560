#!/bin/bash
export TMPDIR=/tmp
for c in c3 c2; do
rm -rf /tmp/pm_$c
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/pm_$c -- python3 tools/ktime.py $c train 30 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pm_$c/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:44]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
for k in acc:
    if "fwd" in k:
        n=cnt[k]; a={c: v/n for c,v in acc[k].items()}
        w=a["SQ_WAVES"]
        print("$c", k, n, "waves", w, {c: round(v/w,1) for c,v in a.items() if c!="SQ_WAVES"})
PY
done
