#!/bin/bash
export TMPDIR=/tmp
rm -rf /tmp/pm_big /tmp/pm_big2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pm_big -- python3 tools/time_full_window.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm_big2 -- python3 tools/time_full_window.py > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("/tmp/pm_big","/tmp/pm_big2"):
    f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    first=None
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]
        if "big_kernel<true>" not in r["Kernel_Name"]: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if first is None: first=r["Counter_Name"]
        if r["Counter_Name"]==first: cnt[k]+=1
    for k in acc: print(k, cnt[k], {c: round(v/cnt[k]) for c,v in acc[k].items()})
PY
