#!/bin/bash
OUT=gpurun_out/r03h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_f2 -- python3 tools/time_backward_extras.py > $OUT/kt_f2.log 2>&1
f=$(find $OUT/kt_f2 -name "*kernel_stats.csv" | head -1); echo $f; head -20 $f
cp $f $OUT/f2_kernel_stats.csv; rm -rf $OUT/kt_f2
