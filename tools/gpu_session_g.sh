#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
A="--steps 40 --warmup 8 --no-cpu-baseline --no-other-configs"
timeout 300 python bench.py $A > $OUT/bench_1rank.json 2> $OUT/bench_1rank.err
DMEL_BENCH_FORCE_DIST=1 timeout 300 python bench.py $A --reducer rccl > $OUT/bench_1rank_rccl.json 2> $OUT/bench_1rank_rccl.err
DMEL_BENCH_FORCE_DIST=1 timeout 300 python bench.py $A --reducer mailbox > $OUT/bench_1rank_mailbox.json 2> $OUT/bench_1rank_mailbox.err
DMEL_BENCH_SHARE_GPU=1 timeout 600 python bench.py $A --gpus 2 --reducer mailbox > $OUT/bench_2ranks_sharedgpu_mailbox.json 2> $OUT/bench_2ranks_sharedgpu_mailbox.err
for f in $OUT/bench_*.json; do echo == $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config'].get('reducer'), d['module_step']['trial_ms_per_step'], d['module_step'].get('graph_captures'))
except Exception as e: print('ERR', e)
"; done
tail -3 $OUT/*.err
