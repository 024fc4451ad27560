#!/bin/bash
# A/B timing of diagnostic builds on one box: tools/ab.sh libA.so libB.so ...  (paths relative to the repo root)
for i in 1 2; do
for lib in "$@"; do
DMEL_LIB=$PWD/$lib python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-module-path 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1], round(d['value']/1e6,1), d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['other_kernels_us']['backward_dot'])"
done; done
