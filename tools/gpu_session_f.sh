#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT
B=$PWD/differentiable-mel-spectrogram_amd/build
for i in 1 2; do for tag in d256 d512 d1024; do DMEL_LIB=$B/libdmel_hip_$tag.so timeout 300 python tools/dtime.py c2 2>&1 | tail -1; done; done > $OUT/dot.txt 2>&1
cat $OUT/dot.txt
