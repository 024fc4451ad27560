#!/bin/bash
OUT=gpurun_out/r03d; mkdir -p $OUT
B=$PWD/differentiable-mel-spectrogram_amd/build
DMEL_LIB=$B/libdmel_hip_abl8.so timeout 300 python tools/ablate.py c2 2>&1 | grep -v amdgpu > $OUT/ablate_w8.txt
cat $OUT/ablate_w8.txt
for tag in w8 w16n; do DMEL_LIB=$B/libdmel_hip_$tag.so timeout 120 python tools/ktime.py c2 train 2>&1 | tail -1; done
