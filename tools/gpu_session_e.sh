#!/bin/bash
OUT=gpurun_out/r03e; mkdir -p $OUT
B=$PWD/differentiable-mel-spectrogram_amd/build
for tag in w8 r16; do echo "== $tag"; DMEL_LIB=$B/libdmel_hip_$tag.so timeout 300 python tools/split_launch.py c2 2>&1 | grep -v amdgpu; done > $OUT/split.txt 2>&1
cat $OUT/split.txt
