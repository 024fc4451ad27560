#!/bin/bash
OUT=gpurun_out/r03c; mkdir -p $OUT
B=$PWD/differentiable-mel-spectrogram_amd/build
for tag in "$@"; do
  DMEL_LIB=$B/libdmel_hip_$tag.so timeout 300 python tools/check_variant.py g2_c2 g6_tone_dc 2>&1 | grep -v amdgpu | tail -4
done > $OUT/check.txt 2>&1
cat $OUT/check.txt
for i in 1 2 3; do
for tag in "$@"; do
  DMEL_LIB=$B/libdmel_hip_$tag.so timeout 120 python tools/ktime.py c2 train 2>&1 | tail -1
done; done > $OUT/ktime_train.txt 2>&1
cat $OUT/ktime_train.txt
for tag in "$@"; do
  DMEL_LIB=$B/libdmel_hip_$tag.so timeout 120 python tools/ktime.py c2 infer 2>&1 | tail -1
done > $OUT/ktime_infer.txt 2>&1
cat $OUT/ktime_infer.txt
