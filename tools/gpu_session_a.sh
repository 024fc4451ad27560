#!/bin/bash
# round-3 session A: n_fft 1024 plan A/B (R = 16 x C = 4 against R = 32 x 32, two frames per wave) + priority / early-twiddle variants
mkdir -p gpurun_out/r03a
OUT=gpurun_out/r03a
B=$PWD/differentiable-mel-spectrogram_amd/build
for tag in r16 r32 r32e r32p r32w; do
  DMEL_LIB=$B/libdmel_hip_$tag.so timeout 300 python tools/check_variant.py g2_c2 g6_pow2_512p9 g6_tone_dc 2>&1 | grep -v amdgpu
done > $OUT/check.txt 2>&1
cat $OUT/check.txt
for i in 1 2 3; do
for tag in r16 r32 r32e r32p r32w; do
  DMEL_LIB=$B/libdmel_hip_$tag.so timeout 120 python tools/ktime.py c2 train 2>&1 | tail -1
done; done > $OUT/ktime_train.txt 2>&1
cat $OUT/ktime_train.txt
for tag in r16 r32; do
  DMEL_LIB=$B/libdmel_hip_$tag.so timeout 120 python tools/ktime.py c2 infer 2>&1 | tail -1
done > $OUT/ktime_infer.txt 2>&1
cat $OUT/ktime_infer.txt
