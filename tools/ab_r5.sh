#!/bin/bash
# A/B on one box: round-4 library against the current one (tools/ktime.py trains), alternating.  usage: tools/ab_r5.sh [configs...]
# (differentiable-mel-spectrogram_amd/build/libdmel_hip_r4.so = the library of the round-4 tree, kept by hand)
cd "$(dirname "$0")/.."
PKG=differentiable-mel-spectrogram_amd
CFGS=${@:-c2 c4 c3 c5 esc_n4096}
for rep in 1 2; do
for c in $CFGS; do
  DMEL_LIB=$PKG/build/libdmel_hip_r4.so python tools/ktime.py $c train 200 2>&1 | tail -1
  DMEL_LIB=$PKG/libdmel_hip.so python tools/ktime.py $c train 200 2>&1 | tail -1
done
done
