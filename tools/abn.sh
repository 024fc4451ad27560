#!/bin/bash
# Alternating trains of several libraries on one box: tools/abn.sh "<tag> <tag> ..." "<config> ..." [rounds]
# (tags name differentiable-mel-spectrogram_amd/build/libdmel_hip_<tag>.so, made by tools/build_variant.sh; "main" = the in-tree library)
cd "$(dirname "$0")/.."
PKG=differentiable-mel-spectrogram_amd
for rep in $(seq 1 ${3:-2}); do
for c in $2; do
for t in $1; do
  if [ "$t" = main ]; then lib=$PKG/libdmel_hip.so; else lib=$PKG/build/libdmel_hip_$t.so; fi
  DMEL_LIB=$lib python tools/ktime.py $c ${MODE:-train} 200 2>&1 | tail -1
done; done; done
