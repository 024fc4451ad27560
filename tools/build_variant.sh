#!/bin/bash
# Compiles the CURRENT csrc/ into differentiable-mel-spectrogram_amd/build/libdmel_hip_<tag>.so (extra flags after the tag),
# for A/B timing on one box with tools/ktime.py (DMEL_LIB selects the library).
set -e
TAG=$1; shift
PKG=$(dirname $0)/../differentiable-mel-spectrogram_amd
mkdir -p $PKG/build
OBJS=""
for f in dmel_fwd.hip dmel_aux.hip dmel_big.hip dmel_xgrad.hip dmel_api.cpp dmel_comm.cpp; do
  o=$PKG/build/${f%.*}_$TAG.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function "$@" -x hip -c $PKG/csrc/$f -o $o &
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/build/libdmel_hip_$TAG.so $OBJS -ldl
echo $PKG/build/libdmel_hip_$TAG.so
