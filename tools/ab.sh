#!/bin/bash
# A/B kernel timing of library variants on one box, alternating: tools/ab.sh "c2 train" base trim1 ...
CFG=${1:-"c2 train"}; shift
for i in 1 2 3; do
for tag in "$@"; do
DMEL_LIB=$PWD/differentiable-mel-spectrogram_amd/build/libdmel_hip_$tag.so python tools/ktime.py $CFG 2>&1 | tail -1
done; done
