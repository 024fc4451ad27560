#!/bin/bash
# gpurun with retries while no GPU slot is free: tools/gpu.sh <timeout-seconds> '<command>'   (development helper, this container only)
T=$1; shift
for i in $(seq 1 30); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$T" -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out" | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname \|^Librccl path"
  exit 0
done
echo "no GPU slot after 30 tries"; exit 3
