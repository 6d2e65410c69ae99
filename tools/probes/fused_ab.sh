#!/bin/bash
# tools/probes/fused_ab.sh <rounds> <mode> lib1.so lib2.so ... : alternating runs of fused_time.py on several builds, one box
rounds=$1; mode=$2; shift; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    printf "%-28s " "$(basename $lib .so)"
    SALSA_HIP_LIB=$PWD/$lib python tools/probes/fused_time.py $mode 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
