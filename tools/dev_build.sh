#!/bin/bash
# tools/dev_build.sh [EXTRA_FLAGS]: incremental build of salsa_amd/lib/libsalsa_hip.so for the edit-compile loop -- one object per
# .hip source under salsa_amd/lib/obj, only stale ones recompiled (in parallel), then one link.  The product's build is
# _lib.build_command() (one hipcc call, what build() and the judge run); this produces the same library faster.
# OUT=<path> selects another output library (A/B variants: SALSA_HIP_LIB=<path> python bench.py ...).
set -e
cd "$(dirname "$0")/.."
OBJ=${OBJDIR:-salsa_amd/lib/obj}; mkdir -p $OBJ
OUT=${OUT:-salsa_amd/lib/libsalsa_hip.so}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $*"
pids=()
for src in salsa_kernels gru_scan nn_ops conv_mfma conv_wide conv_1x1; do
  o=$OBJ/$src.o; s=salsa_amd/csrc/$src.hip
  if [ ! -f $o ] || [ $s -nt $o ] || [ salsa_amd/csrc/salsa_math.h -nt $o ] || [ -n "$(find include -newer $o -name '*.h')" ]; then
    ( hipcc $FL -c -o $o.tmp $s && mv $o.tmp $o ) & pids+=($!)
  fi
done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc = 0 ] || { echo "COMPILE FAILED"; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ/salsa_kernels.o $OBJ/gru_scan.o $OBJ/nn_ops.o $OBJ/conv_mfma.o $OBJ/conv_wide.o $OBJ/conv_1x1.o
echo "built $OUT"
