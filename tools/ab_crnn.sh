#!/bin/bash
# A/B of two builds of libsalsa_hip.so on ONE box: alternating CRNN training benches (and optionally the inference leg).
#   tools/ab_crnn.sh libA.so libB.so [rounds] [extra bench_crnn.py flags]
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do
  for L in $A $B; do
    SALSA_HIP_LIB=$(realpath $L) python bench_crnn.py --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$L', l['value'], l['unit'], l['ms_per_step'], 'ms/step', l.get('final_loss'))"
  done
done
