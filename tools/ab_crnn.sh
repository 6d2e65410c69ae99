#!/bin/bash
# tools/ab_crnn.sh <rounds> lib1.so lib2.so ...: alternating CRNN training-step benches (config 3) of several builds on ONE box
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    SALSA_HIP_LIB=$(realpath $L) python bench_crnn.py --steps 20 --warmup 6 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$L', '%.1f chunks/s  %.3f ms/step' % (l['value'], l['ms_per_step']))"
  done
done
