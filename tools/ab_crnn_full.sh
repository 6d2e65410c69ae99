#!/bin/bash
# tools/ab_crnn_full.sh <rounds> lib1.so lib2.so ...: alternating CRNN training legs of several builds on ONE box THROUGH bench.py
# (the feature leg runs first, so the part is at its loaded clocks: bench_crnn.py alone from a cold process reads ~35 % slower)
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    SALSA_HIP_LIB=$(realpath $L) python bench.py --crnn-steps 60 --no-config4 --no-infer --no-cpu-baseline --blocks 1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['crnn']; print('$L', c['ms_per_step'], 'ms', c['value'], 'chunks/s')"
  done
done
