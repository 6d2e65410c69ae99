#!/bin/bash
# tools/ab_multi.sh <rounds> lib1.so lib2.so ...: alternating feature-bench runs of several builds on ONE box
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    SALSA_HIP_LIB=$(realpath $L) python bench.py --no-crnn --no-config4 --no-infer --no-cpu-baseline --blocks 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k={x['name']:(x['ms_per_launch'], x.get('ms_event_pair')) for x in l['roofline']['kernels']}
print('$L', 'step %.4f ms |' % l['ms_per_step'], ' | '.join('%s %.4f / %.4f' % (n[:8], a, b) for n, (a, b) in k.items()))"
  done
done
