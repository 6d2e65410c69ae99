#!/bin/bash
# A/B of two builds of libsalsa_hip.so on ONE box: alternating short bench.py runs (feature path only), step + per-kernel ms.
#   tools/ab_feature.sh salsa_amd/lib/libsalsa_hip.so salsa_amd/lib/libsalsa_hip_variant.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    SALSA_HIP_LIB=$(realpath $L) python bench.py --no-crnn --no-cpu-baseline --blocks 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k={x['name']:(x['ms_per_launch'], x.get('ms_event_pair')) for x in l['roofline']['kernels']}
print('$L', 'step %.4f ms |' % l['ms_per_step'], ' | '.join('%s prefix %.4f pair %.4f' % (n, a, b) for n, (a, b) in k.items()))"
  done
done
