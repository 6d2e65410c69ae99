# tools/probes/ab_lite.sh libA.so libB.so: bit-identity (lite_std_check.py) + alternating SALSA-Lite bench of two builds on one box
for L in "$@"; do echo "== $L"; SALSA_HIP_LIB=$(realpath $L) python tools/probes/lite_std_check.py 2>&1 | grep -v "NOT built"; done
for i in 1 2 3; do
for L in "$@"; do
  SALSA_HIP_LIB=$(realpath $L) python bench.py --feature salsa_lite --no-crnn --no-config4 --no-infer --no-cpu-baseline --blocks 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$L', 'step %.4f ms' % l['ms_per_step'], [(x['name'][:10], x['ms_per_launch'], x.get('ms_event_pair')) for x in l['roofline']['kernels']])"
done; done
