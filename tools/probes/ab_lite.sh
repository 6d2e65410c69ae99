for i in 1 2 3; do
for L in salsa_amd/lib/libsalsa_hip.so salsa_amd/lib/lite_w3.so; do
  SALSA_HIP_LIB=$(realpath $L) python bench.py --feature salsa_lite --no-crnn --no-config4 --no-infer --no-cpu-baseline --blocks 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$L', 'step %.4f ms' % l['ms_per_step'], [(x['name'][:10], x['ms_per_launch']) for x in l['roofline']['kernels']])"
done; done
