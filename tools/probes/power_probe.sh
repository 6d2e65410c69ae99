#!/bin/bash
# tools/probes/power_probe.sh: package power and clocks (rocm-smi, sampled every ~0.2 s) while (a) the 64 -> 64 forward kernel with real loads, (b) the same
# kernel from a zero-source probe build if SALSA_ZERO_LIB is set, (c) the feature bench run in a loop -- is the part at its power cap when compute and memory "add"?
sample() { for i in $(seq $1); do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" | tr -s ' ' | tr '\n' '|'; echo; sleep 0.2; done; }
cat > /tmp/c64_loop.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = torch.device('cuda:0'); L = _lib.load(); g = torch.Generator(device=dev).manual_seed(0)
N, H, W = 32, 640, 200
xs = [torch.randn((N, 64, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(2)]
w = (torch.randn((64, 64, 3, 3), device=dev, generator=g) * 0.06).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[1]):
    for i in range(50): nn_ops._conv64(xs[i & 1], w)
    torch.cuda.synchronize(); n += 50
print('c64 fwd launches', n, 'avg us', (time.time() - t0) / n * 1e6)
PY
echo "== idle"; sample 3
echo "== c64 forward, real loads"; python /tmp/c64_loop.py 6 & sleep 2.5; sample 12; wait
if [ -n "${SALSA_ZERO_LIB:-}" ]; then echo "== c64 forward, zero-source probe build"; SALSA_HIP_LIB=$SALSA_ZERO_LIB python /tmp/c64_loop.py 6 & sleep 2.5; sample 12; wait; fi
echo "== feature bench in a loop"; (for i in 1 2 3; do python bench.py --no-cpu-baseline --no-crnn --no-config4 --no-infer --blocks 20 > /dev/null 2>&1; done) & sleep 4; sample 12; wait
echo "== CRNN training bench"; python bench.py --no-cpu-baseline --no-config4 --no-infer --crnn-steps 400 --blocks 1 > /dev/null 2>&1 & sleep 8; sample 12; wait
