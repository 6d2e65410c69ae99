#!/bin/bash
# A/B helper (GPU box): tools/ab_build.sh "<hipcc -D flags>" [bench args]  -> rebuilds the library with the flags, runs bench
FLAGS="$1"; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $FLAGS -o salsa_amd/lib/libsalsa_hip.so salsa_amd/csrc/salsa_kernels.hip salsa_amd/csrc/gru_scan.hip salsa_amd/csrc/nn_ops.hip salsa_amd/csrc/conv_mfma.hip salsa_amd/csrc/conv_wide.hip salsa_amd/csrc/conv_1x1.hip 2>/dev/null || { echo BUILD FAIL; exit 1; }
python bench.py --no-cpu-baseline --no-crnn --blocks 3 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$FLAGS', '| ms/step', d['ms_per_step'], [(k['name'][:8],k['ms_per_launch']) for k in d['roofline']['kernels']])"
