#!/bin/bash
# tools/probes/ab_conv.sh "<flags>": rebuild with flags, conv tests, conv64 kernel times in the inference / training profiles
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $1 -o salsa_amd/lib/libsalsa_hip.so salsa_amd/csrc/salsa_kernels.hip salsa_amd/csrc/gru_scan.hip salsa_amd/csrc/nn_ops.hip salsa_amd/csrc/conv_mfma.hip salsa_amd/csrc/conv_wide.hip salsa_amd/csrc/conv_1x1.hip 2>/dev/null || { echo BUILD FAIL; exit 1; }
echo "== $1"; python -m pytest tests/test_crnn_gpu.py -q -x -k "conv" 2>&1 | tail -1
python tools/torch_prof_infer.py 32 2>&1 | grep "total device\|conv3x3_c64"
python tools/torch_prof_crnn.py 2>&1 | grep "total device\|conv3x3_c64_fwd"
