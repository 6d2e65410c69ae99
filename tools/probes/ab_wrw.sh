#!/bin/bash
# probe builds of the wide weight-gradient kernel: tools/probes/ab_wrw.sh "<-D flags>"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $1 -o salsa_amd/lib/libsalsa_hip.so salsa_amd/csrc/salsa_kernels.hip salsa_amd/csrc/gru_scan.hip salsa_amd/csrc/nn_ops.hip salsa_amd/csrc/conv_mfma.hip salsa_amd/csrc/conv_wide.hip salsa_amd/csrc/conv_1x1.hip 2>/dev/null || { echo BUILD FAIL; exit 1; }
echo "== $1"; python tools/probes/conv_wide_probe.py 32 2>&1 | grep wrw | sed -n '2p;4p;6p'
