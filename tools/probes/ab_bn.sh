#!/bin/bash
# tools/probes/ab_bn.sh "<flags>": rebuild with flags and run the BN probe
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $1 -o salsa_amd/lib/libsalsa_hip.so salsa_amd/csrc/salsa_kernels.hip salsa_amd/csrc/gru_scan.hip salsa_amd/csrc/nn_ops.hip salsa_amd/csrc/conv_mfma.hip salsa_amd/csrc/conv_wide.hip salsa_amd/csrc/conv_1x1.hip 2>/dev/null || { echo BUILD FAIL; exit 1; }
echo "== $1"; python tools/probes/bn_probe.py 2>&1 | grep -v Warn | grep -A4 "640, 200" | tail -4
