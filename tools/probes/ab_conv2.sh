#!/bin/bash
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $1 -o salsa_amd/lib/libsalsa_hip.so salsa_amd/csrc/salsa_kernels.hip salsa_amd/csrc/gru_scan.hip salsa_amd/csrc/nn_ops.hip salsa_amd/csrc/conv_mfma.hip salsa_amd/csrc/conv_wide.hip salsa_amd/csrc/conv_1x1.hip 2>/dev/null || { echo BUILD FAIL; exit 1; }
echo "== $1"; python tools/probes/conv_probe.py 8 | tail -1
