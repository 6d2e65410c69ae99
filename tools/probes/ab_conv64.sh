#!/bin/bash
# tools/probes/ab_conv64.sh "<-D flags>" ...: rebuild ONLY conv_mfma.hip per flag set (other objects prebuilt under salsa_amd/lib/obj), run the wrw probe
for FLAGS in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o /tmp/conv_mfma.o salsa_amd/csrc/conv_mfma.hip 2>/dev/null || { echo BUILD FAIL; exit 1; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o salsa_amd/lib/libsalsa_hip.so /tmp/conv_mfma.o salsa_amd/lib/obj/salsa_kernels.o salsa_amd/lib/obj/gru_scan.o salsa_amd/lib/obj/nn_ops.o salsa_amd/lib/obj/conv_wide.o salsa_amd/lib/obj/conv_1x1.o || { echo LINK FAIL; exit 1; }
  echo "== $FLAGS"; python tools/probes/${PROBE:-wrw64_probe.py} 2>&1 | grep -E "wrw64|conv64"
done
