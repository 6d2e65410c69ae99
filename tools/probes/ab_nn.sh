#!/bin/bash
# tools/probes/ab_nn.sh "<-D flags>" ...: for each flag set rebuild ONLY nn_ops.hip (the other objects are prebuilt under
# salsa_amd/lib/obj: `hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c`), link, run the CRNN training bench
for FLAGS in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o /tmp/nn_ops.o salsa_amd/csrc/nn_ops.hip 2>/dev/null || { echo BUILD FAIL; exit 1; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o salsa_amd/lib/libsalsa_hip.so /tmp/nn_ops.o $(ls salsa_amd/lib/obj/*.o | grep -v nn_ops.o) || { echo LINK FAIL; exit 1; }
  python bench_crnn.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$FLAGS', '|', d['value'], 'chunks/s', d['ms_per_step'], 'ms')"
done
