#!/bin/bash
# tools/probes/ab_env.sh VAR=VALUE [rounds]: same-box alternating A/B of the CRNN training step (bench.py --crnn-steps 60): the default
# environment against VAR=VALUE (a switch of salsa_amd/crnn/nn_ops.py, e.g. SALSA_STEM_BN_REDUCE_FUSED=0)
V=$1; R=${2:-3}; O=gpurun_out/ab_env; mkdir -p $O
ms() { python bench.py --crnn-steps 60 --no-config4 --no-infer 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['crnn']; print('$1', c['ms_per_step'], 'ms', c['value'], 'chunks/s')"; }
for i in $(seq $R); do ms default; env $V bash -c "$(declare -f ms); ms $V"; done > $O/ab.txt 2>&1
cat $O/ab.txt
