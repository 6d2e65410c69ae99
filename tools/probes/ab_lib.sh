#!/bin/bash
# tools/probes/ab_lib.sh <variant.so> [rounds]: same-box alternating A/B of the CRNN training step (bench.py --crnn-steps 60), the
# in-tree library against SALSA_HIP_LIB=<variant.so>; optional kernel probe first (PROBE="python tools/probes/x.py").
V=$1; R=${2:-3}; O=gpurun_out/ab_lib; mkdir -p $O
ms() { python bench.py --crnn-steps 60 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['crnn']; print('$1', c['ms_per_step'], 'ms', c['value'], 'chunks/s')"; }
if [ -n "$PROBE" ]; then echo "== in-tree"; $PROBE; echo "== $V"; SALSA_HIP_LIB=$V $PROBE; fi > $O/probe.txt 2>&1
for i in $(seq $R); do ms in-tree; SALSA_HIP_LIB=$V ms variant; done > $O/ab.txt 2>&1
cat $O/probe.txt $O/ab.txt
