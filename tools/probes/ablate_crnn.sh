#!/bin/bash
# same-box ablation of the round-2 training-step changes: each switch off alone, then all off (chunks/s, ms/step, final loss)
run() { env "$@" python bench_crnn.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %7.1f chunks/s  %.3f ms  loss %.4f' % ('$*', d['value'], d['ms_per_step'], d['final_loss']))"; }
run A=1
run SALSA_FILTER_BANK=0
run SALSA_HIP_CONV_1X1=0
run SALSA_HIP_STEM_WRW=0
run SALSA_FUSED_SKIP=0
run SALSA_HIP_BN_POOL=0
run SALSA_CONV_STATS=0
run SALSA_STEM_FUSED_BWD=0
run SALSA_HIP_BN_RES_POOL=0
run SALSA_FILTER_BANK=0 SALSA_HIP_CONV_1X1=0 SALSA_HIP_STEM_WRW=0 SALSA_FUSED_SKIP=0 SALSA_HIP_BN_POOL=0 SALSA_CONV_STATS=0 SALSA_STEM_FUSED_BWD=0 SALSA_HIP_BN_RES_POOL=0
run SALSA_FILTER_BANK=0 SALSA_HIP_CONV_1X1=0 SALSA_HIP_STEM_WRW=0 SALSA_FUSED_SKIP=0 SALSA_HIP_BN_POOL=0 SALSA_CONV_STATS=0 SALSA_STEM_FUSED_BWD=0 SALSA_HIP_BN_RES_POOL=0 SALSA_HIP_CONV_WIDE_WRW=0
run A=1
