#!/bin/bash
# tools/probes/pmc_mem.sh <tag> <kernel substring> <command...>: memory-path counters (TLB, TCP stalls, request latency, TA, L2 stalls,
# vector-memory instructions in flight) of one kernel of any command, one rocprofv3 --pmc pass per group
set -u
TAG=$1; KSUB=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for CTRS in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
            "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum" \
            "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_LATENCY_FIFO_FULL_sum" \
            "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd $GRAFT_REPO_ROOT && timeout 240 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- "$@" > $OUT/p$i.log 2>&1 ); echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in sorted(glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        if '$KSUB' in row['Kernel_Name']:
            agg[(row['Grid_Size'] if 'Grid_Size' in row else '', row['Counter_Name'])].append(float(row['Counter_Value']))
for (g,c),v in sorted(agg.items()): print('$TAG grid %-9s %-42s %.6g (n=%d)'%(g,c,sum(v)/len(v),len(v)))
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
