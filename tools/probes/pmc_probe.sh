#!/bin/bash
# tools/probes/pmc_probe.sh <tag> <kernel-name-substring> <command...>: SQ / TCC counter passes of any command, summarised for one kernel
set -u
TAG=$1; KSUB=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd $GRAFT_REPO_ROOT && timeout 240 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- "$@" > $OUT/p$i.log 2>&1 ); echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in sorted(glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        if '$KSUB' in row['Kernel_Name']:
            agg[(row['Kernel_Name'].split('(')[0][-60:], row['Grid_Size'] if 'Grid_Size' in row else '', row['Counter_Name'])].append(float(row['Counter_Value']))
for (k,g,c),v in sorted(agg.items()): print('%-50s grid %-9s %-32s %.6g (n=%d)'%(k,g,c,sum(v)/len(v),len(v)))
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
