#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/conv_pmc; mkdir -p $OUT; cd /tmp
python $GRAFT_REPO_ROOT/tools/probes/conv_probe.py 8
i=0
for CTRS in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
            "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES" \
            "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA" \
            "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS" \
            "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/probes/conv_probe.py 8 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list); dur={}
for f in sorted(glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        if 'conv3x3_c64_fwd' in row['Kernel_Name']:
            dur.setdefault(row['Counter_Name'], []).append((float(row.get('End_Timestamp', 0)) - float(row.get('Start_Timestamp', 0))))
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
for c,v in sorted(agg.items()): print('%-36s %.4g  (n=%d)  kernel ns %.0f' % (c, sum(v)/len(v), len(v), sum(dur[c])/len(dur[c])))
PY
find $OUT -name '*.csv' -size +2M -delete
