#!/bin/bash
# two counter passes of the 64 -> 64 forward kernel (conv_probe.py 8 = 8 x 2400 x 100): matrix-pipe busy cycles and the clocks
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/conv_pmc2; mkdir -p $OUT; cd /tmp
python $GRAFT_REPO_ROOT/tools/probes/conv_probe.py 8 2>/dev/null
i=0
for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
            "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/probes/conv_probe.py 8 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in sorted(glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        if 'conv3x3_c64_fwd' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
m={c: sum(v)/len(v) for c,v in agg.items()}
for c,v in sorted(m.items()): print('%-34s %.4g' % (c, v))
simd_cycles = m['GRBM_GUI_ACTIVE'] / 8 * 1024      # per-XCD active cycles x 1024 SIMDs
print('SIMD cycles %.4g   matrix-pipe utilisation %.1f %%   MFMAs per launch %.4g' % (simd_cycles, 100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, m['SQ_INSTS_MFMA']))
PY
find $OUT -name '*.csv' -size +2M -delete
