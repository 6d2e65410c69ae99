#!/bin/bash
# PMC counter passes on the GPU box (each pass its own rocprofv3 run, --kernel-trace only, per the microarch guide).
# Usage: tools/pmc_round.sh <tag> [bench args]
set -u
TAG=${1:-pmc}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python -c "import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT'); import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" \
            "WRITE_SIZE GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-crnn --blocks 1 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $CTRS"
done
python - <<PY
import csv, glob, collections, os
out='$OUT'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0].split('<')[0]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
with open(out+'/pmc_summary.csv','w') as fo:
    fo.write('kernel,counter,mean_per_dispatch,n\n')
    for k in agg:
        for c,v in sorted(agg[k].items()):
            fo.write('%s,%s,%.6g,%d\n'%(k,c,sum(v)/len(v),len(v)))
print(open(out+'/pmc_summary.csv').read())
PY
# calibration of FETCH_SIZE / WRITE_SIZE on known byte counts
for CTRS in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/calib_$CTRS -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_calib.py > $OUT/calib_$CTRS.log 2>&1
done
python - <<PY
import csv, glob, collections
out='$OUT'
agg=collections.defaultdict(list)
for f in sorted(glob.glob(out+'/calib_*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0].split('<')[0]
        agg[(k,row['Counter_Name'])].append(float(row['Counter_Value']))
with open(out+'/pmc_calibration.csv','w') as fo:
    fo.write('kernel,counter,mean_per_dispatch_KB,n\n')
    for (k,c),v in sorted(agg.items()):
        fo.write('%s,%s,%.6g,%d\n'%(k,c,sum(v)/len(v),len(v)))
print(open(out+'/pmc_calibration.csv').read())
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +4M -delete
du -sh $OUT
