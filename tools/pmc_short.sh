#!/bin/bash
# tools/pmc_short.sh <tag>: the four counter passes of the feature bench alone behind traffic.json (instruction counts, SQ activity / waits, FETCH_SIZE,
# WRITE_SIZE), each its own rocprofv3 run (--pmc + --kernel-trace only) on a SHORT bench (6 steps: the long runs of pmc_k3.sh were cut by their timeouts
# in round 6) and retried -> gpurun_out/<tag>/pmc_summary.csv in the format tools/pmc_traffic.py reads
set -u
TAG=${1:-pmcshort}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  for try in 1 2 3; do
    rm -rf $OUT/p$i
    timeout 240 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-crnn --no-config4 --no-infer --blocks 1 --steps 6 --warmup 1 > $OUT/p$i.log 2>&1
    rc=$?; echo "pass $i try $try rc=$rc"; [ $rc = 0 ] && break
  done
done
python - <<PY
import csv, glob, collections
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
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*.db' -delete
