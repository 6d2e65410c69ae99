#!/bin/bash
# tools/pmc_sq2.sh <tag>: the SQ activity / wait pass of the feature bench alone with a SHORT run, retried (cut by its timeout at the default step
# count in round 6) -> gpurun_out/<tag>/pmc_summary_sq2.csv, to be appended to the round's pmc_summary.csv
set -u
TAG=${1:-pmcsq2}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for try in 1 2 3; do
  rm -rf $OUT/p2
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-crnn --no-config4 --no-infer --blocks 1 --steps 6 --warmup 1 > $OUT/p2.log 2>&1
  rc=$?; echo "try $try rc=$rc"; [ $rc = 0 ] && break
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/p2/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
with open('$OUT/pmc_summary_sq2.csv', 'w') as fo:
    for (k, c), v in sorted(agg.items()):
        fo.write('%s,%s,%.6g,%d\n' % (k, c, sum(v) / len(v), len(v)))
print(open('$OUT/pmc_summary_sq2.csv').read())
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +2M -delete; find $OUT -name '*.db' -delete
