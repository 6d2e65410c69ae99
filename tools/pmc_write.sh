#!/bin/bash
# tools/pmc_write.sh <tag>: the WRITE_SIZE counter pass of the feature bench alone, retried (this pass has hung under rocprofv3 in
# several rounds and is cut by its timeout) -> gpurun_out/<tag>/pmc_summary_write.csv, to be appended to the round's pmc_summary.csv
set -u
TAG=${1:-pmcw}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for try in 1 2 3; do
  rm -rf $OUT/p4
  timeout 200 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p4 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-crnn --no-config4 --no-infer --blocks 1 --steps 10 > $OUT/p4.log 2>&1
  rc=$?; echo "try $try rc=$rc"; [ $rc = 0 ] && break
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/p4/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
with open('$OUT/pmc_summary_write.csv', 'w') as fo:
    for (k, c), v in sorted(agg.items()):
        fo.write('%s,%s,%.6g,%d\n' % (k, c, sum(v) / len(v), len(v)))
print(open('$OUT/pmc_summary_write.csv').read())
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*.db' -delete
