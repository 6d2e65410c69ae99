#!/bin/bash
# tools/pmc_tcc.sh <tag>: the FETCH_SIZE and WRITE_SIZE passes of the feature bench alone with a SHORT run (the TCC passes serialise
# every dispatch and have been cut by their timeout at the default step count), each retried -> gpurun_out/<tag>/pmc_summary_tcc.csv
set -u
TAG=${1:-pmctcc}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for try in 1 2 3; do
    rm -rf $OUT/$C
    timeout 300 rocprofv3 --pmc $C GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$C -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-crnn --no-config4 --no-infer --blocks 1 --steps 4 --warmup 1 > $OUT/$C.log 2>&1
    rc=$?; echo "$C try $try rc=$rc"; [ $rc = 0 ] && break
  done
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
with open('$OUT/pmc_summary_tcc.csv', 'w') as fo:
    for (k, c), v in sorted(agg.items()):
        if c != 'GRBM_GUI_ACTIVE': fo.write('%s,%s,%.6g,%d\n' % (k, c, sum(v) / len(v), len(v)))
print(open('$OUT/pmc_summary_tcc.csv').read())
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*.db' -delete
