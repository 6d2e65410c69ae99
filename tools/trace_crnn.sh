#!/bin/bash
# tools/trace_crnn.sh <tag>: rocprofv3 kernel trace of the CRNN training bench, aggregated by (kernel, grid) -> per-shape durations
set -u
TAG=${1:-trace}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o crnn -- python $GRAFT_REPO_ROOT/bench_crnn.py --steps 10 --warmup 4 > $OUT/run.log 2>&1
echo "rocprof rc=$?"
python - <<PY
import csv, glob, collections
out='$OUT'
f=glob.glob(out+'/prof/**/*kernel_trace.csv', recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')
    name=name.split('(')[0][:70]
    agg[(name, r.get('Grid_Size_X','?'), r.get('Workgroup_Size_X','?'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
steps=14
rows=sorted(agg.items(), key=lambda kv:-sum(kv[1]))
with open(out+'/by_shape.txt','w') as fo:
    for (n,g,w),v in rows[:120]:
        fo.write('%-72s grid %9s wg %4s calls/step %5.1f  avg %8.1f us  ms/step %7.3f\n'%(n,g,w,len(v)/steps,sum(v)/len(v),sum(v)/steps/1e3))
print(open(out+'/by_shape.txt').read()[:9000])
PY
for f in $(find $OUT/prof -name '*kernel_stats.csv' | head -1); do cp $f $OUT/kernel_stats.csv; done
find $OUT/prof -name '*kernel_trace*' -delete; find $OUT/prof -name '*.db' -delete
