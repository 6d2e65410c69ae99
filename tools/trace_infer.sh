#!/bin/bash
# tools/trace_infer.sh <tag> [clips]: rocprofv3 kernel trace of the batched-inference bench (config 5), aggregated by (kernel, grid)
set -u
TAG=${1:-trace_infer}; CLIPS=${2:-128}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o infer -- python $GRAFT_REPO_ROOT/bench_crnn.py --infer --clips $CLIPS --steps 3 --warmup 1 > $OUT/run.log 2>&1
echo "rocprof rc=$?"; tail -1 $OUT/run.log | cut -c1-300
python - <<PY
import csv, glob, collections
out='$OUT'; clips=$CLIPS
f=glob.glob(out+'/prof/**/*kernel_trace.csv', recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')
    name=name.split('(')[0][:70]
    agg[(name, r.get('Grid_Size_X','?'), r.get('Workgroup_Size_X','?'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
subs=(3+1+1)*clips/32.0      # timed steps + warm-up + the device-only pass, in 32-clip sub-batches (approximate divisor)
rows=sorted(agg.items(), key=lambda kv:-sum(kv[1]))
tot=sum(sum(v) for v in agg.values())
with open(out+'/by_shape.txt','w') as fo:
    fo.write('total kernel time %.1f ms over ~%.0f sub-batches of 32 clips = %.3f ms per sub-batch\n'%(tot/1e3, subs, tot/1e3/subs))
    for (n,g,w),v in rows[:80]:
        fo.write('%-72s grid %9s wg %4s calls %6d  avg %8.1f us  ms/sub-batch %7.3f  %4.1f%%\n'%(n,g,w,len(v),sum(v)/len(v),sum(v)/subs/1e3,100*sum(v)/tot))
print(open(out+'/by_shape.txt').read()[:7000])
PY
find $OUT/prof -name '*kernel_trace*' -delete; find $OUT/prof -name '*.db' -delete
