#!/bin/bash
# rocprofv3 kernel stats of the CRNN training step in steady state (GPU box).  usage: tools/prof_crnn.sh <tag>
# (Do NOT set MIOPEN_FIND_MODE=2 here: without a workspace MIOpen then falls back to CK kernels that are ~50x slower; the
# solver search of the first steps is in the trace instead -- read the per-step numbers from tools/torch_prof_crnn.py.)
TAG=${1:-crnn}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python -c "import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT'); import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
python $GRAFT_REPO_ROOT/bench_crnn.py --steps 10 --warmup 4 | tail -1 | cut -c1-160
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o crnn -- python $GRAFT_REPO_ROOT/bench_crnn.py --steps 10 --warmup 4 > $OUT/run.log 2>&1
f=$(find $OUT/p -name '*kernel_stats.csv' | head -1); cp $f $OUT/crnn_train_kernel_stats.csv
find $OUT/p -name '*kernel_trace.csv' -delete
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/crnn_train_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms (14 steps incl. warm-up):', tot/1e6)
for r in rows[:22]:
    n=r['Name']; n=n if len(n)<70 else n[:34]+'..'+n[-34:]
    print('%-72s calls %6s total %8.2f ms  %5.1f%%'%(n,r['Calls'],float(r['TotalDurationNs'])/1e6,float(r['Percentage'])))
PY
