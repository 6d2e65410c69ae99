#!/bin/bash
# rocprofv3 kernel stats of the CRNN benches (GPU box).  usage: tools/prof_crnn.sh <tag>
TAG=${1:-crnn}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python -c "import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT'); import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for MODE in "--infer --steps 3 --warmup 1 --clips 16" "--steps 6 --warmup 3"; do
  N=$(echo $MODE | cut -c3-7 | tr -d ' -')
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$N -o crnn -- python $GRAFT_REPO_ROOT/bench_crnn.py $MODE > $OUT/run_$N.log 2>&1
  tail -1 $OUT/run_$N.log | cut -c1-200
  f=$(find $OUT/p_$N -name '*kernel_stats.csv' | head -1); cp $f $OUT/crnn_${N}_kernel_stats.csv; head -16 $f | cut -c1-230
  find $OUT/p_$N -name '*kernel_trace.csv' -delete
done
