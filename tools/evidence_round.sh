#!/bin/bash
# tools/evidence_round.sh <tag>: the round's evidence at HEAD, unmixed (round-3 review item 5):
#  (1) rocprofv3 --kernel-trace --stats of the FEATURE bench alone (config 2 launches only: no CRNN / config 4 / inference legs)
#  (2) the PMC passes of the same command (tools/pmc_k3.sh) -> pmc_summary.csv -> (build container) tools/pmc_traffic.py -> traffic.json
#  (3) per-shape kernel trace + stats of the CRNN training bench (tools/trace_crnn.sh)
set -u
TAG=${1:-ev}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_feature -o feat -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-crnn --no-config4 --no-infer > $OUT/feature_bench.json 2> $OUT/feature_bench.err ); echo "feature rocprof rc=$?"
for f in $(find $OUT/prof_feature -name '*kernel_stats.csv' | head -1); do cp $f $OUT/feature_kernel_stats.csv; head -6 $f; done
find $OUT/prof_feature -name '*kernel_trace*' -delete; find $OUT/prof_feature -name '*.db' -delete
$GRAFT_REPO_ROOT/tools/pmc_k3.sh $TAG/pmc | tail -3
$GRAFT_REPO_ROOT/tools/trace_crnn.sh $TAG/crnn | head -40
