#!/bin/bash
# Runs on the GPU box (via gpurun): build, GPU parity tests, smoke, bench, rocprofv3 kernel stats.
# Usage: tools/gpu_round.sh <tag> [bench args...]     outputs under gpurun_out/<tag>/
set -u
TAG=${1:-r}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -1 $OUT/bench.json
timeout 600 python bench.py "$@" --streams 2 --pcie --no-cpu-baseline > $OUT/bench_extra.json 2>> $OUT/bench.err; python -c "import json; d=json.load(open('$OUT/bench_extra.json')); print('pipelined', d.get('pipelined'), 'pcie', d.get('pcie_inclusive'))"
# kernel trace + stats of the same command (no cpu baseline inside the profiled run)
( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o salsa -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_run.log 2>&1 ); echo "rocprof rc=$?"
find $OUT/prof -type f | head -8
for f in $(find $OUT/prof -name '*kernel_stats.csv' | head -1); do head -12 $f; cp $f $OUT/kernel_stats.csv; done
# keep the merge small: drop the raw trace, keep stats
find $OUT/prof -name '*kernel_trace*' -size +8M -delete 2>/dev/null
du -sh $OUT
