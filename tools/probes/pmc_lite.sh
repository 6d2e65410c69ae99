export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_lite_pmc; mkdir -p $OUT; cd /tmp
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((${i:-0}+1))
  timeout 200 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --feature salsa_lite --no-cpu-baseline --no-crnn --no-config4 --no-infer --blocks 1 --steps 6 --warmup 1 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    print('%s,%s,%.6g,%d' % (k[:60], c, sum(v) / len(v), len(v)))
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*.db' -delete
