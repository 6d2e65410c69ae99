#!/bin/bash
# pmc_mini.sh <tag> <lib>: cycles (SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_WAIT_ANY) + kernel-trace durations of the c64 kernels
TAG=$1; LIB=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
SALSA_HIP_LIB=$LIB timeout 240 rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p -o pmc -- python tools/probes/c64_one.py 32 640 200 4 > $OUT/p.log 2>&1
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob('$OUT/p/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'c64' in row['Kernel_Name']:
            agg[(row['Kernel_Name'].split('(')[0][-50:], row['Counter_Name'])].append(float(row['Counter_Value']))
for (k,c),v in sorted(agg.items()): print('$TAG %-50s %-28s %.5g (n=%d)'%(k,c,sum(v)/len(v),len(v)))
dur=collections.defaultdict(list)
for f in glob.glob('$OUT/p/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'c64' in row['Kernel_Name']:
            dur[row['Kernel_Name'].split('(')[0][-50:]].append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e3)
for k,v in sorted(dur.items()): print('$TAG %-50s duration us %.1f (n=%d)'%(k,sum(v)/len(v),len(v)))
PY
