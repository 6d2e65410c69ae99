#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV as a per-step timeline: tools/timeline.py <dir> [n_last_kernels]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
keep = [r for r in rows if any(k in r['Kernel_Name'] for k in ('stft_kernel', 'tracker_kernel', 'cov_eig_kernel'))]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
keep = keep[-n:]
t0 = int(keep[0]['Start_Timestamp'])
for r in keep:
    name = 'K1' if 'stft' in r['Kernel_Name'] else 'K2' if 'tracker' in r['Kernel_Name'] else 'K3'
    a, b = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print('%s  start %8.1f us  end %8.1f us  dur %7.1f us  grid %s' % (name, a, b, b - a, r.get('Grid_Size', '?')))
