#!/usr/bin/env python
"""Busy time against wall time of the CRNN training step from a rocprofv3 --kernel-trace CSV: how much of a step the GPU
idles between kernels (launch gaps).  tools/crnn_gaps.py <dir> [steps]  -- looks at the trace's last `steps` optimizer steps
(the fused-Adam multi_tensor_apply kernel group marks a step's end)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# step boundaries: the first kernel after a run of multi_tensor_apply kernels
bounds, in_opt = [], False
for i, (a, b, n) in enumerate(rows):
    opt = 'multi_tensor_apply' in n
    if in_opt and not opt:
        bounds.append(i)
    in_opt = opt
bounds = bounds[-(steps + 1):]
sel = rows[bounds[0]:bounds[-1]]
span = (sel[-1][1] - sel[0][0]) / 1e6
busy, cur_end, gaps = 0, sel[0][0], []
for a, b, n in sel:
    if a > cur_end:
        gaps.append(((a - cur_end) / 1e3, n))
    busy += max(0, b - max(a, cur_end))
    cur_end = max(cur_end, b)
ns = len(bounds) - 1
print('steps %d  kernels/step %.0f  wall %.3f ms/step  busy %.3f ms/step  idle %.3f ms/step (%.1f %%)' %
      (ns, len(sel) / ns, span / ns, busy / 1e6 / ns, (span - busy / 1e6) / ns, 100 * (1 - busy / 1e6 / span)))
gaps.sort(reverse=True)
print('largest gaps (us, kernel that followed):')
for g, n in gaps[:12]:
    print('  %8.1f  %s' % (g, n[:90]))
import collections
h = collections.Counter(min(int(g), 20) for g, _ in gaps)
print('gap histogram (us -> count/step):', {k: round(v / ns, 1) for k, v in sorted(h.items())})
