#!/usr/bin/env python
"""Static instruction mix of the gfx950 kernels (hipcc -S): tools/asm_stats.py [kernel-name-substring]"""
import collections
import re
import subprocess
import sys
import tempfile
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(), 'k.s')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', out,
                       os.path.join(ROOT, 'salsa_amd', 'csrc', 'salsa_kernels.hip')], stderr=subprocess.DEVNULL)
lines = open(out).read().split('\n')
cur, funcs = None, collections.OrderedDict()
for l in lines:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1)
        funcs[cur] = []
    elif l.startswith('.Lfunc_end'):
        cur = None
    elif cur and l.startswith('\t') and not l.strip().startswith(('.', ';')) and l.strip():
        funcs[cur].append(l.strip().split()[0])
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for name, ins in funcs.items():
    if flt not in name:
        continue
    c = collections.Counter(ins)
    short = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)[:34]
    f64 = sum(v for k, v in c.items() if 'f64' in k)
    print('%-34s total %5d  f64 %5d  ds %4d  global %4d  scratch %3d  waitcnt %4d  div_scale %3d rcp64 %3d rsq64 %3d sqrt64 %3d'
          % (short, sum(c.values()), f64, sum(v for k, v in c.items() if k.startswith('ds_')),
             sum(v for k, v in c.items() if k.startswith('global_')), sum(v for k, v in c.items() if k.startswith('scratch_')),
             c.get('s_waitcnt', 0), c.get('v_div_scale_f64', 0), c.get('v_rcp_f64', 0), c.get('v_rsq_f64', 0), c.get('v_sqrt_f64', 0)))
    if len(sys.argv) > 2:
        for k, v in c.most_common(40):
            print('    %-28s %d' % (k, v))
