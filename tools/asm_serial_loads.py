#!/usr/bin/env python
"""tools/asm_serial_loads.py [source ...]: static scan of the gfx950 ISA of the HIP sources for SERIALISED memory round trips -- a
global / scratch load whose `s_waitcnt vmcnt(0)` follows within a few instructions (the wave then sits out a whole memory latency
for one load), per kernel, with a flag when it happens inside a loop.  Round 4: this pattern (the compiler re-using an earlier
load behind a wait for `cond ? i : 0` indices; a folded-BatchNorm shift fetched from global memory in every tile's epilogue) was
what held the first-layer kernel at 2.5 TB/s."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or [os.path.join(ROOT, 'salsa_amd', 'csrc', f) for f in
                        ('salsa_kernels.hip', 'conv_mfma.hip', 'conv_wide.hip', 'conv_1x1.hip', 'nn_ops.hip', 'gru_scan.hip')]
for src in srcs:
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', out, src],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split('\n')
    cur, in_loop, body = None, False, []
    funcs = {}
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            in_loop = False
        elif l.startswith('.Lfunc_end'):
            cur = None
        elif cur is not None:
            if re.match(r'^\.LBB', l):
                in_loop = 'Loop' in l
            t = l.strip()
            if t and not t.startswith(('.', ';')):
                funcs[cur].append((t, in_loop))
    print('==', os.path.basename(src))
    for name, ins in funcs.items():
        hits, hits_loop = 0, 0
        for i, (t, lp) in enumerate(ins):
            if t.startswith(('global_load', 'scratch_load', 'buffer_load')) and 'lds' not in t:
                for j in range(i + 1, min(i + 7, len(ins))):
                    if ins[j][0].startswith(('global_load', 'scratch_load', 'global_store')):
                        break
                    if re.search(r'vmcnt\(0\)', ins[j][0]):
                        hits += 1
                        hits_loop += lp
                        break
        if hits:
            short = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            short = re.sub(r'\(anonymous namespace\)::', '', short).split('(')[0][:90]
            print('  %-92s serial loads %3d (in loops %3d)' % (short, hits, hits_loop))
