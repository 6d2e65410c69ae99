#!/usr/bin/env python
"""tools/asm_vmcnt0.py [source ...]: every `s_waitcnt vmcnt(0)` (alone or combined) that sits INSIDE A LOOP of a kernel, with the
instruction before it that touches memory -- in the software-pipelined kernels (LDS-direct loads two tiles ahead, counted waits) a
full drain inside the tile loop means the pipeline is being emptied once per iteration (round 5: a spilled address reloaded from
scratch in front of a global_load_lds cost the 64 -> 64 statistics kernel 70 us of 440)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or [os.path.join(ROOT, 'salsa_amd', 'csrc', f) for f in ('conv_mfma.hip', 'conv_wide.hip', 'conv_1x1.hip', 'salsa_kernels.hip')]
for src in srcs:
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', out, src], stderr=subprocess.DEVNULL)
    cur, in_loop, hits, last_mem = None, False, {}, None
    for l in open(out):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur, in_loop, last_mem = m.group(1), False, None
            hits[cur] = []
            continue
        if l.startswith('.Lfunc_end'):
            cur = None
        if cur is None:
            continue
        if re.match(r'^\.LBB', l):
            in_loop = 'Loop' in l
        t = l.strip()
        if t.startswith(('global_', 'scratch_', 'buffer_', 'flat_')):
            last_mem = t.split()[0] + (' (spill reload)' if 'Reload' in t else ' (spill)' if 'Spill' in t else '')
        if in_loop and re.search(r's_waitcnt.*vmcnt\(0\)', t):
            hits[cur].append(last_mem)
    print('==', os.path.basename(src))
    for name, h in hits.items():
        if h:
            short = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            short = re.sub(r'\(anonymous namespace\)::', '', short).split('(')[0][:70]
            import collections
            c = collections.Counter(h)
            print('  %-72s vmcnt(0) in loops: %2d  after: %s' % (short, len(h), ', '.join('%s x%d' % kv for kv in c.most_common(4))))
