#!/usr/bin/env python
"""tools/k1_isa_audit.py [template-argument substring]: ISA category histogram of the STFT kernel, block by block (round-5 review item 3:
"~300 of K1's ~633 instructions per 512-point item are NOT float64 math").  Compiles salsa_kernels.hip to ISA, takes the
instantiation whose demangled name contains the argument (default: the production one, `stft_kernel<512, double, false, 4, 2, false>`),
splits it into basic blocks and counts instruction categories per block; blocks inside the item loop are marked.
  python tools/k1_isa_audit.py > profiles/r6_k1_isa.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CATS = ['f64', 'cvt', 'f32', 'int_valu', 'cmp_sel', 'mov', 'xlane', 'lds', 'global', 'salu', 'wait_nop', 'branch', 'other']


def category(op):
    if op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_sleep')):
        return 'wait_nop'
    if op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc', 's_swappc')):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_bpermute') or op.startswith('ds_permute') or op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane', 'v_permlane')) or '_dpp' in op:
        return 'xlane'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'global'
    if op.startswith('v_cvt'):
        return 'cvt'
    if 'f64' in op:
        return 'f64'
    if op.startswith(('v_cmp', 'v_cndmask')):
        return 'cmp_sel'
    if op.startswith(('v_mov', 'v_accvgpr')):
        return 'mov'
    if re.search(r'_f32|_f16|v_log|v_exp|v_rcp|v_rsq|v_sqrt|v_pk_', op):
        return 'f32'
    if op.startswith('v_'):
        return 'int_valu'
    return 'other'


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else 'stft_kernel<512, double, false, 4, 2, false>'
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', out,
                           os.path.join(ROOT, 'salsa_amd', 'csrc', 'salsa_kernels.hip')], stderr=subprocess.DEVNULL)
    lines = open(out).read().split('\n')
    target = None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            d = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '')
            if want in d:
                target, name = i, d.split('(')[0]
                break
    assert target is not None, want
    end = next(i for i in range(target, len(lines)) if lines[i].startswith('.Lfunc_end'))
    blocks, cur = [], ['entry', False, []]
    depth_note = {}
    for l in lines[target + 1:end]:
        m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), False, []]
            continue
        t = l.strip()
        if t.startswith('; %bb') or t.startswith('; =>') or 'Loop Header' in t or 'Inner Loop' in t or 'in Loop' in t:
            depth_note[cur[0]] = depth_note.get(cur[0], '') + ' ' + t.lstrip('; ')
        if t and not t.startswith(('.', ';')):
            cur[2].append(t)
    blocks.append(cur)
    print('%s: %d instructions in %d basic blocks' % (name, sum(len(b[2]) for b in blocks), len(blocks)))
    print('%-14s %5s | %s | notes' % ('block', 'total', ' '.join('%8s' % c for c in CATS)))
    tot_loop = collections.Counter()
    for label, _, ins in blocks:
        c = collections.Counter(category(x.split()[0]) for x in ins)
        note = depth_note.get(label, '')
        in_loop = 'Loop' in note or 'loop' in note
        if in_loop:
            tot_loop.update(c)
        tail = [x.split()[-1] for x in ins if x.startswith(('s_cbranch', 's_branch'))]
        print('%-14s %5d | %s | %s%s' % (label, len(ins), ' '.join('%8d' % c.get(k, 0) for k in CATS), note.strip()[:70], (' -> ' + ','.join(tail)) if tail else ''))
    print('%-14s %5d | %s | every block the compiler marks as inside a loop' % ('LOOP BLOCKS', sum(tot_loop.values()), ' '.join('%8d' % tot_loop.get(k, 0) for k in CATS)))
    # the most frequent opcodes of the loop blocks
    ops = collections.Counter()
    for label, _, ins in blocks:
        if 'oop' in depth_note.get(label, ''):
            ops.update(x.split()[0] for x in ins)
    print('\nopcodes of the loop blocks:')
    for k, v in ops.most_common(60):
        print('    %-28s %4d  %s' % (k, v, category(k)))


if __name__ == '__main__':
    main()
