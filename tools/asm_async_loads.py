#!/usr/bin/env python
"""tools/asm_async_loads.py [source.hip ...]: the hand-issued asynchronous loads of the convolution kernels, checked in the ISA.

conv_mfma.hip requests a tile's residual pieces with bare `asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) ...)` so that the
compiler's own wait insertion does not drain the LDS-direct prefetch behind them (CONV_RES_EARLY; a hand-written counted
`s_waitcnt vmcnt(n)` after the multiply covers them).  The compiler does not know those destination registers are in flight: a copy,
a live-range split or a spill it places between the load and that wait would capture unfinished data (round-5 advisor finding).
This tool finds every inline-asm global load with a register destination in the compiled listing and walks forward to the first
instruction that touches the destination registers: an `s_waitcnt vmcnt(..)` must sit in between.  Exit code 1 and a listing if not.
tests/test_abi_cpu.py runs it on every build."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(text):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r'\bv(\d+)\b', text))
    return out


def check(src):
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', out, src], stderr=subprocess.DEVNULL)
    return check_lines(open(out).read().split('\n'))


def check_lines(lines):
    func, in_asm, checked, bad = None, False, 0, []
    for i, ln in enumerate(lines):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            func = m.group(1)
        t = ln.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        m = re.match(r'global_load_dword(x\d)?\s+(v\[\d+:\d+\]|v\d+)\s*,', t)
        if not (in_asm and m):
            continue
        dst = regs(m.group(2))
        checked += 1
        waited, asm2 = False, True
        for j in range(i + 1, min(len(lines), i + 6000)):
            u = lines[j].strip()
            if u.startswith(';;#ASMSTART'):
                asm2 = True
                continue
            if u.startswith(';;#ASMEND'):
                asm2 = False
                continue
            if not u or u.startswith((';', '.', '//')) or u.endswith(':'):
                if u.startswith('.Lfunc_end'):
                    break
                continue
            if re.match(r's_waitcnt\b.*vmcnt\(', u):
                waited = True
                continue
            sib = re.match(r'global_load_dword(x\d)?\s+(v\[\d+:\d+\]|v\d+)\s*,\s*(v\[\d+:\d+\])', u) if asm2 else None
            body = u.split('//')[0]
            if sib:                                     # a sibling asm load: only its ADDRESS operand may be one of our registers if
                body = sib.group(2)                     # it was the address of this load too (in-place); its destination is its own
                if regs(body) & dst and not waited:
                    bad.append((func, i + 1, t, j + 1, u))
                continue
            if regs(body) & dst:
                if not waited:
                    bad.append((func, i + 1, t, j + 1, u))
                break
    return checked, bad


if __name__ == '__main__':
    srcs = sys.argv[1:] or [os.path.join(ROOT, 'salsa_amd', 'csrc', 'conv_mfma.hip')]
    rc = 0
    for src in srcs:
        n, bad = check(src)
        print('%s: %d asm-issued register loads checked, %d touched before a vmcnt wait' % (os.path.basename(src), n, len(bad)))
        for func, a, t, b, u in bad:
            print('  %s\n    line %d: %s\n    line %d: %s' % (func, a, t, b, u))
            rc = 1
    sys.exit(rc)
