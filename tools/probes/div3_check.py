#!/usr/bin/env python
"""div3_check.py -- the tracker's three-instruction x / 3 (salsa_kernels.hip div3_exact: q = RN(x y), r = fma(-3, q, x),
RN(q + r y) with y = RN(1/3)) against IEEE division, in exact rational arithmetic (every FMA is one correctly rounded
operation: Fraction -> float rounds to nearest even).  Markstein's theorem says they agree; this checks it over random
exponents, random mantissas, multiples of 3 and the subnormal edge."""
import math
import random
import struct
from fractions import Fraction as F

y = 1.0 / 3.0


def div3(a):
    q = a * y
    r = float(F(a) - 3 * F(q))
    return float(F(q) + F(r) * F(y))


random.seed(1)
n = bad = 0
vals = [0.0, 5e-324, 1e-310, 2.2250738585072014e-308, 1e-300, 1.0, 3.0, 1e300]
for _ in range(300000):
    vals.append(math.ldexp(random.random() + 1.0, random.randint(-250, 250)))
for _ in range(200000):
    a = struct.unpack('d', struct.pack('Q', random.getrandbits(52) | (1023 << 52)))[0]
    vals += [a, a * 3.0]
for a in vals:
    n += 1
    if div3(a) != a / 3.0:
        bad += 1
        print('MISMATCH', a.hex())
print('checked %d values, %d mismatches' % (n, bad))
assert bad == 0
