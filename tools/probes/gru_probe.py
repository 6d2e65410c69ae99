#!/usr/bin/env python
"""gru_probe.py -- the register-resident GRU scans alone (salsa_gru_scan_fwd_regw / _bwd_regw through the C ABI), at the
training shape (T = 40 label-rate steps, B = 32, 2 directions) and the inference shape (T = 300): microseconds per launch and
per step.  SALSA_HIP_LIB selects the build (A/B of kernel variants on one box)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from salsa_amd import _lib  # noqa: E402

L = _lib.load()
dev = torch.device('cuda:0')
H, D = 256, 2
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
for T, B in ((40, 32), (300, 32), (300, 8)):
    g = torch.Generator(device=dev).manual_seed(0)
    gi = torch.randn(T, B, D, 3 * H, device=dev, generator=g)
    whh = torch.randn(D, 3 * H, H, device=dev, generator=g) * 0.05
    bhh = torch.randn(D, 3 * H, device=dev, generator=g) * 0.1
    hs = torch.empty(T, B, D, H, device=dev)
    saved = torch.empty(T, B, D, 4 * H, device=dev)
    dhs = torch.randn(T, B, D, H, device=dev, generator=g)
    dgi = torch.empty(T, B, D, 3 * H, device=dev)
    dgh = torch.empty_like(dgi)
    big = torch.empty(1 << 28, dtype=torch.uint8, device=dev)      # 256 MiB: flushes L2 / Infinity Cache between launches

    def run(fn, reps=10):
        ms = []
        for _ in range(reps):
            big.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert fn() == 0
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        ms.sort()
        return ms[len(ms) // 2] * 1e3

    f = run(lambda: L.salsa_gru_scan_fwd_regw(p(gi), p(whh), p(bhh), p(hs), p(saved), T, B, D, H, st))
    fi = run(lambda: L.salsa_gru_scan_fwd_regw(p(gi), p(whh), p(bhh), p(hs), None, T, B, D, H, st))
    b = run(lambda: L.salsa_gru_scan_bwd_regw(p(dhs), p(whh), p(hs), p(saved), p(dgi), p(dgh), T, B, D, H, st))
    print('T=%3d B=%2d  fwd (training, saves gates) %7.1f us = %.2f us/step   fwd (inference) %7.1f us = %.2f us/step   bwd %7.1f us = %.2f us/step   checksum %.6f %.6f'
          % (T, B, f, f / T, fi, fi / T, b, b / T, float(hs.double().sum()), float(dgi.double().sum())))
