#!/usr/bin/env python
"""64 -> 64 forward kernel with and without the BatchNorm-statistics epilogue, training sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
for H, W in ((640, 200), (320, 100)):
    N = 32
    x = torch.randn((N, 64, H, W), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((64, 64, 3, 3), device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    part = torch.empty(_lib.load().salsa_nn_conv3x3_c64_stats_blocks(N, H, W) * 128, dtype=torch.float64, device=dev)

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    a, b = timed(lambda: nn_ops._conv64(x, w)), timed(lambda: nn_ops._conv64(x, w, stats_part=part))
    print('conv64 %dx%d: plain %.3f ms  with statistics %.3f ms' % (H, W, a, b))
