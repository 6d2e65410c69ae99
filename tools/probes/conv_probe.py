#!/usr/bin/env python
"""The 64 -> 64 convolution kernel alone on an inference-shaped input (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn((n, 64, 2400, 100), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn((64, 64, 3, 3), device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for _ in range(3):
    y = nn_ops._conv64(x, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y = nn_ops._conv64(x, w)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print('conv64 %s: %.3f ms  %.0f TFLOP/s' % (tuple(x.shape), ms, 2 * n * 2400 * 100 * 64 * 64 * 9 / ms / 1e9))
