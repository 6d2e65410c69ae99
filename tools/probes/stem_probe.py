#!/usr/bin/env python
"""Time of the stem convolution kernel on an inference sub-batch (8 x 7 x 4800 x 200) and a training batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
w = torch.randn(64, 7, 3, 3, device=dev)
wq = nn_ops._stem_filter(w)
shift = torch.randn(64, device=dev)
for shape in ((8, 7, 4800, 200), (32, 7, 640, 200)):
    x = torch.randn(shape, device=dev)
    for _ in range(3):
        nn_ops._conv_stem(x, wq, shift, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = nn_ops._conv_stem(x, wq, shift, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = (x.numel() * 4 + y.numel() * 2) / 1e9
    print(shape, '%.3f ms  %.2f TB/s' % (ms, gb / ms))
