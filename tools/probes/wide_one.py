#!/usr/bin/env python
"""wide_one.py <cin> <cout> <H> <W> [reps]: launches of ONE wide-convolution shape (forward + weight gradient) for counter passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd.crnn import nn_ops
cin, cout, H, W = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, cin, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn((cout, cin, 3, 3), device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
gy = torch.randn((32, cout, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for _ in range(reps):
    nn_ops._conv_wide(x, w)
    nn_ops._conv_wide_wrw(x, gy)
torch.cuda.synchronize()
