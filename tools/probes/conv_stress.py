#!/usr/bin/env python
"""Stress of the LDS-direct 64 -> 64 convolution kernel: many random inputs and shapes (few and many tiles per persistent
workgroup, ragged edges) against a float32 convolution of the same bf16-valued operands; any race in the three-buffer
rotation would show as a sporadic mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
torch.manual_seed(0)
worst = 0.0
shapes = [(32, 2400, 100), (8, 4800, 200), (3, 37, 45), (1, 4, 32), (2, 5, 33), (64, 80, 50), (1, 1, 1), (5, 640, 7), (16, 320, 100)]
for n, h, w_ in shapes:
    reps = 4 if n * h * w_ > 4_000_000 else 25
    for r in range(reps):
        x = torch.randn((n, 64, h, w_), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((64, 64, 3, 3), device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = nn_ops._conv64(x, w)
        y2 = nn_ops._conv64(x, w)
        assert torch.equal(y, y2), ('not deterministic', n, h, w_, r)
        ref = F.conv2d(x.float(), w.float(), padding=1)
        err = (y.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
        worst = max(worst, err)
        assert err < 8e-3, (n, h, w_, r, err)
    print((n, h, w_), 'ok', reps, flush=True)
print('worst relative-to-max error %.2e' % worst)
