#!/usr/bin/env python
"""The 1x1 shortcut convolutions at batch 32: hand-written kernels (conv_1x1.hip) against MIOpen through torch, per layer and
direction (forward, data gradient, weight gradient)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
L = _lib.load()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (cin, cout), (h, w) in (((64, 128), (160, 50)), ((128, 256), (80, 25)), ((256, 512), (40, 12))):
    N = 32
    x = torch.randn((N, cin, h, w), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((N, cout, h, w), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wb = (torch.randn((cout, cin, 1, 1), device=dev) * 0.05).to(torch.bfloat16)
    wbt = wb.transpose(0, 1).contiguous()
    gw = torch.zeros((cout, cin, 1, 1), dtype=torch.float32, device=dev)
    M = N * h * w
    mb = 2e-6 * M * (cin + cout)
    t_f = timed(lambda: nn_ops._conv1x1_hip(x, wb))
    t_d = timed(lambda: nn_ops._conv1x1_hip(gy, wbt))
    t_w = timed(lambda: L.salsa_nn_conv1x1_wrw(nn_ops._ptr(x), nn_ops._ptr(gy), nn_ops._ptr(gw), M, cin, cout, nn_ops._stream(x)))
    m_f = timed(lambda: F.conv2d(x, wb))
    m_b = timed(lambda: torch.ops.aten.convolution_backward(gy, x, wb, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, True, False]))
    print('%3d -> %3d @ %3dx%3d (%5.1f MB of activations): fwd %.3f ms (MIOpen %.3f)  dgrad %.3f  wrw %.3f  (MIOpen dgrad + wrw %.3f)   fwd %.2f TB/s'
          % (cin, cout, h, w, mb, t_f, m_f, t_d, t_w, m_b, mb / t_f / 1e3))
