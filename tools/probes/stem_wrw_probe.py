#!/usr/bin/env python
"""The first layer's weight gradient at training size (32 x 7 x 640 x 200): salsa_nn_conv3x3_stem_wrw against MIOpen's
(with the bf16 channels-last copy of the input MIOpen needs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
L = _lib.load()
N, Cin, H, W = 32, 7, 640, 200
x = torch.randn((N, Cin, H, W), device=dev)
gy = torch.randn((N, 64, H, W), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
wb = torch.randn((64, Cin, 3, 3), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
gw = torch.zeros((64, Cin, 3, 3), dtype=torch.float32, device=dev)


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


def own():
    assert L.salsa_nn_conv3x3_stem_wrw(nn_ops._ptr(x), x.stride(0), x.stride(1), nn_ops._ptr(gy), nn_ops._ptr(gw), N, Cin, H, W, nn_ops._stream(x)) == 0


def miopen():
    xb = x.to(dtype=torch.bfloat16, memory_format=torch.channels_last)
    return torch.ops.aten.convolution_backward(gy, xb, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]


gw.zero_(); own()
ref = miopen().float()
print('rel diff own vs MIOpen %.2e' % float((gw - ref).abs().max() / ref.abs().max()))
t_o, t_m = timed(own), timed(miopen)
print('stem wrw: own %.3f ms (%.2f TB/s of dy + x)   MIOpen incl. input copy %.3f ms' % (t_o, (gy.numel() * 2 + x.numel() * 4) / t_o / 1e9, t_m))
