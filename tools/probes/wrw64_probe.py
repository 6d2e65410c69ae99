#!/usr/bin/env python
"""The 64 -> 64 weight-gradient kernel alone: stem-sized (640 x 200) and stage-1-sized (320 x 100) inputs at batch 32,
checked against the float32 torch gradient on a small slice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
L = _lib.load()
for H, W in ((640, 200), (320, 100)):
    N = 32
    x = torch.randn((N, 64, H, W), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((N, 64, H, W), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gw = torch.zeros((64, 64, 3, 3), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)

    def run(xx=x, gg=gy, out=gw):
        rc = L.salsa_nn_conv3x3_c64_wrw(nn_ops._ptr(xx), nn_ops._ptr(gg), nn_ops._ptr(out), xx.shape[0], H, W, nn_ops._stream(xx))
        assert rc == 0
    xs, gs = x[:1].contiguous(memory_format=torch.channels_last), gy[:1].contiguous(memory_format=torch.channels_last)
    small = torch.zeros_like(gw)
    run(xs, gs, small)
    ref = torch.nn.grad.conv2d_weight(xs.float(), (64, 64, 3, 3), gs.float(), padding=1)
    err = float((small - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('wrw64 %dx%d: %.3f ms  %.0f TFLOP/s  rel err %.1e' % (H, W, ms, 2 * N * H * W * 64 * 64 * 9 / ms / 1e9, err))
