#!/usr/bin/env python
"""Round 5: the first layer's TRAINING kernels alone at 32 x 7 x 640 x 200 -- forward with the statistics epilogue
(salsa_nn_conv3x3_stem_stats) and the weight gradient with the BatchNorm backward formed on load (salsa_nn_conv3x3_stem_wrw_bn) --
event-timed over 20 launches each; SALSA_HIP_LIB selects a probe build.  Prints a checksum of each result so that variants can be
compared for equality."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
L = _lib.load()
torch.manual_seed(0)
N, Cin, H, W = 32, 7, 640, 200
x = torch.randn((N, Cin, H, W), device=dev)
w = torch.randn(64, Cin, 3, 3, device=dev) * 0.1
wq = nn_ops._stem_filter(w)
x1 = torch.empty((N, 64, H, W), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
g = torch.randn((N, 64, H, W), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
nb = L.salsa_nn_conv3x3_stem_stats_blocks(N, H, W)
part = torch.empty(nb * 128, dtype=torch.float64, device=dev)
coef = (torch.rand(7 * 64, device=dev) + 0.5).float()
gw = torch.zeros((64, Cin, 3, 3), dtype=torch.float32, device=dev)
nn_ops.set_deterministic(True, dev)


def fwd():
    assert L.salsa_nn_conv3x3_stem_stats(nn_ops._ptr(x), x.stride(0), x.stride(1), nn_ops._ptr(wq), nn_ops._ptr(x1), nn_ops._ptr(part),
                                         N, Cin, H, W, nn_ops._stream(x)) == 0


def wrw():
    assert L.salsa_nn_conv3x3_stem_wrw_bn(nn_ops._ptr(x), x.stride(0), x.stride(1), nn_ops._ptr(g), nn_ops._ptr(x1), nn_ops._ptr(coef), 1,
                                          nn_ops._ptr(gw), N, Cin, H, W, nn_ops._stream(x)) == 0


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


fwd()
t_f = timed(fwd)
s_f = float(x1.float().abs().sum()), float(part.sum())
gw.zero_(); wrw()
s_w = float(gw.abs().sum())
t_w = timed(wrw)
gbf = (x.numel() * 4 + x1.numel() * 2) / 1e9
gbw = (x.numel() * 4 + 2 * x1.numel() * 2) / 1e9
print('stem fwd+stats %.1f us (%.2f TB/s)  wrw_bn %.1f us (%.2f TB/s, slab reduction included)  checks %.6e %.6e %.6e'
      % (t_f * 1e3, gbf / t_f, t_w * 1e3, gbw / t_w, s_f[0], s_f[1], s_w), flush=True)
