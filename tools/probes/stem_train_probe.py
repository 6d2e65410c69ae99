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


stat = [torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5, torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3]
dwb = torch.empty((2, 64), device=dev)
ws = torch.empty(L.salsa_nn_bn_workspace_bytes(1, N * H * W, 64) // 8 + 1, dtype=torch.float64, device=dev)
coef2 = torch.empty(7 * 64, device=dev)
HAS_BNF = hasattr(L, 'salsa_nn_conv3x3_stem_wrw_bnf')
if HAS_BNF:
    nbytes = L.salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes(N, H, W)
    slabs = torch.empty(nbytes // 4, device=dev)


def pair():  # the BatchNorm backward's reduction (coefficients only) + the weight gradient that forms dx on load
    assert L.salsa_nn_bn_bwd(nn_ops._ptr(g), None, nn_ops._ptr(x1), None, None, 1, N * H * W, 64, nn_ops._ptr(stat[2]), nn_ops._ptr(stat[3]),
                             nn_ops._ptr(stat[0]), nn_ops._ptr(stat[1]), 1, nn_ops._ptr(dwb[0]), nn_ops._ptr(dwb[1]), nn_ops._ptr(ws),
                             nn_ops._ptr(coef2), 0.0, 0, nn_ops._stream(x)) == 0
    assert L.salsa_nn_conv3x3_stem_wrw_bn(nn_ops._ptr(x), x.stride(0), x.stride(1), nn_ops._ptr(g), nn_ops._ptr(x1), nn_ops._ptr(coef2), 1,
                                          nn_ops._ptr(gw), N, Cin, H, W, nn_ops._stream(x)) == 0


def bnf():   # both in one pass (round 5)
    assert L.salsa_nn_conv3x3_stem_wrw_bnf(nn_ops._ptr(x), x.stride(0), x.stride(1), nn_ops._ptr(g), nn_ops._ptr(x1), nn_ops._ptr(stat[0]),
                                           nn_ops._ptr(stat[1]), nn_ops._ptr(stat[2]), nn_ops._ptr(stat[3]), 1, nn_ops._ptr(gw),
                                           nn_ops._ptr(dwb[0]), nn_ops._ptr(dwb[1]), nn_ops._ptr(slabs), nbytes, N, Cin, H, W,
                                           nn_ops._stream(x)) == 0


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

fwd()                                                                  # (x1 = a real convolution output for the pair below)
gw.zero_(); pair(); torch.cuda.synchronize()
ref = (gw.clone(), dwb.clone())
t_p = timed(pair)
line = 'BatchNorm reduction + weight gradient: two passes %.1f us' % (t_p * 1e3)
if HAS_BNF:
    gw.zero_(); bnf(); torch.cuda.synchronize()
    err = [float((a - b).abs().max() / b.abs().max()) for a, b in zip((gw, dwb), ref)]
    t_b = timed(bnf)
    line += ' | one pass %.1f us (max rel diff dW %.1e, dgamma/dbeta %.1e)' % (t_b * 1e3, err[0], err[1])
print(line, flush=True)
