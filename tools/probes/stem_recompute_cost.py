#!/usr/bin/env python
"""Round-5 review item 4, costed with measurements: the first layer's training forward is
  (a) salsa_nn_conv3x3_stem_stats   reads x (115 MB), writes the pre-BatchNorm map x1 (524 MB) + statistics
  (b) salsa_nn_bn_train_fwd         reads x1 (524 MB), writes y = relu(bn(x1)) (524 MB)
and the proposal is (d) a statistics-only pass (no store) + (c) a second launch that recomputes the convolution and stores
relu(scale * conv + shift) directly.  (c) is the inference stem kernel (conv + folded shift + ReLU -> y), (d) is the statistics
kernel built with -DSTEM_NO_STORE -DSALSA_PROBE_BUILD (run this script once per library: SALSA_HIP_LIB=...).  The backward
(salsa_nn_conv3x3_stem_wrw_bnf) READS x1; so either (c) stores x1 as well (+524 MB written) or the weight-gradient kernel
recomputes it (+33 GFLOP of MFMA and an LDS transpose in a kernel that is issue-bound at 288 us)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = torch.device('cuda:0')
L = _lib.load()
print('library flags: %r' % _lib.build_flags())
N, Cin, H, W = 32, 7, 640, 200
x = torch.randn((N, Cin, H, W), device=dev)
conv, bn = nn_ops.Conv3x3(Cin, 64, 3, padding=1, bias=False).to(dev), nn_ops.BatchNormAct2d(64).to(dev)
wq = nn_ops._stem_filter(conv.weight)
shift = torch.randn(64, device=dev)
ptr, stream = nn_ops._ptr, nn_ops._stream
nb = L.salsa_nn_conv3x3_stem_stats_blocks(N, H, W)
x1 = torch.empty((N, 64, H, W), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
y = torch.empty_like(x1, memory_format=torch.channels_last)
part = torch.empty(nb * 128, dtype=torch.float64, device=dev)
save = torch.empty((2, 64), dtype=torch.float32, device=dev)
M = N * H * W
ws = torch.empty(L.salsa_nn_bn_workspace_bytes(1, M, 64) // 8 + 1, dtype=torch.float64, device=dev)
nbt = torch.zeros((), dtype=torch.long, device=dev)

def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

a = timeit(lambda: L.salsa_nn_conv3x3_stem_stats(ptr(x), x.stride(0), x.stride(1), ptr(wq), ptr(x1), ptr(part), N, Cin, H, W, stream(x)))
b = timeit(lambda: L.salsa_nn_bn_train_fwd(ptr(x1), ptr(y), None, 1, M, 64, ptr(bn.weight), ptr(bn.bias), 1e-5, 0.1, ptr(bn.running_mean),
                                           ptr(bn.running_var), ptr(save[0]), ptr(save[1]), ptr(ws), 1, 0.0, 0, ptr(nbt), ptr(part), nb, stream(x)))
c = timeit(lambda: nn_ops._conv_stem(x, wq, shift, True))
print('(a) statistics kernel as built here   %7.1f us' % a)
print('(b) BatchNorm apply pass              %7.1f us' % b)
print('(c) conv + shift + ReLU -> y          %7.1f us' % c)
print('(a) + (b) = %.1f us; (a) + (c) = %.1f us' % (a + b, a + c))
