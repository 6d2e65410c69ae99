#!/usr/bin/env python
"""c64_rot_time.py: the 64 -> 64 kernels at the training shapes with inputs ROTATING through buffers larger than the Infinity Cache
(what a training step sees), forward plain / with statistics / weight gradient"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = torch.device('cuda:0')
nn_ops.set_deterministic(os.environ.get('SALSA_DETERMINISTIC', '1') != '0', dev)
L = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
tag = os.path.basename(os.environ.get('SALSA_HIP_LIB', 'default'))
for N, H, W, R in ((32, 320, 100, 4), (32, 640, 200, 2)):
    xs = [torch.randn((N, 64, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(R)]
    gys = [torch.randn((N, 64, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(R)]
    w = (torch.randn((64, 64, 3, 3), device=dev, generator=g) * 0.06).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    nb = L.salsa_nn_conv3x3_c64_stats_blocks(N, H, W)
    part = torch.empty(nb * 128, dtype=torch.float64, device=dev)
    gw = torch.zeros((64, 3, 3, 64), dtype=torch.float32, device=dev)
    fns = {'fwd': lambda i: nn_ops._conv64(xs[i % R], w), 'fwd+stats': lambda i: nn_ops._conv64(xs[i % R], w, stats_part=part),
           'wrw': lambda i: L.salsa_nn_conv3x3_c64_wrw(nn_ops._ptr(xs[i % R]), nn_ops._ptr(gys[i % R]), nn_ops._ptr(gw), N, H, W, nn_ops._stream(w))}
    out = []
    for name, fn in fns.items():
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(24):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 24
        out.append('%s %.1f us (%.0f TF/s)' % (name, t * 1e3, 2.0 * N * H * W * 64 * 64 * 9 / t / 1e9))
    print(tag, '%d x %d x %d:' % (N, H, W), '  '.join(out), flush=True)
