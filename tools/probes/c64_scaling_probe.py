#!/usr/bin/env python
"""Launch time of the 64 -> 64 kernels against the batch size at the stage-1 map (320 x 100) and the stem map (640 x 200): a
straight-line fit separates the per-launch fixed cost (filter load, pipeline fill, tail) from the per-tile rate."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops

dev = torch.device('cuda:0')
nn_ops.set_deterministic(os.environ.get('SALSA_DETERMINISTIC', '1') != '0', dev)
print('deterministic slabs' if nn_ops.is_deterministic() else 'float atomics')
L = _lib.load()
g = torch.Generator(device='cpu').manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, W in ((320, 100), (640, 200)):
    for N in (2, 4, 8, 16, 32, 64):
        if N * H * W * 64 * 2 > 3 << 30:
            continue
        x = torch.randn((N, 64, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((64, 64, 3, 3), generator=g) * 0.06).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gy = torch.randn((N, 64, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        nb = L.salsa_nn_conv3x3_c64_stats_blocks(N, H, W)
        part = torch.empty(nb * 128, dtype=torch.float64, device=dev)
        t_f = timed(lambda: nn_ops._conv64(x, w))
        t_s = timed(lambda: nn_ops._conv64(x, w, stats_part=part))
        gw = torch.zeros((64, 3, 3, 64), dtype=torch.float32, device=dev)
        t_w = timed(lambda: L.salsa_nn_conv3x3_c64_wrw(nn_ops._ptr(x), nn_ops._ptr(gy), nn_ops._ptr(gw), N, H, W, nn_ops._stream(x)))
        fl = 2.0 * N * H * W * 64 * 64 * 9
        print('%3d x %3d x %3d: fwd %7.1f us (%4.0f TF/s)  fwd+stats %7.1f us (%4.0f)  wrw %7.1f us (%4.0f)   tiles %6d  wgs %d'
              % (N, H, W, t_f, fl / t_f / 1e6, t_s, fl / t_s / 1e6, t_w, fl / t_w / 1e6, N * H * W // 128, nb), flush=True)
