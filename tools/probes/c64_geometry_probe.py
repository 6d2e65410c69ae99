#!/usr/bin/env python
"""The 64 -> 64 kernels (forward plain / with statistics / folded-BatchNorm + residual / pooled, and the weight gradient) at the
CRNN's map shapes and a few ragged ones: results against torch (float32 on the bf16 inputs) and time per launch.  A/B of the
transposed tile geometry (conv_mfma.hip struct Geo): SALSA_HIP_LIB=<a build with -DC64_TRANSPOSE=0> for the other side."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops

dev = torch.device('cuda:0')
L = _lib.load()
g = torch.Generator(device='cpu').manual_seed(0)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(32, 320, 100), (32, 640, 200), (8, 2400, 100), (4, 4800, 200), (3, 37, 50), (2, 130, 66), (5, 64, 8)]
for N, H, W in shapes:
    x = torch.randn((N, 64, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((64, 64, 3, 3), generator=g) * 0.06).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((N, 64, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn((N, 64, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    shift = torch.randn(64, generator=g).to(dev)
    small = N * H * W <= 2_000_000
    ref = F.conv2d(x.float(), w.float(), padding=1) if small else None
    errs = {}
    y = nn_ops._conv64(x, w)
    if small:
        errs['fwd'] = float((y.float() - ref).abs().max() / ref.abs().max())
    nb = L.salsa_nn_conv3x3_c64_stats_blocks(N, H, W)
    part = torch.empty(nb * 128, dtype=torch.float64, device=dev)
    ys = nn_ops._conv64(x, w, stats_part=part)
    errs['stats_y'] = 0.0 if torch.equal(ys, y) else 1.0
    p2 = part.view(nb, 2, 64).sum(0)
    yd = y.double()
    errs['stats_sum'] = float((p2[0] - yd.sum(dim=(0, 2, 3))).abs().max() / yd.abs().sum(dim=(0, 2, 3)).max())
    errs['stats_sq'] = float(((p2[1] - (yd * yd).sum(dim=(0, 2, 3))).abs() / (yd * yd).sum(dim=(0, 2, 3))).max())
    # folded BatchNorm + residual + ReLU
    yb = torch.empty_like(y)
    assert L.salsa_nn_conv3x3_c64_bias_act(nn_ops._ptr(x), nn_ops._ptr(w), nn_ops._ptr(shift), nn_ops._ptr(res), nn_ops._ptr(yb), 1, N, H, W,
                                           nn_ops._stream(x)) == 0
    if small:
        want = torch.relu(ref + shift.view(1, -1, 1, 1) + res.float())
        errs['bias_res'] = float((yb.float() - want).abs().max() / want.abs().max())
    t_pool = None
    if H % 2 == 0 and W % 2 == 0:
        yp = torch.empty((N, 64, H // 2, W // 2), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
        run_pool = lambda: L.salsa_nn_conv3x3_c64_bias_act_pool(nn_ops._ptr(x), nn_ops._ptr(w), nn_ops._ptr(shift), None, nn_ops._ptr(yp), 1,  # noqa: E731
                                                                N, H, W, nn_ops._stream(x))
        assert run_pool() == 0
        if small:
            want = F.avg_pool2d(torch.relu(ref + shift.view(1, -1, 1, 1)), 2)
            errs['pool'] = float((yp.float() - want).abs().max() / want.abs().max())
        t_pool = timed(run_pool)
    gw = torch.zeros((64, 3, 3, 64), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
    assert L.salsa_nn_conv3x3_c64_wrw(nn_ops._ptr(x), nn_ops._ptr(gy), nn_ops._ptr(gw), N, H, W, nn_ops._stream(x)) == 0
    if small:
        rw = torch.nn.grad.conv2d_weight(x.float(), (64, 64, 3, 3), gy.float(), padding=1)
        errs['wrw'] = float((gw - rw).abs().max() / rw.abs().max())
    t_f = timed(lambda: nn_ops._conv64(x, w))
    t_s = timed(lambda: nn_ops._conv64(x, w, stats_part=part))
    gw2 = torch.zeros_like(gw)
    t_w = timed(lambda: L.salsa_nn_conv3x3_c64_wrw(nn_ops._ptr(x), nn_ops._ptr(gy), nn_ops._ptr(gw2), N, H, W, nn_ops._stream(x)))
    fl = 2 * N * H * W * 64 * 64 * 9 / 1e9
    print('%2d x %4d x %3d  fwd %.3f ms (%4.0f TF)  +stats %.3f  wrw %.3f (%4.0f TF)  pooled %s   max rel err: %s' %
          (N, H, W, t_f, fl / t_f, t_s, t_w, fl / t_w, ('%.3f' % t_pool) if t_pool else '  -  ',
           ' '.join('%s %.1e' % kv for kv in errs.items())), flush=True)
