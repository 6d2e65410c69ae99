#!/usr/bin/env python
"""The wide 3x3 convolution kernel (conv_wide.hip) against MIOpen through torch, layer by layer at the training shapes
(batch 32): correctness (vs float32 F.conv2d of the bf16-rounded operands) and time of forward and data gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from salsa_amd.crnn import nn_ops

dev = 'cuda:0'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(64, 128, 160, 50), (128, 128, 160, 50), (128, 256, 80, 25), (256, 256, 80, 25), (256, 512, 40, 12), (512, 512, 40, 12),
          (128, 64, 160, 50), (256, 128, 80, 25), (512, 256, 40, 12)]


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
    return e0.elapsed_time(e1) / n


g = torch.Generator(device=dev).manual_seed(0)
for cin, cout, H, W in shapes:
    x = torch.randn((N, cin, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, 3, 3), device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ok = bool(nn_ops._lib.load().salsa_nn_conv3x3_wide_supported(N, H, W, cin, cout))
    if not ok:
        print('%4d -> %4d @ %3dx%3d: not supported' % (cin, cout, H, W))
        continue
    y = nn_ops._conv_wide(x, w)
    ref = F.conv2d(x[:2].float(), w.float(), padding=1)
    err = float((y[:2].float() - ref).abs().max() / ref.abs().max())
    t_hip = timed(lambda: nn_ops._conv_wide(x, w))
    t_mio = timed(lambda: F.conv2d(x, w, padding=1))
    fl = 2.0 * N * H * W * cin * cout * 9
    if nn_ops._lib.load().salsa_nn_conv3x3_wide_wrw_supported(N, H, W, cin, cout):
        gy = torch.randn((N, cout, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gw = nn_ops._conv_wide_wrw(x, gy)
        ref_w = torch.ops.aten.convolution_backward(gy[:2].float(), x[:2].float(), w.float(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        gw2 = nn_ops._conv_wide_wrw(x[:2].contiguous(memory_format=torch.channels_last), gy[:2].contiguous(memory_format=torch.channels_last))
        werr = float((gw2 - ref_w).abs().max() / ref_w.abs().max())
        t_w = timed(lambda: nn_ops._conv_wide_wrw(x, gy))
        t_wm = timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
        print('      wrw: rel err %.2e   hip %.3f ms (%4.0f TF/s)   MIOpen %.3f ms (%4.0f TF/s)   ratio %.2f' % (werr, t_w, fl / t_w / 1e9, t_wm, fl / t_wm / 1e9, t_wm / t_w))
    print('%4d -> %4d @ %3dx%3d: rel err %.2e   hip %.3f ms (%4.0f TF/s)   MIOpen %.3f ms (%4.0f TF/s)   ratio %.2f'
          % (cin, cout, H, W, err, t_hip, fl / t_hip / 1e9, t_mio, fl / t_mio / 1e9, t_mio / t_hip), flush=True)
