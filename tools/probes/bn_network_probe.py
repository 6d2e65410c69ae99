#!/usr/bin/env python
"""bn_probe.py -- every fused BatchNorm pass of the training step, shape by shape: milliseconds and achieved HBM rate on the
bytes it has to move (round 3: where do the 3.4 ms of BatchNorm passes go, and which kernel is furthest below the ~6.3 TB/s
read / ~5 TB/s copy rate of the box?).  Forward and backward are timed separately with events around 20 repetitions."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from salsa_amd.crnn.nn_ops import BatchNormAct2d  # noqa: E402

dev = torch.device('cuda:0')
B = int(os.environ.get('BATCH', 32))
REPS = 20


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


def case(name, C, H, W, residual=False, relu=True, drop=0.0, pool=False):
    bn = BatchNormAct2d(C).to(dev).train()
    x = torch.randn(B, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    res = torch.randn(B, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True) if residual else None
    S = x.numel() * 2 / 1e9                       # GB of one full-resolution bf16 tensor

    def fwd():
        if pool:
            return bn.relu_pool(x, residual=res)
        return bn(x, residual=res, relu=relu, dropout_p=drop)

    y = fwd()
    g = torch.randn_like(y)
    t_f = timed(fwd)

    def fb():
        yy = fwd()
        yy.backward(g)
        x.grad = None
        if res is not None:
            res.grad = None
        bn.weight.grad = bn.bias.grad = None

    t_fb = timed(fb)
    t_b = t_fb - t_f
    So = S / 4 if pool else S
    # forward: statistics pass reads x; apply reads x (+res) writes y
    bytes_f = S + S + (S if residual else 0) + So
    # backward: reduce reads dy, x (+ y or res for the mask); apply reads the same, writes dx (+ dres)
    mask_extra = S if residual else 0
    bytes_b = (So + S + mask_extra) * 2 + S + (S if residual else 0)
    print('%-34s C=%3d %4dx%-3d  fwd %.3f ms %5.2f TB/s   bwd %.3f ms %5.2f TB/s   (S = %.0f MB)'
          % (name, C, H, W, t_f, bytes_f / t_f, t_b, bytes_b / t_b, S * 1e3))
    return t_f, t_b


tot_f = tot_b = 0.0
plan = [
    ('stem bn1 (relu)', 64, 640, 200, dict()),
    ('stem bn2 (relu+pool)', 64, 640, 200, dict(pool=True)),
    ('stage1 bn1 (relu+dropout) x2', 64, 320, 100, dict(drop=0.1)),
    ('stage1.0 bn2 (+res, relu)', 64, 320, 100, dict(residual=True)),
    ('stage1.1 bn2 (+res, relu, pool)', 64, 320, 100, dict(residual=True, pool=True)),
    ('stage2 bn1 (relu+dropout) x2', 128, 160, 50, dict(drop=0.1)),
    ('stage2 short_bn', 128, 160, 50, dict(relu=False)),
    ('stage2.0 bn2 (+res, relu)', 128, 160, 50, dict(residual=True)),
    ('stage2.1 bn2 (+res, relu, pool)', 128, 160, 50, dict(residual=True, pool=True)),
    ('stage3 bn1 x2', 256, 80, 25, dict(drop=0.1)),
    ('stage3 short_bn', 256, 80, 25, dict(relu=False)),
    ('stage3.0 bn2 (+res)', 256, 80, 25, dict(residual=True)),
    ('stage3.1 bn2 (+res, pool)', 256, 80, 25, dict(residual=True, pool=True)),
    ('stage4 bn1 x2', 512, 40, 12, dict(drop=0.1)),
    ('stage4 short_bn', 512, 40, 12, dict(relu=False)),
    ('stage4 bn2 (+res) x2', 512, 40, 12, dict(residual=True)),
]
for name, C, H, W, kw in plan:
    mult = 2 if 'x2' in name else 1
    f, b = case(name, C, H, W, **kw)
    tot_f += mult * f
    tot_b += mult * b
print('sum over the network (standalone statistics passes included): forward %.3f ms, backward %.3f ms' % (tot_f, tot_b))
