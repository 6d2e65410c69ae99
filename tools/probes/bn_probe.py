#!/usr/bin/env python
"""Per-kernel time of the fused BatchNorm passes on the CRNN's largest activations (torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
from salsa_amd.crnn.nn_ops import BatchNormAct2d

dev = 'cuda:0'
for shape, res in (((32, 64, 640, 200), False), ((32, 64, 320, 100), True), ((32, 128, 160, 50), False), ((32, 256, 80, 25), True), ((32, 512, 40, 12), True)):
    bn = BatchNormAct2d(shape[1]).to(dev).train()
    x = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
    gy = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        bn(x, residual=r, relu=True).backward(gy)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            bn(x, residual=r, relu=True).backward(gy)
        torch.cuda.synchronize()
    mb = x.numel() * 2 / 1e6
    print(shape, 'residual' if res else '', '%.0f MB per tensor' % mb)
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
        if 'bn_' in e.key:
            name = e.key.split('::')[-1].split('<')[0].split('(')[0]
            ms = e.device_time_total / e.count / 1e3
            n_t = {'bn_stats_kernel': 1, 'bn_apply_kernel': 2 + res, 'bn_bwd_reduce_kernel': 2 + res, 'bn_bwd_apply_kernel': 3 + 2 * res}.get(name, 0)
            print('   %-24s %.3f ms  %s' % (name, ms, '%.2f TB/s' % (n_t * mb / ms / 1e3) if n_t else ''))
