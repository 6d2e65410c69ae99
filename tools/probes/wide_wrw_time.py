#!/usr/bin/env python
"""wide_wrw_time.py: time of the wide weight-gradient launch alone (no slab reduction: SALSA_DETERMINISTIC=0) at the CRNN's shapes"""
import os, sys
os.environ.setdefault('SALSA_DETERMINISTIC', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
nn_ops.set_deterministic(os.environ.get('SALSA_DETERMINISTIC', '1') != '0', dev)
print('deterministic slabs' if nn_ops.is_deterministic() else 'float atomics')
g = torch.Generator(device=dev).manual_seed(0)
for cin, cout, H, W in ((128, 128, 160, 50), (256, 256, 80, 25), (512, 512, 40, 12)):
    xs = [torch.randn((32, cin, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    gys = [torch.randn((32, cout, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    for _ in range(3):
        nn_ops._conv_wide_wrw(xs[0], gys[0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(30):
        nn_ops._conv_wide_wrw(xs[i % 3], gys[i % 3])
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30
    ws = [(torch.randn((cout, cin, 3, 3), device=dev, generator=g) * 0.02).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    for _ in range(3):
        nn_ops._conv_wide(xs[0], ws[0])
    torch.cuda.synchronize()
    e0.record()
    for i in range(30):
        nn_ops._conv_wide(xs[i % 3], ws[i % 3])
    e1.record()
    torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / 30
    print('%s %d->%d %dx%d: forward %.1f us  %.0f TF/s' % (os.path.basename(os.environ.get('SALSA_HIP_LIB', 'default')), cin, cout, H, W, tf * 1e3, 2.0 * 32 * H * W * cin * cout * 9 / tf / 1e9), flush=True)
    print('%s %d->%d %dx%d: %.1f us  %.0f TF/s' % (os.path.basename(os.environ.get('SALSA_HIP_LIB', 'default')), cin, cout, H, W, t * 1e3, 2.0 * 32 * H * W * cin * cout * 9 / t / 1e9), flush=True)
