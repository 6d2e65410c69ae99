#!/usr/bin/env python
"""c64_one.py <N> <H> <W> [reps]: launches of the 64 -> 64 forward (plain, with statistics) and weight-gradient kernels at ONE
shape, each on a FRESH input (rotating buffers larger than the MALL) -- for counter passes (tools/probes/pmc_probe.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
N, H, W = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device('cuda:0')
L = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn((N, 64, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(4)]
gys = [torch.randn((N, 64, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(4)]
w = (torch.randn((64, 64, 3, 3), device=dev, generator=g) * 0.06).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
nb = L.salsa_nn_conv3x3_c64_stats_blocks(N, H, W)
part = torch.empty(nb * 128, dtype=torch.float64, device=dev)
gw = torch.zeros((64, 3, 3, 64), dtype=torch.float32, device=dev)
for i in range(reps):
    x, gy = xs[i % 4], gys[i % 4]
    nn_ops._conv64(x, w)
    nn_ops._conv64(x, w, stats_part=part)
    L.salsa_nn_conv3x3_c64_wrw(nn_ops._ptr(x), nn_ops._ptr(gy), nn_ops._ptr(gw), N, H, W, nn_ops._stream(x))
torch.cuda.synchronize()
