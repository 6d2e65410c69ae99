import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd.crnn import nn_ops
N, H, W = 32, 640, 200
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn((N, 64, H, W), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(2)]
w = (torch.randn((64, 64, 3, 3), device=dev, generator=g) * 0.06).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for i in range(6):
    nn_ops._conv64(xs[i % 2], w)
torch.cuda.synchronize()
