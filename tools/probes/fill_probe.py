#!/usr/bin/env python
"""Write-only / read-only / copy bandwidth of plain torch kernels on ~1 GB (what a store-bound kernel can hope for)."""
import torch
dev = 'cuda:0'
n = 983_040_000 // 2
y = torch.empty(n, dtype=torch.bfloat16, device=dev)
x = torch.randn(n, dtype=torch.float32, device=dev).to(torch.bfloat16)
def t(f, nbytes, name):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('%-12s %.3f ms  %.2f TB/s' % (name, ms, nbytes / ms / 1e9))
t(lambda: y.fill_(1.0), n * 2, 'fill')
t(lambda: y.zero_(), n * 2, 'zero')
t(lambda: x.sum(), n * 2, 'sum (read)')
t(lambda: y.copy_(x), n * 4, 'copy')
t(lambda: torch.relu_(x), n * 4, 'relu_ (rw)')
