#!/usr/bin/env python
"""Steady-state kernel breakdown of one CRNN training step (torch.profiler, after MIOpen's solver search has run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from salsa_amd.crnn.train import Trainer, synthetic_batch

tr = Trainer('cuda:0')
x, sed, doa = synthetic_batch(32, 'cuda:0', seed=1)
for _ in range(6):
    tr.train_step(x, sed, doa)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        tr.train_step(x, sed, doa)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print('total device ms per step: %.2f' % (tot / 3e3))
for e in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 28]:
    print('%-80s n=%5d  %7.2f ms/step  %5.1f%%' % (e.key[:80], e.count // 3, e.device_time_total / 3e3, 100 * e.device_time_total / tot))
