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
OPS = '--ops' in sys.argv                                   # table of the aten / autograd operators that launched the kernels
if OPS:
    sys.argv.remove('--ops')
with profile(activities=[ProfilerActivity.CUDA] + ([ProfilerActivity.CPU] if OPS else [])) as prof:
    for _ in range(3):
        tr.train_step(x, sed, doa)
    torch.cuda.synchronize()
if OPS:
    evs = [e for e in prof.key_averages() if e.self_device_time_total > 0 and e.cpu_time_total > 0]
    evs.sort(key=lambda e: -e.count)
    print('operators that launch kernels: calls/step, self device ms/step')
    for e in evs[:60]:
        print('%-60s n=%5d  %7.3f ms/step' % (e.key[:60], e.count // 3, e.self_device_time_total / 3e3))
    sys.exit(0)
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print('total device ms per step: %.2f' % (tot / 3e3))
for e in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 28]:
    print('%-80s n=%5d  %7.2f ms/step  %5.1f%%' % (e.key[:80], e.count // 3, e.device_time_total / 3e3, 100 * e.device_time_total / tot))
