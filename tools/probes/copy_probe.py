#!/usr/bin/env python
"""Which tensors do the aten::copy_ / add_ / fill_ / sum kernels of one CRNN training step move?  (torch.profiler, by input shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
from salsa_amd.crnn.train import Trainer, synthetic_batch
tr = Trainer('cuda:0')
x, sed, doa = synthetic_batch(32, 'cuda:0', seed=1)
for _ in range(6):
    tr.train_step(x, sed, doa)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    for _ in range(3):
        tr.train_step(x, sed, doa)
    torch.cuda.synchronize()
evs = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0 and e.key in sys.argv[1:]]
evs.sort(key=lambda e: -e.self_device_time_total)
for e in evs[:int(os.environ.get('TOP', 40))]:
    print('%-12s n=%3d %8.1f us/step  %s' % (e.key[6:], e.count // 3, e.self_device_time_total / 3, str(e.input_shapes)[:150]))
