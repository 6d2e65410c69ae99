#!/usr/bin/env python
"""Kernel breakdown of one inference sub-batch (SALSA features + CRNN forward), torch.profiler.  usage: [clips=32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from salsa_amd.crnn.train import Trainer
from salsa_amd.extractor import SalsaExtractor

dev = torch.device('cuda:0')
tr = Trainer(dev)
ex = SalsaExtractor(audio_format='foa', fmax_doa=9000, device=dev)
ex.set_scaler(torch.full((4, 1, 200), -60.0, device=dev), torch.full((4, 1, 200), 12.0, device=dev))
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
audio = 0.1 * torch.randn(NB, 4, 60 * 24000, device=dev)
def step():
    f = ex.extract(audio)[:, :, :4800]
    return tr.infer(f)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print('total device ms per %d-clip sub-batch: %.2f' % (NB, tot / 3e3))
for e in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 20]:
    print('%-80s n=%5d  %7.2f ms  %5.1f%%' % (e.key[:80], e.count // 3, e.device_time_total / 3e3, 100 * e.device_time_total / tot))
