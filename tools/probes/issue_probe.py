#!/usr/bin/env python
"""Is the CRNN training step bound by the host's launch rate?  Time to ISSUE 20 steps (no sync) against the time until they
have finished, at batch 32 and at batch 4 (whose wall time is the host floor)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd.crnn.train import Trainer, synthetic_batch
dev = torch.device('cuda:0')
for batch in (32, 4):
    tr = Trainer(dev)
    x, sed, doa = synthetic_batch(batch, dev, seed=1)
    for _ in range(5):
        tr.train_step(x, sed, doa)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.train_step(x, sed, doa)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('batch %2d: issue %.2f ms/step, finished %.2f ms/step' % (batch, (t1 - t0) * 50, (t2 - t0) * 50))
