#!/usr/bin/env python
"""Training trajectories with the fused GRU scans against torch.nn.GRU (MIOpen) in the same model: losses of the first steps and
every 10th of 60, with the GRU's inter-layer dropout on (0.3, the reference's) and off.  With dropout off the two must agree to
rounding; with it on they differ only by the masks drawn."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd.crnn import model as M
from salsa_amd.crnn.train import Trainer, synthetic_batch

x, sed, doa = synthetic_batch(32, 'cuda:0', seed=1)
for p in (0.0, 0.3):
    for fused in (True, False):
        M.FUSED_GRU = fused
        torch.manual_seed(0)
        tr = Trainer('cuda:0')
        tr.raw_model.decoder.gru.dropout = p
        out = []
        for i in range(60):
            loss = tr.train_step(x, sed, doa)[0]
            if i < 3 or i % 10 == 9:
                out.append('%.4f' % float(loss))
        print('gru dropout %.1f  %-22s %s' % (p, 'fused scans' if fused else 'torch.nn.GRU (MIOpen)', ' '.join(out)), flush=True)
