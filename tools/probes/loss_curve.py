#!/usr/bin/env python
"""Loss of a fixed synthetic batch over 60 training steps (seeded): a coarse end-to-end check that a kernel change did not
break a gradient.  Run with SALSA_* switches to compare configurations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd.crnn.train import Trainer, synthetic_batch
torch.manual_seed(0)
tr = Trainer('cuda:0')
x, sed, doa = synthetic_batch(32, 'cuda:0', seed=1)
out = []
for i in range(60):
    loss = tr.train_step(x, sed, doa)[0]
    if i % 10 == 9:
        out.append('%.4f' % float(loss))
print(' '.join(out))
