#!/usr/bin/env python
"""bn_bits_ab.py: the training step with the ReLU bit planes on / off, alternating inside ONE process (same box, same clocks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd.crnn import nn_ops
from salsa_amd.crnn.train import Trainer, synthetic_batch
tr = Trainer('cuda:0', total_steps=1000)
batches = [synthetic_batch(32, 'cuda:0', seed=s) for s in range(4)]
for i in range(8):
    tr.train_step(*batches[i % 4])
torch.cuda.synchronize()
res = {True: [], False: []}
for rnd in range(6):
    for on in (False, True):
        nn_ops.USE_BN_RELU_BITS = on
        for i in range(3):
            tr.train_step(*batches[i % 4])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            tr.train_step(*batches[i % 4])
        e1.record()
        torch.cuda.synchronize()
        res[on].append(e0.elapsed_time(e1) / 12)
for on in (False, True):
    print('bits', on, ' '.join('%.3f' % t for t in res[on]), ' median %.3f ms' % sorted(res[on])[len(res[on]) // 2])
