#!/usr/bin/env python
"""fused_nt_debug.py: where do the fused schedule and the three-kernel path differ (session 3: after the NT stores)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from salsa_amd.synth import synth_clip
from salsa_amd.extractor import SalsaExtractor
dev = torch.device('cuda:0')
ys = np.stack([synth_clip(7 + i, 48000) for i in range(3)])
a = torch.from_numpy(ys).to(dev)
ex = SalsaExtractor()
refs = [ex.extract(a).clone() for _ in range(3)]
print('three-kernel path, run to run equal:', [bool(torch.equal(refs[0], r)) for r in refs[1:]])
ex.set_fused(1)
outs = [ex.extract(a).clone() for _ in range(3)]
print('fused, run to run equal:', [bool(torch.equal(outs[0], r)) for r in outs[1:]])
d = (outs[0] != refs[0])
print('differ:', int(d.sum()), 'by channel', [int(d[:, c].sum()) for c in range(7)])
idx = d.nonzero()
if len(idx):
    print('clips', idx[:, 0].unique().tolist(), 'frames', idx[:, 2].min().item(), '..', idx[:, 2].max().item(), 'bins', idx[:, 3].min().item(), '..', idx[:, 3].max().item())
    fr = idx[:, 2].unique().tolist()
    print('n frames', len(fr), fr[:40])
    i = idx[0].tolist()
    print('first', i, float(outs[0][tuple(i)]), float(refs[0][tuple(i)]))
    nzr = (refs[0][:, 4:] != 0).sum().item(); nzo = (outs[0][:, 4:] != 0).sum().item()
    print('nonzero ch4-6: ref', nzr, 'fused', nzo)
