#!/usr/bin/env python
"""timing_probe.py -- how should the feature step be attributed to its three kernels?  (round 3, VERDICT r2 item 1a: the
event-bracketed per-kernel times of round 2 summed to 1.30 ms against a 1.14-ms step.)  On one 32 x 60-s batch:
  plain     K steps, wall clock                                   -> the step
  pairs     an event pair around every launch in the real sequence (salsa_plan_set_timing(1)); also the wall time of THAT
  repeat    K back-to-back launches of each kernel between one event pair (set_timing(K))
  prefix    wall time of K issues of [STFT], [STFT, tracker], [STFT, tracker, cov_eig] (set_timing(-1 / -2 / 0)): differences
            add up to the step exactly
  empty     elapsed time of an event pair with nothing between
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_batch  # noqa: E402
from salsa_amd.extractor import SalsaExtractor  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
host = make_batch(2021, 32, 60 * 24000)
audio = torch.from_numpy(host).to(dev)
ex = SalsaExtractor(audio_format='foa', fmax_doa=9000, device=dev)
out = torch.empty((32,) + tuple(ex.output_shape(60 * 24000)), device=dev)


def wall(n=K, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ex.extract(audio, out=out)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return float(np.median(ts))


for _ in range(5):
    ex.extract(audio, out=out)
res = {'K': K}
res['plain_ms'] = wall()
ex.set_timing(1)
res['pairs_step_wall_ms'] = wall()
acc = {}
for _ in range(K):
    ex.extract(audio, out=out)
    for name, ms in ex.read_timing():
        acc.setdefault(name, []).append(ms)
res['pairs'] = {k: float(np.mean(v)) for k, v in acc.items()}
res['pairs_sum'] = sum(res['pairs'].values())
ex.set_timing(K)
ex.extract(audio, out=out)
res['repeat'] = dict(ex.read_timing())
res['repeat_sum'] = sum(res['repeat'].values())
ex.set_timing(-1)
p1 = wall()
ex.set_timing(-2)
p2 = wall()
ex.set_timing(0)
p3 = wall()
res['prefix'] = {'stft_logspec': p1, 'noise_floor_tracker': p2 - p1, 'cov_eig': p3 - p2}
res['prefix_sum'] = p3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
em = []
for _ in range(50):
    e0.record(); e1.record()
    torch.cuda.synchronize()
    em.append(e0.elapsed_time(e1))
res['empty_event_pair_ms'] = float(np.median(em))
print(json.dumps(res, indent=1))
