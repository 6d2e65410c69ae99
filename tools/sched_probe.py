#!/usr/bin/env python
"""Probe (GPU box): wall time per step of salsa_extract_batch under the pipelined schedules (clip groups x STFT split by
channel pair x eager / hipGraph replay), each checked bit-identical to the plain three-kernel run.
  python tools/sched_probe.py                 -> sweep, one line per schedule
  python tools/sched_probe.py G SPLIT GRAPH   -> 6 steps of that one schedule (run under rocprofv3 --kernel-trace for a timeline)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from salsa_amd.extractor import SalsaExtractor
from bench import make_batch

host = make_batch(2021, 32, 1440000)
dev = torch.device('cuda:0')
audio = torch.from_numpy(host).to(dev)
ex = SalsaExtractor(device=dev)
out = torch.empty((32,) + tuple(ex.output_shape(1440000)), dtype=torch.float32, device=dev)
ex.extract(audio, out=out)
ref = out.clone()


def timed(n=20, reps=3):
    best = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ex.extract(audio, out=out)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / n)
    return sorted(best)[len(best) // 2]


if len(sys.argv) > 3:
    G, split, graph = int(sys.argv[1]), bool(int(sys.argv[2])), bool(int(sys.argv[3]))
    ex.set_pipeline(G, split, graph)
    for _ in range(6):
        ex.extract(audio, out=out)
    torch.cuda.synchronize()
    print('G %d split %d graph %d equal %s' % (G, split, graph, bool(torch.equal(out, ref))))
    sys.exit(0)

for G, split, graph in [(1, 0, 0), (1, 1, 0), (1, 1, 1), (2, 0, 1), (2, 1, 1), (4, 0, 1), (4, 1, 1), (8, 0, 1), (8, 1, 1), (16, 1, 1), (2, 1, 0), (4, 1, 0)]:
    ex.set_pipeline(G, bool(split), bool(graph))
    out.zero_()
    for _ in range(3):
        ex.extract(audio, out=out)
    torch.cuda.synchronize()
    eq = bool(torch.equal(out, ref))
    print('groups %2d  split %d  graph %d   %.4f ms/step   equal %s' % (G, split, graph, 1e3 * timed(), eq), flush=True)
