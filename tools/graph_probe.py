#!/usr/bin/env python
"""Probe: salsa_extract_batch captured in a HIP graph with the clip-group pipeline at different depths."""
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
for G in (1, 2, 3, 4, 8):
    ex.set_groups(G)
    for _ in range(3):
        ex.extract(audio, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ex.extract(audio, out=out)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            ex.extract(audio, out=out)
    out.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 20
    print('groups %d  eager %.4f ms  graph replay %.4f ms  equal %s' % (G, 1e3 * eager, 1e3 * rep, bool(torch.equal(out, ref))))
