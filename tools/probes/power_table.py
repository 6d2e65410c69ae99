#!/usr/bin/env python
"""power_table.py <seconds>: one kernel at a time in a loop while rocm-smi is sampled (package power, shader clock): a device copy, the
BatchNorm apply pass, the 64 -> 64 forward convolution, a wide convolution, the three feature kernels -- which ones sit at the 1400-W cap?"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
from salsa_amd.extractor import SalsaExtractor
from salsa_amd.synth import synth_clip
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device('cuda:0'); L = _lib.load(); g = torch.Generator(device=dev).manual_seed(0)
def bf(shape): return torch.randn(shape, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
samples = []
def sampler(stop):
    while not stop.is_set():
        o = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
        p = re.search(r'Package Power \(W\): ([\d.]+)', o); s = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', o)
        if p and s: samples.append((time.time(), float(p.group(1)), int(s.group(1))))
def run(name, fn, work_note):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    t1 = time.time()
    mid = [(p, s) for (t, p, s) in samples if t0 + 1.0 < t < t1 - 0.3]
    us = (t1 - t0) / n * 1e6
    if mid: print('%-44s %8.1f us/launch   power %5.0f W (max %5.0f)   sclk %4.0f MHz   %s' % (name, us, np.mean([m[0] for m in mid]), max(m[0] for m in mid), np.mean([m[1] for m in mid]), work_note(us)), flush=True)
    else: print(name, us, 'no samples')
stop = threading.Event(); th = threading.Thread(target=sampler, args=(stop,)); th.start()
a = torch.empty(1 << 30, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
run('device copy 1 GiB', lambda: b.copy_(a), lambda us: '%.2f TB/s read + write' % (2 * (1 << 30) / us / 1e6))
del a, b
xs = [bf((32, 64, 640, 200)) for _ in range(2)]; w64 = (torch.randn((64, 64, 3, 3), device=dev, generator=g) * 0.06).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
it = [0]
def c64():
    it[0] += 1; nn_ops._conv64(xs[it[0] & 1], w64)
run('64 -> 64 forward, 32 x 640 x 200', c64, lambda us: '%.0f TFLOP/s, %.2f TB/s' % (2.0 * 32 * 640 * 200 * 64 * 64 * 9 / us / 1e6, 2 * xs[0].numel() * 2 / us / 1e6))
y = torch.empty_like(xs[0]); mean = torch.zeros(64, device=dev); inv = torch.ones(64, device=dev); ga = torch.ones(64, device=dev); be = torch.zeros(64, device=dev)
M = 32 * 640 * 200
def bn():
    it[0] += 1
    L.salsa_nn_bn_eval_fwd(nn_ops._ptr(xs[it[0] & 1]), nn_ops._ptr(y), None, 1, M, 64, nn_ops._ptr(ga), nn_ops._ptr(be), nn_ops._ptr(mean), nn_ops._ptr(inv), 1, nn_ops._stream(y))
run('BatchNorm apply pass, same map', bn, lambda us: '%.2f TB/s read + write' % (2 * xs[0].numel() * 2 / us / 1e6))
del xs, y
for cin, H, W in ((128, 160, 50), (256, 80, 25), (512, 40, 12)):
    xw = bf((32, cin, H, W)); ww = (torch.randn((cin, cin, 3, 3), device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    run('wide forward %d -> %d, 32 x %d x %d' % (cin, cin, H, W), lambda: nn_ops._conv_wide(xw, ww), lambda us: '%.0f TFLOP/s' % (2.0 * 32 * H * W * cin * cin * 9 / us / 1e6))
ys = np.stack([synth_clip(2021 + i) for i in range(32)]); au = torch.from_numpy(ys).to(dev); ex = SalsaExtractor()
run('feature path (three kernels), 32 x 60 s', lambda: ex.extract(au), lambda us: '%.2f TB/s algorithmic, 4.34 GB real' % (1597619200 / us / 1e6))
with ex.issue_prefix(1):
    run('STFT kernel alone (prefix issue)', lambda: ex.extract(au), lambda us: '%.2f TB/s real (2.34 GB)' % (2.34e9 / us / 1e6))
with ex.issue_prefix(2):
    run('STFT + tracker (prefix issue)', lambda: ex.extract(au), lambda us: '')
ex.set_fused(1)
run('fused schedule (STFT, tracker, fused STFT + cov/eig)', lambda: ex.extract(au), lambda us: '')
stop.set(); th.join()
