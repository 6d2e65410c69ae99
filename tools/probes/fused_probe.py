"""Round 5: the fused STFT + covariance / eigen kernel against the three-kernel path on one box -- bit-equality of the outputs
(same arithmetic, so nothing less) on the bench batch, ragged and MIC batches, then per-kernel times (event pairs, K launches
per pair) and whole-step wall times of both schedules, alternating.
  python tools/probes/fused_probe.py [--quick] [--modes 0,1,...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from bench import make_batch  # noqa: E402
from salsa_amd.extractor import SalsaExtractor  # noqa: E402
from salsa_amd.synth import synth_clip  # noqa: E402

dev = torch.device('cuda:0')
quick = '--quick' in sys.argv
modes = [0, 1]
for a in sys.argv:
    if a.startswith('--modes='):
        modes = [int(x) for x in a.split('=')[1].split(',')]


def run(ex, a, mode):
    ex.set_fused(mode)
    return ex.extract(a).clone()


def check(name, ys, **kw):
    a = torch.from_numpy(ys).to(dev)
    ex = SalsaExtractor(device=dev, **kw)
    ref = run(ex, a, 0)
    for m in modes[1:]:
        out = run(ex, a, m)
        same = torch.equal(out, ref)
        d = (out - ref).abs()
        nz = int((d > 0).sum())
        print('%-28s mode %d: bit-equal %s  (differing elements %d, max |diff| %.3g; spec %d, spatial %d)' % (
            name, m, same, nz, float(d.max()), int((d[:, :4] > 0).sum()), int((d[:, 4:] > 0).sum())), flush=True)


check('foa 3 x 2.0 s', np.stack([synth_clip(7 + i, 48000) for i in range(3)]))
check('foa 2 x 0.61 s (ragged T)', np.stack([synth_clip(17 + i, 14700) for i in range(2)]))
check('mic 4 x 8 s', np.stack([synth_clip(60 + i, 8 * 24000) for i in range(4)]), audio_format='mic', fmax_doa=4000)
check('foa interleaved 2 x 3 s', np.ascontiguousarray(np.stack([synth_clip(27 + i, 72000) for i in range(2)]).transpose(0, 2, 1)),
      audio_layout='interleaved')
B = 8 if quick else 32
ys = make_batch(2021, B, 60 * 24000)
check('foa %d x 60 s (config 2)' % B, ys)

a = torch.from_numpy(ys).to(dev)
ex = SalsaExtractor(device=dev)
for rnd in range(2 if quick else 3):
    for m in modes:
        ex.set_fused(m)
        ex.set_timing(0)
        for _ in range(3):
            ex.extract(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            ex.extract(a)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / K * 1e3
        ex.set_timing(10)
        ex.extract(a)
        tm = ex.read_timing()
        ex.set_timing(0)
        print('round %d mode %d: step %.4f ms | %s' % (rnd, m, wall, '  '.join('%s %.4f' % kv for kv in tm)), flush=True)
