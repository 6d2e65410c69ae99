"""Which host lines launch the small torch kernels (fills, copies, reductions) of one CRNN training step?
torch.profiler with stacks, a few steps; prints, per aten op that reaches the GPU, the count per step, the device time per step
and the innermost frame under salsa_amd/ (or bench_crnn.py) that issued it.

    python tools/probes/small_kernels_probe.py [--steps 4] > gpurun_out/small_kernels.txt
"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--only-small', action='store_true', help='list only the fills / copies / small elementwise ops')
    ap.add_argument('--config4', action='store_true', help='on-the-fly SALSA-MIC extraction + device augmentation in front of the step')
    a = ap.parse_args()
    from salsa_amd.crnn.train import Trainer, synthetic_batch
    dev = torch.device('cuda:0')
    tr = Trainer(dev)
    x, sed, doa = synthetic_batch(a.batch, dev, seed=2021)
    step = lambda: tr.train_step(x, sed, doa)   # noqa: E731
    if a.config4:
        import numpy as np
        from salsa_amd.augment import augment_batch
        from salsa_amd.extractor import SalsaExtractor
        from salsa_amd.synth import synth_clip
        ex = SalsaExtractor(audio_format='mic', fmax_doa=4000, device=dev)
        ex.set_scaler(torch.full((4, 1, 200), -60.0, device=dev), torch.full((4, 1, 200), 12.0, device=dev))
        audio = torch.from_numpy(np.stack([synth_clip(4021 + i, 8 * 24000) for i in range(a.batch)])).to(dev)
        gen = torch.Generator().manual_seed(2021)

        def step():
            xb = ex.extract(audio)[:, :, :640]
            xb, sb, db = augment_batch(xb, sed, doa, 'mic', gen=gen)
            return tr.train_step(xb, sb, db)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, 'self_device_time_total', None)
        if dt is None:
            dt = getattr(ev, 'self_cuda_time_total', 0)
        if not dt or ev.device_type != torch.autograd.DeviceType.CPU:
            continue
        site = '?'
        for fr in ev.stack or []:
            if 'salsa_amd' in fr and 'small_kernels_probe' not in fr:
                site = fr.split('salsa_amd/')[-1]
                break
        if site == '?' and a.only_small:
            site = ' <- '.join(f.split('/')[-1] for f in (ev.stack or [])[:3])
        shapes = str(ev.input_shapes)[:60] if ev.input_shapes else ''
        k = (ev.name, site, shapes)
        agg[k][0] += 1
        agg[k][1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values())
    print('device time attributed to aten ops: %.3f ms/step' % (tot / a.steps / 1e3))
    print('%-34s %7s %9s  %s' % ('op', 'n/step', 'us/step', 'site  shapes'))
    small = ('fill', 'zero', 'copy', 'cat', 'add', 'mul', 'div', 'sum', 'mean', 'where', 'clone', 'dropout', 'sigmoid', 'tanh', 'sub', 'neg', 'sqrt')
    if a.only_small:
        rows = [r for r in rows if any(t in r[0][0] for t in small)]
        print('small ops: %.3f ms/step' % (sum(v[1] for _, v in rows) / a.steps / 1e3))
    for (name, site, shapes), (n, us) in rows[:120]:
        print('%-34s %7.1f %9.1f  %s  %s' % (name[:34], n / a.steps, us / a.steps, site, shapes))


if __name__ == '__main__':
    main()
