#!/usr/bin/env python
"""power_phases.py [K]: package power / shader clock of EACH feature kernel: the library's timing mode launches every kernel of one call K times back to back
(K = 2000: the STFT kernel for ~1 s, the tracker ~0.35 s, cov_eig ~0.7 s), the hwmon files are polled beside it (bench_crnn.PowerSampler) and the
timeline is cut at the kernels' own event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench_crnn import PowerSampler
from salsa_amd.extractor import SalsaExtractor
from salsa_amd.synth import synth_clip
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device('cuda:0')
au = torch.from_numpy(np.stack([synth_clip(2021 + i) for i in range(32)])).to(dev)
ex = SalsaExtractor()
for _ in range(3): ex.extract(au)
torch.cuda.synchronize()
ps = PowerSampler(dev); time.sleep(0.3)
ex.set_timing(K)
t0 = time.time(); ex.extract(au); torch.cuda.synchronize(); t1 = time.time()
tm = ex.read_timing(); ex.set_timing(0)
print('wall %.3f s;' % (t1 - t0), ', '.join('%s %.4f ms' % (n, ms) for n, ms in tm))
t = t0
for n, ms in tm:
    d = ms * K / 1e3
    st = ps.stats(t + 0.45 * d, t + 0.95 * d)      # the second half of the phase: the power figure is a slow average
    print('%-22s %6.3f s   %s' % (n, d, st and {k: st[k] for k in ('mean_w', 'max_w', 'sclk_mhz_mean', 'samples')}))
    t += d
ps.close()
