"""Where the file-to-file harness spends its time (round 6): a 64-clip tree of 60-s clips on tmpfs through salsa_amd.features._extract_split
with the stage timers of _FilePipeline, plus the pieces around it (rmtree of the previous output, listing)."""
import os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_batch
from salsa_amd import features
from salsa_amd.extractor import SalsaExtractor
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
host = make_batch(2021, 32, 60 * 24000)
tmp = tempfile.mkdtemp(dir='/dev/shm')
d = os.path.join(tmp, 'in'); os.makedirs(d)
for i in range(n_clips):
    np.save(os.path.join(d, 'clip%03d.npy' % i), host[i % 32])
out = os.path.join(tmp, 'out')
ex = SalsaExtractor()
for rep in range(3):
    t0 = time.perf_counter(); features._rmtree_parallel(out); t_rm = time.perf_counter() - t0
    stats = {}
    t0 = time.perf_counter()
    features._extract_split(ex, d, out, 24000, 32, clear=False, stats=stats)
    dt = time.perf_counter() - t0
    print('rep %d: rmtree %.3f s, split %.3f s (%.0f audio-s/s incl. rmtree), stats %s' % (rep, t_rm, dt, n_clips * 60 / (dt + t_rm), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in stats.items()}))
shutil.rmtree(tmp)
