"""Per-kernel times (K launches per event pair) + step wall time of one schedule of the config-2 batch; SALSA_HIP_LIB selects a
probe build.  python tools/probes/fused_time.py <mode> [B]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from bench import make_batch  # noqa: E402
from salsa_amd.extractor import SalsaExtractor  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device('cuda:0')
a = torch.from_numpy(make_batch(2021, B, 60 * 24000)).to(dev)
ex = SalsaExtractor(device=dev)
ex.set_fused(mode)
for _ in range(3):
    ex.extract(a)
torch.cuda.synchronize()
res = []
for rnd in range(2):
    t0 = time.perf_counter()
    for _ in range(20):
        ex.extract(a)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    ex.set_timing(10)
    ex.extract(a)
    tm = ex.read_timing()
    ex.set_timing(0)
    res.append('step %.4f | %s' % (wall, '  '.join('%s %.4f' % (k[:12], v) for k, v in tm)))
print(' || '.join(res), flush=True)
