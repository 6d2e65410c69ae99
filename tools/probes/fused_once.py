"""A few plain extract() calls of the config-2 batch in one schedule (for counter passes): python tools/probes/fused_once.py <mode>"""
import sys

import torch

sys.path.insert(0, '.')
from bench import make_batch  # noqa: E402
from salsa_amd.extractor import SalsaExtractor  # noqa: E402

dev = torch.device('cuda:0')
a = torch.from_numpy(make_batch(2021, 32, 60 * 24000)).to(dev)
ex = SalsaExtractor(device=dev)
ex.set_fused(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for _ in range(4):
    ex.extract(a)
torch.cuda.synchronize()
