#!/usr/bin/env python
"""Known-byte-count workloads for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (the microarch guide: FETCH
reads 1/2 for wide coalesced streams; other widths and WRITE_SIZE must be calibrated in your own access pattern).
 1. torch copy of 1 GiB float32 (vectorised elementwise kernel): reads 1 GiB, writes 1 GiB.
 2. salsa_logspec_batch on 32 x 60-s clips: the STFT kernel's own access pattern (4 B/lane strided-frame reads, 4 B/lane
    row writes) with NO spill: reads >= 737 280 000 B of audio, writes exactly 491 622 400 B."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from salsa_amd.extractor import SalsaExtractor

dev = torch.device('cuda:0')
a = torch.randn(256 * 1024 * 1024, device=dev)
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
ex = SalsaExtractor(device=dev)
audio = 0.1 * torch.randn(32, 4, 1440000, device=dev)
for _ in range(3):
    ex.logspec(audio)
torch.cuda.synchronize()
print('calib done')
