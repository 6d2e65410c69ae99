#!/usr/bin/env python
"""profiles/<tag>_pmc_summary.csv -> profiles/traffic.json: measured HBM bytes per launch of each kernel, from the
rocprofv3 PMC passes of tools/pmc_round.sh (separate --pmc runs, --kernel-trace only), corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 64 B per 128-B request (x2) -- confirmed here by profiles/*_pmc_calibration.csv (a 720 000 KiB
elementwise read reports 360 036; the STFT kernel's 4-B/lane audio reads obey the same factor) -- and WRITE_SIZE needs
no factor (720 000 KiB written reports 720 000; the STFT kernel's row writes report 1.034x their exact byte count).
usage: tools/pmc_traffic.py profiles/r1_v7_pmc_summary.csv [batch] [older_summary.csv]
(a counter missing from the first summary -- a pass that timed out -- is taken from the older one and named in 'source')"""
import csv
import json
import os
import sys

src = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
name = {'stft_kernel': 'stft_logspec', 'tracker_kernel': 'noise_floor_tracker', 'cov_eig_kernel': 'cov_eig'}
acc, origin = {}, {}
for path in [src] + sys.argv[3:4]:
    for row in csv.DictReader(open(path)):
        k = name.get(row['kernel'])
        if k and row['counter'] in ('FETCH_SIZE', 'WRITE_SIZE') and row['counter'] not in acc.get(k, {}):
            acc.setdefault(k, {})[row['counter']] = float(row['mean_per_dispatch'])
            origin[row['counter']] = os.path.basename(path)
out = {'source': ', '.join('%s from %s' % (c, f) for c, f in sorted(origin.items())), 'batch_clips_per_launch': batch, 'correction': 'FETCH_SIZE KiB x2 (gfx950), WRITE_SIZE KiB x1',
       'kernels': {}}
for k, v in acc.items():
    rd, wr = v.get('FETCH_SIZE', 0) * 1024 * 2, v.get('WRITE_SIZE', 0) * 1024
    out['kernels'][k] = {'hbm_read_bytes': int(rd), 'hbm_write_bytes': int(wr), 'hbm_bytes': int(rd + wr)}
path = os.path.join(os.path.dirname(src), 'traffic.json')
json.dump(out, open(path, 'w'), indent=1)
print(json.dumps(out, indent=1))
