#!/usr/bin/env python
"""profiles/<tag>_pmc_summary.csv -> profiles/traffic.json: measured HBM bytes per launch of each kernel, from the
rocprofv3 PMC passes of tools/pmc_round.sh (separate --pmc runs, --kernel-trace only), corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 64 B per 128-B request (x2) -- confirmed here by profiles/*_pmc_calibration.csv (a 720 000 KiB
elementwise read reports 360 036; the STFT kernel's 4-B/lane audio reads obey the same factor) -- and WRITE_SIZE needs
no factor (720 000 KiB written reports 720 000; the STFT kernel's row writes report 1.034x their exact byte count).
Also derives each kernel's VALU utilisation = SQ_ACTIVE_INST_VALU (quad-cycles, summed over the chip's SIMDs) x 4 /
(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the fraction of SIMD cycles in which a vector instruction was executing -- for these
float64 kernels the "what bounds it besides HBM" figure SURVEY section 8(d) asks for.
usage: tools/pmc_traffic.py profiles/r1_v7_pmc_summary.csv [batch] [older_summary.csv]   (run in the build container right after the
counter passes: stamps the commit and a hash of salsa_kernels.hip + salsa_math.h, which bench.py compares with the sources it runs)
(a counter missing from the first summary -- a pass that timed out -- is taken from the older one and named in 'source')"""
import csv
import json
import os
import sys

src = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
name = {'stft_kernel': 'stft_logspec', 'tracker_kernel': 'noise_floor_tracker', 'cov_eig_kernel': 'cov_eig'}
# (gate_doubt_kernel, launched behind cov_eig_kernel since round 6, reads the 3.7-MB doubt mask and exits: its bytes are added to cov_eig's)
extra = {'gate_doubt_kernel': 'cov_eig'}
acc, origin, addl = {}, {}, {}
for path in [src] + sys.argv[3:4]:
    rows = []
    for ln in open(path).read().splitlines()[1:]:            # kernel names may carry commas (template arguments): split from the right
        parts = ln.rsplit(',', 3)
        if len(parts) == 4:
            rows.append({'kernel': parts[0], 'counter': parts[1], 'mean_per_dispatch': parts[2]})
    for row in rows:
        ke = next((v for kk, v in extra.items() if row['kernel'].startswith(kk)), None)
        if ke and path == src and row['counter'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            addl.setdefault(ke, {}).setdefault(row['counter'], 0.0)
            addl[ke][row['counter']] += float(row['mean_per_dispatch'])
        k = next((v for kk, v in name.items() if row['kernel'].startswith(kk)), None)   # (summaries may keep the template arguments)
        if k and row['counter'] in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_ACTIVE_INST_VALU', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_VALU') and row['counter'] not in acc.get(k, {}):
            acc.setdefault(k, {})[row['counter']] = float(row['mean_per_dispatch'])
            origin[row['counter']] = os.path.basename(path)
import hashlib
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    commit = subprocess.check_output(['git', '-C', ROOT, 'rev-parse', '--short=12', 'HEAD']).decode().strip()
except Exception:
    commit = None
ksha = hashlib.sha256(b''.join(open(os.path.join(ROOT, 'salsa_amd', 'csrc', f), 'rb').read() for f in ('salsa_kernels.hip', 'salsa_math.h'))).hexdigest()[:16]
out = {'source': ', '.join('%s from %s' % (c, f) for c, f in sorted(origin.items())), 'batch_clips_per_launch': batch,
       'commit': commit, 'kernel_sources_sha16': ksha, 'correction': 'FETCH_SIZE KiB x2 (gfx950), WRITE_SIZE KiB x1; valu_util = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)',
       'kernels': {}}
for k, v in acc.items():
    rd = (v.get('FETCH_SIZE', 0) + addl.get(k, {}).get('FETCH_SIZE', 0.0)) * 1024 * 2
    wr = (v.get('WRITE_SIZE', 0) + addl.get(k, {}).get('WRITE_SIZE', 0.0)) * 1024
    out['kernels'][k] = {'hbm_read_bytes': int(rd), 'hbm_write_bytes': int(wr), 'hbm_bytes': int(rd + wr)}
    if v.get('SQ_ACTIVE_INST_VALU') and v.get('GRBM_GUI_ACTIVE'):
        out['kernels'][k]['valu_util'] = round(v['SQ_ACTIVE_INST_VALU'] * 4 / (v['GRBM_GUI_ACTIVE'] / 8 * 1024), 4)
        out['kernels'][k]['valu_wave_instructions'] = int(v.get('SQ_INSTS_VALU', 0))
        out['kernels'][k]['gpu_cycles'] = int(v['GRBM_GUI_ACTIVE'] / 8)
path = os.path.join(os.path.dirname(src), 'traffic.json')
json.dump(out, open(path, 'w'), indent=1)
print(json.dumps(out, indent=1))
