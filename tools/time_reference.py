#!/usr/bin/env python
"""Same-container timing of the REFERENCE Python beside the port (round-3 review, "missing 1"; BASELINE.md:51-52).

Imports /root/reference unmodified through tools/ref_shims.py (build container only -- the reference never travels), runs its
own per-file loop ``extract_features(task='feature')`` (dataset/salsa_feature_extraction.py:339-385, which prints its own
wall clock) on a tree holding ONE seeded synthetic 60-s 4-channel clip (the bench's clip, seed 2021, BASELINE config 2's
parameters: FOA, n_fft 512, hop 300, fmax_doa 9000, cond 5, tracking on, compression on), then times the oracle
(oracle/salsa_oracle.c, one thread) on the SAME clip on the SAME core set, and checks the two outputs agree.  Writes
profiles/ref_vs_port.json: audio-s/s of both, and ratio = port / reference -- what bench.py's ``cpu_baseline`` divides its
on-box port figure by to quote ``reference_python_equiv`` (the reference's speed cannot be measured on the GPU box)."""
import contextlib
import io
import json
import os
import re
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
os.environ.setdefault('OMP_NUM_THREADS', '1')          # numpy / LAPACK of the reference: one thread, like the port's figure
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
os.environ.setdefault('MKL_NUM_THREADS', '1')

import make_golden as mg  # noqa: E402  (installs the shims, imports the reference; its __main__ block does not run)
from oracle import oracle as orc  # noqa: E402
from oracle.cpu_bench import effective_cpus, host_cpu  # noqa: E402
from salsa_amd.synth import synth_clip  # noqa: E402


def main(seconds=60.0, seed=2021, repeats=3):
    n = int(seconds * 24000)
    y = synth_clip(seed, n)
    ref_times, ref_printed, feat = [], [], None
    for _ in range(repeats):
        tmp = tempfile.mkdtemp()
        mg.ref_shims.H5_STORE.clear()
        cfg_path, feat_dir = mg.make_tree(tmp, 'foa', {'fold1_room1_mix001': y}, {}, fmax=9000)
        buf = io.StringIO()
        t0 = time.perf_counter()
        with np.errstate(all='ignore'), contextlib.redirect_stdout(buf):
            mg.ref_salsa.extract_features(data_config=cfg_path, cond_num=5, n_hopframes=3, is_tracking=True,
                                          is_compress_high_freq=True, task='feature')
        ref_times.append(time.perf_counter() - t0)
        ref_printed += [float(x) for x in re.findall(r'Elapsed time: ([0-9.]+) s', buf.getvalue())]
        tree = mg.collect(feat_dir)
        feat = [v for k, v in tree.items() if k.endswith('|feature')][0]
    orc.build()
    orc.set_threads(1)
    orc.extract_salsa(synth_clip(1, 24000))            # warm-up (library load)
    port_times = []
    for _ in range(max(3, repeats)):
        t0 = time.perf_counter()
        out = orc.extract_salsa(y, fmax_doa=9000, audio_format='foa')
        port_times.append(time.perf_counter() - t0)
    # the two must be computing the same thing
    assert out.shape == feat.shape, (out.shape, feat.shape)
    np.testing.assert_allclose(out[:4], feat[:4], rtol=1e-5, atol=2e-5)
    bad = np.abs(out[4:] - feat[4:]) > 1e-6 + 1e-5 * np.abs(feat[4:])
    assert bad.mean() < 1e-6, float(bad.mean())
    ref_s, port_s = float(np.median(ref_times)), float(np.median(port_times))
    model, phys, logical = host_cpu()
    rep = {
        'clip': {'seed': seed, 'seconds': seconds, 'format': 'foa', 'fmax_doa': 9000, 'cond_num': 5, 'tracking': True},
        'host': {'cpu': model, 'physical_cores': phys, 'logical_cpus': logical, **effective_cpus()},
        'threads': 1,
        'reference_python': {'wall_s_per_clip': ref_s, 'all_runs_s': ref_times, 'own_elapsed_print_s': ref_printed,
                             'audio_s_per_s': seconds / ref_s,
                             'what': 'dataset/salsa_feature_extraction.py extract_features(task="feature") on a one-clip tree, imported '
                                     'unmodified (tools/ref_shims.py stands in for librosa 0.8.0 / h5py / fire)'},
        'port_oracle': {'wall_s_per_clip': port_s, 'all_runs_s': port_times, 'audio_s_per_s': seconds / port_s,
                        'what': 'oracle/salsa_oracle.c extract (one thread) on the same clip'},
        'ratio_port_over_reference': ref_s / port_s,
        'outputs_agree': True,
        'numpy': np.__version__,
    }
    path = os.path.join(ROOT, 'profiles', 'ref_vs_port.json')
    json.dump(rep, open(path, 'w'), indent=1)
    print(json.dumps({k: rep[k] for k in ('ratio_port_over_reference',)}), rep['reference_python']['audio_s_per_s'],
          rep['port_oracle']['audio_s_per_s'], path)


if __name__ == '__main__':
    main(*(float(a) for a in sys.argv[1:2]))
