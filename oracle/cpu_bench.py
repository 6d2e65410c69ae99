"""CPU-baseline worker for bench.py: times the oracle (the CPU restatement of the reference) on whole clips, one
single-threaded process per clip.  TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/salsa_oracle.c header)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_clip(job):
    seed, n_samples, feature, fmt, fmax = job
    from oracle import oracle as orc
    from salsa_amd.synth import synth_clip
    orc.set_threads(1)
    y = synth_clip(seed, n_samples)                 # synthesis is not billed
    t0 = time.perf_counter()
    if feature == 'salsa':
        orc.extract_salsa(y, fmax_doa=fmax, audio_format=fmt)
    else:
        orc.extract_lite(y, fmax_doa=fmax, feature_type=feature)
    return time.perf_counter() - t0


def run(feature, fmt, fmax, n_samples, workers=None):
    """-> dict for bench.py's cpu_baseline: `workers` clips processed concurrently, one core each."""
    import multiprocessing as mp
    from oracle import oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    workers = workers or max(1, min(cores, 32))
    jobs = [(2021 + i, n_samples, feature, fmt, fmax) for i in range(workers)]
    with mp.get_context('spawn').Pool(workers) as pool:
        pool.map(run_clip, [(1, 24000, feature, fmt, fmax)] * workers)      # warm-up: imports + page-in
        t0 = time.perf_counter()
        per = pool.map(run_clip, jobs, chunksize=1)
        wall = time.perf_counter() - t0
    secs = n_samples / 24000.0
    busy = max(per)
    return {'value': round(workers * secs / busy, 2), 'unit': 'audio-seconds/s', 'cores': workers, 'kind': 'port',
            'sample': '%d x %.0f-s clips (seeds 2021..), one single-threaded process per clip running '
                      'oracle/salsa_oracle.c (float64 C restatement of the reference); slowest clip %.2f s, '
                      '%.1f core-s total, host has %d cores' % (workers, secs, busy, sum(per), cores),
            'single_core_value': round(secs / (sum(per) / len(per)), 2)}
