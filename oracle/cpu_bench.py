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


def host_cpu():
    """-> (model string, physical cores, logical CPUs) of this host from /proc/cpuinfo."""
    model, phys = 'unknown', set()
    try:
        pid = cid = None
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name') and model == 'unknown':
                model = ln.split(':', 1)[1].strip()
            elif ln.startswith('physical id'):
                pid = ln.split(':', 1)[1].strip()
            elif ln.startswith('core id'):
                cid = ln.split(':', 1)[1].strip()
            elif not ln.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(phys) or logical), logical


def _timed_pool(workers, jobs, feature, fmt, fmax):
    import multiprocessing as mp
    with mp.get_context('spawn').Pool(workers) as pool:
        pool.map(run_clip, [(1, 24000, feature, fmt, fmax)] * workers)      # warm-up: imports + page-in
        t0 = time.perf_counter()
        per = pool.map(run_clip, jobs, chunksize=1)
        wall = time.perf_counter() - t0
    return per, wall


def run(feature, fmt, fmax, n_samples, workers=None):
    """-> dict for bench.py's cpu_baseline.  Two bounded samples of the same workload, one 60-s clip per single-threaded
    worker process: (i) ALL logical CPUs of the host busy at once (`value`, `cores`), (ii) 32 workers (`cores32`, less
    memory-bandwidth contention per core).  Rates are clips x seconds / wall time of the slowest worker."""
    from oracle import oracle as orc
    orc.build()
    model, phys, logical = host_cpu()
    secs = n_samples / 24000.0
    out = {}
    for tag, w in (('all', workers or logical), ('c32', max(1, min(logical, 32)))):
        if tag == 'c32' and w >= (workers or logical):
            continue
        jobs = [(2021 + i, n_samples, feature, fmt, fmax) for i in range(w)]
        per, wall = _timed_pool(w, jobs, feature, fmt, fmax)
        out[tag] = (w, per, wall)
    w, per, wall = out['all']
    busy = max(per)
    res = {'value': round(w * secs / busy, 2), 'unit': 'audio-seconds/s', 'cores': w, 'kind': 'port',
           'cpu_model': model, 'physical_cores': phys, 'logical_cpus': logical,
           'sample': '%d x %.0f-s clips (seeds 2021..), one single-threaded process per logical CPU running '
                     'oracle/salsa_oracle.c (float64 C restatement of the reference) concurrently; slowest clip %.2f s, '
                     '%.1f core-s total, pool wall %.2f s' % (w, secs, busy, sum(per), wall),
           'single_core_value': round(secs / (sum(per) / len(per)), 2)}
    if 'c32' in out:
        w2, per2, _ = out['c32']
        res['cores32'] = {'value': round(w2 * secs / max(per2), 2), 'cores': w2,
                          'per_core': round(secs / (sum(per2) / len(per2)), 2),
                          'note': '%d workers only (round 1\'s figure): less contention per core' % w2}
    return res
