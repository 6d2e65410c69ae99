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


def effective_cpus():
    """-> dict: the CPUs this process may actually use.  os.cpu_count() reports the host's logical CPUs even inside a
    container whose cgroup grants a fraction of them: the scheduler affinity mask and the cgroup CPU quota (v2 cpu.max, v1
    cfs_quota_us / cfs_period_us) say what this process really gets."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = logical
    quota, quota_src = None, None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        quota_src = 'cgroup v2 cpu.max = %s %s' % (q, per)
        if q != 'max':
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            quota_src = 'cgroup v1 cfs_quota_us / cfs_period_us = %d / %d' % (q, per)
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    eff = affinity if quota is None else max(1, min(affinity, int(quota)))
    return {'effective_cores': eff, 'affinity_cpus': affinity, 'cgroup_cpu_quota': quota, 'cgroup_source': quota_src,
            'os_cpu_count': logical}


def _timed_pool(workers, jobs, feature, fmt, fmax):
    import multiprocessing as mp
    with mp.get_context('spawn').Pool(workers) as pool:
        pool.map(run_clip, [(1, 24000, feature, fmt, fmax)] * workers)      # warm-up: imports + page-in
        t0 = time.perf_counter()
        per = pool.map(run_clip, jobs, chunksize=1)
        wall = time.perf_counter() - t0
    return per, wall


def run(feature, fmt, fmax, n_samples, workers=None):
    """-> dict for bench.py's cpu_baseline.  Bounded samples of the same workload, one 60-s clip per single-threaded worker
    process, at up to three occupancies: every CPU this process may use (scheduler affinity and cgroup quota -- NOT
    os.cpu_count(), which reports the host even inside a container that owns a fraction of it), every physical core, and 32
    workers (round 1's figure), each capped at the effective count.  `value` / `cores` are the FASTEST of them; every
    configuration is listed under `configs`.  Rates are clips x seconds / the slowest worker's time."""
    from oracle import oracle as orc
    orc.build()
    model, phys, logical = host_cpu()
    secs = n_samples / 24000.0
    eff = effective_cpus()
    cap = eff['effective_cores']
    counts = [workers] if workers else sorted({max(1, min(logical, cap)), max(1, min(phys, cap)), max(1, min(cap, 32))}, reverse=True)
    configs = []
    for w in counts:
        jobs = [(2021 + i, n_samples, feature, fmt, fmax) for i in range(w)]
        per, wall = _timed_pool(w, jobs, feature, fmt, fmax)
        configs.append({'cores': w, 'value': round(w * secs / max(per), 2), 'per_core': round(secs / (sum(per) / len(per)), 2),
                        'slowest_clip_s': round(max(per), 2), 'core_s_total': round(sum(per), 1), 'pool_wall_s': round(wall, 2)})
    best = max(configs, key=lambda c: c['value'])
    # The reference Python cannot run on the GPU box (it never travels).  Its speed RELATIVE to this port was measured once, in the
    # build container, both on the same clip, core and numpy (tools/time_reference.py -> profiles/ref_vs_port.json: a committed data
    # file, read here); dividing the on-box port figure by that ratio quotes the port figure in "reference Python" terms.
    ref_equiv = None
    try:
        import json
        rv = json.load(open(os.path.join(ROOT, 'profiles', 'ref_vs_port.json')))
        ratio = float(rv['ratio_port_over_reference'])
        if feature == 'salsa' and ratio > 0:
            ref_equiv = {'value': round(best['value'] / ratio, 2), 'single_core_value': round(best['per_core'] / ratio, 2),
                         'ratio_port_over_reference': round(ratio, 3),
                         'provenance': 'profiles/ref_vs_port.json: reference extract_features() %.2f vs oracle %.2f audio-s/s on one core of '
                                       '%s (build container, clip seed %d, same numpy); assumes the ratio carries over to this host'
                                       % (rv['reference_python']['audio_s_per_s'], rv['port_oracle']['audio_s_per_s'], rv['host']['cpu'],
                                          rv['clip']['seed'])}
    except Exception:
        pass
    return {'value': best['value'], 'unit': 'audio-seconds/s', 'cores': best['cores'], 'kind': 'port',
            'reference_python_equiv': ref_equiv,
            'cpu_model': model, 'physical_cores': phys, 'logical_cpus': logical,
            'effective_cores': cap, 'affinity_cpus': eff['affinity_cpus'], 'cgroup_cpu_quota': eff['cgroup_cpu_quota'],
            'cgroup_source': eff['cgroup_source'],
            'sample': '%d x %.0f-s clips (seeds 2021..), one single-threaded process per worker running oracle/salsa_oracle.c '
                      '(float64 C restatement of the reference) concurrently; slowest clip %.2f s, %.1f core-s total; '
                      'fastest of the occupancies in `configs`' % (best['cores'], secs, best['slowest_clip_s'], best['core_s_total']),
            'single_core_value': best['per_core'], 'configs': configs}
