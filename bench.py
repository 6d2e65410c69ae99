#!/usr/bin/env python
"""bench.py -- BASELINE.json's two-part metric on N MI355X: SALSA feature-extraction audio-s/s (config 2, the headline
`value`) and CRNN training clips/s (config 3, the `crnn` object of the same JSON line).

Feature step = one pass of the hot path (STFT -> log-spectrogram + noise gate + 4x4 covariance -> principal
eigenvector) over one batch of 32 synthetic 60-s 4-channel 24-kHz clips already resident in HBM, FOA parameters
(fmax_doa 9000, cond 5, tracking on, high-frequency compression on), output [32][7][4801][200] float32 in HBM.  Clips
shard across ranks with no collective on the data path (weak scaling: 32 clips per GPU).  The K-step timed block
(barrier + synchronize on both sides, MAX over ranks) is repeated `--blocks` times and the MEDIAN block is reported
(box-to-box and run-to-run spread of a 24-ms region is several percent); every block's time is in `blocks_ms`.

CRNN step (after the feature path, its own timed region) = forward + loss + backward + Adam on 32 chunks (7,640,200) per
GPU, bf16 autocast, bucketed gradient all-reduce over RCCL for N > 1 (bench_crnn.train_bench, salsa_amd/crnn/grad_sync.py);
at N = 1 the same step on stock PyTorch-ROCm kernels is timed beside it (`crnn.torch_baseline`, a stated baseline).
Further legs of the same line (reported, never `value`): `config4` = on-the-fly SALSA-MIC + augmentation + training step;
`inference` = BASELINE config 5 at its real size, --infer-clips (1024) distinct 60-s clips sharded over the ranks through
SALSA-FOA + CRNN forward + DCASE rows on the host, with per-clip latency percentiles (bench_crnn.infer_bench).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus 8            # no launcher needed: re-executes itself under torch.distributed.run, one rank per GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md; ~6.3 TB/s achievable)
F64_VALU_PEAK_TFLOPS = 78.6  # float64 vector peak, same guide (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)


def _clip(args):
    from salsa_amd.synth import synth_clip
    return synth_clip(*args)


def make_batch(seed0, batch, n_samples):
    import multiprocessing as mp
    jobs = [(seed0 + i, n_samples) for i in range(batch)]
    try:
        with mp.get_context('fork').Pool(min(batch, os.cpu_count() or 1, 16)) as pool:
            clips = pool.map(_clip, jobs)
    except Exception:
        clips = [_clip(j) for j in jobs]
    return np.stack(clips)


def algorithmic_bytes(batch, n_samples, T, F):
    """audio read once + feature array written once, per kernel (DESIGN.md 'Roofline')."""
    return {
        'stft_logspec': batch * (4 * n_samples * 4 + 4 * T * F * 4),
        'noise_floor_tracker': 0,
        'cov_eig': batch * (3 * T * F * 4),
    }


def cpu_baseline(feature, fmt, fmax, n_samples):
    """The oracle (CPU restatement of the reference) timed on this host's cores on a bounded sample of the same
    workload (one 60-s clip per single-threaded worker process; all cores, and 32 cores).  Reported beside the GPU
    number, never the target."""
    from oracle import cpu_bench
    return cpu_bench.run(feature, fmt, fmax, n_samples)



def harness_bench(fmt, fmax, batch, n_samples, host):
    """file tree -> feature-file tree through salsa_amd.features.extract_features (the reference's Makefile entry point)"""
    import shutil
    import tempfile

    import yaml

    from salsa_amd import features
    from salsa_amd import io as sio
    base = '/dev/shm' if os.path.isdir('/dev/shm') and shutil.disk_usage('/dev/shm').free > 12e9 else None
    tmp = tempfile.mkdtemp(dir=base)
    try:
        n_clips = 2 * batch
        d = os.path.join(tmp, 'data', fmt + '_dev')
        os.makedirs(d)
        os.makedirs(os.path.join(tmp, 'data', fmt + '_eval'))
        for i in range(n_clips):
            np.save(os.path.join(d, 'fold1_room1_mix%03d.npy' % i), host[i % batch])
        cfg = {'data_dir': os.path.join(tmp, 'data'), 'feature_dir': os.path.join(tmp, 'feat'),
               'data': {'format': fmt, 'fs': 24000, 'n_fft': 512, 'win_len': 512, 'hop_len': 300, 'fmin_doa': 50, 'fmax_doa': fmax}}
        cfg_path = os.path.join(tmp, 'cfg.yml')
        yaml.safe_dump(cfg, open(cfg_path, 'w'))
        secs = n_clips * n_samples / 24000.0
        res = {'clips': n_clips, 'tree': 'float32 .npy clips on %s, feature files %s' % ('tmpfs' if base else 'the temp dir', ('HDF5 (h5py)' if sio.HAVE_H5PY else 'HDF5 (libhdf5 %s through ctypes)' % sio._hdf5.version()) if sio.HAVE_HDF5 else 'raw .npy (no HDF5 library here)')}
        for name, flag in (('pipelined', True), ('serial', False)):
            features.USE_FILE_PIPELINE = flag
            features.extract_features(data_config=cfg_path, task='feature', batch_size=batch)        # warm-up: plans, pinned slots
            t0 = time.perf_counter()
            features.extract_features(data_config=cfg_path, task='feature', batch_size=batch)
            dt = time.perf_counter() - t0
            res[name] = {'s': round(dt, 3), 'audio_s_per_s': round(secs / dt, 1)}
        features.USE_FILE_PIPELINE = True
        # the dataset's own format: 16-bit PCM WAV clips (TNSSE2021).  Round 6: the file's data chunk is uploaded as it is and converted on
        # the device (salsa_pcm_to_planar); `host_decode` = the same tree with the numpy decoder (SALSA_RAW_PCM=0)
        try:
            from scipy.io import wavfile
            dw = os.path.join(tmp, 'data_wav', fmt + '_dev')
            os.makedirs(dw)
            os.makedirs(os.path.join(tmp, 'data_wav', fmt + '_eval'))
            for i in range(n_clips):
                y = host[i % batch]
                wavfile.write(os.path.join(dw, 'fold1_room1_mix%03d.wav' % i), 24000, np.clip(y.T * (30000.0 / np.abs(y).max()), -32768, 32767).astype(np.int16))
            cfg_w = dict(cfg, data_dir=os.path.join(tmp, 'data_wav'), feature_dir=os.path.join(tmp, 'feat_wav'))
            cfg_w_path = os.path.join(tmp, 'cfg_wav.yml')
            yaml.safe_dump(cfg_w, open(cfg_w_path, 'w'))
            for name, flag in (('wav16_pipelined', True), ('wav16_pipelined_host_decode', False)):
                features.RAW_PCM = flag
                features.extract_features(data_config=cfg_w_path, task='feature', batch_size=batch)
                t0 = time.perf_counter()
                features.extract_features(data_config=cfg_w_path, task='feature', batch_size=batch)
                dt = time.perf_counter() - t0
                res[name] = {'s': round(dt, 3), 'audio_s_per_s': round(secs / dt, 1)}
            shutil.rmtree(os.path.join(tmp, 'feat_wav'), ignore_errors=True)
            shutil.rmtree(os.path.join(tmp, 'data_wav'), ignore_errors=True)
        except Exception as e:      # (a reported extra)
            res['wav16_pipelined'] = {'error': '%s: %s' % (type(e).__name__, e)}
        finally:
            features.RAW_PCM = True
        # the reference's default task: features AND the scaler (round 6: its statistics are taken on the device while the dev split
        # is extracted; SALSA_FUSED_SCALER=0 / the second figure: compute_scaler reads every feature file back, as the reference does)
        for name, flag in (('feature_scaler', True), ('feature_scaler_rereading_files', False)):
            features.FUSED_SCALER = flag
            t0 = time.perf_counter()
            features.extract_features(data_config=cfg_path, task='feature_scaler', batch_size=batch)
            dt = time.perf_counter() - t0
            res[name] = {'s': round(dt, 3), 'audio_s_per_s': round(secs / dt, 1)}
        features.FUSED_SCALER = True
        # the file system's share: read every clip, write every feature file, nothing else (one thread)
        feat_dir = os.path.join(tmp, 'feat')
        files = [os.path.join(r, f) for r, _, fs_ in os.walk(feat_dir) for f in fs_ if 'feature_scaler' not in f]   # (clips' files, not the scaler)
        one = sio.load_arrays(files[0])['feature']
        t0 = time.perf_counter()
        for i in range(n_clips):
            np.load(os.path.join(d, 'fold1_room1_mix%03d.npy' % i))
        t_r = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in range(8):
            sio.save_arrays(os.path.join(tmp, 'probe_%d.h5' % i), feature=one)
        t_w = (time.perf_counter() - t0) / 8 * n_clips
        res['file_io_alone'] = {'read_s': round(t_r, 3), 'write_s_one_thread': round(t_w, 3),
                                'note': 'np.load of every clip; save_arrays of every feature file, both from ONE thread (the pipeline reads and writes with pools of threads)'}
        res['timed_region'] = ('one extract_features(task=\'feature\') call: YAML parse, emptying the previous run\'s split folder (the reference does, '
                               ':344), plan lookup, reads -> pinned slots -> device -> features -> pinned slots -> feature files; the pipeline\'s pinned slots '
                               'and plans are cached per process (the warm-up call built them)')
        res['pcie_bound_s'] = round(n_clips / batch * (host.nbytes + one.nbytes * batch) / 55e9, 3)
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _flush_c_stdio():
    """push out whatever native libraries (RCCL's version banner) left in libc's stdout buffer -- every rank, as soon as its
    process group is gone, so that nothing of it can land after rank 0's result line"""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass



_TERM_KEEP = []


def _arm_termination(callback):
    """Run `callback` (from a helper thread) as soon as this process receives SIGTERM -- also while the main thread sits in a
    blocking native call (a collective whose peer died), where a Python signal handler would never get to run: the C-level handler
    writes the signal number to a wake-up socket at once, and the helper thread is blocked on the other end.  torch.distributed.run
    sends SIGTERM to the surviving ranks when one rank fails.  Returns a function that disarms."""
    import signal
    import socket
    import threading
    r, w = socket.socketpair()
    _TERM_KEEP.append((r, w))      # (the wake-up descriptor must outlive this frame whether or not the caller keeps `disarm`)
    w.setblocking(False)
    old_handler = signal.signal(signal.SIGTERM, lambda *_: None)   # (a Python-level handler must exist for the wake-up byte)
    old_fd = signal.set_wakeup_fd(w.fileno())
    armed = [True]

    def wait():
        try:
            data = r.recv(1)
        except OSError:
            return
        if armed[0] and data:
            callback()

    threading.Thread(target=wait, daemon=True).start()

    def disarm():
        armed[0] = False
        signal.set_wakeup_fd(old_fd)
        signal.signal(signal.SIGTERM, old_handler if old_handler is not None else signal.SIG_DFL)
        try:
            w.close()      # wakes the helper thread (recv returns b'')
        except OSError:
            pass
    return disarm

def _emit(line):
    """print the ONE JSON line as the LAST thing on stdout: RCCL writes a version banner through C stdio, which on a pipe sits
    in libc's buffer until exit and would otherwise land after (and be taken for) the result line"""
    _flush_c_stdio()
    sys.stdout.write(json.dumps(line) + '\n')
    sys.stdout.flush()

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--blocks', type=int, default=5, help='timed blocks of --steps steps; the median block is reported')
    ap.add_argument('--batch', type=int, default=32, help='clips per GPU per step')
    ap.add_argument('--seconds', type=float, default=60.0)
    ap.add_argument('--feature', default='salsa', choices=['salsa', 'salsa_lite', 'salsa_ipd'])
    ap.add_argument('--format', default=None, choices=['foa', 'mic'])
    ap.add_argument('--fmax-doa', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-crnn', action='store_true', help='skip the CRNN-training half of the metric')
    ap.add_argument('--no-infer', action='store_true', help='skip the batched-inference leg (BASELINE config 5)')
    ap.add_argument('--no-torch-baseline', action='store_true', help='skip the stock-PyTorch training step timed beside the CRNN leg (N = 1 only)')
    ap.add_argument('--infer-torch-baseline', action='store_true', help='also time config 5 with the CRNN forward on stock PyTorch-ROCm kernels (inference.torch_baseline; ~70 s more)')
    ap.add_argument('--no-config4', action='store_true', help='skip the on-the-fly SALSA-MIC + augmentation training leg (BASELINE config 4)')
    ap.add_argument('--tolerate-crnn-failure', action='store_true', help='exit 0 even if a CRNN-side leg (crnn / config4 / inference) failed')
    ap.add_argument('--infer-steps', type=int, default=5, help='timed steps of the inference leg; a step is the WHOLE --infer-clips job once through')
    ap.add_argument('--infer-clips', type=int, default=1024, help='clips of the whole inference job, sharded over the GPUs (BASELINE config 5: 1024)')
    ap.add_argument('--crnn-steps', type=int, default=20)
    ap.add_argument('--crnn-warmup', type=int, default=5)
    ap.add_argument('--streams', type=int, default=1, help='extra leg: K steps round-robin over this many HIP streams / plans (reported as pipelined, never `value`)')
    ap.add_argument('--pcie', action='store_true', help='also time host->device->extract->device->host (reported, never `value`)')
    ap.add_argument('--groups', type=int, default=0, help='clip-group pipelining depth (0 = library default)')
    ap.add_argument('--fused', type=int, default=0, help='measurement: salsa_plan_set_fused mode (1 = STFT -> tracker -> fused STFT + '
                                                         'covariance / eigen kernel; bit-identical, measured slower: DESIGN section 6)')
    args = ap.parse_args()

    from bench_crnn import infer_bench, self_spawn, train_bench
    self_spawn(args.gpus, __file__)           # `--gpus N` without a launcher: become N ranks (does not return then)

    fmt = args.format or ('foa' if args.feature == 'salsa' else 'mic')
    fmax = args.fmax_doa or (9000 if args.feature == 'salsa' and fmt == 'foa' else 4000 if args.feature == 'salsa' else 2000)
    n_samples = int(round(args.seconds * 24000))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))

    # synthesise this rank's clips before touching the GPU (fork-safe); seeds 2021.. as SURVEY 8(d) config 2
    host = make_batch(2021 + rank * args.batch, args.batch, n_samples)

    import torch
    import torch.distributed as dist
    from salsa_amd.extractor import SalsaExtractor

    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback exists)'
    # SALSA_BENCH_SHARE_GPU=1 (functional test of the N > 1 path on a one-GPU box): ranks share the visible devices round-robin
    # and the process group is gloo (RCCL refuses two ranks on one device) -- the timing means nothing then, the control flow
    # (sharding, barriers, max over ranks, the gradient synchroniser, rank-0-only line) is the real one
    share = os.environ.get('SALSA_BENCH_SHARE_GPU', '0') == '1'
    if share:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo') if share else dist.init_process_group('nccl', device_id=dev)
        rccl_ranks = dist.get_world_size()          # read back from the process group, not from the command line

    from salsa_amd import _lib as _salsa_lib
    build_flags = _salsa_lib.build_flags()
    ex = SalsaExtractor(audio_format=fmt, feature_type=args.feature, fmax_doa=fmax, device=dev)
    if args.fused:
        ex.set_fused(args.fused)
    if args.groups:
        ex.set_groups(args.groups)
    audio = torch.from_numpy(host).to(dev)
    Cn, T, F = ex.output_shape(n_samples)
    out = torch.empty((args.batch, Cn, T, F), dtype=torch.float32, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    power = None
    if rank == 0:                                       # package power / shader clock over the timed regions (bench_crnn.PowerSampler)
        from bench_crnn import PowerSampler
        power = PowerSampler(dev)
        import atexit
        atexit.register(power.close)
    for _ in range(args.warmup):
        ex.extract(audio, out=out)
    block_s = []
    for _ in range(max(1, args.blocks)):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ex.extract(audio, out=out)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        block_s.append(el)
    elapsed = float(np.median(block_s))             # the median K-step block (max over ranks within each block)

    # per-kernel durations: HIP events recorded by the library on the launch stream, separate pass (not in `value`)
    kernels = []
    roofline = None
    cpu = None
    pcie = None
    pipelined = None
    if rank == 0:
        # Per-kernel attribution, two ways (tools/timing_probe.py, profiles/r3_timing_schemes.json):
        #  (1) `ms_per_launch` = PREFIX DIFFERENCES of the real launch sequence, wall clock, no events: K issues of [STFT],
        #      of [STFT, tracker] and of the whole path on the buffers a full call left behind (every kernel is idempotent on
        #      them); a kernel's figure is what the step gains when it is added, and the three add up to the step EXACTLY;
        #  (2) `ms_event_pair` = a HIP event pair around every launch of the plain sequence on the launch stream (what a
        #      rocprofv3 kernel trace shows; a marker between two kernels makes the later one wait for the earlier one's
        #      write-back drain, which otherwise overlaps it, so these can add up to a few percent MORE than the step).
        n_t = max(3, min(args.steps, 10))

        def wall_ms(n):
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    ex.extract(audio, out=out)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / n * 1e3)
            return float(np.median(ts))

        prefix, prefix_note = {}, None
        power_k1 = power_step = None

        def power_loop(fn, seconds=2.0):
            """the hwmon power figure is a moving average with a time constant of several hundred ms: the timed blocks (0.1 s) are over
            before it has risen.  So, OUTSIDE every timed region, the same launches run for ~2 s more and the last 60 % of that is read."""
            if power is None:
                return None
            t_start = time.time()
            while time.time() - t_start < seconds:
                for _ in range(args.steps):
                    fn()
                torch.cuda.synchronize()
            return power.stats(t_start + 0.4 * seconds, time.time())
        if args.feature == 'salsa':
            with ex.issue_prefix(1):                           # (restores plain issue on exit, exceptions included)
                p1 = wall_ms(args.steps)
                power_k1 = power_loop(lambda: ex.extract(audio, out=out))
            with ex.issue_prefix(2):
                p2 = wall_ms(args.steps)
            p3 = wall_ms(args.steps)                           # the whole path, timed the same way as its prefixes
            power_step = power_loop(lambda: ex.extract(audio, out=out))
            prefix = {'stft_logspec': p1, 'noise_floor_tracker': p2 - p1, 'cov_eig': p3 - p2}
            scale = (1e3 * elapsed / args.steps) / p3          # ... then scaled so the three add up to the headline step
            prefix = {k: v * scale for k, v in prefix.items()}
            if min(prefix.values()) <= 0:                      # a noisy box: differences of separately timed loops can cross
                prefix_note = 'prefix differences not monotone on this run (%s): event-pair times used instead' % \
                    {k: round(v, 4) for k, v in prefix.items()}
                prefix = {}
        ex.set_timing(1)
        tot, cnt = {}, {}
        for _ in range(n_t):
            ex.extract(audio, out=out)
            for name, ms in ex.read_timing():
                tot[name] = tot.get(name, 0.0) + ms
                cnt[name] = cnt.get(name, 0) + 1
        ex.set_timing(False)
        pair_ms = {name: tot[name] / cnt[name] for name in tot}
        ab = algorithmic_bytes(args.batch, n_samples, T, F)
        if args.feature != 'salsa':
            ab = {'stft_logspec': args.batch * (4 * n_samples * 4 + 7 * T * F * 4)}
        for name in tot:
            launches = cnt[name] // n_t                      # launches per step
            ms = prefix.get(name, pair_ms[name]) / launches  # duration of ONE launch (prefix difference; lite: the event pair)
            b = ab.get(name, 0) / launches                   # algorithmic bytes ONE launch moves
            kernels.append({'name': name, 'launches_per_step': launches, 'ms_per_launch': round(ms, 4),
                            'ms_event_pair': round(pair_ms[name], 4),
                            'algorithmic_bytes_per_launch': int(b),
                            'GBps': round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                            'frac': round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None})
        # measured HBM bytes and VALU utilisation per launch (rocprofv3 PMC, collected offline by tools/pmc_round.sh +
        # tools/pmc_traffic.py: counters cannot be sampled from inside this process); only valid for the batch size and
        # kernel version they were taken at (profiles/traffic.json names both)
        traffic, valu, traffic_src = {}, {}, None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
            if tj.get('batch_clips_per_launch') == args.batch and args.feature == 'salsa' and abs(args.seconds - 60) < 1e-9:
                traffic = {k: v['hbm_bytes'] for k, v in tj['kernels'].items()}
                valu = {k: v.get('valu_util') for k, v in tj['kernels'].items()}
                import hashlib
                ksha = hashlib.sha256(b''.join(open(os.path.join(ROOT, 'salsa_amd', 'csrc', f), 'rb').read()
                                               for f in ('salsa_kernels.hip', 'salsa_math.h'))).hexdigest()[:16]
                fresh = tj.get('kernel_sources_sha16') == ksha
                traffic_src = ('offline PMC: profiles/traffic.json, commit %s, %s; rocprofv3 --pmc passes (%s) of the same kernels on another run'
                               % (tj.get('commit'), 'taken at EXACTLY the kernel sources this run executes' if fresh else
                                  'STALE: the kernel sources changed since the counters were taken', tj.get('source', 'tools/pmc_round.sh')))
        except Exception:
            pass
        for k in kernels:
            k['traffic'] = traffic.get(k['name'])
            k['f64_valu_util'] = valu.get(k['name'])
        # the box's attainable HBM rate ("confirm with a device copy and use the measured peak as denominator too")
        a1 = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        a2 = torch.empty_like(a1)
        a2.copy_(a1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            a2.copy_(a1)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * a1.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a1, a2
        # headline fraction = the WHOLE path (49 925 600 B per clip x B / step time / 8 TB/s); every kernel's
        # own figure is in `kernels`, and `dominant` names the kernel with the most time per step (no tie-breaking rule)
        t_of = lambda k: k['ms_per_launch'] * k['launches_per_step']
        dom = max(kernels, key=t_of)
        step_ms = 1e3 * elapsed / args.steps
        pipe_bytes = sum(ab.values())
        achieved = pipe_bytes / (step_ms * 1e-3) / 1e9
        tr_known = [k['traffic'] * k['launches_per_step'] for k in kernels if k.get('traffic')]
        roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4),
                    'traffic': int(sum(tr_known)) if len(tr_known) == len(kernels) and tr_known else None,
                    'traffic_source': traffic_src,
                    'kernel_ms_sum': round(sum(t_of(k) for k in kernels), 4),
                    'kernel_ms_sum_event_pairs': round(sum(k['ms_event_pair'] * k['launches_per_step'] for k in kernels), 4),
                    'kernel_timing_note': prefix_note,
                    'kernel_timing': 'ms_per_launch = prefix differences of the real launch sequence (wall clock over %d issues of [STFT], [STFT, tracker], whole path; they add up to the step); ms_event_pair = HIP event pair around each launch of the plain sequence on the launch stream, %d calls' % (args.steps, n_t),
                    'scope': 'whole step: algorithmic bytes of the path (%d) / median wall time per step' % pipe_bytes,
                    'algorithmic_bytes': pipe_bytes, 'ms': round(step_ms, 4),
                    'peak_measured': round(copy_gbs, 1), 'frac_of_measured': round(achieved / copy_gbs, 4),
                    'peak_measured_note': '1 GiB device-to-device copy, read + write bytes / time',
                    # the dominant kernel as the bench contract words it: algorithmic bytes of one launch / its average launch duration
                    # by HIP events on the launch stream (the event pair; what rocprofv3 shows) -- the prefix-difference figure, which
                    # also bills the kernel for the write-back drain that follows it, is given beside it
                    'dominant': {'kernel': dom['name'], 'ms_per_launch': dom['ms_event_pair'],
                                 'achieved': round(dom['algorithmic_bytes_per_launch'] / (dom['ms_event_pair'] * 1e-3) / 1e9, 1),
                                 'frac': round(dom['algorithmic_bytes_per_launch'] / (dom['ms_event_pair'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 'ms_per_launch_prefix': dom['ms_per_launch'], 'frac_prefix': dom['frac'], 'traffic': dom.get('traffic'),
                                 'achieved_on_traffic': (round(dom['traffic'] / (dom['ms_event_pair'] * 1e-3) / 1e9, 1) if dom.get('traffic') else None),
                                 'timing': 'HIP event pair around the launch on the launch stream, mean over %d calls' % n_t},
                    'secondary_bound': {'unit': 'fraction of SIMD cycles issuing a VALU instruction (float64 in the STFT and the tracker: peak %.1f TFLOP/s; packed float32 in cov_eig since round 4)' % F64_VALU_PEAK_TFLOPS,
                                        'per_kernel': {k['name']: k['f64_valu_util'] for k in kernels}},
                    # what the package drew over the timed blocks, and while the dominant kernel ran ALONE (the [STFT]-only prefix
                    # issues above): at the power cap with the shader clock below its 2.4 GHz, energy -- not overlap -- is the bound
                    'power': {'whole_step': power_step,
                              'stft_logspec_alone': power_k1},
                    'kernels': kernels}
        if args.streams > 1:
            # independent batches on separate streams: the latency-bound tracker of one step hides under the STFT /
            # eigen kernels of its neighbours (bulk extraction of a dataset is exactly this situation)
            exs = [SalsaExtractor(audio_format=fmt, feature_type=args.feature, fmax_doa=fmax, device=dev) for _ in range(args.streams)]
            outs = [torch.empty_like(out) for _ in range(args.streams)]
            sts = [torch.cuda.Stream() for _ in range(args.streams)]
            def run(n):
                for i in range(n):
                    j = i % args.streams
                    with torch.cuda.stream(sts[j]):
                        exs[j].extract(audio, out=outs[j])
            run(2 * args.streams)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            tp = (time.perf_counter() - tp) / args.steps
            pipelined = {'streams': args.streams, 'ms_per_step': round(1e3 * tp, 4),
                         'audio_s_per_s': round(args.batch * args.seconds / tp, 1)}
            del exs, outs
        if args.pcie:
            # boundary note: when the caller hands HOST buffers (the Python surface does), every step also
            # moves the clips in and the features out over PCIe; pinned memory, async copies on the compute stream
            hp = torch.from_numpy(host).pin_memory()
            op = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
            for _ in range(2):
                audio.copy_(hp, non_blocking=True); ex.extract(audio, out=out); op.copy_(out, non_blocking=True)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            n_p = max(3, min(args.steps, 5))
            for _ in range(n_p):
                audio.copy_(hp, non_blocking=True); ex.extract(audio, out=out); op.copy_(out, non_blocking=True)
            torch.cuda.synchronize()
            tp = (time.perf_counter() - tp) / n_p
            pcie = {'ms_per_step': round(1e3 * tp, 3), 'audio_s_per_s': round(args.batch * args.seconds / tp, 1),
                    'bytes_h2d': int(hp.numel() * 4), 'bytes_d2h': int(op.numel() * 4),
                    'note': 'serial H2D + extract + D2H per step, pinned host memory'}
            # the same host-to-host job through HostPipeline: 3 slots, copy-in / compute / copy-out on their own streams
            from salsa_amd.extractor import HostPipeline
            del hp, op
            pipe = HostPipeline(depth=3, audio_format=fmt, feature_type=args.feature, fmax_doa=fmax, device=dev)
            # (a reader would decode audio straight into the pinned slot: every slot is filled once here, outside the timing)
            for _ in pipe.run(fill=lambda b, i: b.__setitem__(Ellipsis, host), n_batches=3, shape=host.shape):
                pass
            tp = time.perf_counter()
            n_h = 12
            for _ in pipe.run(fill=lambda b, i: None, n_batches=n_h, shape=host.shape):
                pass
            tp = (time.perf_counter() - tp) / n_h
            pcie['overlapped'] = {'ms_per_step': round(1e3 * tp, 3), 'audio_s_per_s': round(args.batch * args.seconds / tp, 1),
                                  'note': 'HostPipeline(depth=3): pinned host slot -> device -> features -> pinned host slot, the transfers of neighbouring batches overlapped on separate streams'}
            del pipe
            # the drop-in harness itself, files to files (reference: salsa_feature_extraction.py:351-383): a synthetic tree of
            # 2 x batch 60-s clips as float32 .npy files on tmpfs -> extract_features(task='feature') -> one feature file per
            # clip.  Next to it: the same with the pipeline off (read, copy, extract, copy back, write, serially) and what the
            # file system alone does with these bytes.
            if args.feature == 'salsa':
                pcie['harness'] = harness_bench(fmt, fmax, args.batch, n_samples, host)

    lite_leg = None
    if world == 1 and args.feature == 'salsa':
        # ---- SURVEY 8(a8) / BASELINE config 1's feature on the GPU (reported, never `value`): SALSA-Lite MIC on the same 32 x 60-s batch --
        # one launch of the STFT kernel's Lite STD instantiation per step; algorithmic bytes 48 715 748 per clip
        try:
            lex = SalsaExtractor(audio_format='mic', feature_type='salsa_lite', fmax_doa=2000, device=dev)
            lout = torch.empty((args.batch,) + tuple(lex.output_shape(n_samples)), dtype=torch.float32, device=dev)
            for _ in range(args.warmup):
                lex.extract(audio, out=lout)
            lt = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    lex.extract(audio, out=lout)
                torch.cuda.synchronize()
                lt.append((time.perf_counter() - t0) / args.steps)
            lms = 1e3 * float(np.median(lt))
            lbytes = args.batch * (4 * n_samples * 4 + 7 * int(lout.shape[2]) * int(lout.shape[3]) * 4)
            lite_leg = {'workload': 'SALSA-Lite MIC (fmax_doa 2000): batch %dx%.0f-s 4-ch clips, 1 GPU' % (args.batch, args.seconds),
                        'ms_per_step': round(lms, 4), 'value': round(args.batch * args.seconds / (lms * 1e-3), 1), 'unit': 'audio-seconds/s',
                        'algorithmic_bytes': lbytes, 'achieved_gbs': round(lbytes / (lms * 1e-3) / 1e9, 1),
                        'frac_of_8tbs': round(lbytes / (lms * 1e-3) / 8e12, 4), 'timing': 'median of 3 blocks of %d steps, wall clock' % args.steps}
            del lex, lout
            torch.cuda.empty_cache()
        except Exception as e:                      # (a reported extra: never takes the line down)
            lite_leg = {'error': '%s: %s' % (type(e).__name__, e)}
    # ---- second half of the metric: CRNN training (its own timed region; every rank takes part in the data-parallel run)
    del ex, out, audio
    torch.cuda.empty_cache()
    failures = []
    done = {}                                    # the CRNN-side legs as they finish (what a bail-out can still report)

    def make_line(cpu_, status=None):
        audio_s = world * args.batch * args.seconds * args.steps
        ln = {
            'metric': 'SALSA feat-extract audio-s/s',
            'value': round(audio_s / elapsed, 1),
            'unit': 'audio-seconds/s',
            'n_gpus': world,
            'rccl_ranks': rccl_ranks,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4),
            'blocks_ms': [round(1e3 * b / args.steps, 4) for b in block_s],
            'blocks_note': 'ms per step of each timed %d-step block (max over ranks); value / ms_per_step = the median block' % args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            # what the path computes in (round-4 review): the STFT, the tracker and the cold list of the eigen-solve in float64; the
            # covariance + eigen-solve of the other ~99.7 % of the gated frames as a packed-float32 pair solve whose every kept
            # decision is certified against float64; SALSA-Lite: float64 STFT, float32 phases
            'dtype': ('f64 (STFT, noise-floor tracker, eigen-solve cold list) + packed f32 (covariance and eigen-solve of the certified frames)'
                      if args.feature == 'salsa' else 'f64 (STFT) + f32 (log-spectrogram, phase differences)'),
            'data': 'synthetic',
            'build_flags': build_flags or 'product build (no -D)',
            'config': {'workload': 'Full SALSA %s (eigenvector path): batch %dx%.0f-s 4-ch 24 kHz clips per GPU, '
                                   'feature-extract only, n_fft 512 hop 300 fmax_doa %d cond 5 tracking on'
                                   % (fmt.upper(), args.batch, args.seconds, fmax) if args.feature == 'salsa' else
                                   '%s MIC: batch %dx%.0f-s clips per GPU' % (args.feature, args.batch, args.seconds),
                       'clips_per_gpu': args.batch, 'clip_seconds': args.seconds, 'feature': args.feature,
                       'format': fmt, 'sharding': 'clips/%d (no collective)' % world,
                       'clip_groups': args.groups or 'default'},
            'roofline': roofline,
            'cpu_baseline': cpu_,
            'crnn': done.get('crnn'),
            'config4': done.get('config4'),
            'inference': done.get('inference'),
            'salsa_lite': lite_leg,
        }
        if status or failures:
            ln['status'] = status or 'partial: %s failed' % ', '.join(failures)
        if pcie:
            ln['pcie_inclusive'] = pcie
        if pipelined:
            ln['pipelined'] = pipelined
        return ln

    def bail(why):
        """N > 1 only: a CRNN-side leg failed or hung on some rank.  The feature half of the line is complete by now, so rank 0
        still prints it (the unfinished legs are null, `status` says why) before the job is torn down with a non-zero exit;
        torch.distributed.run then ends the other ranks."""
        if rank == 0:
            _emit(make_line(None, status='partial: %s (the feature-path half above is complete)' % why))
        os._exit(3)

    watchdog, disarm_term = None, None
    if world > 1 and not args.no_crnn and args.feature == 'salsa':
        import threading
        limit = float(os.environ.get('SALSA_BENCH_LEG_TIMEOUT', '600'))
        watchdog = threading.Timer(limit, bail, args=('the CRNN-side legs did not finish within %.0f s at N = %d' % (limit, world),))
        watchdog.daemon = True
        watchdog.start()
        disarm_term = _arm_termination(lambda: bail('terminated by the launcher: another rank failed in a CRNN-side leg'))

    def leg(name, fn):
        """One of the CRNN-side legs; a failure is recorded in the line AND makes the process exit non-zero (unless
        --tolerate-crnn-failure): the CRNN number is half of BASELINE.json's metric.  At N > 1 the ranks cannot go on after one of
        them failed (the others sit in a collective): see bail()."""
        try:
            if os.environ.get('SALSA_BENCH_FAIL_LEG') == name and rank == world - 1:   # (test hook: tests/test_gpu_configs.py)
                raise RuntimeError('injected failure')
            done[name] = fn()
        except Exception as e:
            if world > 1:
                import traceback
                traceback.print_exc()
                bail('%s failed on rank %d: %s: %s' % (name, rank, type(e).__name__, e))
            failures.append(name)
            done[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        return done[name]

    crnn = config4 = infer = None
    if not args.no_crnn and args.feature == 'salsa':
        crnn = leg('crnn', lambda: train_bench(rank, world, dev, batch=32, steps=args.crnn_steps, warmup=args.crnn_warmup, power=power))
        # ---- BASELINE config 4: raw 8-s MIC chunks -> SALSA-MIC on device -> the reference's augmentation -> training step
        if not args.no_config4:
            torch.cuda.empty_cache()
            config4 = leg('config4', lambda: train_bench(rank, world, dev, batch=32, steps=args.crnn_steps, warmup=args.crnn_warmup,
                                                         on_the_fly=True, augment=True, power=power))
        # ---- BASELINE config 5 (reported, not part of the metric): --infer-clips (1024) distinct 60-s clips in all, sharded over
        # the ranks, through SALSA + CRNN forward + DCASE rows, sub-batches of 32 (bench_crnn.infer_bench)
        if not args.no_infer:
            def _infer():
                from types import SimpleNamespace
                from salsa_amd.crnn.train import Trainer
                torch.cuda.empty_cache()
                return infer_bench(SimpleNamespace(clips=args.infer_clips, sub_batch=32, steps=args.infer_steps, warmup=1), rank, world, dev,
                                   Trainer(dev, ddp=False), power=power)
            infer = leg('inference', _infer)

    if world > 1:
        dist.barrier()
        if watchdog is not None:
            watchdog.cancel()
            disarm_term()
        dist.destroy_process_group()
        _flush_c_stdio()
    if rank != 0:
        return
    # the same training step on stock PyTorch-ROCm kernels, same box (a stated baseline; N = 1 only: the GPU is idle now)
    if world == 1 and isinstance(done.get('crnn'), dict) and 'value' in done['crnn'] and not args.no_torch_baseline:
        from bench_crnn import torch_baseline
        torch.cuda.empty_cache()
        done['crnn']['torch_baseline'] = torch_baseline()
        tb = done['crnn']['torch_baseline']
        if tb and tb.get('value'):
            done['crnn']['speedup_vs_torch_baseline'] = round(done['crnn']['value'] / tb['value'], 3)
    # (opt-in: the child process costs ~70 s -- MIOpen's solver search on 60-s maps -- and the default run is kept near two minutes)
    if world == 1 and isinstance(done.get('inference'), dict) and 'value' in done['inference'] and args.infer_torch_baseline:
        from bench_crnn import torch_infer_baseline
        torch.cuda.empty_cache()
        done['inference']['torch_baseline'] = torch_infer_baseline()
        tb = done['inference']['torch_baseline']
        if tb and tb.get('value'):
            done['inference']['speedup_vs_torch_baseline'] = round(done['inference']['value'] / tb['value'], 3)
    # CPU baseline: rank 0 only, for every N (after the process group is gone, so the other ranks are not kept waiting on a
    # collective); the same bounded sample at every N
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(args.feature, fmt, fmax, n_samples)

    line = make_line(cpu)
    _emit(line)
    if failures and not args.tolerate_crnn_failure:
        sys.stdout.flush()
        sys.exit(3)


if __name__ == '__main__':
    main()
