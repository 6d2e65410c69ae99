#!/usr/bin/env python
"""bench_crnn.py -- secondary bench (BASELINE.json config 3/4): SELD CRNN training throughput on N MI355X.

A step = forward + loss + backward + Adam update on a batch of 32 synthetic 8-s SALSA chunks (7,640,200) per GPU, bf16
autocast, channels-last; data-parallel ranks all-reduce gradients over RCCL (salsa_amd/crnn/grad_sync.py, overlapped with backward).
--on-the-fly adds the feature extraction in front of every step (config 4: SALSA-MIC from raw 8-s audio on device).
Prints ONE JSON line on rank 0.  (The headline bench of this repo is bench.py: the feature path.)

  python bench_crnn.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_crnn.py --gpus N
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def self_spawn(n_gpus: int, script: str) -> None:
    """`python <script> --gpus N` with no torch.distributed environment: become N ranks (one process per GPU, RCCL) by
    re-executing this command under torch.distributed.run on a free loopback port.  Does not return when it spawns."""
    if n_gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(script)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


MFMA_BF16_PEAK_TFLOPS = 2500.0           # dense bf16 MFMA peak, MI355X_MICROARCH.md
GFLOP_PER_CHUNK_FWD = 44.74 + 0.19       # conv + GRU forward per 8-s chunk (probed)


def infer_bench(args, rank, world, dev, tr, audio=None, power=None):
    """Batched inference (BASELINE config 5: 1024 concurrent 60-s clips sharded over the GPUs of the node, sub-batches of 32).
    `args.clips` is the WHOLE JOB's clip count (default 1024): rank r takes the r-th contiguous range (salsa_amd.distributed.
    shard_range; 128 per GPU at N = 8, all 1024 = 23.6 GB of audio at N = 1), every clip DISTINCT and seeded (seed 2021 + its
    global index, synthesised on the device: salsa_amd.synth.synth_clips_device) and resident in HBM when the timed region
    starts.  A step = every clip of the job once through: raw audio -> SALSA-FOA features (HIP) with normalise-on-load fused ->
    CRNN forward (bf16) -> sigmoid / xyz -> pinned host -> combine_chunks -> DCASE rows ON THE HOST, by the product's engine
    (salsa_amd.crnn.infer.infer_pipelined: two sub-batches in flight, no device-wide synchronize).  No collective on the data
    path.  Per-clip latency comes from the timed steps themselves (the engine stamps every sub-batch), two ways: from the ISSUE of
    the clip's sub-batch to its rows on the host, and from the step's start (all clips present at once, queueing included).
    `audio`: this rank's clips as a device tensor (tests); None -> synthesised here."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from salsa_amd.crnn.infer import infer_pipelined
    from salsa_amd.distributed import shard_range
    from salsa_amd.extractor import SalsaExtractor
    total = int(args.clips)
    lo, hi = shard_range(total, rank, world)
    n_mine = hi - lo
    n_samples = int(getattr(args, 'n_samples', 60 * 24000))
    ex = SalsaExtractor(audio_format='foa', fmax_doa=9000, device=dev)
    if audio is None:
        from salsa_amd.synth import synth_clips_device
        t_s = time.perf_counter()
        audio = synth_clips_device(2021 + lo, n_mine, n_samples, device=dev)
        torch.cuda.synchronize()
        t_synth = time.perf_counter() - t_s
    else:
        t_synth = None
    assert audio.shape[0] == n_mine
    mean = torch.full((4, 1, 200), -60.0, device=dev)
    std = torch.full((4, 1, 200), 12.0, device=dev)
    ex.set_scaler(mean, std)                                  # normalise-on-load fused into the extraction kernel
    sub = args.sub_batch
    n_lab = (n_samples // 300) // 8                           # label frames per clip (600 for 60 s)
    featurize = lambda a, b: ex.extract(audio[a:b])[:, :, :8 * n_lab]      # a view: the stem kernel takes the strides
    # untrained weights answer ~0.5 everywhere: the SED threshold is put at the 95th percentile of one sub-batch's
    # probabilities so that ~5 % of the (frame, class) pairs are active -- a DCASE-like row density (hundreds of rows per clip)
    p0, _ = tr.infer(featurize(0, min(sub, n_mine)))
    thr = float(torch.quantile(p0.flatten()[:1 << 20].float(), 0.95).item())
    stamps, starts = [], []

    def step():
        t0 = time.perf_counter()
        n0 = len(stamps)
        rows = infer_pipelined(n_mine, featurize, tr.infer, sub_batch=sub, depth=2, sed_threshold=thr, n_label_frames=n_lab,
                               as_array=True, stamps=stamps)
        starts.extend([t0] * (len(stamps) - n0))
        return rows

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    del stamps[:], starts[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lat_issue = [t_done - t_issue for (a, b, t_issue, t_done) in stamps for _ in range(b - a)]
    lat_arrival = [t_done - ts for (a, b, t_issue, t_done), ts in zip(stamps, starts) for _ in range(b - a)]
    n_rows = int(sum(len(r) for r in rows))
    pw = None
    if power is not None and world == 1:   # outside the timed region (the hwmon power figure is a slow moving average): ~2 s more, last 60 % read
        tw0 = time.time()
        while time.time() - tw0 < 2.0:
            step()
        torch.cuda.synchronize()
        pw = power.stats(tw0 + 0.8, time.time())
    # the device alone (no D2H of the outputs, no rows): what the host-side post-processing costs the job, outside the timed region
    torch.cuda.synchronize()
    td = time.perf_counter()
    for a in range(0, n_mine, sub):
        tr.infer(featurize(a, min(n_mine, a + sub)))
    torch.cuda.synchronize()
    td = time.perf_counter() - td
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        lats = [None] * world
        dist.all_gather_object(lats, (lat_issue, lat_arrival))
        lat_issue = [v for part in lats for v in part[0]]
        lat_arrival = [v for part in lats for v in part[1]]
    if rank != 0:
        return None
    q = lambda v, f: round(1e3 * float(np.quantile(np.asarray(v), f)), 2) if len(v) else None
    return ({
        'metric': 'SALSA+CRNN inference clips/s', 'value': round(total * args.steps / elapsed, 2),
        'unit': '60-s clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True,
        'scaling': 'strong (the job is %d clips whatever N is; %d per GPU here)' % (total, n_mine),
        'vs_baseline': None, 'dtype': 'bf16 (CRNN) / f64 + packed f32 (features: see the feature line)',
        'data': 'synthetic (%d distinct seeded clips: noise + three AR(1) bursts each, synthesised on the device)' % total,
        'p50_latency_ms_per_clip': q(lat_issue, 0.5), 'p90_latency_ms_per_clip': q(lat_issue, 0.9),
        'p50_latency_from_arrival_ms': q(lat_arrival, 0.5), 'p90_latency_from_arrival_ms': q(lat_arrival, 0.9),
        'max_latency_from_arrival_ms': q(lat_arrival, 1.0),
        'latency_samples': len(lat_issue),
        'latency_note': 'per clip, from the timed steps themselves, all ranks: *_latency_ms_per_clip = issue of the clip\'s %d-clip sub-batch '
                        '(two sub-batches in flight) -> its DCASE rows exist on the host; *_from_arrival = the same end point measured from the '
                        'start of the step, when all %d clips are present (queueing behind the earlier sub-batches included)' % (sub, total),
        'device_only_clips_per_s': round(n_mine / td, 2),
        'device_only_note': 'this rank\'s clips once through extraction + forward with no output copy and no rows (outside the timed region)',
        'dcase_rows_per_clip': round(n_rows / max(1, n_mine), 1), 'sed_threshold': round(thr, 4),
        'synth_s': None if t_synth is None else round(t_synth, 2),
        'power': pw,
        'config': {'workload': 'batched inference (BASELINE config 5): %d distinct 60-s 4-ch clips / %d GPU%s (%d per GPU, %.1f GB of audio '
                               'resident), SALSA-FOA + CRNN forward + DCASE rows on the host, sub-batches of %d'
                               % (total, world, '' if world == 1 else 's', n_mine, n_mine * 4 * n_samples * 4 / 1e9, sub),
                   'clips': total, 'clips_per_gpu': n_mine, 'sub_batch': sub,
                   'parallelism': 'clips sharded over %d GPU%s, no collective on the data path' % (world, '' if world == 1 else 's')}})


N_ROT = 4          # distinct device-resident batches rotating through the timed training loops


class PowerSampler:
    """Package power and shader clock of the GPU while a timed region runs (round 5: the STFT kernel and the 64 -> 64 convolutions hold
    the package AT its 1400-W cap with the clock throttled, profiles/r5_power_probe.txt -- so the bench line carries what the part drew).
    A helper PROCESS polls the amdgpu hwmon files (power1_input in microwatts, freq1_input = sclk in Hz) of every card it can see at
    ~200 Hz into a temporary file; the card is this process's GPU, matched by PCI address (every card and the busiest one as a fallback); stats(t0, t1) takes time.time() stamps.  Measurement
    garnish only: any failure makes it report None, never an exception."""
    _SRC = ("import glob,os,sys,time\n"
            "ds=[d for d in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')) if os.path.exists(d+'/power1_input')]\n"
            "own=[d for d in ds if len(sys.argv)>2 and sys.argv[2] and sys.argv[2] in os.path.realpath(d)]\n"
            "ds=own or ds\n"
            "fs=[(open(d+'/power1_input'),open(d+'/freq1_input') if os.path.exists(d+'/freq1_input') else None) for d in ds]\n"
            "caps=[open(d+'/power1_cap').read().strip() if os.path.exists(d+'/power1_cap') else '0' for d in ds]\n"
            "o=open(sys.argv[1],'w'); o.write('# '+' '.join(caps)+'\\n')\n"
            "def rd(f):\n"
            "    f.seek(0); return f.read().strip() or '0'\n"
            "pp=os.getppid()\n"
            "while os.getppid()==pp:\n"
            "    t=time.time(); o.write('%.6f %s\\n'%(t,' '.join(rd(p)+' '+(rd(q) if q else '0') for p,q in fs))); o.flush()\n"
            "    time.sleep(max(0.0,0.005-(time.time()-t)))\n")

    def __init__(self, device=None):
        self.proc, self.path, self.slot = None, None, ''
        try:
            try:    # this process's GPU by its PCI address (a node shows the hwmon files of every GPU, other tenants' included)
                import torch
                pr = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device())
                self.slot = '%04x:%02x:%02x.' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            except Exception:
                self.slot = ''
            import glob
            import subprocess
            import tempfile
            if not glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'):
                return
            fd, self.path = tempfile.mkstemp(prefix='salsa_power_', suffix='.txt')
            os.close(fd)
            self.proc = subprocess.Popen([sys.executable, '-c', self._SRC, self.path, self.slot], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            self.proc = None

    def stats(self, t0, t1):
        if self.proc is None:
            return None
        try:
            import numpy as np
            caps, rows = None, []
            for ln in open(self.path):
                if ln.startswith('#'):
                    caps = [float(x) / 1e6 for x in ln[1:].split()]
                    continue
                v = ln.split()
                if len(v) >= 3 and t0 <= float(v[0]) <= t1:
                    rows.append([float(x) for x in v[1:]])
            if len(rows) < 3:
                return None
            a = np.asarray(rows)
            pw, ck = a[:, 0::2] / 1e6, a[:, 1::2] / 1e6
            c = int(np.argmax(pw.mean(axis=0)))
            return {'cap_w': caps[c] if caps and c < len(caps) else None, 'mean_w': round(float(pw[:, c].mean()), 1),
                    'max_w': round(float(pw[:, c].max()), 1), 'sclk_mhz_mean': round(float(ck[:, c].mean()), 0),
                    'sclk_mhz_min': round(float(ck[:, c].min()), 0), 'samples': int(len(rows)),
                    'pci': self.slot or None, 'cards_polled': int(pw.shape[1]),
                    'source': 'amdgpu hwmon power1_input / freq1_input of this GPU (matched by PCI address; the busiest card if that failed), polled at ~200 Hz by a helper process over ~1.2 s of the same launches run right after the timed region (the figure is a slow moving average)'}
        except Exception:
            return None

    def close(self):
        try:
            if self.proc is not None:
                self.proc.kill()
                self.proc.wait(timeout=5)
            if self.path:
                os.unlink(self.path)
        except Exception:
            pass
        self.proc = None



def train_bench(rank, world, dev, batch=32, steps=20, warmup=5, on_the_fly=False, augment=False, fp32_grads=False,
                n_frames=640, amp_dtype='default', force_ddp=False, power=None):
    """CRNN training throughput (BASELINE.json config 3; config 4 with on_the_fly): forward + loss + backward + Adam on
    `batch` 8-s chunks per GPU per step, bf16 autocast; for world > 1 a bucketed gradient all-reduce on RCCL (grad_sync.py; SALSA_GRAD_SYNC=ddp: torch DDP)
    overlapped with the backward.  The process group must already be initialised for world > 1.  Every rank calls this;
    rank 0 gets the result dict, the others None.  `n_frames` / `amp_dtype` / a CPU `dev` exist for the world-size-2 gloo test
    of this very function (tests/test_crnn_cpu.py): the data-parallel branch, the barrier / max-over-ranks timing and the result dict are
    the ones the GPU run uses."""
    import torch
    import torch.distributed as dist
    from salsa_amd.crnn.train import Trainer, synthetic_batch

    dev = torch.device(dev)
    on_gpu = dev.type == 'cuda'
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    kw = {} if amp_dtype == 'default' else {'amp_dtype': amp_dtype}
    if force_ddp:
        kw['ddp'] = True                          # (a 1-rank group: --force-ddp)
    tr = Trainer(dev, bf16_grad_allreduce=not fp32_grads, **kw)
    # N_ROT distinct device-resident batches rotate through the loop (round-3 review: ONE 115-MB batch for all steps stays in the
    # 256-MB Infinity Cache, and config 4 re-extracted the same 32 chunks every step): a step never sees the data of the step
    # before, 4 x 115 MB of features / 4 x 98 MB of audio do not fit the cache.  Every rank draws its own batches.
    n_rot = N_ROT if on_gpu else 1
    data = [synthetic_batch(batch, dev, seed=2021 + rank + 1000 * j, n_frames=n_frames) for j in range(n_rot)]
    ex, audios = None, None
    if on_the_fly:
        from salsa_amd.extractor import SalsaExtractor
        ex = SalsaExtractor(audio_format='mic', fmax_doa=4000, device=dev)
        # normalise-on-load (dataset/database.py:197-202) fused into the extraction, as the precomputed path's loader does
        ex.set_scaler(torch.full((4, 1, 200), -60.0, device=dev), torch.full((4, 1, 200), 12.0, device=dev))
        import numpy as np
        from salsa_amd.synth import synth_clip                        # seeded 8-s chunks (bursts + noise), 32 per rank
        audios = [torch.from_numpy(np.stack([synth_clip(4021 + (rank * n_rot + j) * batch + i, 8 * 24000) for i in range(batch)])).to(dev)
                  for j in range(n_rot)]
    aug_gen = torch.Generator().manual_seed(2021 + rank)
    it = [0]

    def step():
        j = it[0] % n_rot
        it[0] += 1
        xb, sb, db = data[j]
        if ex is not None:
            xb = ex.extract(audios[j])[:, :, :640]                  # (B,7,641,200) -> 640 frames
        if augment:
            from salsa_amd.augment import augment_batch
            xb, sb, db = augment_batch(xb, sb, db, 'mic' if ex is not None else 'foa', gen=aug_gen)
        return tr.train_step(xb, sb, db)

    for _ in range(warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()[0]
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    pw = None
    if power is not None and world == 1:   # OUTSIDE the timed region: the hwmon power figure is a slow moving average, so ~2 s more of
        tw0 = time.time()                  # the same steps run and the last 60 % of that is read (bench.py power_loop, same reason)
        while time.time() - tw0 < 2.0:
            for _ in range(10):
                step()
            sync()
        pw = power.stats(tw0 + 0.8, time.time())
    ranks = 1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ranks = dist.get_world_size()
    if rank != 0:
        return None
    from salsa_amd.crnn import nn_ops
    clips = world * batch * steps
    cps = clips / elapsed
    tflops = 3 * GFLOP_PER_CHUNK_FWD * cps / 1e3                     # fwd + bwd ~ 3x forward
    return {
        'metric': 'CRNN train clips/s', 'value': round(cps, 1), 'unit': '8-s chunks/s', 'n_gpus': world,
        'rccl_ranks': ranks, 'backend': (dist.get_backend() if (world > 1 or force_ddp) else None), 'steps': steps, 'warmup': warmup, 'ms_per_step': round(1e3 * elapsed / steps, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'CRNN (ResNet22 + BiGRU) training on %s SALSA chunks (7,%d,200), batch %d per GPU, Adam'
                               % ('on-the-fly extracted MIC (raw [B][4][192000] audio -> SALSA-MIC on device, scaler fused)' if on_the_fly else 'precomputed-FOA-shaped', n_frames, batch)
                               + (' + device augmentation' if augment else '') + '; %d distinct device-resident batches rotate' % n_rot,
                   'parallelism': 'dp%d' % world, 'grad_allreduce': 'fp32' if fp32_grads else 'bf16'},
        'roofline': {'bound': 'mfma', 'achieved': round(tflops / world, 1), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(tflops / world / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None,
                     'power': pw,
                     'note': 'per GPU; 134.8 GFLOP per chunk (3x the 44.93 GFLOP forward); convolutions: %s' % nn_ops.conv_backend_note()},
        'final_loss': float(loss)}



STOCK_ENV = {'SALSA_HIP_POOL': '0', 'SALSA_HIP_BN': '0', 'SALSA_HIP_CONV': '0', 'SALSA_HIP_CONV_WIDE': '0', 'SALSA_HIP_CONV_WIDE_WRW': '0',
             'SALSA_HIP_CONV_1X1': '0', 'SALSA_HIP_STEM_WRW': '0', 'SALSA_HIP_BN_POOL': '0', 'SALSA_HIP_BN_RES_POOL': '0', 'SALSA_HIP_FREQ_MEAN': '0',
             'SALSA_FUSED_GRU': '0', 'SALSA_FUSED_LOSS': '0', 'SALSA_HIP_ADAM': '0', 'SALSA_CONV_STATS': '0', 'SALSA_WIDE_CONV_STATS': '0',
             'SALSA_STEM_FUSED_BWD': '0', 'SALSA_FILTER_BANK': '0', 'SALSA_FUSED_SKIP': '0', 'SALSA_GEMM_1X1': '0'}


def torch_baseline(steps=8, warmup=3, batch=32, timeout=420):
    """The SAME training step on this box with every hand-written CRNN kernel switched off: torch / MIOpen convolutions, torch
    BatchNorm, pooling, GRU (MIOpen), loss and (fused) Adam, bf16 autocast, channels-last -- what `pip install torch` gives the
    reference's model on an MI355X.  A stated baseline (like cpu_baseline), measured in a child process (the switches are read at
    import) after this process's own legs; None on any failure, never an exception."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(STOCK_ENV)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(steps), '--warmup', str(warmup), '--batch', str(batch)],
                           capture_output=True, text=True, timeout=timeout, env=env)
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
        return {'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'], 'steps': steps, 'warmup': warmup,
                'kind': 'stock PyTorch-ROCm (MIOpen / hipBLASLt / torch kernels), same model, batch, dtype and box',
                'convolutions': line['roofline']['note'].split('convolutions: ')[-1], 'switches': 'SALSA_HIP_* = 0 (bench_crnn.STOCK_ENV)'}
    except Exception as e:  # noqa: BLE001 - a baseline, not the product
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}


def torch_infer_baseline(clips=64, steps=2, warmup=1, timeout=300):
    """Config 5's leg with every hand-written CRNN kernel switched off (the feature extraction stays ours: stock PyTorch has none): torch /
    MIOpen convolutions with folded BatchNorm, MIOpen GRU, bf16 autocast -- a stated baseline for `inference`, measured in a child process on
    a smaller job (the rate does not depend on the job size: sub-batches of 32); an error object on any failure, never an exception."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(STOCK_ENV)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--gpus', '1', '--infer', '--clips', str(clips), '--steps', str(steps),
                            '--warmup', str(warmup)], capture_output=True, text=True, timeout=timeout, env=env)
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
        return {'value': line['value'], 'unit': line['unit'], 'clips': clips, 'steps': steps, 'warmup': warmup,
                'p50_latency_ms_per_clip': line.get('p50_latency_ms_per_clip'),
                'kind': 'stock PyTorch-ROCm CRNN forward (MIOpen / hipBLASLt / torch kernels) behind the same HIP feature extraction, same model, '
                        'sub-batch, dtype and box', 'switches': 'SALSA_HIP_* = 0 (bench_crnn.STOCK_ENV)'}
    except Exception as e:  # noqa: BLE001 - a baseline, not the product
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}


def _flush_c_stdio():
    """push out whatever native libraries (RCCL's version banner) left in libc's stdout buffer -- every rank, as soon as its
    process group is gone, so that nothing of it can land after rank 0's result line"""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _emit(line):
    """print the ONE JSON line as the LAST thing on stdout: RCCL writes a version banner through C stdio, which on a pipe sits
    in libc's buffer until exit and would otherwise land after (and be taken for) the result line"""
    _flush_c_stdio()
    sys.stdout.write(json.dumps(line) + '\n')
    sys.stdout.flush()

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='default 20 (training) / 5 (--infer: a step is the whole 1024-clip job)')
    ap.add_argument('--warmup', type=int, default=None, help='default 5 (training) / 1 (--infer)')
    ap.add_argument('--batch', type=int, default=32, help='chunks per GPU per step')
    ap.add_argument('--on-the-fly', action='store_true', help='extract SALSA-MIC features from raw audio every step')
    ap.add_argument('--augment', action='store_true', help='apply the reference training augmentation on device every step')
    ap.add_argument('--fp32-grads', action='store_true', help='all-reduce fp32 gradients instead of bf16-compressed')
    ap.add_argument('--infer', action='store_true', help='config 5: batched inference, SALSA + CRNN forward on 60-s clips')
    ap.add_argument('--force-ddp', action='store_true',
                    help='one rank only: run the data-parallel gradient path over a 1-rank RCCL group anyway (what the N > 1 path '
                         'costs BEFORE any communication: hooks, bucket copies, the bf16 compression)')
    ap.add_argument('--clips', type=int, default=1024, help='--infer: 60-s clips of the WHOLE job per step, sharded over the GPUs (config 5: 1024)')
    ap.add_argument('--sub-batch', type=int, default=32,
                    help='--infer: clips per extraction + CRNN forward (config 5 says 32; 8 gives a third of the latency at 77 %% of the rate)')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 5 if args.infer else 20
    if args.warmup is None:
        args.warmup = 1 if args.infer else 5
    self_spawn(args.gpus, __file__)                                  # --gpus N without a launcher: become N ranks
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))

    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), 'bench_crnn.py needs an MI355X'
    # SALSA_BENCH_SHARE_GPU=1 (functional test of the N > 1 path on a one-GPU box): ranks share the visible devices round-robin
    # and the process group is gloo (RCCL refuses two ranks on one device) -- the timing means nothing then, the control flow
    # (sharding, barriers, max over ranks, the gradient synchroniser, rank-0-only line) is the real one
    share = os.environ.get('SALSA_BENCH_SHARE_GPU', '0') == '1'
    if share:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo') if share else dist.init_process_group('nccl', device_id=dev)
    elif args.force_ddp:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    if args.infer:
        from salsa_amd.crnn.train import Trainer
        line = infer_bench(args, rank, world, dev, Trainer(dev, ddp=False))
        if world > 1:
            dist.destroy_process_group()
            _flush_c_stdio()
        if line is not None:
            _emit(line)
        return
    line = train_bench(rank, world, dev, args.batch, args.steps, args.warmup, args.on_the_fly, args.augment, args.fp32_grads,
                       force_ddp=args.force_ddp)
    if world > 1 or args.force_ddp:
        dist.destroy_process_group()
        _flush_c_stdio()
    if line is not None:
        _emit(line)


if __name__ == '__main__':
    main()
