"""GPU tests (run with -m gpu) of the BASELINE.json configurations at their FULL sizes, closing round 1's untested list:
config 2 at batch 32, 60-s SALSA-Lite and SALSA-MIC clips against the oracle, config 4's on-the-fly 8-s MIC features against
the oracle, config 5 end to end (extract -> fused normalise -> CRNN forward on (7,4800,200) -> DCASE rows) against the same
weights on torch / MIOpen layers, a 2-rank RCCL DDP step (skipped below 2 GPUs), and the error bound of the kernels' float32
dB conversion over the whole float32 exponent range."""
import ctypes as C
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch

from conftest import ROOT
from salsa_amd.synth import synth_clip
from test_gpu_parity import ATOL_DB, RTOL, _check, _check_lite, _extractor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def test_config2_batch32_full_size_properties(dev, oracle):
    """BASELINE config 2 exactly as bench.py runs it: 32 x 60-s FOA clips in ONE call (939 MB of spill, 32-bit per-clip
    offsets at their largest).  Size-independent properties: determinism, the first and the last clip bit-equal to their
    solo runs (batch invariance across the whole workspace), zero band above upper_bin, unit-norm FOA vectors, and ALL 32
    clips of the batch against the oracle."""
    from bench import make_batch
    B, n = 32, 60 * 24000
    ys = make_batch(2021, B, n)
    ex = _extractor()
    a = torch.from_numpy(ys).to(dev)
    out = ex.extract(a)
    out2 = ex.extract(a).clone()
    assert torch.equal(out, out2)                                          # deterministic at full size
    assert out.shape == (B, 7, 4801, 200) and bool(torch.isfinite(out).all())
    assert not bool(out[:, 4:, :, 191:].any())                             # zero above upper_bin (:373-374)
    nrm = torch.sqrt((out[:, 4:].double() ** 2).sum(dim=1))
    emitted = nrm > 0
    frac = float(emitted[:, :, :191].double().mean())
    assert 0.02 < frac < 0.9, frac
    assert float((nrm[emitted] - 1.0).abs().max()) < 1e-6                  # FOA eigenvector features are unit vectors
    for i in (0, B - 1):
        solo = ex.extract(a[i:i + 1].contiguous())
        assert torch.equal(solo[0], out2[i]), 'clip %d depends on its batch neighbours' % i
    for i in range(B):                                                     # EVERY clip of the headline workload against the oracle (~0.7 s of CPU each)
        ref, aux = oracle.extract_salsa(ys[i], return_aux=True)
        _check(out2[i].cpu().numpy(), ref, aux['margin'])


def _both_instantiations(dev, ys, **kw):
    """The production covariance / eigen instantiation (packed float32 + float64 cold list) and the all-float64 one
    (FLAG_FORCE_F64) on ONE batch -- same STFT kernel, same tracker, so the same spill and masks: gate pattern bit-equal,
    features inside the bar, spectrogram channels bit-equal; returns the production run's solver counters."""
    from salsa_amd import _lib
    a = torch.from_numpy(ys).to(dev)
    ex = _extractor(**kw)
    ex.set_stats(True)
    out = ex.extract(a)
    st = ex.read_stats()
    ref = _extractor(flags=_lib.FLAG_FORCE_F64, **kw).extract(a)
    assert torch.equal(out[:, :4], ref[:, :4])
    g0, g1 = (out[:, 4:] != 0).any(dim=1), (ref[:, 4:] != 0).any(dim=1)
    assert torch.equal(g0, g1), 'gate patterns differ in %d TF bins' % int((g0 != g1).sum())
    err = (out[:, 4:] - ref[:, 4:]).abs() - (1e-6 + 1e-5 * ref[:, 4:].abs())
    assert float(err.max()) <= 0.0, float(err.max())
    st['emitted'] = int(g0.sum())
    return st


def test_config2_packed_solver_equals_float64_instantiation(dev):
    """Round-4 review item 2(i): the 32 clips of config 2 through both instantiations on one spill; 32 MIC 8-s chunks likewise."""
    from bench import make_batch
    st = _both_instantiations(dev, make_batch(2021, 32, 60 * 24000))
    assert 0 < st['cold_frames'] < 0.05 * st['gated_frames'], st
    print('config 2, FOA 32 x 60 s: %(gated_frames)d gated frames in %(items)d items, %(cold_frames)d to the float64 cold list, '
          '%(emitted)d emitted' % st)
    ys = np.stack([synth_clip(4021 + i, 8 * 24000) for i in range(32)])
    st = _both_instantiations(dev, ys, audio_format='mic', fmax_doa=4000)
    assert st['cold_frames'] < 0.1 * st['gated_frames'], st
    print('config 4 chunks, MIC 32 x 8 s: %(gated_frames)d gated frames, %(cold_frames)d to the float64 cold list, %(emitted)d emitted' % st)


def test_config2_fused_schedule_is_bit_identical(dev):
    """config 2 at full size through the fused STFT + covariance / eigen kernel (1024 segments of 152 frames): bit-identical."""
    from bench import make_batch
    a = torch.from_numpy(make_batch(2021, 32, 60 * 24000)).to(dev)
    ex = _extractor()
    ref = ex.extract(a).clone()
    ex.set_fused(1)
    assert torch.equal(ex.extract(a), ref)


def test_full_size_lite_and_mic_clips_against_oracle(dev, oracle):
    """One 60-s clip through SALSA-Lite (config 1's shape) and one through full SALSA MIC (fmax_doa 4 kHz, config 4's
    extractor) against the oracle -- the goldens hold 3-s clips only."""
    n = 60 * 24000
    y = synth_clip(2021, n)
    out = _extractor(audio_format='mic', feature_type='salsa_lite', fmax_doa=2000).extract(torch.from_numpy(y[None]).to(dev))
    assert out.shape == (1, 7, 4801, 191)
    _check_lite(out[0].cpu().numpy(), oracle.extract_lite(y, fmax_doa=2000))
    y = synth_clip(2022, n)
    out = _extractor(audio_format='mic', fmax_doa=4000).extract(torch.from_numpy(y[None]).to(dev))
    ref, aux = oracle.extract_salsa(y, fmax_doa=4000, audio_format='mic', return_aux=True)
    _check(out[0].cpu().numpy(), ref, aux['margin'])
    assert (out[0, 4:, :, :84] != 0).any() and not bool(out[0, 4:, :, 84:].any())


def test_on_the_fly_mic_chunks_against_oracle(dev, oracle):
    """config 4's feature leg: a batch of 8-s MIC chunks extracted on device (what the training step consumes) equals the
    oracle chunk by chunk; the 640-frame crop the trainer takes is a view of it."""
    ys = np.stack([synth_clip(60 + i, 8 * 24000) for i in range(4)])
    feats = _extractor(audio_format='mic', fmax_doa=4000).extract(torch.from_numpy(ys).to(dev))
    assert feats.shape == (4, 7, 641, 200)
    for i in range(4):
        ref, aux = oracle.extract_salsa(ys[i], fmax_doa=4000, audio_format='mic', return_aux=True)
        _check(feats[i].cpu().numpy(), ref, aux['margin'])


def test_config4_on_the_fly_augmented_training_leg(dev, oracle):
    """BASELINE config 4's per-step path, link by link (reference: SeldDataset.__getitem__ dataset/dataloader.py:37-62 under
    datamodule.py:137-142 into training_step models/seld_models.py:68-76), exactly as bench.py's `config4` leg runs it:
    (1) raw 8-s MIC chunks -> SALSA-MIC on device, each chunk equal to the oracle, and with the scaler attached equal to the
        oracle's features normalised on load (database.py:197-202);
    (2) the 640-frame crop (a strided view, no copy) -> augment_batch with a fixed generator: the one-pass HIP kernel equals
        the torch-operator composite of the same draws bit for bit, and so do the targets;
    (3) ten optimizer steps on that fixed augmented batch reduce the loss."""
    from salsa_amd.augment import apply_augment_torch, augment_batch, draw_augment
    from salsa_amd.crnn.train import Trainer, synthetic_batch
    B = 4
    ys = np.stack([synth_clip(4021 + i, 8 * 24000) for i in range(B)])
    ex = _extractor(audio_format='mic', fmax_doa=4000)
    a = torch.from_numpy(ys).to(dev)
    raw = ex.extract(a).clone()
    refs = []
    for i in range(B):
        ref, aux = oracle.extract_salsa(ys[i], fmax_doa=4000, audio_format='mic', return_aux=True)
        _check(raw[i].cpu().numpy(), ref, aux['margin'])
        refs.append(ref)
    mean = torch.full((4, 1, 200), -60.0, device=dev)
    std = torch.full((4, 1, 200), 12.0, device=dev)
    ex.set_scaler(mean, std)
    feats = ex.extract(a)
    assert feats.shape == (B, 7, 641, 200)
    want = raw.clone()
    want[:, :4] = (raw[:, :4] - mean) / std
    np.testing.assert_allclose(feats.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-6)
    x = feats[:, :, :640]                                                  # the trainer's crop: a view with its own strides
    assert not x.is_contiguous()
    _, sed, doa = synthetic_batch(B, dev, seed=7)
    # (2) same generator state -> same draws -> the kernel must equal the torch composite
    xa, sa, da = augment_batch(x, sed, doa, 'mic', gen=torch.Generator().manual_seed(99))
    d = draw_augment(B, 640, 200, 'mic', torch.Generator().manual_seed(99))
    xt, dt = apply_augment_torch(x.contiguous(), doa, d, 'mic')
    assert torch.equal(xa, xt) and torch.equal(da, dt) and torch.equal(sa, sed)
    assert not torch.equal(xa, x.contiguous())                             # the draws did something
    assert bool((d['h'] > 0).any()) and bool((d['m'] > 0).any() or (d['shift'] > 0).any())
    # (3) the step itself
    tr = Trainer(dev, total_steps=100)
    losses = [float(tr.train_step(xa, sa, da)[0]) for _ in range(10)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_decibel_conversion_error_bound(dev):
    """10*log10(max(1e-10, p)) in the STFT kernel is 3.0103 * v_log_f32(p): sweep EVERY float32 exponent with a dense
    mantissa grid (plus the neighbours of 1.0 and of the 1e-10 clamp) against float64.  Bar: the reference's own float32
    result is only defined to 1e-5 relative (north_star); in the range 4-channel audio in [-1, 1) can reach
    (-100 .. +60 dB) the absolute error must stay below the 2e-5 dB the parity tests use."""
    from salsa_amd import _lib
    mant = np.concatenate([np.arange(0, 1 << 23, 1 << 11, dtype=np.uint32),                # 4096 evenly spaced mantissas
                           np.arange(0, 64, dtype=np.uint32), (1 << 23) - 1 - np.arange(0, 64, dtype=np.uint32)])
    expo = np.arange(1, 255, dtype=np.uint32)                                                # all normal exponents
    bits = (expo[:, None] << 23) | mant[None, :]
    p = bits.reshape(-1).view(np.float32)
    extra = np.array([0.0, 1e-10, np.nextafter(np.float32(1e-10), np.float32(1)), np.nextafter(np.float32(1e-10), np.float32(0)),
                      1e-45, 1e-40, 1.0], np.float32)                                        # clamp edge, denormals, exact 1
    p = np.concatenate([p, extra])
    d_p = torch.from_numpy(p).to(dev)
    d_o = torch.empty_like(d_p)
    rc = _lib.load().salsa_selftest_decibel(C.c_void_p(d_p.data_ptr()), C.c_void_p(d_o.data_ptr()), d_p.numel(),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0, _lib.last_error()
    got = d_o.cpu().numpy().astype(np.float64)
    ref = 10.0 * np.log10(np.maximum(1e-10, p.astype(np.float64)))
    err = np.abs(got - ref)
    assert np.all(err <= ATOL_DB + RTOL * np.abs(ref)), 'worst %g dB at p=%g' % (err.max(), p[err.argmax()])
    audible = (ref >= -100.0) & (ref <= 60.0)
    assert err[audible].max() < ATOL_DB, 'worst in the audio range: %g dB at p=%g' % (err[audible].max(), p[audible][err[audible].argmax()])
    assert got[len(p) - len(extra)] == pytest.approx(-100.0, abs=1e-5)                      # p = 0 sits on the amin clamp
    print('db10: max abs error %.3g dB overall, %.3g dB within [-100, 60] dB' % (err.max(), err[audible].max()))


def test_config5_end_to_end_inference_matches_torch_layers(dev):
    """BASELINE config 5 / reference test_step (models/seld_models.py:110-117): raw 60-s FOA clips -> SALSA features with
    normalise-on-load fused -> CRNN forward on (7,4800,200) -> sigmoid / xyz -> combine_chunks -> DCASE rows.  The same
    weights and features through torch / MIOpen layers (every hand-written CRNN kernel off) must give the same file-level
    outputs within bf16 noise, and the same rows wherever the SED probability is not within that noise of the threshold."""
    from salsa_amd.crnn import model as M, nn_ops
    from salsa_amd.crnn.postprocess import combine_chunks, to_dcase_rows
    from salsa_amd.crnn.testing import seeded_fill
    from salsa_amd.crnn.train import Trainer
    B = 2
    ys = np.stack([synth_clip(3000 + i, 60 * 24000) for i in range(B)])
    ex = _extractor()
    raw = ex.extract(torch.from_numpy(ys).to(dev)).clone()
    rng = np.random.RandomState(5)
    mean = (raw[:, :4].mean(dim=(0, 2), keepdim=False).cpu().numpy()[:, None, :]).astype(np.float32)     # (4,1,200): a plausible scaler
    std = (raw[:, :4].std(dim=(0, 2)).cpu().numpy()[:, None, :] + 1.0 + rng.rand(4, 1, 200)).astype(np.float32)
    ex.set_scaler(mean, std)
    feats = ex.extract(torch.from_numpy(ys).to(dev))
    ref_norm = (raw[:, :4].cpu().numpy() - mean[None]) / std[None]
    np.testing.assert_allclose(feats[:, :4].cpu().numpy(), ref_norm, rtol=1e-6, atol=1e-6)               # database.py:197-202
    assert torch.equal(feats[:, 4:], raw[:, 4:])                                                         # spatial channels untouched
    x = feats[:, :, :4800]                                                                               # database.py:203-207 trim
    tr = Trainer(dev, total_steps=10)
    seeded_fill(tr.raw_model, 11)
    with torch.no_grad():
        for mod in tr.raw_model.modules():                 # non-trivial running statistics for the folded BatchNorm path
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.uniform_(-0.1, 0.1)
                mod.running_var.uniform_(0.8, 1.2)

    def run(on):
        nn_ops.USE_HIP_POOL = nn_ops.USE_HIP_BN = nn_ops.USE_HIP_CONV = on
        M.FUSED_GRU = on
        p, d = tr.infer(x if on else x.contiguous())
        return p.cpu().numpy(), d.cpu().numpy()

    try:
        p_on, d_on = run(True)
        p_off, d_off = run(False)
    finally:
        nn_ops.USE_HIP_POOL = nn_ops.USE_HIP_BN = nn_ops.USE_HIP_CONV = True
        M.FUSED_GRU = True
    assert p_on.shape == (B, 600, 12) and d_on.shape == (B, 600, 36)
    np.testing.assert_allclose(p_on, p_off, rtol=0, atol=3e-2)
    np.testing.assert_allclose(d_on, d_off, rtol=0, atol=6e-2)
    for i in range(B):
        # one 4800-frame chunk per file (test_chunk_len = 4800): combine_chunks is the identity placement
        file_p_on, file_p_off = combine_chunks(p_on[i][None], 600, 600), combine_chunks(p_off[i][None], 600, 600)
        assert np.array_equal(file_p_on, p_on[i])
        thr = float(np.median(p_on[i]))                    # untrained weights: put the threshold where classes are active
        rows_on = to_dcase_rows(file_p_on, d_on[i], sed_threshold=thr)
        rows_off = to_dcase_rows(file_p_off, d_off[i], sed_threshold=thr)
        sure = np.abs(p_on[i] - thr) > 3e-2                # decisions outside bf16 noise of the threshold must agree
        act_on = {(r[0], r[1]) for r in rows_on if sure[r[0], r[1]]}
        act_off = {(r[0], r[1]) for r in rows_off if sure[r[0], r[1]]}
        assert act_on == act_off and len(rows_on) > 100
        ang_on = {(r[0], r[1]): (r[3], r[4]) for r in rows_on}
        ang_off = {(r[0], r[1]): (r[3], r[4]) for r in rows_off}
        both = sorted(set(ang_on) & set(ang_off))
        # xyz within 6e-2 => angles within a few degrees unless the vector is tiny (untrained tanh heads): compare medians
        da = np.array([abs(((ang_on[k][0] - ang_off[k][0] + 180) % 360) - 180) for k in both])
        assert np.median(da) <= 5, np.median(da)


def test_config5_one_gpu_share_128_distinct_clips_sharded(dev):
    """BASELINE config 5 at ONE GPU's share (1024 clips / 8 GPUs = 128, sub-batches of 32) through the product's sharded engine
    (salsa_amd.crnn.infer.infer_clips_sharded): 128 DISTINCT seeded clips resident in HBM -> features -> CRNN forward -> DCASE
    rows on the host.  Checked: every clip answered, in the sorted-name order, distinct clips give distinct rows; three clips of
    different sub-batches against their SOLO runs (batch invariance through the extractor, the CRNN and the pipelined engine);
    one clip against the same weights on torch / MIOpen layers; latency stamps cover every clip exactly once."""
    from salsa_amd.crnn import model as M, nn_ops
    from salsa_amd.crnn.infer import infer_clips_sharded
    from salsa_amd.crnn.testing import seeded_fill
    from salsa_amd.crnn.train import Trainer
    from salsa_amd.synth import synth_clips_device
    n_clips, sub = 128, 32
    audio = synth_clips_device(7000, n_clips, 60 * 24000, device=dev)
    assert not torch.equal(audio[0], audio[1])
    ex = _extractor()
    ex.set_scaler(np.full((4, 1, 200), -60.0, np.float32), np.full((4, 1, 200), 12.0, np.float32))
    tr = Trainer(dev, total_steps=10)
    seeded_fill(tr.raw_model, 11)
    names = ['clip%04d' % i for i in range(n_clips)]
    index = {n: i for i, n in enumerate(names)}
    seen = {}

    def featurize(group):
        idx = [index[n] for n in group]
        assert idx == list(range(idx[0], idx[0] + len(idx)))               # contiguous ranges of the sorted list
        return ex.extract(audio[idx[0]:idx[0] + len(idx)])[:, :, :4800]

    def forward(x):
        p, d = tr.infer(x)
        seen[len(seen)] = (p.clone(), d.clone())
        return p, d

    p0, _ = tr.infer(featurize(names[:sub]))
    thr = float(torch.quantile(p0.flatten().float()[:1 << 20], 0.9).item())
    seen.clear()
    stamps = []
    rows = infer_clips_sharded(names[::-1], featurize, forward, rank=0, world=1, sub_batch=sub, sed_threshold=thr, stamps=stamps)
    assert sorted(rows) == names and len(seen) == n_clips // sub
    assert sorted((a, b) for a, b, _, _ in stamps) == [(i, i + sub) for i in range(0, n_clips, sub)]
    assert all(t1 > t0 for _, _, t0, t1 in stamps)
    assert all(len(r) > 50 for r in rows.values())
    assert rows[names[0]] != rows[names[1]] and rows[names[40]] != rows[names[100]]
    # the second rank's share of a 2-rank job is the second half of the sorted list, with the same answers
    half = infer_clips_sharded(names, featurize, tr.infer, rank=1, world=2, sub_batch=sub, sed_threshold=thr, gather=False)
    assert sorted(half) == names[64:] and all(half[n] == rows[n] for n in half)
    # batch invariance: clips 3, 45 and 127 alone
    for i in (3, 45, 127):
        p_b, d_b = seen[i // sub][0][i % sub], seen[i // sub][1][i % sub]
        p_s, d_s = tr.infer(ex.extract(audio[i:i + 1])[:, :, :4800])
        assert float((p_s[0] - p_b).abs().max()) <= 2e-2 and float((d_s[0] - d_b).abs().max()) <= 4e-2, i
        solo = infer_clips_sharded([names[i]], lambda g: ex.extract(audio[i:i + 1])[:, :, :4800], tr.infer, sub_batch=sub,
                                   sed_threshold=thr)[names[i]]
        sure = (p_b - thr).abs().cpu().numpy() > 2e-2
        assert {(r[0], r[1]) for r in solo if sure[r[0], r[1]]} == {(r[0], r[1]) for r in rows[names[i]] if sure[r[0], r[1]]}
    # one clip against torch / MIOpen layers (every hand-written CRNN kernel off)
    x = ex.extract(audio[77:78])[:, :, :4800].contiguous()
    try:
        nn_ops.USE_HIP_POOL = nn_ops.USE_HIP_BN = nn_ops.USE_HIP_CONV = False
        M.FUSED_GRU = False
        p_off, d_off = tr.infer(x)
    finally:
        nn_ops.USE_HIP_POOL = nn_ops.USE_HIP_BN = nn_ops.USE_HIP_CONV = True
        M.FUSED_GRU = True
    p_on, d_on = seen[77 // sub][0][77 % sub], seen[77 // sub][1][77 % sub]
    assert float((p_off[0] - p_on).abs().max()) <= 3e-2 and float((d_off[0] - d_on).abs().max()) <= 6e-2


def test_bench_eight_ranks_on_one_gpu_with_the_drivers_command():
    """The driver's scaling run, `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 8 --steps K --warmup W`, dry-run on the ONE GPU of this box (SALSA_BENCH_SHARE_GPU=1: gloo process group, the
    eight ranks share cuda:0; the timing means nothing, the control flow is the real one): eight ranks in every leg, world read back
    from the process group, the 1024-clip inference job sharded 128 per rank, ONE JSON line, last on stdout, and the CPU-baseline
    leg (rank 0, after the process group is gone) inside the cores this container may use."""
    import json
    import subprocess
    from oracle.cpu_bench import effective_cpus
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(SALSA_BENCH_SHARE_GPU='1', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1',
           '--blocks', '2', '--crnn-steps', '2', '--crnn-warmup', '1', '--infer-steps', '1']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    json_lines = [ln for ln in out_lines if ln.startswith('{')]
    assert len(json_lines) == 1 and out_lines[-1] == json_lines[0]          # ONE line, from rank 0 only, last
    line = json.loads(json_lines[0])
    assert line['n_gpus'] == 8 and line['rccl_ranks'] == 8 and line['value'] > 0 and line['scaling'] == 'weak'
    assert len(line['blocks_ms']) == 2 and line['config']['sharding'].startswith('clips/8')
    for leg in ('crnn', 'config4'):
        assert line[leg]['n_gpus'] == 8 and line[leg]['rccl_ranks'] == 8 and line[leg]['value'] > 0, leg
        assert line[leg]['config']['parallelism'] == 'dp8'
    inf = line['inference']
    assert inf['n_gpus'] == 8 and inf['config']['clips'] == 1024 and inf['config']['clips_per_gpu'] == 128
    assert inf['latency_samples'] == 1024 and inf['value'] > 0              # every rank's clips stamped exactly once per step
    assert inf['p50_latency_ms_per_clip'] <= inf['p50_latency_from_arrival_ms'] <= inf['max_latency_from_arrival_ms']
    cpu = line['cpu_baseline']
    assert cpu is not None and cpu['value'] > 0 and 1 <= cpu['cores'] <= effective_cpus()['effective_cores']
    assert all(c['cores'] <= effective_cpus()['effective_cores'] for c in cpu.get('configs', []))
    assert 'status' not in line


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _nccl_ddp_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from salsa_amd.crnn.train import Trainer, synthetic_batch
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    d = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=d)
    tr = Trainer(d, total_steps=10, bf16_grad_allreduce=False)
    before = torch.cat([p.detach().flatten() for p in tr.raw_model.parameters()]).clone()
    x, sed, doa = synthetic_batch(2, d, seed=100 + rank)                    # different chunks per rank
    loss, _, _ = tr.train_step(x, sed, doa)
    after = torch.cat([p.detach().flatten() for p in tr.raw_model.parameters()])
    torch.save({'after': after.cpu(), 'moved': float((after - before).abs().max()), 'loss': float(loss),
                'ranks': dist.get_world_size(), 'backend': dist.get_backend()}, os.path.join(tmp, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs on one node (RCCL over xGMI)')
def test_two_rank_rccl_ddp_step_keeps_replicas_identical(tmp_path):
    """config 4's parallelism on real hardware: two processes, one GPU each, DDP gradient all-reduce on RCCL."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_nccl_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'r0.pt'), torch.load(tmp_path / 'r1.pt')
    assert r0['ranks'] == 2 and r0['backend'] == 'nccl'
    assert r0['moved'] > 0 and np.isfinite(r0['loss']) and np.isfinite(r1['loss']) and r0['loss'] != r1['loss']
    assert torch.equal(r0['after'], r1['after'])                            # the all-reduced update is identical


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs on one node')
def test_bench_self_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher must come back with n_gpus = rccl_ranks = 2 and both metric halves."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--blocks', '1', '--seconds', '10', '--crnn-steps', '2', '--crnn-warmup', '1'],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2
    assert line['crnn']['n_gpus'] == 2 and line['crnn']['rccl_ranks'] == 2 and line['crnn']['value'] > 0


def _shared_gpu_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from salsa_amd.crnn.train import Trainer, synthetic_batch
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    torch.cuda.set_device(0)
    d = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)           # (RCCL refuses two ranks on one device)
    out = {}
    for k in range(2):                                                     # two Trainers one after the other, as bench.py's legs
        tr = Trainer(d, total_steps=10)
        assert tr.grad_sync is not None
        x, sed, doa = synthetic_batch(2, d, seed=100 + rank + 10 * k)      # different chunks per rank
        t0 = time.perf_counter()
        for _ in range(3):
            loss, _, _ = tr.train_step(x, sed, doa)
        torch.cuda.synchronize()
        out['after%d' % k] = torch.cat([p.detach().flatten() for p in tr.raw_model.parameters()]).cpu()
        out['loss%d' % k], out['s%d' % k] = float(loss), time.perf_counter() - t0
    torch.save(out, os.path.join(tmp, 'g%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharing_the_gpu_keep_replicas_identical(tmp_path):
    """The N > 1 training path on the real HIP kernels with ONE GPU: two processes on cuda:0, gloo process group, the bucketed
    gradient synchroniser (salsa_amd/crnn/grad_sync.py) on CUDA tensors, two Trainers in a row per process (bench.py's `crnn`
    and `config4` legs).  Replicas stay bit-identical although the ranks see different chunks, and the second Trainer is as
    fast as the first (a synchroniser that ran its collectives from a private stream made every second one 100x slower)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_shared_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'g0.pt'), torch.load(tmp_path / 'g1.pt')
    for k in range(2):
        assert np.isfinite(r0['loss%d' % k]) and r0['loss%d' % k] != r1['loss%d' % k]
        assert torch.equal(r0['after%d' % k], r1['after%d' % k])
    assert r0['s1'] < 5 * r0['s0'] + 5.0, (r0['s0'], r0['s1'])


def test_bench_two_ranks_on_one_gpu():
    """`SALSA_BENCH_SHARE_GPU=1 python bench.py --gpus 2` (no launcher: it re-executes itself under torch.distributed.run): the
    whole N = 2 control flow of the driver's command on one device -- n_gpus = ranks = 2 in both halves, every leg present,
    the JSON line last on stdout."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['SALSA_BENCH_SHARE_GPU'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--blocks', '1', '--crnn-steps', '2', '--crnn-warmup', '1', '--infer-steps', '2', '--infer-clips', '128', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)                                                 # the result is the LAST line
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['value'] > 0
    for leg in ('crnn', 'config4'):
        assert line[leg]['n_gpus'] == 2 and line[leg]['rccl_ranks'] == 2 and line[leg]['value'] > 0, leg
    assert line['inference']['value'] > 0 and line['inference']['config']['clips_per_gpu'] == 64
    assert line['inference']['latency_samples'] == 2 * 128


def _sharded_harness_worker(rank, world, port, cfg):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from salsa_amd.distributed import extract_features_sharded
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    extract_features_sharded(data_config=cfg, batch_size=2)
    dist.destroy_process_group()


def test_sharded_harness_with_two_ranks_on_one_gpu_equals_the_single_process_tree(tmp_path):
    """salsa_amd.distributed.extract_features_sharded (rank r extracts the r-th contiguous range of every split's sorted file
    list on ITS process's extractor; the scaler's partial sums are all-reduced, rank 0 writes it) run by two gloo ranks that share
    cuda:0, against salsa_amd.features.extract_features in one process: the same files, bit-identical features, the same scaler."""
    import torch.multiprocessing as mp
    from salsa_amd import io as sio
    from salsa_amd.features import extract_features
    from test_gpu_parity import _make_tree
    clips = {('dev|fold%d_clip%d' % (1 + i % 2, i)): synth_clip(900 + i, 24000 + 3000 * i) for i in range(5)}
    clips['eval|mix001'] = synth_clip(950, 30000)
    trees = {}
    for mode in ('single', 'sharded'):
        d = tmp_path / mode
        d.mkdir()
        cfg, feat_dir = _make_tree(str(d), 'foa', clips, 9000)
        if mode == 'single':
            extract_features(data_config=cfg, batch_size=2)
        else:
            mp.spawn(_sharded_harness_worker, args=(2, _free_port(), cfg), nprocs=2, join=True)
        found = {}
        for root, _, files in os.walk(feat_dir):
            for f in files:
                found[os.path.relpath(os.path.join(root, f), feat_dir)] = sio.load_arrays(os.path.join(root, f))
        trees[mode] = found
    assert sorted(trees['single']) == sorted(trees['sharded']) and len(trees['single']) == 7       # 6 clips + the scaler
    for rel, arrs in trees['single'].items():
        for k, v in arrs.items():
            w = trees['sharded'][rel][k]
            if rel.endswith('scaler.h5'):
                np.testing.assert_allclose(w, v, rtol=1e-6, atol=1e-6)        # (float64 partial sums added in another order)
            else:
                assert np.array_equal(w, v), (rel, k)


def test_bench_n2_failure_in_a_crnn_leg_still_prints_the_feature_half():
    """N = 2 (both ranks on the one GPU): the last rank fails in the `config4` leg while rank 0 waits in that leg's collectives.
    The job must end non-zero AND rank 0 must still print the line -- feature-path half complete, `crnn` present, `config4`
    null, `status` saying why -- instead of losing the headline number to a failure on the consumer side."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(SALSA_BENCH_SHARE_GPU='1', SALSA_BENCH_FAIL_LEG='config4', SALSA_BENCH_LEG_TIMEOUT='45')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--blocks', '1', '--crnn-steps', '2', '--crnn-warmup', '1', '--infer-steps', '1', '--infer-clips', '64', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert lines, r.stderr[-2000:]
    line = json.loads(lines[-1])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['roofline']['frac'] > 0
    assert line['crnn'] is not None and line['crnn']['value'] > 0 and line['config4'] is None
    assert line['status'].startswith('partial')
