#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the UPSTREAM reference (imported from /root/reference, unmodified) in the
build container.  Run:  python tools/make_golden.py          (needs /root/reference; never runs on the GPU box)

The reference has no tests, golden vectors or fixtures of its own, so these vectors ARE the pin:
every array below is an output of the reference's own functions
  dataset/salsa_feature_extraction.py   extract_normalized_eigenvector (:17-129), MagStftExtractor (:132-201),
                                         compute_scaler (:204-262), extract_features (:265-391)
  dataset/salsa_lite_feature_extraction.py  extract_features (:18-137)
  contrib/salsa_flexible.py             SalsaFeatures / SalsaLiteFeatures (:271-400), the on-the-fly surface
  utilities/transforms.py               the SALSA training augmentations (datamodule.py:45-52, :73-82)
  metrics/SELD2021_evaluation_metrics.py, metrics/dcase_utils.py   SELD scores of DCASE-format CSV rows
driven through tools/ref_shims.py (librosa 0.8.0 / h5py / fire stand-ins -- third-party arithmetic restated there).
Inputs are regenerated from seeds by salsa_amd/synth.py; each fixture stores the SHA-256 of every input so a drifted
generator is detected.  Fixtures hold data only (inputs' hashes, parameters, expected outputs).
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402

ref_shims.install()
from dataset import salsa_feature_extraction as ref_salsa  # noqa: E402  (the reference)
from dataset import salsa_lite_feature_extraction as ref_lite  # noqa: E402  (the reference)

from salsa_amd.synth import sha256_of, synth_clip, synth_stft_block  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def save(name, meta, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print('%-34s %8.1f KB  %s' % (name, os.path.getsize(path) / 1024, {k: v.shape for k, v in arrays.items()}))


# ----------------------------------------------------------------------------------------------- G5: W + bin limits
def g5_w_and_bins():
    arrays, meta = {}, {'what': 'MagStftExtractor.W and bin-limit table', 'bins': []}
    for n_fft, hop in ((512, 300), (256, 150)):
        for comp in (True, False):
            ex = ref_salsa.MagStftExtractor(n_fft=n_fft, hop_length=hop, win_length=n_fft, is_compress_high_freq=comp)
            arrays['W_%d_%s' % (n_fft, 'c' if comp else 'n')] = ex.W
    for fs, n_fft, fmin, fmax in ((24000, 512, 50, 9000), (24000, 512, 50, 4000), (24000, 512, 50, 2000),
                                  (24000, 256, 50, 9000), (24000, 256, 50, 4000), (24000, 512, 0, 20000),
                                  (48000, 512, 50, 9000), (16000, 512, 100, 7999), (32000, 256, 125, 5000)):
        # exact statements of salsa_feature_extraction.py:298-304 (same in lite :50-54, cutoff :57-58)
        fmax_c = np.min((fmax, fs // 2))
        lower = int(np.floor(fmin * n_fft / float(fs)))
        upper = int(np.floor(fmax_c * n_fft / float(fs)))
        lower = int(np.max((1, lower)))
        cutoff = int(np.floor(9000 * n_fft / float(fs)))
        meta['bins'].append([fs, n_fft, fmin, fmax, lower, upper, cutoff])
    save('g5_w_bins', meta, **arrays)


# ----------------------------------------------------------------------------------------------- G1/G2/G7: eigvec
def run_eig(X, fmt, track, cond=5.0, lower_bin=1, fs=24000, n_fft=512):
    with np.errstate(all='ignore'):
        return ref_salsa.extract_normalized_eigenvector(
            X.astype(complex), condition_number=cond, n_hopframes=3, is_tracking=track, audio_format=fmt,
            fs=fs, n_fft=n_fft, lower_bin=lower_bin)


def g1_eigvec():
    for seed, kind, nb, nt in ((0, 'mixed', 24, 96), (1, 'mixed', 16, 64), (2, 'noise', 12, 48)):
        X = synth_stft_block(seed, nb, nt, kind=kind)
        arrays, meta = {}, {'seed': seed, 'kind': kind, 'n_bins': nb, 'n_frames': nt, 'sha': sha256_of(X),
                            'lower_bin': 1, 'fs': 24000, 'n_fft': 512, 'cond': 5.0}
        for fmt in ('foa', 'mic'):
            for track in (True, False):
                arrays['%s_%s' % (fmt, 'track' if track else 'notrack')] = run_eig(X, fmt, track)
        # G7: cond=0 makes the coherence gate vacuous (s0 > 0), so the non-zero pattern IS the tracker's indicator_sig
        arrays['sig_mask'] = (np.abs(run_eig(X, 'foa', True, cond=0.0)).sum(axis=0) > 0)
        arrays['foa_track_cond2'] = run_eig(X, 'foa', True, cond=2.0)
        arrays['mic_track_lb7'] = run_eig(X, 'mic', True, lower_bin=7)
        save('g1_eigvec_s%d' % seed, meta, **arrays)


def g2_adversarial():
    rng = np.random.RandomState(77)
    nb, nt = 10, 40
    cases = {}
    # (a) exact rank-1: every frame a scalar multiple of one steering vector
    steer = np.array([1.0, 0.5 - 0.25j, -0.75 + 0.1j, 0.2 + 0.6j])
    s = rng.randn(nb, nt) + 1j * rng.randn(nb, nt)
    cases['rank1'] = (s[:, :, None] * steer[None, None, :])
    # (b) two orthogonal sources with amplitude ratio around sqrt(cond) -> lambda1/lambda2 around 5
    a = np.array([1, 1, 1, 1]) / 2.0
    b = np.array([1, -1, 1, -1]) / 2.0
    Xb = np.zeros((nb, nt, 4), complex)
    for ib in range(nb):
        ratio = np.sqrt(5.0) * (0.98 + 0.004 * ib)      # straddles the gate
        ph = np.exp(1j * rng.uniform(-np.pi, np.pi, nt))
        ph2 = np.exp(1j * rng.uniform(-np.pi, np.pi, nt))
        Xb[ib] = ratio * ph[:, None] * a[None, :] + ph2[:, None] * b[None, :]
    cases['margin'] = Xb
    # (c) silent block then signal: floor clamps at 1e-6, all-zero covariance
    Xc = np.zeros((nb, nt, 4), complex)
    Xc[:, 25:, :] = (rng.randn(nb, 15, 4) + 1j * rng.randn(nb, 15, 4)) * 1e-3
    cases['silent'] = Xc
    # (d) FOA W-channel component tiny: division by u[0,0] blows up (reference has no guard, :118)
    Xd = (rng.randn(nb, nt, 1) + 1j * rng.randn(nb, nt, 1)) * np.array([1e-7, 1.0, 0.5, -0.3])[None, None, :]
    Xd = Xd + 1e-9 * (rng.randn(nb, nt, 4) + 1j * rng.randn(nb, nt, 4))
    cases['w_tiny'] = Xd
    # (e) constant magnitude: tracker slow-rise path (countdown < 0) then falls behind
    ph = np.exp(1j * rng.uniform(-np.pi, np.pi, (nb, nt, 1)))
    cases['const_mag'] = ph * np.array([1.0, 0.3, 0.2, 0.1])[None, None, :] * \
        (1.0 + 0.3 * np.sin(np.arange(nt) / 3.0))[None, :, None]
    # (f) level steps: above/below-floor transitions reset the countdown
    env = np.where((np.arange(nt) // 5) % 2 == 0, 1.0, 0.05)
    cases['steps'] = (rng.randn(nb, nt, 4) + 1j * rng.randn(nb, nt, 4)) * env[None, :, None]
    arrays, meta = {}, {'lower_bin': 1, 'fs': 24000, 'n_fft': 512, 'cond': 5.0, 'cases': sorted(cases)}
    for name, X in cases.items():
        X = X.astype(np.complex64)
        arrays['X_' + name] = X          # adversarial inputs are stored (tiny), not regenerated
        for fmt in ('foa', 'mic'):
            for track in (True, False):
                arrays['%s_%s_%s' % (name, fmt, 'track' if track else 'notrack')] = run_eig(X, fmt, track)
        arrays['%s_sig_mask' % name] = (np.abs(run_eig(X, 'mic', True, cond=0.0)).sum(axis=0) != 0) | \
                                       (np.abs(run_eig(X, 'foa', True, cond=0.0)).sum(axis=0) > 0)
    save('g2_adversarial', meta, **arrays)


# ----------------------------------------------------------------------------------------------- harness runs
def make_tree(tmp, fmt, dev_clips, eval_clips, fs=24000, n_fft=512, hop=300, fmin=50, fmax=9000):
    data_dir = os.path.join(tmp, 'data')
    feat_dir = os.path.join(tmp, 'feat')
    for split, clips in ((fmt + '_dev', dev_clips), (fmt + '_eval', eval_clips)):
        d = os.path.join(data_dir, split)
        os.makedirs(d, exist_ok=True)
        for name, audio in clips.items():
            p = os.path.join(d, name + '.wav')
            open(p, 'wb').close()
            ref_shims.register_audio(p, audio)
    cfg = {'data_dir': data_dir, 'feature_dir': feat_dir,
           'data': {'format': fmt, 'fs': fs, 'n_fft': n_fft, 'win_len': n_fft, 'hop_len': hop,
                    'fmin_doa': fmin, 'fmax_doa': fmax}}
    cfg_path = os.path.join(tmp, 'cfg.yml')
    with open(cfg_path, 'w') as f:
        yaml.safe_dump(cfg, f)
    return cfg_path, feat_dir


def collect(feat_dir):
    out = {}
    for path, dsets in ref_shims.H5_STORE.items():
        if path.startswith(os.path.abspath(feat_dir)):
            rel = os.path.relpath(path, feat_dir)
            for k, v in dsets.items():
                out[rel.replace(os.sep, '|') + '|' + k] = v
    return out


def g3_end_to_end():
    specs = {
        'foa': dict(fmax=9000, dev={'fold1_room1_mix001': (101, 36000), 'fold1_room1_mix002': (102, 24000),
                                    'fold2_room1_mix003': (103, 19200)}, ev={'mix001': (104, 12000)}),
        'mic': dict(fmax=4000, dev={'fold1_room1_mix001': (111, 24000), 'fold3_room2_mix007': (112, 19200)},
                    ev={'mix001': (113, 9000)}),
    }
    for fmt, sp in specs.items():
        tmp = tempfile.mkdtemp()
        ref_shims.H5_STORE.clear()
        dev = {k: synth_clip(s, n) for k, (s, n) in sp['dev'].items()}
        ev = {k: synth_clip(s, n) for k, (s, n) in sp['ev'].items()}
        cfg_path, feat_dir = make_tree(tmp, fmt, dev, ev, fmax=sp['fmax'])
        with np.errstate(all='ignore'):
            ref_salsa.extract_features(data_config=cfg_path, cond_num=5, n_hopframes=3, is_tracking=True,
                                       is_compress_high_freq=True, task='feature_scaler')
        arrays = collect(feat_dir)
        meta = {'format': fmt, 'fmax_doa': sp['fmax'], 'fmin_doa': 50, 'fs': 24000, 'n_fft': 512, 'hop': 300,
                'cond_num': 5, 'clips': {('dev|' + k): [s, n, sha256_of(dev[k])] for k, (s, n) in sp['dev'].items()}}
        meta['clips'].update({('eval|' + k): [s, n, sha256_of(ev[k])] for k, (s, n) in sp['ev'].items()})
        save('g3_salsa_%s' % fmt, meta, **arrays)
        shutil.rmtree(tmp)
    # variants: no tracking + no compression + n_fft 256 on one short FOA clip (dir-name suffixes are part of the pin)
    for tag, kw, nfft, hop in (('notrack_nocompress', dict(is_tracking=False, is_compress_high_freq=False), 512, 300),
                               ('nfft256', dict(is_tracking=True, is_compress_high_freq=True), 256, 150)):
        tmp = tempfile.mkdtemp()
        ref_shims.H5_STORE.clear()
        dev = {'fold1_a': synth_clip(121, 9600)}
        cfg_path, feat_dir = make_tree(tmp, 'foa', dev, {}, n_fft=nfft, hop=hop)
        with np.errstate(all='ignore'):
            ref_salsa.extract_features(data_config=cfg_path, cond_num=5, n_hopframes=3, task='feature', **kw)
        arrays = collect(feat_dir)
        meta = {'format': 'foa', 'fmax_doa': 9000, 'fmin_doa': 50, 'fs': 24000, 'n_fft': nfft, 'hop': hop,
                'cond_num': 5, 'kw': kw, 'clips': {'dev|fold1_a': [121, 9600, sha256_of(dev['fold1_a'])]}}
        save('g3_salsa_foa_%s' % tag, meta, **arrays)
        shutil.rmtree(tmp)
    # one longer clip (6 s) for tracker drift: spatial channels only, to keep the fixture small
    tmp = tempfile.mkdtemp()
    ref_shims.H5_STORE.clear()
    dev = {'long': synth_clip(131, 144000)}
    cfg_path, feat_dir = make_tree(tmp, 'foa', dev, {})
    with np.errstate(all='ignore'):
        ref_salsa.extract_features(data_config=cfg_path, cond_num=5, n_hopframes=3, task='feature')
    arrays = collect(feat_dir)
    (k, v), = arrays.items()
    meta = {'format': 'foa', 'fmax_doa': 9000, 'fmin_doa': 50, 'fs': 24000, 'n_fft': 512, 'hop': 300, 'cond_num': 5,
            'clips': {'dev|long': [131, 144000, sha256_of(dev['long'])]}, 'key': k,
            'logspec_sum': float(v[:4].astype(np.float64).sum())}
    save('g3_salsa_foa_long', meta, spatial=v[4:], logspec_stride8=v[:4, ::8])
    shutil.rmtree(tmp)


def g4_lite():
    for ftype, clips in (('salsa_lite', {'fold1_room1_mix001': (141, 24000), 'fold2_room1_mix009': (142, 19200)}),
                         ('salsa_ipd', {'fold1_room1_mix001': (143, 14400)})):
        tmp = tempfile.mkdtemp()
        ref_shims.H5_STORE.clear()
        dev = {k: synth_clip(s, n) for k, (s, n) in clips.items()}
        ev = {'mix001': synth_clip(149, 9000)}
        cfg_path, feat_dir = make_tree(tmp, 'mic', dev, ev, fmax=2000)
        with np.errstate(all='ignore'):
            ref_lite.extract_features(data_config=cfg_path, feature_type=ftype, task='feature_scaler')
        arrays = collect(feat_dir)
        meta = {'format': 'mic', 'fmax_doa': 2000, 'fmin_doa': 50, 'fs': 24000, 'n_fft': 512, 'hop': 300,
                'feature_type': ftype,
                'clips': {('dev|' + k): [s, n, sha256_of(dev[k])] for k, (s, n) in clips.items()}}
        meta['clips']['eval|mix001'] = [149, 9000, sha256_of(ev['mix001'])]
        save('g4_%s' % ftype, meta, **arrays)
        shutil.rmtree(tmp)


def g8_stft():
    """STFT boundary itself (third-party arithmetic, restated in ref_shims._stft): cross-checked against torch.stft
    and an explicit per-frame rfft so the restatement is pinned by two independent implementations."""
    import torch
    y = synth_clip(151, 6000)[0]
    S = ref_shims._stft(y, n_fft=512, hop_length=300, win_length=512)
    St = torch.stft(torch.from_numpy(y).double(), 512, 300, 512, torch.hann_window(512, periodic=True,
                    dtype=torch.float64), center=True, pad_mode='reflect', return_complex=True).numpy()
    err = np.abs(S - St).max() / np.abs(St).max()
    assert err < 1e-6, err
    save('g8_stft', {'seed': 151, 'n': 6000, 'sha': sha256_of(y), 'torch_rel_err': float(err)}, stft=S)


# ----------------------------------------------------------------------------------------------- G10: contrib on-the-fly
def g10_flexible(fixture='g10_flexible', n=27000, cases=None):
    """contrib/salsa_flexible.py: SalsaFeatures / SalsaLiteFeatures called as their docstrings show.
    Spectrogram channels are stored as float32 (their working dtype), spatial channels as float64.
    fixture g13_flexible_multi: the same for 5 - 8 microphones (shorter clips: 15 planes per case)."""
    import importlib.util
    if not hasattr(np, 'bool'):
        np.bool = bool        # removed in numpy >= 1.24; used at contrib/salsa_flexible.py:335
    spec = importlib.util.spec_from_file_location('ref_salsa_flexible',
                                                  os.path.join(ref_shims.REF_ROOT, 'contrib', 'salsa_flexible.py'))
    flex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(flex)
    cases = cases or [  # name, kind, seed, n_ch, ctor kwargs, call kwargs;  n = 27000: 1.125 s -> 91 frames
        ('salsa_default', 'salsa', 171, 4, {}, dict(clip_freqs=True, clip_spatial_alias=False)),
        ('salsa_alias', 'salsa', 172, 4, {}, dict(clip_freqs=True, clip_spatial_alias=True)),
        ('salsa_full_axis', 'salsa', 173, 4, {}, dict(clip_freqs=False, clip_spatial_alias=True)),
        ('salsa_notrack', 'salsa', 174, 4, {}, dict(clip_freqs=True, clip_spatial_alias=False, is_tracking=False)),
        ('salsa_kwargs', 'salsa', 175, 4, dict(fmin_doa=100, fmax_doa=4000, fmax_spec=8000),
         dict(clip_freqs=True, clip_spatial_alias=True, ew_thresh=3.0, covmat_avg_neighbours=2, floor_mask_ratio=2.0)),
        ('salsa_3mics', 'salsa', 176, 3, {}, dict(clip_freqs=True, clip_spatial_alias=False)),
        ('salsa_2mics', 'salsa', 177, 2, {}, dict(clip_freqs=True, clip_spatial_alias=True)),
        ('lite_default', 'lite', 178, 4, {}, dict(clip_freqs=True, clip_spatial_alias=False)),
        ('lite_alias_full_axis', 'lite', 179, 4, {}, dict(clip_freqs=False, clip_spatial_alias=True)),
        ('lite_2mics', 'lite', 180, 2, {}, dict(clip_freqs=True, clip_spatial_alias=True)),
    ]
    arrays, meta = {}, {'n': n, 'fs': 24000, 'stft_winsize': 512, 'hop_length': 300, 'cases': {}}
    for name, kind, seed, n_ch, ckw, kw in cases:
        y = synth_clip(seed, n, n_ch=n_ch)
        ctor = dict(fs=24000, stft_winsize=512, hop_length=300, fmin_doa=50, fmax_doa=2000, fmax_spec=9000)
        ctor.update(ckw)
        obj = (flex.SalsaFeatures if kind == 'salsa' else flex.SalsaLiteFeatures)(**ctor)
        out = obj(y, **kw)
        assert out.shape[0] == 2 * n_ch - 1 and out.dtype == np.float64
        arrays[name + '_spec'] = out[:n_ch].astype(np.float32)
        assert np.array_equal(arrays[name + '_spec'].astype(np.float64), out[:n_ch])   # spectrograms ARE float32 values
        arrays[name + '_spatial'] = out[n_ch:]
        meta['cases'][name] = {'kind': kind, 'seed': seed, 'n_ch': n_ch, 'sha': sha256_of(y), 'ctor': ctor, 'call': kw,
                               'nonzero': float((out[n_ch:] != 0).mean())}
    save(fixture, meta, **arrays)
    for k, v in meta['cases'].items():
        print('   ', k, 'spatial non-zero fraction %.3f' % v['nonzero'])


G13_CASES = [
    # more than 4 microphones ("arbitrary channels", stacked_covmat_eigh :52-77): N x N eigenproblems
        ('salsa_6mics', 'salsa', 181, 6, {}, dict(clip_freqs=True, clip_spatial_alias=False)),
        ('salsa_8mics_alias', 'salsa', 182, 8, {}, dict(clip_freqs=True, clip_spatial_alias=True)),
        ('salsa_5mics_kwargs', 'salsa', 183, 5, dict(fmin_doa=100, fmax_doa=4000, fmax_spec=8000),
         dict(clip_freqs=True, clip_spatial_alias=True, ew_thresh=3.0, covmat_avg_neighbours=2, floor_mask_ratio=2.0)),
        ('salsa_7mics_notrack', 'salsa', 184, 7, {}, dict(clip_freqs=True, clip_spatial_alias=False, is_tracking=False)),
        ('lite_8mics', 'lite', 185, 8, {}, dict(clip_freqs=True, clip_spatial_alias=False)),
        ('lite_5mics_full_axis', 'lite', 186, 5, {}, dict(clip_freqs=False, clip_spatial_alias=True)),
]


def g13_flexible_multi():
    g10_flexible('g13_flexible_multi', 13500, G13_CASES)              # 0.5625 s -> 46 frames


G15_CASES = [
    # 9 - 16 microphones: the run-time-sized instantiations of the MI355X kernels (include/salsa_hip.h SALSA_MAX_MICS)
        ('salsa_9mics', 'salsa', 191, 9, {}, dict(clip_freqs=True, clip_spatial_alias=False)),
        ('salsa_12mics_alias_notrack', 'salsa', 192, 12, {}, dict(clip_freqs=True, clip_spatial_alias=True, is_tracking=False)),
        ('salsa_16mics_kwargs', 'salsa', 193, 16, dict(fmin_doa=100, fmax_doa=4000, fmax_spec=8000),
         dict(clip_freqs=True, clip_spatial_alias=True, ew_thresh=3.0, covmat_avg_neighbours=2, floor_mask_ratio=2.0)),
        ('lite_11mics', 'lite', 194, 11, {}, dict(clip_freqs=True, clip_spatial_alias=False)),
]


def g15_flexible_many():
    g10_flexible('g15_flexible_many', 9000, G15_CASES)                # 0.375 s -> 31 frames


# ----------------------------------------------------------------------------------------------- G11: train augmentation
def g11_augment():
    """The reference's training augmentation for SALSA features (dataset/datamodule.py:45-52 FOA, :73-82 MIC) applied the
    way SeldDataset.__getitem__ does (dataloader.py:56-60), under np.random.seed(s): SHA-256 of the outputs for 48 seeds
    per format + three full outputs per format."""
    import hashlib
    from utilities import transforms as T
    rng = np.random.RandomState(7)
    x = rng.randn(7, 48, 64).astype(np.float32)
    y_sed = (rng.rand(6, 12) < 0.2).astype(np.float32)
    y_doa = rng.randn(6, 36).astype(np.float32)
    arrays = {'x': x, 'y_sed': y_sed, 'y_doa': y_doa}
    meta = {'image_aspect_ratio': 48 / 200, 'n_zero_channels': 3, 'seeds': list(range(1000, 1048)), 'sha': {}}
    for fmt in ('foa', 'mic'):
        if fmt == 'foa':
            joint = T.ComposeMapTransform([T.TfmapRandomSwapChannelFoa(n_classes=12)])
            plain = T.ComposeTransformNp([T.RandomShiftUpDownNp(freq_shift_range=10)])
        else:
            joint = T.ComposeMapTransform([T.TfmapRandomSwapChannelMic(n_classes=12)])
            plain = T.ComposeTransformNp([T.RandomShiftUpDownNp(freq_shift_range=10),
                                          T.CompositeCutout(image_aspect_ratio=48 / 200, n_zero_channels=3)])
        hs = []
        for s in meta['seeds']:
            np.random.seed(s)
            xo, so, do = joint(x, y_sed, y_doa)
            xo = plain(xo)
            assert xo.dtype == np.float32 and do.dtype == np.float32 and so is y_sed
            hs.append([hashlib.sha256(np.ascontiguousarray(xo).tobytes()).hexdigest(),
                       hashlib.sha256(np.ascontiguousarray(do).tobytes()).hexdigest()])
            if s < 1003:
                arrays['%s_x_%d' % (fmt, s)] = xo
                arrays['%s_doa_%d' % (fmt, s)] = do
        meta['sha'][fmt] = hs
        print('    %s: %d distinct outputs of %d' % (fmt, len({h[0] for h in hs}), len(hs)))
    save('g11_augment', meta, **arrays)


# ----------------------------------------------------------------------------------------------- G12: SELD metrics
def g12_metrics():
    """metrics/SELD2021_evaluation_metrics.py + metrics/dcase_utils.py on synthetic DCASE-format rows (CSV round trip
    included): per-file cumulative (ER, F, LE, LR) and the raw counters."""
    import types
    sys.modules.setdefault('IPython', types.SimpleNamespace(embed=lambda *a, **k: None))   # imported, never called
    from metrics import SELD2021_evaluation_metrics as M, dcase_utils as U
    rng = np.random.RandomState(12)
    tmp = tempfile.mkdtemp()
    ev = M.SELDMetrics(nb_classes=12, doa_threshold=20)
    arrays, cum = {}, []
    for fi in range(4):
        gt, pred = [], []
        for _ in range(14):                                        # events: class, onset, duration, direction, drift
            c, t0, dur = rng.randint(12), rng.randint(0, 560), rng.randint(5, 90)
            azi, ele, tr = rng.randint(-180, 180), rng.randint(-45, 46), rng.randint(3)
            fate = rng.rand()                                      # detected / missed / detected late with a big error
            for t in range(t0, min(600, t0 + dur)):
                gt.append([t, c, tr, azi, ele])
                if fate < 0.6 or (fate > 0.8 and t > t0 + dur // 2):
                    err = 8 if fate < 0.6 else 60
                    a = int(azi + rng.randint(-err, err + 1))
                    a = (a + 180) % 360 - 180
                    pred.append([t, c, 0, a, int(np.clip(ele + rng.randint(-err, err + 1), -90, 90))])
        for _ in range(3):                                         # false alarms
            c, t0, dur = rng.randint(12), rng.randint(0, 560), rng.randint(5, 40)
            for t in range(t0, min(600, t0 + dur)):
                pred.append([t, c, 0, int(rng.randint(-180, 180)), int(rng.randint(-45, 46))])
        gt.sort(key=lambda r: r[0])
        pred.sort(key=lambda r: r[0])
        for name, rows in (('gt', gt), ('pred', pred)):
            np.savetxt(os.path.join(tmp, '%s%d.csv' % (name, fi)), np.array(rows, dtype=np.int64), fmt='%d', delimiter=',')
            arrays['%s%d' % (name, fi)] = np.array(rows, dtype=np.int64)
        g = U.segment_labels(U.load_output_format_file(os.path.join(tmp, 'gt%d.csv' % fi)), _max_frames=600, _nb_label_frames_1s=10)
        p = U.segment_labels(U.load_output_format_file(os.path.join(tmp, 'pred%d.csv' % fi)), _max_frames=600, _nb_label_frames_1s=10)
        ev.update_seld_scores(p, g)
        cum.append([float(v) for v in ev.compute_seld_scores()] +
                   [float(v) for v in (ev._TP, ev._FP, ev._FN, ev._S, ev._D, ev._I, ev._Nref, ev._DE_TP, ev._DE_FP, ev._DE_FN, ev._total_DE)])
    empty = M.SELDMetrics(nb_classes=12, doa_threshold=20)
    save('g12_metrics', {'n_files': 4, 'columns': 'ER F LE LR TP FP FN S D I Nref DE_TP DE_FP DE_FN total_DE',
                         'no_data_scores': [float(v) for v in empty.compute_seld_scores()]},
         cumulative=np.array(cum), **arrays)
    print('    final ER %.4f F %.4f LE %.3f LR %.4f' % tuple(cum[-1][:4]))
    shutil.rmtree(tmp)


def g17_labels():
    """Database.load_classwise_gt (dataset/database.py:253-296) on synthetic DCASE2021 metadata CSVs: the label half of
    load_chunk_data.  Cases pin the write order -- tracks from the shortest to the longest, rows in file order -- with the same
    class active on two tracks in the same frame, a repeated (frame, class) inside one track, a gap in the track numbering,
    negative and +-180 / +-90 degree angles, unsorted rows, a full 600-frame clip and a short one.  The CSV text is stored with
    the expected arrays (fixture = data)."""
    import types
    from dataset.database import Database  # (the reference; h5py is the shim, nothing is read from disk but the CSVs)
    tmp = tempfile.mkdtemp()
    rng = np.random.RandomState(17)
    cases = {}
    # (a) two tracks of the SAME class overlapping in frames 10..19: track 0 is longer (30 rows) than track 1 (12 rows)
    rows = [(f, 3, 0, 10 * (f % 7) - 30, 5) for f in range(0, 30)] + [(f, 3, 1, -170 + f, -20) for f in range(10, 22)]
    cases['same_class_two_tracks'] = (rows, 4800)
    # (b) the shorter track is listed LAST in the file and has the higher id; a third id (2) never appears, id 3 does
    rows = [(f, 5, 3, 45, 10) for f in range(100, 140)] + [(f, 5, 0, -90, -45) for f in range(120, 130)] + \
           [(f, 7, 0, 180, 90) for f in range(0, 5)] + [(f, 7, 3, -180, -90) for f in range(3, 4)]
    cases['order_and_gaps'] = (rows, 4800)
    # (c) a repeated (frame, class) inside one track: the later row stays
    rows = [(50, 2, 0, 10, 0), (50, 2, 0, 20, 0), (50, 2, 0, 30, 15), (51, 2, 0, 30, 15), (51, 11, 1, -1, -1)]
    cases['repeat_in_track'] = (rows, 640)
    # (d) random polyphony, rows shuffled, 60-s clip
    rows = []
    for trk in range(4):
        t0 = int(rng.randint(0, 400))
        cls = int(rng.randint(0, 12))
        for f in range(t0, t0 + int(rng.randint(20, 200))):
            rows.append((f, cls, trk, int(rng.randint(-180, 181)), int(rng.randint(-90, 91))))
    rng.shuffle(rows)
    cases['random_polyphony'] = ([tuple(int(v) for v in r) for r in rows], 4800)
    # (e) a short clip (chunk-sized, 8 s)
    cases['short_clip'] = ([(f, f % 12, f % 2, 3 * f - 100, f - 40) for f in range(0, 80, 3)], 640)
    ns = types.SimpleNamespace(label_upsample_ratio=8, n_classes=12, output_format='reg_xyz', sort_tracks=Database.sort_tracks)
    arrays, meta = {}, {'what': 'Database.load_classwise_gt', 'cases': {}, 'label_upsample_ratio': 8, 'n_classes': 12}
    for name, (rows, n_frames) in cases.items():
        fn = os.path.join(tmp, name + '.csv')
        text = ''.join('%d,%d,%d,%d,%d\n' % r for r in rows)
        open(fn, 'w').write(text)
        sed, doa = Database.load_classwise_gt(ns, fn, n_frames)
        meta['cases'][name] = {'n_frames': n_frames, 'csv': text}
        arrays[name + '_sed'], arrays[name + '_doa'] = sed, doa
    save('g17_labels', meta, **arrays)
    shutil.rmtree(tmp)


def g18_feature_tree_rng(seed, T, F):
    """the synthetic feature file of fixture g18 (a function of its seed: the test regenerates it)"""
    rng = np.random.RandomState(seed)
    f = np.empty((7, T, F), np.float32)
    f[:4] = (-40 + 12 * rng.standard_normal((4, T, F))).astype(np.float32)        # dB-like spectrogram channels
    f[4:] = (rng.uniform(-1, 1, (3, T, F)) * (rng.uniform(size=(3, T, F)) < 0.25)).astype(np.float32)
    return f


def g18_feature_tree():
    """Database.get_split('train', stage='fit') of the reference (dataset/database.py:120-231: load_feature_scaler :87-96, load_chunk_data
    :175-240 -- read 'feature', normalise the first 4 channels, trim to 4800 frames, labels from the metadata CSV, segment indices -- and
    the concatenation) on a synthetic tree of PRECOMPUTED feature files: BASELINE config 3's data path.  Three clips: two of 4801 frames
    (a 60-s clip: trimmed to 4800) and one of 1280 frames; F = 6 keeps the fixture small.  The feature files are functions of their seeds
    (g18_feature_tree_rng, restated by the test); stored: the split's chunk indices, targets, file names, and the normalised features (every 16th frame + the SHA-256 of all of them)."""
    from dataset.database import Database  # the reference (h5py = the in-memory shim of ref_shims)
    import h5py
    tmp = tempfile.mkdtemp()
    F, clips = 6, [('fold1_room1_mix001', 181, 4801), ('fold1_room1_mix002', 182, 1280), ('fold2_room1_mix003', 183, 4801)]
    root = os.path.join(tmp, 'feat')
    os.makedirs(os.path.join(root, 'foa_dev'))
    os.makedirs(os.path.join(tmp, 'gt', 'metadata_dev'))
    os.makedirs(os.path.join(tmp, 'meta', 'original'))
    rng = np.random.RandomState(18)
    mean = (-40 + rng.standard_normal((4, 1, F))).astype(np.float32)
    std = (10 + rng.uniform(size=(4, 1, F))).astype(np.float32)
    with h5py.File(os.path.join(root, 'foa_feature_scaler.h5'), 'w') as hf:
        hf.create_dataset('mean', data=mean, dtype=np.float32)
        hf.create_dataset('std', data=std, dtype=np.float32)
    csv_text = {}
    for name, seed, T in clips:
        with h5py.File(os.path.join(root, 'foa_dev', name + '.h5'), 'w') as hf:
            hf.create_dataset('feature', data=g18_feature_tree_rng(seed, T, F), dtype=np.float32)
        r = np.random.RandomState(seed)
        rows = []
        for trk in range(3):
            t0, cls = int(r.randint(0, T // 8 - 60)), int(r.randint(0, 12))
            rows += [(f, cls, trk, int(r.randint(-180, 181)), int(r.randint(-90, 91))) for f in range(t0, t0 + int(r.randint(10, 60)))]
        csv_text[name] = ''.join('%d,%d,%d,%d,%d\n' % x for x in rows)
        open(os.path.join(tmp, 'gt', 'metadata_dev', name + '.csv'), 'w').write(csv_text[name])
    open(os.path.join(tmp, 'meta', 'original', 'train.csv'), 'w').write('filename\n' + ''.join(n + '\n' for n, _, _ in clips))
    db = Database(feature_root_dir=root, gt_meta_root_dir=os.path.join(tmp, 'gt'), audio_format='foa', n_classes=12, fs=24000, n_fft=512,
                  hop_len=300, label_rate=10, train_chunk_len_s=8.0, train_chunk_hop_len_s=0.5)
    d = db.get_split('train', split_meta_dir=os.path.join(tmp, 'meta', 'original'), stage='fit')
    meta = {'what': "Database.get_split('train', stage='fit')", 'clips': [[n, s, t] for n, s, t in clips], 'F': F, 'csv': csv_text,
            'filename_list': list(d['filename_list']), 'test_batch_size': int(d['test_batch_size']),
            'feature_chunk_len': int(d['feature_chunk_len']), 'gt_chunk_len': int(d['gt_chunk_len'])}
    feats = d['features'].astype(np.float32)
    meta['features_shape'] = list(feats.shape)
    meta['features_sha256'] = sha256_of(feats)
    save('g18_feature_tree', meta, mean=mean, std=std, features_every_16th_frame=feats[:, ::16].copy(), sed_targets=d['sed_targets'].astype(np.float32),
         doa_targets=d['doa_targets'].astype(np.float32), feature_chunk_idxes=np.asarray(d['feature_chunk_idxes'], np.int64),
         gt_chunk_idxes=np.asarray(d['gt_chunk_idxes'], np.int64))
    print('    %d chunks, features %s, last file has %d chunks' % (len(d['feature_chunk_idxes']), d['features'].shape, d['test_batch_size']))
    shutil.rmtree(tmp)


if __name__ == '__main__':
    g5_w_and_bins()
    g1_eigvec()
    g2_adversarial()
    g3_end_to_end()
    g4_lite()
    g8_stft()
    g10_flexible()
    g13_flexible_multi()
    g15_flexible_many()
    g11_augment()
    g12_metrics()
    g17_labels()
    g18_feature_tree()
