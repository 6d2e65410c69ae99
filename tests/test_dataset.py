"""Data-layer mirror: segment index arithmetic against an independent statement of database.py:98-119 (CPU) and the
on-device feature bank against the oracle + numpy normalisation (GPU)."""
import os

import numpy as np
import pytest


def _ref_segments(n_frames, chunk_len, hop, ds, pointer):
    # direct restatement of dataset/database.py:98-119
    n_crop = n_frames // ds
    idx = list(np.arange(pointer, pointer + n_crop - chunk_len // ds + 1, hop // ds))
    if (n_crop - chunk_len // ds) % (hop // ds) != 0:
        idx.append(pointer + n_crop - chunk_len // ds)
    return [int(i) for i in idx], pointer + n_crop


def test_segment_indices_match_reference_arithmetic():
    from salsa_amd.dataset import get_segment_idxes, second2frame
    assert second2frame(8.0, 24000, 300) == 640 and second2frame(0.5, 24000, 300) == 40
    for n_frames in (4800, 4000, 648, 640):
        for ds in (1, 8):
            p = 0
            for _ in range(3):
                got, p2 = get_segment_idxes(n_frames, 640, 40, ds, p)
                ref, p3 = _ref_segments(n_frames, 640, 40, ds, p)
                assert got == ref and p2 == p3
                p = p2
    idx, _ = get_segment_idxes(4800, 640, 40, 1, 0)
    assert len(idx) == 105                                                  # 105 chunks per 60-s clip


def test_load_classwise_gt_reproduces_the_reference(tmp_path):
    """salsa_amd.dataset.load_classwise_gt against golden g17 (Database.load_classwise_gt of the imported reference on synthetic
    DCASE metadata CSVs): bit-equal sed and xyz targets, including which track wins where two tracks hold a class in one frame."""
    from conftest import load_golden
    from salsa_amd.dataset import load_classwise_gt, sort_tracks
    meta, a = load_golden('g17_labels')
    assert set(meta['cases']) >= {'same_class_two_tracks', 'order_and_gaps', 'repeat_in_track', 'random_polyphony', 'short_clip'}
    for name, c in meta['cases'].items():
        fn = tmp_path / (name + '.csv')
        fn.write_text(c['csv'])
        sed, doa = load_classwise_gt(str(fn), c['n_frames'], meta['n_classes'], meta['label_upsample_ratio'])
        assert sed.dtype == np.float32 and doa.dtype == np.float32
        assert np.array_equal(sed, a[name + '_sed']), name
        assert np.array_equal(doa, a[name + '_doa']), name
    # the overlap of case (a): frames 10..21 hold class 3 on both tracks; the longer track (0) is written last and stays
    doa = a['same_class_two_tracks_doa']
    f = 15
    azi, ele = np.float32((10 * (f % 7) - 30) * np.pi / 180.0), np.float32(5 * np.pi / 180.0)
    assert doa[f, 3] == np.cos(azi) * np.cos(ele) and doa[20, 3] != 0 and doa[21, 3] != 0
    assert list(sort_tracks(np.array([0, 0, 0, 3, 3]))) == [1, 2, 3, 0]
    with pytest.raises(AssertionError):
        load_classwise_gt(str(fn), 641)
    with pytest.raises(ValueError):
        load_classwise_gt(str(fn), 640, output_format='polar')


@pytest.mark.gpu
def test_feature_bank_takes_labels_from_dcase_metadata_csvs(tmp_path):
    """GpuFeatureBank.add_clips(gt_meta=...) -- config 4 trained from a DCASE metadata tree: the chunks' targets are the golden
    g17 arrays cut with the reference's segment arithmetic (database.py:209-219)."""
    import torch
    from conftest import load_golden
    from salsa_amd.dataset import GpuFeatureBank
    from salsa_amd.extractor import SalsaExtractor
    from salsa_amd.synth import synth_clip
    meta, a = load_golden('g17_labels')
    names = ['repeat_in_track', 'short_clip']
    fns = []
    for n in names:
        fn = tmp_path / (n + '.csv')
        fn.write_text(meta['cases'][n]['csv'])
        fns.append(str(fn))
    ys = np.stack([synth_clip(90 + i, 640 * 300) for i in range(2)])           # 641 frames -> 640
    bank = GpuFeatureBank(SalsaExtractor(audio_format='mic', fmax_doa=4000), max_clip_s=60)
    bank.add_clips(ys, names, gt_meta=fns)
    bank.fit_scaler()
    bank.finalize()
    assert len(bank) == 2
    for i, n in enumerate(names):
        x, sed, doa, name = bank[i]
        assert name == n and x.shape == (7, 640, 200)
        assert torch.equal(sed.cpu(), torch.from_numpy(a[n + '_sed'])) and torch.equal(doa.cpu(), torch.from_numpy(a[n + '_doa']))
    with pytest.raises(AssertionError):
        bank.add_clips(ys, names, sed=[a['short_clip_sed']] * 2, gt_meta=fns)


@pytest.mark.gpu
def test_feature_bank_matches_oracle_and_numpy_normalisation(oracle):
    import torch
    from salsa_amd.dataset import GpuFeatureBank
    from salsa_amd.extractor import SalsaExtractor
    from salsa_amd.synth import synth_clip
    ys = np.stack([synth_clip(80 + i, 10 * 24000) for i in range(3)])
    bank = GpuFeatureBank(SalsaExtractor(), max_clip_s=60)
    bank.add_clips(ys[:2], ['a', 'b'])
    bank.add_clips(ys[2:], ['c'])
    mean, std = bank.fit_scaler()
    bank.finalize()
    refs = [oracle.extract_salsa(y)[:, :800] for y in ys]                  # 801 frames -> 800 (multiple of 8)
    rmean, rstd = oracle.compute_scaler(refs)
    np.testing.assert_allclose(mean.cpu().numpy(), rmean, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(std.cpu().numpy(), rstd, rtol=1e-5, atol=1e-4)
    cat = np.concatenate(refs, axis=1)
    cat[:4] = (cat[:4] - rmean) / rstd
    assert len(bank) == 3 * len(range(0, 800 - 640 + 1, 40))
    x, sed, doa, name = bank[len(bank) - 1]
    assert x.shape == (7, 640, 200) and sed.shape == (80, 12) and doa.shape == (80, 36) and name == 'c'
    s = bank.chunk_idx[-1]
    np.testing.assert_allclose(x[:4].cpu().numpy(), cat[:4, s:s + 640], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(x[4:].cpu().numpy(), cat[4:, s:s + 640], rtol=1e-5, atol=1e-6)
    xb, _, _, names = bank.batch([0, 5, len(bank) - 1])
    assert xb.shape == (3, 7, 640, 200) and names[0] == 'a'


@pytest.mark.gpu
def test_feature_bank_from_a_precomputed_feature_tree(tmp_path):
    """BASELINE config 3's data path: extract_features() writes the tree (feature files + scaler file), a bank built from those FILES
    (Database.load_chunk_data :190-219 + load_feature_scaler :87-96) yields the same chunks as the reference's numpy arithmetic on the
    files -- (x - mean) / std on the first 4 channels, trim to a multiple of 8 frames, segment indices, labels from the CSVs -- and the
    same as a bank fed the audio with that scaler."""
    import torch
    import yaml
    from scipy.io import wavfile
    from conftest import load_golden
    from salsa_amd import io as sio
    from salsa_amd.dataset import GpuFeatureBank, get_segment_idxes
    from salsa_amd.extractor import SalsaExtractor
    from salsa_amd.features import extract_features
    from salsa_amd.synth import synth_clip
    meta, a = load_golden('g17_labels')
    data_dir, feat_dir = str(tmp_path / 'data'), str(tmp_path / 'feat')
    os.makedirs(os.path.join(data_dir, 'foa_dev'))
    os.makedirs(os.path.join(data_dir, 'foa_eval'))
    names = ['repeat_in_track', 'short_clip', 'third']
    lens = [640 * 300, 640 * 300, 700 * 300 + 17]
    ys = [synth_clip(70 + i, n) for i, n in enumerate(lens)]
    for n, y in zip(names, ys):
        wavfile.write(os.path.join(data_dir, 'foa_dev', n + '.wav'), 24000, y.T)
    csvs = []
    for n in names:
        fn = tmp_path / (n + '.csv')
        fn.write_text(meta['cases'][n if n != 'third' else 'short_clip']['csv'])
        csvs.append(str(fn))
    cfg = {'data_dir': data_dir, 'feature_dir': feat_dir,
           'data': {'format': 'foa', 'fs': 24000, 'n_fft': 512, 'win_len': 512, 'hop_len': 300, 'fmin_doa': 50, 'fmax_doa': 9000}}
    with open(tmp_path / 'cfg.yml', 'w') as f:
        yaml.safe_dump(cfg, f)
    extract_features(data_config=str(tmp_path / 'cfg.yml'))                   # task='feature_scaler': files + scaler
    root = os.path.join(feat_dir, 'salsa', 'foa', '24000fs_512nfft_300nhop_5cond_9000fmaxdoa')
    files = [os.path.join(root, 'foa_dev', n + '.h5') for n in names]
    bank = GpuFeatureBank(max_clip_s=60)                                      # no extractor: files only
    bank.load_feature_scaler(os.path.join(root, 'foa_feature_scaler.h5'))
    bank.add_feature_files(files, gt_meta=csvs)
    bank.finalize()
    with pytest.raises(AssertionError):
        bank.add_clips(np.stack(ys[:2]), names[:2])
    # the reference's arithmetic on the files (database.py:190-231)
    sc = sio.load_arrays(os.path.join(root, 'foa_feature_scaler.h5'))
    pointer, cat, idx, who = 0, [], [], []
    for n, fpath in zip(names, files):
        feat = sio.load_arrays(fpath)['feature'].copy()
        feat[:4] = (feat[:4] - sc['mean']) / sc['std']
        n_frames = min(feat.shape[1], 600 * 8)
        n_frames -= n_frames % 8                                             # (whole label frames: what the 4800-frame trim gives 60-s clips)
        cat.append(feat[:, :n_frames])
        ii, pointer = get_segment_idxes(n_frames, 640, 40, 1, pointer)
        idx += ii
        who += [n] * len(ii)
    cat = np.concatenate(cat, axis=1)
    assert bank.chunk_idx == idx and bank.chunk_name == who and len(bank) == len(idx) >= 4
    for i in (0, 1, len(bank) - 1):
        x, sed, doa, name = bank[i]
        assert name == who[i] and x.shape == (7, 640, 200)
        np.testing.assert_array_equal(x[4:].cpu().numpy(), cat[4:, idx[i]:idx[i] + 640])
        np.testing.assert_allclose(x[:4].cpu().numpy(), cat[:4, idx[i]:idx[i] + 640], rtol=1e-6, atol=1e-6)   # float32 (x - mean) / std on the device
    x0, sed0, doa0, _ = bank[0]
    assert torch.equal(sed0.cpu(), torch.from_numpy(a['repeat_in_track_sed'])) and torch.equal(doa0.cpu(), torch.from_numpy(a['repeat_in_track_doa']))
    # ... and the same chunks as a bank fed the AUDIO with that scaler
    bank2 = GpuFeatureBank(SalsaExtractor(), max_clip_s=60)
    bank2.add_clips(np.stack(ys[:2]), names[:2], gt_meta=csvs[:2])
    bank2.add_clips(ys[2][None], names[2:], gt_meta=csvs[2:])
    bank2.load_feature_scaler(os.path.join(root, 'foa_feature_scaler.h5'))
    bank2.finalize()
    assert bank2.chunk_idx == bank.chunk_idx
    for i in (0, len(bank) - 1):
        assert torch.equal(bank2[i][0], bank[i][0]) and torch.equal(bank2[i][1], bank[i][1])


def _g18_feature(seed, T, F):
    # tools/make_golden.py::g18_feature_tree_rng restated: the fixture's feature files are functions of their seeds
    rng = np.random.RandomState(seed)
    f = np.empty((7, T, F), np.float32)
    f[:4] = (-40 + 12 * rng.standard_normal((4, T, F))).astype(np.float32)
    f[4:] = (rng.uniform(-1, 1, (3, T, F)) * (rng.uniform(size=(3, T, F)) < 0.25)).astype(np.float32)
    return f


@pytest.mark.gpu
def test_feature_bank_from_files_reproduces_the_references_get_split(tmp_path):
    """golden g18 = the REFERENCE's Database.get_split('train', stage='fit') on a synthetic tree of precomputed feature files (database.py:
    120-231).  The bank built from the same files (real HDF5 through salsa_amd.io) must hold the same split: chunk start indices of
    features and labels, file name per chunk, targets bit for bit, normalised features to float32 round-off (and, restated in numpy, to
    the reference's SHA-256), 60-s clips trimmed to 4800 frames."""
    import hashlib
    import torch
    from conftest import load_golden
    from salsa_amd import io as sio
    from salsa_amd.dataset import GpuFeatureBank
    meta, a = load_golden('g18_feature_tree')
    F = meta['F']
    files, csvs, cat = [], [], []
    for name, seed, T in meta['clips']:
        f = _g18_feature(seed, T, F)
        files.append(sio.save_arrays(str(tmp_path / (name + '.h5')), feature=f))
        (tmp_path / (name + '.csv')).write_text(meta['csv'][name])
        csvs.append(str(tmp_path / (name + '.csv')))
        g = f.copy()
        g[:4] = (g[:4] - a['mean']) / a['std']                                 # database.py:197-202
        cat.append(g[:, :min(T, 4800)])
    cat = np.concatenate(cat, axis=1)
    assert list(cat.shape) == meta['features_shape'] and hashlib.sha256(cat.tobytes()).hexdigest() == meta['features_sha256']
    scaler_file = sio.save_arrays(str(tmp_path / 'foa_feature_scaler.h5'), mean=a['mean'], std=a['std'])
    bank = GpuFeatureBank(max_clip_s=60)
    bank.load_feature_scaler(scaler_file)
    bank.add_feature_files(files, names=[c[0] for c in meta['clips']], gt_meta=csvs)
    bank.finalize()
    assert bank.chunk_idx == a['feature_chunk_idxes'].tolist() and bank.gt_idx == a['gt_chunk_idxes'].tolist()
    assert bank.chunk_name == meta['filename_list'] and bank.chunk_len == meta['feature_chunk_len']
    assert bank.chunk_name.count(meta['clips'][-1][0]) == meta['test_batch_size']
    assert torch.equal(bank.sed_all.cpu(), torch.from_numpy(a['sed_targets'])) and torch.equal(bank.doa_all.cpu(), torch.from_numpy(a['doa_targets']))
    feats = bank.features.cpu().numpy()
    assert list(feats.shape) == meta['features_shape']
    np.testing.assert_array_equal(feats[4:, ::16], a['features_every_16th_frame'][4:])
    np.testing.assert_allclose(feats[:4, ::16], a['features_every_16th_frame'][:4], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(feats[:4], cat[:4], rtol=1e-6, atol=1e-6)
    x, sed, doa, name = bank[len(bank) - 1]
    s, g = int(a['feature_chunk_idxes'][-1]), int(a['gt_chunk_idxes'][-1])
    assert name == meta['filename_list'][-1] and x.shape == (7, 640, F) and sed.shape == (meta['gt_chunk_len'], 12)
    assert torch.equal(sed.cpu(), torch.from_numpy(a['sed_targets'][g:g + 80])) and np.array_equal(x[4:].cpu().numpy(), cat[4:, s:s + 640])
