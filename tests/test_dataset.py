"""Data-layer mirror: segment index arithmetic against an independent statement of database.py:98-119 (CPU) and the
on-device feature bank against the oracle + numpy normalisation (GPU)."""
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
