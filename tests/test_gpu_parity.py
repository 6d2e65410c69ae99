"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI (libsalsa_hip.so via
salsa_amd.extractor), against (i) the golden vectors produced by the reference itself and (ii) the CPU oracle on
seeded inputs.  Tolerances (BASELINE.json north_star): bit-exact frame/bin indexing and gates, 1e-5 relative on
floats.  A gate disagreement is admissible only inside float64 round-off of the threshold (|margin| < 1e-9)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_clip, load_golden
from salsa_amd.synth import sha256_of, synth_clip, synth_stft_block

pytestmark = pytest.mark.gpu

RTOL, ATOL_DB, ATOL_SP = 1e-5, 2e-5, 1e-6


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _extractor(**kw):
    from salsa_amd.extractor import SalsaExtractor
    return SalsaExtractor(**kw)


def _gpu_features(y, dev, **kw):
    ex = _extractor(**kw)
    a = torch.from_numpy(np.ascontiguousarray(y[None])).to(dev)
    return ex.extract(a)[0].cpu().numpy()


def _check(out, ref, margin=None, n_spec=4):
    """out/ref (7,T,F).  margin: oracle gate margins (nd,T) to justify gate flips, or None (no flips allowed)."""
    assert out.shape == ref.shape and out.dtype == np.float32
    np.testing.assert_allclose(out[:n_spec], ref[:n_spec], rtol=RTOL, atol=ATOL_DB)
    with np.errstate(invalid='ignore'):
        bad = np.abs(out[n_spec:] - ref[n_spec:]) > ATOL_SP + RTOL * np.abs(ref[n_spec:])
    bad = bad.any(axis=0)                              # (T, F)
    if bad.any():
        assert margin is not None, 'spatial channels differ in %d TF bins' % bad.sum()
        nd = margin.shape[0]
        assert not bad[:, nd:].any()
        m = np.abs(margin.T[bad[:, :nd]])
        assert np.all(m < 1e-9), 'gate/values differ outside round-off of the threshold: %s' % m[:8]
        assert bad.sum() <= max(2, int(2e-5 * bad.size))


def _check_lite(out, ref, ftype='salsa_lite', lower=1):
    """SALSA-Lite / IPD (7,T,F).  Mirror-symmetric frames (frame 0; the last frame when (N-1) % hop == 0) have a
    REAL spectrum up to 1e-14 round-off, so an inter-channel phase of +-pi has a sign decided by that round-off in the
    reference too: phases are compared modulo one turn, and at most a handful of elements may use the wrap."""
    assert out.shape == ref.shape and out.dtype == np.float32
    np.testing.assert_allclose(out[:4], ref[:4], rtol=RTOL, atol=ATOL_DB)
    F = out.shape[2]
    k = np.arange(lower, lower + F, dtype=np.float64)
    period = 2.0 * np.ones(F) if ftype == 'salsa_ipd' else 2 * np.pi / (2 * np.pi * 24000 / (512 * 343.0) * k)
    d = out[4:].astype(np.float64) - ref[4:]
    wraps = np.round(d / period)
    d = d - period * wraps
    assert np.all(np.abs(d) <= ATOL_SP + RTOL * np.abs(ref[4:]))
    wrapped_frames = np.unique(np.nonzero(wraps)[1])
    assert len(wrapped_frames) <= 2, 'phase wraps outside the mirror-symmetric frames: %s' % wrapped_frames


# ----------------------------------------------------------------------------------------------- eigenvector stage
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_eigvec_matches_reference_golden(dev, seed):
    meta, a = load_golden('g1_eigvec_s%d' % seed)
    X = synth_stft_block(seed, meta['n_bins'], meta['n_frames'], kind=meta['kind'])
    assert sha256_of(X) == meta['sha']
    Xd = torch.from_numpy(X[None]).to(dev)
    for fmt in ('foa', 'mic'):
        for track in (True, False):
            ex = _extractor(audio_format=fmt, is_tracking=track, fmax_doa=9000 if fmt == 'foa' else 4000)
            out, gate = ex.eigvec(Xd, lower_bin=1, return_gate=True)
            ref = a['%s_%s' % (fmt, 'track' if track else 'notrack')]
            np.testing.assert_allclose(out[0].cpu().numpy(), ref, rtol=1e-8, atol=1e-9)
            if track:
                g = gate[0].cpu().numpy()
                assert np.array_equal(g > 0, a['sig_mask'])         # tracker indicator bit-exact
    ex = _extractor(audio_format='foa', cond_num=2.0)
    np.testing.assert_allclose(ex.eigvec(Xd, 1)[0].cpu().numpy(), a['foa_track_cond2'], rtol=1e-8, atol=1e-9)
    ex = _extractor(audio_format='mic', fmax_doa=4000)
    np.testing.assert_allclose(ex.eigvec(Xd, 7)[0].cpu().numpy(), a['mic_track_lb7'], rtol=1e-8, atol=1e-9)


def test_eigvec_adversarial_golden(dev, oracle):
    meta, a = load_golden('g2_adversarial')
    for case in meta['cases']:
        X = a['X_' + case]
        Xd = torch.from_numpy(np.ascontiguousarray(X[None])).to(dev)
        for fmt in ('foa', 'mic'):
            for track in (True, False):
                ref = a['%s_%s_%s' % (case, fmt, 'track' if track else 'notrack')]
                ex = _extractor(audio_format=fmt, is_tracking=track, fmax_doa=9000 if fmt == 'foa' else 4000)
                out = ex.eigvec(Xd, 1)[0].cpu().numpy()
                fin = np.isfinite(ref)
                assert np.array_equal(np.isfinite(out), fin), (case, fmt, track)
                tol = 1e-5 if case == 'w_tiny' else 1e-8
                with np.errstate(invalid='ignore'):
                    bad = ((np.abs(out - ref) > tol * (0.1 + np.abs(ref))) & fin).any(axis=0)
                if case == 'margin' and track:
                    _, aux = oracle.extract_normalized_eigenvector(X, 5.0, 3, True, fmt, fs=24000, n_fft=512,
                                                                   lower_bin=1, return_aux=True)
                    assert np.all(np.abs(aux['margin'][bad]) < 1e-9)
                else:
                    assert not bad.any(), (case, fmt, track, int(bad.sum()))
        ex = _extractor(audio_format='foa', cond_num=0.0)
        _, gate = ex.eigvec(Xd, 1, return_gate=True)
        assert np.array_equal(gate[0].cpu().numpy() > 0, a[case + '_sig_mask']), case


# ------------------------------------------------------------------ the PRODUCTION solver (packed float32 + float64 cold list)
def _feat_vs_ref(out_f32, ref_f64, margin=None, tol_scale=1.0):
    """out (3, T, nb) float32 from the feature kernel vs ref (3, nb, T) float64 of the reference: the feature bar
    1e-6 + 1e-5 |ref|; where a margin array is given, a disagreement is admitted only inside round-off of the gate."""
    out = np.transpose(out_f32, (0, 2, 1)).astype(np.float64)
    fin = np.isfinite(ref_f64)
    assert np.array_equal(np.isfinite(out), fin)
    with np.errstate(invalid='ignore'):
        bad = ((np.abs(out - ref_f64) > tol_scale * (ATOL_SP + RTOL * np.abs(ref_f64))) & fin).any(axis=0)
    if bad.any():
        assert margin is not None, '%d TF bins outside the bar' % bad.sum()
        assert np.all(np.abs(margin[bad]) < 1e-9), np.abs(margin[bad])[:8]
    return int(bad.sum())


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_production_solver_matches_reference_golden(dev, oracle, seed):
    """Round-4 review, missing 1 / weak 1: goldens g1 reached the GPU only through salsa_eigvec_batch, which is hard-wired to the
    float64 instantiation.  salsa_eigvec_feature_batch runs the instantiation salsa_extract_batch launches -- the packed-float32
    pair solve with its float64 cold list -- on the same blocks; FLAG_FORCE_F64 runs the float64 one through the same entry."""
    from salsa_amd import _lib
    meta, a = load_golden('g1_eigvec_s%d' % seed)
    X = synth_stft_block(seed, meta['n_bins'], meta['n_frames'], kind=meta['kind'])
    Xd = torch.from_numpy(X[None]).to(dev)
    cold = 0
    for fmt, key, kw, lb in (('foa', 'foa_track', {}, 1), ('mic', 'mic_track', {}, 1), ('foa', 'foa_track_cond2', dict(cond_num=2.0), 1),
                             ('mic', 'mic_track_lb7', {}, 7)):
        cond = kw.get('cond_num', 5.0)
        _, aux = oracle.extract_normalized_eigenvector(X, cond, 3, True, fmt, fs=24000, n_fft=512, lower_bin=lb, return_aux=True)
        outs = []
        for flags in (0, _lib.FLAG_FORCE_F64):
            ex = _extractor(audio_format=fmt, fmax_doa=9000 if fmt == 'foa' else 4000, flags=flags, **kw)
            ex.set_stats(True)
            out = ex.eigvec_features(Xd, lb)[0].cpu().numpy()
            st = ex.read_stats()
            _feat_vs_ref(out, a[key], aux['margin'])
            outs.append(out)
            if flags == 0:
                cold += st['cold_frames']
                assert st['gated_frames'] == int(a['sig_mask'].sum()), (st, int(a['sig_mask'].sum()))   # the tracker's gate, bit for bit
        # the two instantiations against each other: same gate pattern, features inside the bar
        assert np.array_equal((outs[0] != 0).any(axis=0), (outs[1] != 0).any(axis=0))
        np.testing.assert_allclose(outs[0], outs[1], rtol=RTOL, atol=ATOL_SP)
    print('g1 seed %d: %d frames handed to the float64 cold list' % (seed, cold))


def test_production_solver_adversarial_golden(dev, oracle):
    """Goldens g2 (exact rank-1 windows, eigenvalue ratios 4.999 / 5.001 around cond_num, u[0] ~ 0, silence, constant magnitude,
    level steps) through the packed-float32 instantiation and through the float64 one."""
    from salsa_amd import _lib
    meta, a = load_golden('g2_adversarial')
    report = {}
    for case in meta['cases']:
        X = a['X_' + case]
        Xd = torch.from_numpy(np.ascontiguousarray(X[None])).to(dev)
        for fmt in ('foa', 'mic'):
            ref = a['%s_%s_track' % (case, fmt)]
            _, aux = oracle.extract_normalized_eigenvector(X, 5.0, 3, True, fmt, fs=24000, n_fft=512, lower_bin=1, return_aux=True)
            for flags in (0, _lib.FLAG_FORCE_F64):
                ex = _extractor(audio_format=fmt, fmax_doa=9000 if fmt == 'foa' else 4000, flags=flags)
                ex.set_stats(True)
                out = ex.eigvec_features(Xd, 1)[0].cpu().numpy()
                st = ex.read_stats()
                # u[0] ~ 0: Re(u/u[0]) is ill-conditioned in the reference itself; the float64 test above admits 1e-5 (0.1 + |ref|) there
                nbad = _feat_vs_ref(out, ref, aux['margin'] if case == 'margin' else None, tol_scale=10.0 if case == 'w_tiny' else 1.0)
                assert st['gated_frames'] == int(a[case + '_sig_mask'].sum())
                if flags == 0:
                    report[(case, fmt)] = (st['gated_frames'], st['cold_frames'], nbad)
    print('g2 through the packed solve: (gated frames, float64 cold-list frames, margin-admitted) per case: %s' % report)


def test_fused_schedule_is_bit_identical(dev):
    """Round 5, stage (a) of the spill removal: STFT -> tracker -> ONE fused kernel (STFT of a segment of frames into an LDS ring,
    packed solve from the ring, float64 records solved at the end of the segment) does the three-kernel path's arithmetic in
    the same order, so its features are bit-identical -- FOA / MIC, ragged lengths, interleaved audio, a fused scaler, and with
    the record slice switched off (mode 2: the in-step fallback)."""
    cases = [('foa 3 x 2 s', np.stack([synth_clip(7 + i, 48000) for i in range(3)]), {}),
             ('foa ragged', np.stack([synth_clip(17 + i, 14700) for i in range(2)]), {}),
             ('mic 4 x 8 s', np.stack([synth_clip(60 + i, 8 * 24000) for i in range(4)]), dict(audio_format='mic', fmax_doa=4000)),
             ('interleaved', np.ascontiguousarray(np.stack([synth_clip(27 + i, 72000) for i in range(2)]).transpose(0, 2, 1)),
              dict(audio_layout='interleaved')),
             ('one 20-s clip', synth_clip(33, 20 * 24000)[None], {})]
    for name, ys, kw in cases:
        a = torch.from_numpy(ys).to(dev)
        ex = _extractor(**kw)
        ref = ex.extract(a).clone()
        assert bool((ref[:, 4:] != 0).any()), name
        for mode in (1, 2):
            ex.set_fused(mode)
            ex.set_stats(True)
            out = ex.extract(a).clone()
            st = ex.read_stats()
            ex.set_stats(False)
            assert torch.equal(out, ref), '%s, mode %d: %d elements differ' % (name, mode, int((out != ref).sum()))
            assert st['tiles'] > 0 and st['gated_frames'] > 0, (name, st)      # (the fused kernel ran: its counters moved)
        ex.set_fused(0)
    # with the scaler fused into the stores (config 4's extractor)
    ys = np.stack([synth_clip(90 + i, 8 * 24000) for i in range(3)])
    a = torch.from_numpy(ys).to(dev)
    ex = _extractor(audio_format='mic', fmax_doa=4000)
    rng = np.random.default_rng(5)
    ex.set_scaler(rng.normal(-60, 5, (4, 200)).astype(np.float32), rng.uniform(5, 15, (4, 200)).astype(np.float32))
    ref = ex.extract(a).clone()
    ex.set_fused(1)
    assert torch.equal(ex.extract(a), ref)


# ----------------------------------------------------------------------------------------------- end-to-end goldens
def _golden_items(meta, a):
    clips = {k: golden_clip(*v) for k, v in meta['clips'].items()}
    for key, ref in a.items():
        parts = key.split('|')
        if parts[-1] != 'feature':
            continue
        split, name = parts[3], parts[4][:-3]
        yield key, clips[('dev|' if split.endswith('_dev') else 'eval|') + name], ref


@pytest.mark.parametrize('fmt', ['foa', 'mic'])
def test_full_salsa_matches_reference_golden(dev, oracle, fmt):
    meta, a = load_golden('g3_salsa_%s' % fmt)
    for key, y, ref in _golden_items(meta, a):
        out = _gpu_features(y, dev, audio_format=fmt, fmax_doa=meta['fmax_doa'])
        _, aux = oracle.extract_salsa(y, fmax_doa=meta['fmax_doa'], audio_format=fmt, return_aux=True)
        _check(out, ref, aux['margin'])


def test_full_salsa_variants_match_reference_golden(dev):
    meta, a = load_golden('g3_salsa_foa_notrack_nocompress')
    for key, y, ref in _golden_items(meta, a):
        _check(_gpu_features(y, dev, is_tracking=False, is_compress_high_freq=False), ref)
    meta, a = load_golden('g3_salsa_foa_nfft256')
    for key, y, ref in _golden_items(meta, a):
        _check(_gpu_features(y, dev, n_fft=256, hop_len=150), ref)
    meta, a = load_golden('g3_salsa_foa_long')
    y = golden_clip(*meta['clips']['dev|long'])
    out = _gpu_features(y, dev)
    np.testing.assert_allclose(out[4:], a['spatial'], rtol=RTOL, atol=ATOL_SP)
    np.testing.assert_allclose(out[:4, ::8], a['logspec_stride8'], rtol=RTOL, atol=ATOL_DB)


@pytest.mark.parametrize('ftype', ['salsa_lite', 'salsa_ipd'])
def test_salsa_lite_matches_reference_golden(dev, ftype):
    meta, a = load_golden('g4_%s' % ftype)
    for key, y, ref in _golden_items(meta, a):
        out = _gpu_features(y, dev, audio_format='mic', feature_type=ftype, fmax_doa=2000)
        _check_lite(out, ref, ftype)
        assert not out[4:, :, 42:].any()


# ----------------------------------------------------------------------------------------------- vs oracle, seeded
@pytest.mark.parametrize('fmt,fmax,layout', [('foa', 9000, 'planar'), ('mic', 4000, 'interleaved')])
def test_batched_clips_against_oracle(dev, oracle, fmt, fmax, layout):
    n = 10 * 24000
    ys = np.stack([synth_clip(500 + i, n) for i in range(3)])
    ex = _extractor(audio_format=fmt, fmax_doa=fmax, audio_layout=layout)
    a = torch.from_numpy(ys).to(dev)
    if layout == 'interleaved':
        a = a.permute(0, 2, 1).contiguous()
    out = ex.extract(a).cpu().numpy()
    assert out.shape == (3, 7, 801, 200)
    for i in range(3):
        ref, aux = oracle.extract_salsa(ys[i], fmax_doa=fmax, audio_format=fmt, return_aux=True)
        _check(out[i], ref, aux['margin'])
    # batch invariance: a clip's features do not depend on its batch neighbours (bit-identical)
    solo = ex.extract(a[1:2].contiguous()).cpu().numpy()
    assert np.array_equal(solo[0], out[1])


@pytest.mark.parametrize('ftype,fmax', [('salsa_lite', 2000), ('salsa_ipd', 2000), ('salsa_lite', 4000), ('salsa_lite', 9000)])
def test_lite_std_instantiation_equals_the_general_kernel(dev, oracle, ftype, fmax):
    """Round 6: planar 4-channel audio takes the Lite STD instantiation of the STFT kernel (channel 0's spectrum through LDS, per-register
    band tests, 3 waves per SIMD); the same clips INTERLEAVED take the general kernel.  Bit-equal outputs -- incl. ragged lengths (reflect
    path), a silent stretch, spectra tiny enough for the scaled phase product -- and one clip against the oracle."""
    ys = np.stack([synth_clip(900 + i, 24000 * 5 + 137) for i in range(6)])
    ys[5, :, :3000] = 0.0
    ys[4] *= 1e-18
    a = torch.from_numpy(ys).to(dev)
    std = _extractor(audio_format='mic', feature_type=ftype, fmax_doa=fmax).extract(a)
    gen = _extractor(audio_format='mic', feature_type=ftype, fmax_doa=fmax, audio_layout='interleaved').extract(a.permute(0, 2, 1).contiguous())
    assert std.shape == gen.shape == (6, 7, 401, 191)
    assert torch.equal(std, gen), 'differing elements: %d' % int((std != gen).sum())
    _check_lite(std[1].cpu().numpy(), oracle.extract_lite(ys[1], fmax_doa=fmax, feature_type=ftype), ftype=ftype)
    for n in (300, 813, 3001):                                    # tiny clips: every frame on the reflect path
        y = torch.from_numpy(synth_clip(40 + n, n)[None]).to(dev)
        s1 = _extractor(audio_format='mic', feature_type=ftype, fmax_doa=fmax).extract(y)
        g1 = _extractor(audio_format='mic', feature_type=ftype, fmax_doa=fmax, audio_layout='interleaved').extract(y.permute(0, 2, 1).contiguous())
        assert torch.equal(s1, g1), n


def test_ragged_and_tiny_clips_against_oracle(dev, oracle):
    for n in (300, 511, 512, 813, 2999, 3000, 3001):       # around hop / n_fft boundaries; heavy reflect padding
        y = synth_clip(900 + n, n)
        ref, aux = oracle.extract_salsa(y, return_aux=True)
        _check(_gpu_features(y, dev), ref, aux['margin'])
        ref = oracle.extract_lite(y, fmax_doa=2000)
        out = _gpu_features(y, dev, audio_format='mic', feature_type='salsa_lite', fmax_doa=2000)
        _check_lite(out, ref)


def test_other_hop_frames_and_cond(dev, oracle):
    """n_hopframes != 3 takes the generic (non-unrolled) covariance path; cond_num 0 disables the coherence gate."""
    y = synth_clip(321, 3 * 24000)
    for nh, cond in ((2, 5.0), (1, 3.0), (3, 0.0), (4, 8.0)):
        ref, aux = oracle.extract_salsa(y, n_hopframes=nh, cond_num=cond, return_aux=True)
        _check(_gpu_features(y, dev, n_hopframes=nh, cond_num=cond), ref, aux['margin'])


def test_silent_and_constant_clips(dev, oracle):
    z = np.zeros((4, 6000), np.float32)
    out = _gpu_features(z, dev)
    assert np.abs(out[:4] + 100.0).max() < 2e-5 and not out[4:].any()    # 10*log10(1e-10); nothing passes the gates
    c = np.ones((4, 6000), np.float32) * 0.25
    ref, aux = oracle.extract_salsa(c, return_aux=True)
    _check(_gpu_features(c, dev), ref, aux['margin'])


def test_logspec_entry_point(dev, oracle):
    y = synth_clip(77, 24000)
    ex = _extractor()
    out = ex.logspec(torch.from_numpy(y[None]).to(dev))[0].cpu().numpy()
    np.testing.assert_allclose(out, oracle.logspec(y), rtol=RTOL, atol=ATOL_DB)
    ex = _extractor(is_compress_high_freq=False)
    out = ex.logspec(torch.from_numpy(y[None]).to(dev))[0].cpu().numpy()
    np.testing.assert_allclose(out, oracle.logspec(y, compress=False), rtol=RTOL, atol=ATOL_DB)


# ----------------------------------------------------------------------------------------------- full size
def test_full_size_60s_batch_properties(dev, oracle):
    """BASELINE config 2 shape (60-s clips), reduced batch: one clip checked against the oracle in full, the rest by
    size-independent properties (determinism, batch invariance, zero band, finite, spatial unit norm)."""
    B, n = 4, 60 * 24000
    ys = np.stack([synth_clip(2021 + i, n) for i in range(B)])
    ex = _extractor()
    a = torch.from_numpy(ys).to(dev)
    out1 = ex.extract(a)
    out2 = ex.extract(a)
    assert torch.equal(out1, out2)                                        # deterministic
    o = out1.cpu().numpy()
    assert o.shape == (B, 7, 4801, 200) and np.isfinite(o).all()
    assert not o[:, 4:, :, 191:].any()                                    # zero above upper_bin (:373-374)
    nrm = np.sqrt((o[:, 4:].astype(np.float64) ** 2).sum(axis=1))
    emitted = nrm > 0
    assert 0.02 < emitted[:, :, :191].mean() < 0.9
    assert np.abs(nrm[emitted] - 1.0).max() < 1e-6                        # FOA eigenvector features are unit vectors
    ref, aux = oracle.extract_salsa(ys[0], return_aux=True)
    _check(o[0], ref, aux['margin'])
    perm = torch.flip(a, dims=[0]).contiguous()
    assert torch.equal(torch.flip(ex.extract(perm), dims=[0]), out1)      # batch-order invariance


def test_errors_mirror_reference(dev):
    with pytest.raises(AssertionError):
        _extractor(n_fft=1024)
    with pytest.raises(ValueError):
        _extractor(audio_format='xyz')
    with pytest.raises(AssertionError):
        _extractor(feature_type='salsa_lite', audio_format='foa')          # lite :72
    with pytest.raises(AssertionError):
        _extractor(feature_type='salsa_lite', audio_format='mic', fmax_doa=9500)   # lite :59


def test_clip_group_pipeline_is_bit_identical(dev):
    """The multi-stream clip-group pipeline only reorders independent clips: outputs must equal the single-stream run."""
    ys = np.stack([synth_clip(700 + i, 4 * 24000) for i in range(5)])
    a = torch.from_numpy(ys).to(dev)
    ex = _extractor()
    ex.set_groups(1)
    ref = ex.extract(a).clone()
    for g in (2, 3, 8):
        ex.set_groups(g)
        for _ in range(3):
            assert torch.equal(ex.extract(a), ref)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                    # called from a non-default stream
        o2 = ex.extract(a)
    side.synchronize()
    assert torch.equal(o2, ref)


# ----------------------------------------------------------------------------------------------- reference surface
def _make_tree(tmp, fmt, clips, fmax, n_fft=512, hop=300):
    import yaml
    from scipy.io import wavfile
    data_dir, feat_dir = os.path.join(tmp, 'data'), os.path.join(tmp, 'feat')
    for key, y in clips.items():
        split, name = key.split('|')
        d = os.path.join(data_dir, '%s_%s' % (fmt, split))
        os.makedirs(d, exist_ok=True)
        wavfile.write(os.path.join(d, name + '.wav'), 24000, y.T)          # float32 WAV: samples survive exactly
    os.makedirs(os.path.join(data_dir, fmt + '_eval'), exist_ok=True)
    cfg = {'data_dir': data_dir, 'feature_dir': feat_dir,
           'data': {'format': fmt, 'fs': 24000, 'n_fft': n_fft, 'win_len': n_fft, 'hop_len': hop, 'fmin_doa': 50,
                    'fmax_doa': fmax}}
    path = os.path.join(tmp, 'cfg.yml')
    with open(path, 'w') as f:
        yaml.safe_dump(cfg, f)
    return path, feat_dir


@pytest.mark.parametrize('fmt', ['foa', 'mic'])
def test_extract_features_harness_reproduces_reference_tree(dev, oracle, tmp_path, fmt):
    """salsa_amd.features.extract_features against the reference's own run (fixture g3): directory naming, one
    feature file per clip, (7,T,200) float32 values, and the scaler over all dev files."""
    from salsa_amd import io as sio
    from salsa_amd.features import extract_features
    meta, a = load_golden('g3_salsa_%s' % fmt)
    clips = {k: golden_clip(*v) for k, v in meta['clips'].items()}
    cfg, feat_dir = _make_tree(str(tmp_path), fmt, clips, meta['fmax_doa'])
    extract_features(data_config=cfg, batch_size=2)
    n_checked = 0
    for key, ref in a.items():
        parts = key.split('|')
        got = sio.load_arrays(os.path.join(feat_dir, *parts[:-1]))[parts[-1]]
        if parts[-1] == 'feature':
            split, name = parts[3], parts[4][:-3]
            y = clips[('dev|' if split.endswith('_dev') else 'eval|') + name]
            _, aux = oracle.extract_salsa(y, fmax_doa=meta['fmax_doa'], audio_format=fmt, return_aux=True)
            _check(got, ref, aux['margin'])
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
        n_checked += 1
    assert n_checked == len(a)
    # round 6: task='feature_scaler' takes the scaler's statistics from the features while they are on the device; the file-based
    # compute_scaler (what task='scaler' alone runs, and what the reference does) must give the same file
    from salsa_amd.features import compute_scaler
    desc = [d for d in os.listdir(os.path.join(feat_dir, 'salsa', fmt))][0]
    root = os.path.join(feat_dir, 'salsa', fmt, desc)
    fused = sio.load_arrays(os.path.join(root, fmt + '_feature_scaler.h5'))
    compute_scaler(feature_dir=root, audio_format=fmt)
    files = sio.load_arrays(os.path.join(root, fmt + '_feature_scaler.h5'))
    for k in ('mean', 'std'):
        assert fused[k].shape == files[k].shape == (4, 1, 200) and fused[k].dtype == np.float32
        np.testing.assert_allclose(fused[k], files[k], rtol=2e-6, atol=2e-6)
    # the tree as the REFERENCE reads it: h5py (the image's conda interpreter has 3.3.0), hf['feature'][:] / hf['mean'][:] (database.py:
    # 87-96, :193-195) on the files the harness wrote through libhdf5
    from test_host_logic_cpu import CONDA_PY, _conda_h5py
    if sio.HAVE_HDF5 and not sio.HAVE_H5PY and _conda_h5py():
        import hashlib
        import json
        import subprocess
        dev_dir = os.path.join(root, fmt + '_dev')
        names = sio.feature_files(dev_dir)
        assert names and all(n.endswith('.h5') for n in names)
        code = ("import h5py, sys, json, hashlib\n"
                "out = {}\n"
                "for fn in sys.argv[1:]:\n"
                "    with h5py.File(fn, 'r') as hf:\n"
                "        out[fn] = {k: [list(hf[k].shape), str(hf[k].dtype), hashlib.sha256(hf[k][:].tobytes()).hexdigest()] for k in hf.keys()}\n"
                "print(json.dumps(out))\n")
        paths = [os.path.join(dev_dir, names[0]), os.path.join(root, fmt + '_feature_scaler.h5')]
        r = subprocess.run([CONDA_PY, '-c', code] + paths, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1000:]
        seen = json.loads(r.stdout.strip().splitlines()[-1])
        for pth in paths:
            mine = sio.load_arrays(pth)
            assert set(seen[pth]) == set(mine)
            for k, v in mine.items():
                assert seen[pth][k] == [list(v.shape), 'float32', hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()], (pth, k)


@pytest.mark.timeout(120)
def test_file_pipeline_raises_instead_of_hanging(dev, tmp_path):
    """Round-4 advice: a failure in any stage of the extraction harness's file pipeline -- the device issue in the caller's thread,
    the writer (e.g. a full disk), the reader -- must stop the other stages and come out of run() as the exception, on a tree long
    enough that the reader is waiting for a free slot when it happens."""
    from salsa_amd import features, io as sio
    from salsa_amd.extractor import SalsaExtractor
    adir, fdir = tmp_path / 'audio', tmp_path / 'feat'
    adir.mkdir(); fdir.mkdir()
    for i in range(12):
        np.save(adir / ('clip%02d.npy' % i), synth_clip(300 + i, 24000))
    todo = list(enumerate(sorted(os.listdir(adir))))

    class Boom(RuntimeError):
        pass

    # (i) extract() raises in the caller's thread on the third batch
    ex = SalsaExtractor(device=dev)
    pipe = features._FilePipeline(ex)
    calls = [0]
    real = pipe.slots[0]['ex'].extract
    def bad_extract(*a, **k):
        calls[0] += 1
        if calls[0] == 3:
            raise Boom('device issue failed')
        return real(*a, **k)
    for sl in pipe.slots:
        sl['ex'].extract = bad_extract
    with pytest.raises(Boom):
        pipe.run(todo, str(adir), str(fdir), 24000, 2)
    # (ii) the writer fails
    pipe = features._FilePipeline(SalsaExtractor(device=dev))
    real_save = sio.save_arrays
    def bad_save(path, **kw):
        if path.endswith('clip03.h5') or 'clip03' in os.path.basename(path):
            raise Boom('no space left on device')
        return real_save(path, **kw)
    sio.save_arrays = bad_save
    try:
        with pytest.raises(Boom):
            pipe.run(todo, str(adir), str(fdir), 24000, 2)
    finally:
        sio.save_arrays = real_save
    # (iii) the reader fails (a missing file), and the pipeline object is still usable afterwards
    with pytest.raises(Exception):
        pipe.run(todo + [(99, 'missing.npy')], str(adir), str(fdir), 24000, 2)
    pipe.run(todo, str(adir), str(fdir), 24000, 2)
    assert len(os.listdir(fdir)) == 12


def test_lite_harness_and_python_surface(dev, tmp_path):
    from salsa_amd import io as sio
    from salsa_amd import lite_features
    from salsa_amd.features import MagStftExtractor, extract_normalized_eigenvector
    meta, a = load_golden('g4_salsa_lite')
    clips = {k: golden_clip(*v) for k, v in meta['clips'].items()}
    cfg, feat_dir = _make_tree(str(tmp_path), 'mic', clips, 2000)
    lite_features.extract_features(data_config=cfg, feature_type='salsa_lite', batch_size=3)
    for key, ref in a.items():
        parts = key.split('|')
        got = sio.load_arrays(os.path.join(feat_dir, *parts[:-1]))[parts[-1]]
        if parts[-1] == 'feature':
            _check_lite(got, ref)
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    # numpy-in / numpy-out functions with the reference's signatures
    meta, g = load_golden('g1_eigvec_s1')
    X = synth_stft_block(1, meta['n_bins'], meta['n_frames'], kind=meta['kind'])
    out = extract_normalized_eigenvector(X.astype(complex), 5.0, 3, True, 'mic', fs=24000, n_fft=512, lower_bin=1)
    np.testing.assert_allclose(out, g['mic_track'], rtol=1e-8, atol=1e-9)
    y = synth_clip(5, 12000)
    ex = MagStftExtractor(n_fft=512, hop_length=300, win_length=512)
    assert ex.W.shape == (200, 257)
    assert ex.extract(y[:3]).shape == (3, 41, 200)                          # any channel count, like the reference


def test_streamed_bulk_extraction_matches_single_stream(dev):
    from salsa_amd.extractor import StreamedExtractor
    batches = [torch.from_numpy(np.stack([synth_clip(800 + 3 * i + j, 3 * 24000) for j in range(3)])).to(dev)
               for i in range(5)]
    ex = _extractor()
    ref = [ex.extract(b).clone() for b in batches]
    got = [o.clone() for o in StreamedExtractor(n_streams=2).extract_many(batches)]
    assert len(got) == len(ref) and all(torch.equal(a, b) for a, b in zip(got, ref))


def test_extract_is_hip_graph_capturable(dev):
    """include/salsa_hip.h promises that the extract call allocates and synchronises nothing: capture it in a HIP graph
    (torch.cuda.CUDAGraph), replay on fresh input, compare with the eager result."""
    ys = [np.stack([synth_clip(900 + 2 * i + j, 2 * 24000) for j in range(2)]) for i in range(2)]
    ex = _extractor()
    static_in = torch.from_numpy(ys[0]).to(dev)
    static_out = torch.empty((2,) + tuple(ex.output_shape(2 * 24000)), dtype=torch.float32, device=dev)
    ex.extract(static_in, out=static_out)                                  # warm-up: sizes the workspace
    ref0 = static_out.clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            ex.extract(static_in, out=static_out)
    static_out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, ref0)
    static_in.copy_(torch.from_numpy(ys[1]).to(dev))
    g.replay()
    torch.cuda.synchronize()
    eager = _extractor().extract(torch.from_numpy(ys[1]).to(dev))
    assert torch.equal(static_out, eager)


def test_full_size_scaling_invariance(dev):
    """Size-independent property at BASELINE clip length: scaling the audio by an exact power of two scales every
    spectrum, the 3-frame RMS and the noise floor exactly (as long as the 1e-6 floor clamp is not hit), so every gate
    decision and every eigenvector is bit-identical, while the log-spectrogram moves by 20*log10(4) dB."""
    ys = np.stack([synth_clip(2040 + i, 60 * 24000) for i in range(2)])
    ex = _extractor()
    a = torch.from_numpy(ys).to(dev)
    f1 = ex.extract(a).clone()
    f4 = ex.extract(a * 4.0)
    assert torch.equal(f1[:, 4:], f4[:, 4:])
    lo = f1[:, :4].cpu().numpy()
    d = f4[:, :4].cpu().numpy() - lo
    free = lo > -99.0                                    # bins not sitting on the amin = 1e-10 clamp (-100 dB)
    assert free.mean() > 0.99 and np.abs(d[free] - 20 * np.log10(4.0)).max() < 5e-5


def test_channel_permutation_equivariance(dev):
    """Swapping FOA channels Y and Z (inputs 2 and 3) swaps log-spectrogram rows 2/3 and eigenvector components 5/6 and
    changes nothing else (the covariance is permuted, its spectrum and gates are not) -- to float64 round-off."""
    y = synth_clip(2050, 10 * 24000)
    ex = _extractor()
    f = ex.extract(torch.from_numpy(y[None]).to(dev))[0].cpu().numpy()
    g = ex.extract(torch.from_numpy(np.ascontiguousarray(y[[0, 1, 3, 2]])[None]).to(dev))[0].cpu().numpy()
    assert np.array_equal(f[:2], g[:2])                                    # pair 0 untouched: bit-identical
    np.testing.assert_allclose(f[[3, 2]], g[2:4], rtol=0, atol=2e-5)       # pair 1 packs (y3 + i*y2): other rounding path
    np.testing.assert_allclose(f[[4, 6, 5]], g[4:], rtol=1e-5, atol=1e-6)


def test_fused_scaler_equals_separate_normalisation(dev):
    from salsa_amd.extractor import normalize_
    ys = np.stack([synth_clip(950 + i, 3 * 24000) for i in range(2)])
    a = torch.from_numpy(ys).to(dev)
    rng = np.random.RandomState(3)
    mean = (rng.randn(4, 1, 200) * 5 - 50).astype(np.float32)
    std = (rng.rand(4, 1, 200) * 10 + 5).astype(np.float32)
    ex = _extractor()
    raw = ex.extract(a).clone()
    ref = normalize_(raw.clone(), torch.from_numpy(mean), torch.from_numpy(std))
    ex.set_scaler(mean, std)
    fused = ex.extract(a).clone()
    assert torch.equal(fused, ref)                               # same float32 arithmetic, bit-identical
    np.testing.assert_allclose(ref[:, :4].cpu().numpy(), (raw[:, :4].cpu().numpy() - mean) / std, rtol=1e-6, atol=1e-6)
    ex.set_scaler(None)
    assert torch.equal(ex.extract(a), raw)


# ----------------------------------------------------------------------------------------------- contrib on-the-fly surface
@pytest.mark.parametrize('fixture', ['g10_flexible', 'g13_flexible_multi', 'g15_flexible_many'])
def test_contrib_flexible_surface_matches_reference_golden(dev, fixture):
    """SalsaFeatures / SalsaLiteFeatures of contrib/salsa_flexible.py, called like the reference's docstrings:
    2 - 4 microphones (g10, the 4-channel kernels) and 5 - 8 (g13, salsa_extract_multichannel: N x N Jacobi eigen-solver; 9 - 16
    microphones: g15, the run-time-sized instantiations)."""
    from flex_compare import compare_flexible
    from salsa_amd.flexible import SalsaFeatures, SalsaLiteFeatures
    meta, g = load_golden(fixture)
    for name, c in meta['cases'].items():
        y = synth_clip(c['seed'], meta['n'], n_ch=c['n_ch'])
        assert sha256_of(y) == c['sha']
        obj = (SalsaFeatures if c['kind'] == 'salsa' else SalsaLiteFeatures)(**c['ctor'])
        out = obj(y, **c['call'])
        assert out.dtype == np.float64, name
        compare_flexible(out, g[name + '_spec'], g[name + '_spatial'], c, meta['n'], spec_tol=(RTOL, ATOL_DB),
                         spat_tol=ATOL_SP, spat_rtol=RTOL)


def test_contrib_flexible_batch_against_oracle(dev, oracle):
    from flex_compare import compare_flexible
    from salsa_amd.flexible import SalsaFeatures, SalsaLiteFeatures
    n = 3 * 24000 + 123
    ys = np.stack([synth_clip(300 + i, n) for i in range(3)])
    a = torch.from_numpy(ys).to(dev)
    ctor = dict(fs=24000, stft_winsize=512, hop_length=300, fmin_doa=50, fmax_doa=4000, fmax_spec=9000)
    for kind, cls, call in (('salsa', SalsaFeatures, dict(clip_freqs=True, clip_spatial_alias=True, ew_thresh=4.0)),
                            ('salsa', SalsaFeatures, dict(clip_freqs=False, clip_spatial_alias=False, is_tracking=False,
                                                          ew_thresh=1.05)),
                            ('lite', SalsaLiteFeatures, dict(clip_freqs=False, clip_spatial_alias=False))):
        out = cls(**ctor).extract_batch(a, **call).cpu().numpy()
        for i in range(3):
            kw = dict(ctor)
            kw.update(call)
            ref = oracle.flexible(ys[i], kind=kind, **kw)
            case = {'n_ch': 4, 'ctor': ctor, 'call': {'clip_freqs': call['clip_freqs']}}
            compare_flexible(out[i], ref[:4].astype(np.float32), ref[4:], case, n, spec_tol=(RTOL, ATOL_DB),
                             spat_tol=ATOL_SP, spat_rtol=RTOL)
        if kind == 'salsa':
            assert (out[:, 4:] != 0).any(), 'degenerate case: nothing passed the gates'


def test_contrib_flexible_nfft256_and_long_clip_against_oracle(dev, oracle):
    from flex_compare import compare_flexible
    from salsa_amd.flexible import SalsaFeatures
    ctor = dict(fs=24000, stft_winsize=256, hop_length=150, fmin_doa=50, fmax_doa=3000, fmax_spec=9000)
    call = dict(clip_freqs=True, clip_spatial_alias=True)
    n = 2 * 24000 + 31
    y = synth_clip(411, n)
    out = SalsaFeatures(**ctor)(y, **call)
    kw = dict(ctor)
    kw.update(call)
    ref = oracle.flexible(y, kind='salsa', **kw)
    compare_flexible(out, ref[:4].astype(np.float32), ref[4:], {'n_ch': 4, 'ctor': ctor, 'call': call}, n,
                     spec_tol=(RTOL, ATOL_DB), spat_tol=ATOL_SP, spat_rtol=RTOL)


def test_long_single_clip_against_oracle(dev, oracle):
    """200 s in one clip (33x the chunks of the goldens): tracker hand-over across 250 chunks, 32-bit offsets."""
    n = 200 * 24000 + 123
    y = synth_clip(77, n)
    out = _gpu_features(y, dev, audio_format='foa')
    ref, aux = oracle.extract_salsa(y, audio_format='foa', return_aux=True)
    _check(out, ref, margin=aux['margin'])


def test_to_freq_major(dev):
    from salsa_amd.flexible import to_freq_major
    x = torch.randn(3, 7, 131, 77, device=dev)
    assert torch.equal(to_freq_major(x), x.permute(0, 1, 3, 2).double())
    with pytest.raises(ValueError):
        from salsa_amd.flexible import SalsaFeatures
        SalsaFeatures().extract_batch(torch.zeros(1, 17, 4000, device=dev))         # 2 - 16 microphones
    with pytest.raises(AssertionError):
        from salsa_amd.flexible import SalsaFeatures
        SalsaFeatures(fmax_doa=9500)                                            # contrib :183


def test_torch_op_equals_the_extractor(dev):
    import salsa_amd.torch_ops  # noqa: F401
    a = torch.from_numpy(np.stack([synth_clip(500 + i, 30000) for i in range(2)])).to(dev)
    assert torch.equal(torch.ops.salsa.extract(a), _extractor().extract(a))
    assert torch.equal(torch.ops.salsa.extract(a, 'mic', 'salsa_lite', 24000, 512, 300, 50, 2000),
                       _extractor(audio_format='mic', feature_type='salsa_lite', fmax_doa=2000).extract(a))


@pytest.mark.parametrize('fmt', ['foa', 'mic'])
def test_one_pass_augmentation_kernel_equals_the_torch_composite(dev, fmt):
    """salsa_augment_batch against swap -> shift -> cutout done with torch operators (which golden g11 ties to the
    reference's transforms): bit-identical features for the same draws, every swap combination covered."""
    from salsa_amd.augment import apply_augment_hip, apply_augment_torch, augment_batch, draw_augment
    g = torch.Generator().manual_seed(5)
    B, T, F = 48, 80, 200
    x = torch.randn((B, 7, T, F), generator=g).to(dev)
    doa = torch.randn((B, 10, 36), generator=g).to(dev)
    d = draw_augment(B, T, F, fmt, gen=g, p=0.8)
    nb = 4 if fmt == 'foa' else 3
    for i in range(2 ** nb):                                        # all swap-bit combinations on the first samples
        d['m'][i, :nb] = torch.tensor([(i >> k) & 1 for k in range(nb)])
    d['shift'][:4] = torch.tensor([0, 9, 9, 1])
    d['up'][:4] = torch.tensor([True, True, False, False])
    ref, _ = apply_augment_torch(x, doa, d, fmt)
    out = apply_augment_hip(x, d, fmt)
    assert torch.equal(out, ref)
    assert not torch.equal(out, x)
    big = torch.randn((B, 7, T + 1, F), generator=g).to(dev)            # a time-cropped view, like extract(...)[:, :, :640]
    view = big[:, :, :T]
    assert torch.equal(apply_augment_hip(view, d, fmt), apply_augment_torch(view, doa, d, fmt)[0])
    xa, sed, ya = augment_batch(x, doa[:, :, :12], doa, fmt, gen=torch.Generator().manual_seed(9))
    xb, yb = apply_augment_torch(x, doa, draw_augment(B, T, F, fmt, gen=torch.Generator().manual_seed(9)), fmt)
    assert torch.equal(xa, xb) and torch.equal(ya, yb)


def test_augmentation_kernel_reproduces_the_reference_samples_directly(dev):
    """Golden g11 THROUGH the HIP kernel (round-5 review, weak 4: it used to reach salsa_augment_batch in two hops).  For each of
    the 48 seeds x 2 formats the reference's own draws -- np.random.seed(s), consumed in the reference's call order by
    salsa_amd.augment.reference_draws -- become the kernel's parameters; the augmented features must hash to what the reference's
    SeldDataset produced (and equal the stored arrays), the targets likewise through the branch-free swap."""
    import hashlib
    from salsa_amd.augment import apply_augment_hip, reference_draws, swap_targets
    meta, a = load_golden('g11_augment')
    x = torch.from_numpy(a['x'])[None].to(dev).contiguous()
    y_doa = torch.from_numpy(a['y_doa'])[None].to(dev)
    T, F = x.shape[2:]
    for fmt in ('foa', 'mic'):
        changed = cut = 0
        for s, (hx, hd) in zip(meta['seeds'], meta['sha'][fmt]):
            np.random.seed(s)

            def minmax_after(d):                                   # the sample after swap + shift: a first pass of the kernel, no cutout
                y = apply_augment_hip(x, d, fmt)
                return float(y.min()), float(y.max())
            d = reference_draws(np.random, T, F, fmt, minmax_after, image_aspect_ratio=meta['image_aspect_ratio'])
            xo = apply_augment_hip(x, d, fmt)[0].cpu().numpy()
            do = swap_targets(y_doa, d['m'].to(dev), fmt)[0].cpu().numpy()
            if ('%s_x_%d' % (fmt, s)) in a:
                assert np.array_equal(xo, a['%s_x_%d' % (fmt, s)]) and np.array_equal(do, a['%s_doa_%d' % (fmt, s)]), (fmt, s)
            assert hashlib.sha256(np.ascontiguousarray(xo).tobytes()).hexdigest() == hx, (fmt, s)
            assert hashlib.sha256(np.ascontiguousarray(do).tobytes()).hexdigest() == hd, (fmt, s)
            changed += not np.array_equal(xo, a['x'])
            cut += int(d['h'].sum() > 0)
        assert changed > len(meta['seeds']) // 2 and (cut > 8 if fmt == 'mic' else cut == 0), (fmt, changed, cut)


def test_host_pipeline_matches_direct_extraction(dev):
    from salsa_amd.extractor import HostPipeline
    ys = [np.stack([synth_clip(700 + 10 * k + i, 40000) for i in range(3)]) for k in range(7)]
    ex = _extractor()
    want = [ex.extract(torch.from_numpy(y).to(dev)).cpu().numpy() for y in ys]
    pipe = HostPipeline(depth=3)
    got = [o.copy() for o in pipe.run(ys)]
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    got2 = [o.copy() for o in pipe.run(ys[:2])]                                      # fewer batches than slots; reuse of the slots
    assert all(np.array_equal(a, b) for a, b in zip(got2, want[:2]))
    got3 = [o.copy() for o in pipe.run(fill=lambda buf, i: buf.__setitem__(Ellipsis, ys[i]), n_batches=5, shape=ys[0].shape)]
    assert all(np.array_equal(a, b) for a, b in zip(got3, want[:5]))


@pytest.mark.parametrize('n_ch', [5, 6, 8, 9, 12, 16])
def test_contrib_multichannel_batch_against_oracle(dev, oracle, n_ch):
    """salsa_extract_multichannel (5 - 16 microphones: N x N Hermitian eigenproblem by Jacobi; 6 / 8 unrolled instantiations,
    10 - 16 the run-time-sized one) on a batch of longer clips,
    tracking on and off, against the oracle (which golden g13 ties to the reference)."""
    from flex_compare import compare_flexible
    from salsa_amd.flexible import SalsaFeatures, SalsaLiteFeatures
    n = 2 * 24000 + 77
    ys = np.stack([synth_clip(600 + 10 * n_ch + i, n, n_ch=n_ch) for i in range(2)])
    a = torch.from_numpy(ys).to(dev)
    ctor = dict(fs=24000, stft_winsize=512, hop_length=300, fmin_doa=50, fmax_doa=4000, fmax_spec=9000)
    for kind, cls, call in (('salsa', SalsaFeatures, dict(clip_freqs=True, clip_spatial_alias=True, ew_thresh=4.0)),
                            ('salsa', SalsaFeatures, dict(clip_freqs=False, clip_spatial_alias=False, is_tracking=False,
                                                          ew_thresh=1.05)),
                            ('lite', SalsaLiteFeatures, dict(clip_freqs=True, clip_spatial_alias=False))):
        out = cls(**ctor).extract_batch(a, **call).cpu().numpy()
        assert out.shape[1] == 2 * n_ch - 1
        for i in range(2):
            kw = dict(ctor)
            kw.update(call)
            ref = oracle.flexible(ys[i], kind=kind, **kw)
            case = {'n_ch': n_ch, 'ctor': ctor, 'call': {'clip_freqs': call['clip_freqs']}}
            compare_flexible(out[i], ref[:n_ch].astype(np.float32), ref[n_ch:], case, n, spec_tol=(RTOL, ATOL_DB),
                             spat_tol=ATOL_SP, spat_rtol=RTOL)
        if kind == 'salsa' and call.get('is_tracking', True):
            assert (out[:, n_ch:] != 0).any(), 'degenerate case: nothing passed the gates'


def test_timing_modes_leave_results_untouched(dev):
    """The measurement knobs of salsa_plan_set_timing (include/salsa_hip.h): an event pair per launch (1), K launches of every
    kernel between one event pair (K > 1), and the plain issue of a PREFIX of the path (STFT alone, STFT + tracker; Python:
    ``issue_prefix``, the library answers SALSA_PARTIAL) that bench.py uses to attribute the step to its kernels.  Every kernel is idempotent on (audio, spill, masks), so the output of
    any of them -- on buffers a full call left behind -- must be the plain call's bit for bit, and the timing modes must report
    one positive duration per kernel."""
    ys = np.stack([synth_clip(300 + i, 4 * 24000) for i in range(3)])
    a = torch.from_numpy(ys).to(dev)
    ex = _extractor()
    ref = ex.extract(a).clone()
    out = torch.empty_like(ref)
    for mode in (1, 4):
        ex.set_timing(mode)
        ex.extract(a, out=out)
        t = ex.read_timing()
        assert [n for n, _ in t] == ['stft_logspec', 'noise_floor_tracker', 'cov_eig'] and all(ms > 0 for _, ms in t), t
        assert torch.equal(out, ref)
    ex.set_timing(0)
    assert ex.read_timing() == []
    ex.extract(a, out=out)                       # full call: spill and masks in place
    spatial = out[:, 4:].clone()
    out[:, 4:] = 7.0                             # poison the channels the skipped kernel would have written
    out[:, :4] = 0.0
    with pytest.raises(ValueError):
        ex.set_timing(-1)                        # the prefix mode is not reachable through the timing switch any more
    for kernels in (1, 2):
        with ex.issue_prefix(kernels):
            ex.extract(a, out=out)
            torch.cuda.synchronize()
            assert torch.equal(out[:, :4], ref[:, :4])                     # the STFT launch ran and rewrote the spectrograms
            assert bool((out[:, 4:] == 7.0).all())                         # ... and the covariance kernel did not
        # outside the context the plan is back to plain issue, and the library's PARTIAL code is never swallowed there
    from salsa_amd import _lib
    ex.L.salsa_plan_set_timing(ex._plan, -1)     # (raw ABI: what a careless caller could do)
    with pytest.raises(RuntimeError):
        ex.extract(a, out=out)                   # SALSA_PARTIAL (1) is not SALSA_OK: extract() refuses the incomplete result
    assert _lib.PARTIAL == 1
    ex.L.salsa_plan_set_timing(ex._plan, 0)
    ex.extract(a, out=out)
    assert torch.equal(out, ref) and torch.equal(out[:, 4:], spatial)
