"""Shared comparison rule for the contrib on-the-fly surface (golden g10, contrib/salsa_flexible.py)."""
import numpy as np


def compare_flexible(out, spec_ref, spat_ref, case, n_samples, spec_tol=(2e-6, 2e-5), spat_tol=1e-9, spat_rtol=0.0,
                     gate_exact=True):
    """out: (2C-1, F, T).  Spectrograms to float32 accuracy; spatial channels: zero pattern exact (gates), values to
    spat_tol -- except where the spectra are REAL up to round-off (frame 0 and, when the last frame is centred on the
    last sample, that frame: mirror-symmetric about the reflect point; bins 0 and n_fft/2), whose 0-or-+-pi phase has
    a sign decided by that round-off: there the comparison is modulo one phase turn."""
    n_ch = case['n_ch']
    ctor, call = case['ctor'], case['call']
    nb = ctor['stft_winsize'] // 2 + 1
    lo = max(1, int(np.floor(ctor['fmin_doa'] * ctor['stft_winsize'] / float(ctor['fs'])))) if call['clip_freqs'] else 0
    F, T = spat_ref.shape[1:]
    assert out.shape == (2 * n_ch - 1, F, T), (out.shape, spat_ref.shape)
    np.testing.assert_allclose(out[:n_ch], spec_ref, rtol=spec_tol[0], atol=spec_tol[1])
    k = np.arange(lo, lo + F)
    delta = np.float32(2 * np.pi * ctor['fs'] / (ctor['stft_winsize'] * 343.0))
    nf = (np.where(k == 0, 1, k).astype(np.float32) * delta).astype(np.float64)
    real_tf = np.zeros((F, T), bool)
    real_tf[:, 0] = True
    if (n_samples - 1) % ctor['hop_length'] == 0 or n_samples % ctor['hop_length'] == 0:
        real_tf[:, -1] = True
    real_tf[(k == 0) | (k == nb - 1), :] = True
    o, r = out[n_ch:].astype(np.float64), spat_ref.astype(np.float64)
    if gate_exact:
        generic = ~np.broadcast_to(real_tf, r.shape)
        assert np.array_equal((o != 0) & generic, (r != 0) & generic), 'gate pattern differs'
    d = o - r
    period = (2 * np.pi / nf)[None, :, None]
    dw = d - period * np.round(d / period)
    err = np.abs(np.where(real_tf[None], dw, d))
    assert np.all(err <= spat_tol + spat_rtol * np.abs(r)), err.max()
