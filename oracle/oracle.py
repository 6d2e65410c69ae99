"""ctypes front-end of oracle/libsalsa_oracle.so (CPU restatement of the reference; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product package
(salsa_amd) never does.  Function names mirror the reference functions they restate (file:line in salsa_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libsalsa_oracle.so')
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'salsa_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, '-B', 'libsalsa_oracle.so'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        fp, dp, ip, up = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_ubyte)
        L.salsa_oracle_n_frames.restype = C.c_long
        L.salsa_oracle_n_frames.argtypes = [C.c_long, C.c_int]
        L.salsa_oracle_bin_limits.argtypes = [C.c_int] * 4 + [ip, ip, ip]
        L.salsa_oracle_W.argtypes = [C.c_int, C.c_int, fp]
        L.salsa_oracle_stft.argtypes = [fp, C.c_long, C.c_int, C.c_int, C.c_int, fp]
        L.salsa_oracle_logspec.argtypes = [fp, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.salsa_oracle_eigvec.argtypes = [fp, C.c_int, C.c_long, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, dp, up, up, dp]
        L.salsa_oracle_extract_salsa.argtypes = [fp, C.c_long] + [C.c_int] * 6 + [C.c_double] + [C.c_int] * 4 + \
                                                [fp, up, dp]
        L.salsa_oracle_extract_lite.argtypes = [fp, C.c_long] + [C.c_int] * 7 + [fp]
        L.salsa_oracle_flex_bins.argtypes = [C.c_int] * 5 + [ip, ip, ip]
        L.salsa_oracle_flex.argtypes = [fp, C.c_int, C.c_long] + [C.c_int] * 9 + [C.c_double, C.c_int, C.c_int,
                                                                                   C.c_double, fp, dp]
        L.salsa_oracle_resample.argtypes = [fp, C.c_int, C.c_long, fp, C.c_long, C.c_long, C.c_double, dp, dp, C.c_int, C.c_int]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def set_threads(n: int) -> None:
    lib().salsa_oracle_set_threads(int(n))


def max_threads() -> int:
    return int(lib().salsa_oracle_max_threads())


def n_frames(n_samples: int, hop: int) -> int:
    return int(lib().salsa_oracle_n_frames(n_samples, hop))


def bin_limits(fs, n_fft, fmin_doa, fmax_doa):
    lo, up, cut = C.c_int(), C.c_int(), C.c_int()
    lib().salsa_oracle_bin_limits(fs, n_fft, int(fmin_doa), int(fmax_doa), C.byref(lo), C.byref(up), C.byref(cut))
    return lo.value, up.value, cut.value


def freq_dim(n_fft, compress=True):
    return int(lib().salsa_oracle_freq_dim(n_fft, int(compress)))


def W_matrix(n_fft, compress=True):
    F = freq_dim(n_fft, compress)
    assert F > 0, 'nfft is not 512 or 256'
    W = np.zeros((F, n_fft // 2 + 1), np.float32)
    lib().salsa_oracle_W(n_fft, int(compress), _fp(W))
    return W


def stft(y, n_fft=512, hop=300, win=None):
    y = np.ascontiguousarray(y, np.float32)
    win = n_fft if win is None else win
    T = n_frames(y.shape[0], hop)
    out = np.zeros((n_fft // 2 + 1, T), np.complex64)
    rc = lib().salsa_oracle_stft(_fp(y), y.shape[0], n_fft, hop, win, _fp(out.view(np.float32)))
    assert rc == 0
    return out


def logspec(audio, n_fft=512, hop=300, win=None, compress=True):
    audio = np.ascontiguousarray(audio, np.float32)
    win = n_fft if win is None else win
    Cn, N = audio.shape
    out = np.zeros((Cn, n_frames(N, hop), freq_dim(n_fft, compress)), np.float32)
    rc = lib().salsa_oracle_logspec(_fp(audio), Cn, N, n_fft, hop, win, int(compress), _fp(out))
    assert rc == 0
    return out


def extract_normalized_eigenvector(X, condition_number=5.0, n_hopframes=3, is_tracking=True, audio_format='foa',
                                   fs=None, n_fft=None, lower_bin=None, return_aux=False):
    """X (n_bins, n_frames, 4) complex -> (3, n_bins, n_frames) float64.  Mirrors salsa_feature_extraction.py:17."""
    if audio_format not in ('foa', 'mic'):
        raise ValueError('audio format {} is not valid'.format(audio_format))
    X = np.ascontiguousarray(X, np.complex64)
    nb, nt, nc = X.shape
    assert nc == 4
    out = np.zeros((3, nb, nt), np.float64)
    sig = np.zeros((nb, nt), np.uint8)
    rank = np.zeros((nb, nt), np.uint8)
    margin = np.zeros((nb, nt), np.float64)
    rc = lib().salsa_oracle_eigvec(_fp(X.view(np.float32)), nb, nt, float(condition_number), int(n_hopframes),
                                   int(bool(is_tracking)), 0 if audio_format == 'foa' else 1, int(fs or 0),
                                   int(n_fft or 0), int(lower_bin or 0), _dp(out), _up(sig), _up(rank), _dp(margin))
    assert rc == 0
    if return_aux:
        return out, dict(sig=sig.astype(bool), rank=rank, margin=margin)
    return out


def extract_salsa(audio, fs=24000, n_fft=512, hop=300, win=None, fmin_doa=50, fmax_doa=9000, cond_num=5.0,
                  n_hopframes=3, is_tracking=True, is_compress_high_freq=True, audio_format='foa', return_aux=False):
    """audio (4, N) float32 -> (7, T, F) float32: the per-file body of salsa_feature_extraction.py:353-377."""
    if audio_format not in ('foa', 'mic'):
        raise ValueError('Unknown audio format {}'.format(audio_format))
    audio = np.ascontiguousarray(audio, np.float32)
    win = n_fft if win is None else win
    N = audio.shape[1]
    T, F = n_frames(N, hop), freq_dim(n_fft, is_compress_high_freq)
    assert F > 0, 'only 256 or 512 fft is supported'
    lo, up, _ = bin_limits(fs, n_fft, fmin_doa, fmax_doa)
    out = np.zeros((7, T, F), np.float32)
    rank = np.zeros((max(up - lo, 1), T), np.uint8)
    margin = np.zeros((max(up - lo, 1), T), np.float64)
    rc = lib().salsa_oracle_extract_salsa(_fp(audio), N, fs, n_fft, hop, win, int(fmin_doa), int(fmax_doa),
                                          float(cond_num), int(n_hopframes), int(bool(is_tracking)),
                                          int(bool(is_compress_high_freq)), 0 if audio_format == 'foa' else 1,
                                          _fp(out), _up(rank), _dp(margin))
    assert rc == 0, rc
    if return_aux:
        return out, dict(rank=rank, margin=margin, lower_bin=lo, upper_bin=up)
    return out


def extract_lite(audio, fs=24000, n_fft=512, hop=300, win=None, fmin_doa=50, fmax_doa=2000,
                 feature_type='salsa_lite'):
    """audio (4, N) float32 -> (7, T, cutoff-lower) float32: salsa_lite_feature_extraction.py:94-123."""
    assert feature_type in ['salsa_lite', 'salsa_ipd'], 'Invalid feature type {}'.format(feature_type)
    audio = np.ascontiguousarray(audio, np.float32)
    win = n_fft if win is None else win
    N = audio.shape[1]
    lo, up, cut = bin_limits(fs, n_fft, fmin_doa, fmax_doa)
    assert up <= cut, 'Upper bin for spatial feature is higher than cutoff bin for spectrogram!'
    out = np.zeros((7, n_frames(N, hop), cut - lo), np.float32)
    rc = lib().salsa_oracle_extract_lite(_fp(audio), N, fs, n_fft, hop, win, int(fmin_doa), int(fmax_doa),
                                         int(feature_type == 'salsa_ipd'), _fp(out))
    assert rc == 0, rc
    return out


def compute_scaler(features):
    """Iterable of (7,T,F) arrays -> (mean, std) each (4,1,F) float32: salsa_feature_extraction.py:204-256
    (sklearn StandardScaler.partial_fit per channel over all dev files; population variance)."""
    n, s, ss = 0, None, None
    for f in features:
        assert f.shape[0] == 7, 'only support n_channels = 7, got {}'.format(f.shape[0])
        x = np.asarray(f[:4], np.float64)
        if s is None:
            s, ss = np.zeros((4, x.shape[2])), np.zeros((4, x.shape[2]))
            shift = x.mean(axis=1)          # shifted sums keep the one-pass variance well conditioned
        d = x - shift[:, None, :]
        s += d.sum(axis=1)
        ss += (d * d).sum(axis=1)
        n += x.shape[1]
    mean = shift + s / n
    var = ss / n - (s / n) ** 2
    return mean[:, None, :].astype(np.float32), np.sqrt(var)[:, None, :].astype(np.float32)


def flexible(audio, kind='salsa', fs=24000, stft_winsize=512, hop_length=300, fmin_doa=50, fmax_doa=2000,
             fmax_spec=9000, clip_freqs=True, clip_spatial_alias=False, ew_thresh=5.0, covmat_avg_neighbours=3,
             is_tracking=True, floor_mask_ratio=1.5):
    """contrib/salsa_flexible.py SalsaFeatures / SalsaLiteFeatures __call__ (:237-265): (C, N) float32 ->
    (2C-1, F, T) float64 (freq-major, like the reference)."""
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    Cn, N = audio.shape
    lo, up, cut = C.c_int(), C.c_int(), C.c_int()
    if lib().salsa_oracle_flex_bins(fs, stft_winsize, fmin_doa, fmax_doa, fmax_spec, C.byref(lo), C.byref(up), C.byref(cut)):
        raise AssertionError('Upper bin for spatial feature is higher than cutoff bin for spectrogram!')
    nb = stft_winsize // 2 + 1
    F = (min(cut.value, nb) - lo.value) if clip_freqs else nb
    T = n_frames(N, hop_length)
    spec = np.empty((Cn, F, T), np.float32)
    spat = np.empty((Cn - 1, F, T), np.float64)
    rc = lib().salsa_oracle_flex(_fp(audio), Cn, N, fs, stft_winsize, hop_length, fmin_doa, fmax_doa, fmax_spec,
                                 int(kind == 'lite'), int(clip_freqs), int(clip_spatial_alias), float(ew_thresh),
                                 int(covmat_avg_neighbours), int(is_tracking), float(floor_mask_ratio), _fp(spec), _dp(spat))
    if rc != F:
        raise ValueError('salsa_oracle_flex failed: %d' % rc)
    return np.concatenate([spec.astype(np.float64), spat])


def kaiser_best_filter():
    """resampy 0.2.2 filters.py: the 'kaiser_best' table = sinc_window(num_zeros=64, precision=9, window=kaiser(beta=14.769656459379492),
    rolloff=0.9475937167399596) -> (half window float64 [64 * 512 + 1], 512).  (resampy ships it precomputed; regenerated here.)"""
    import scipy.signal
    num_zeros, precision, beta, rolloff = 64, 9, 14.769656459379492, 0.9475937167399596
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = scipy.signal.windows.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def librosa_resample(y, orig_sr, target_sr):
    """librosa 0.8.0 core/audio.py::resample(y, orig_sr, target_sr, res_type='kaiser_best', fix=True, scale=False) of a float32
    (..., n) array -- what librosa.load(sr=target_sr) applies to a file of another rate (salsa_feature_extraction.py:353).
    PARITY UNPINNED (librosa / resampy absent): see salsa_oracle.c."""
    y = np.ascontiguousarray(y, np.float32)
    if orig_sr == target_sr:
        return y
    ratio = float(target_sr) / orig_sr
    n_in = y.shape[-1]
    n_samples = int(np.ceil(n_in * ratio))                 # librosa: fix_length target
    n_out = int(n_in * ratio)                              # resampy core.py: shape[axis] = int(shape[axis] * sample_ratio)
    interp_win, num_table = kaiser_best_filter()
    interp_win = interp_win.copy()
    if ratio < 1:
        interp_win *= ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    x2 = y.reshape(-1, n_in)
    out = np.zeros((x2.shape[0], n_samples), np.float32)
    rc = lib().salsa_oracle_resample(_fp(x2), x2.shape[0], n_in, _fp(out), n_out, n_samples, ratio, _dp(interp_win), _dp(interp_delta),
                                     interp_win.shape[0], num_table)
    assert rc == 0
    return out.reshape(y.shape[:-1] + (n_samples,))
