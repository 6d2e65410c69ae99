"""Resampling on load, on the device: what ``librosa.load(path, sr=fs, mono=False, dtype=np.float32)`` does to a file whose
native rate is not ``fs`` (dataset/salsa_feature_extraction.py:353, salsa_lite_feature_extraction.py:93).

librosa 0.8.0 (``core/audio.py::resample``, res_type='kaiser_best', fix=True, scale=False) hands the float32 samples to
resampy 0.2.2 (requirements.yml:181) and pads / trims the result to ``ceil(n * ratio)`` samples.  resampy's 'kaiser_best'
filter is ``sinc_window(num_zeros=64, precision=9, window=kaiser(beta=14.769656459379492), rolloff=0.9475937167399596)``
(resampy/filters.py; shipped there as a precomputed table): a half-window of 64 * 512 + 1 float64 values, linearly interpolated
between entries by ``interpn.py::resample_f``.  This module builds that table, its first differences and the reference's
sequentially accumulated read positions on the host (a few hundred KB, cached per rate pair / length); the arithmetic on the
samples is the HIP kernel behind ``salsa_resample_batch`` (include/salsa_hip.h), which reproduces resample_f's float32
in-place accumulation tap for tap.  No CPU path: without the library / a GPU this raises like the rest of the package.

Neither librosa nor resampy is in /root/reference or in the image: PARITY UNPINNED for this step (restated from the published
algorithm; tests hold the kernel to the CPU oracle's restatement bit for bit and both to analytic properties).
"""
import ctypes as C
import functools
import math

import numpy as np
import torch

from . import _lib

KAISER_BEST = dict(num_zeros=64, precision=9, beta=14.769656459379492, rolloff=0.9475937167399596)


@functools.lru_cache(maxsize=None)
def kaiser_best_filter():
    """-> (half_window float64 [num_zeros * 2**precision + 1], num_table = 2**precision): resampy.filters.sinc_window with the
    'kaiser_best' parameters (half of a Kaiser-tapered sinc, sampled 512 times per zero crossing)."""
    from scipy.signal import windows
    p = KAISER_BEST
    num_bits = 2 ** p['precision']
    n = num_bits * p['num_zeros']
    sinc_win = p['rolloff'] * np.sinc(p['rolloff'] * np.linspace(0, p['num_zeros'], num=n + 1, endpoint=True))
    taper = windows.kaiser(2 * n + 1, p['beta'])[n:]
    half = taper * sinc_win
    half.setflags(write=False)
    return half, num_bits


def output_lengths(n_in: int, sr_orig: int, sr_new: int):
    """-> (n_out, n_fixed): resampy computes int(n * ratio) samples, librosa's fix_length makes that ceil(n * ratio)."""
    ratio = float(sr_new) / sr_orig
    return int(n_in * ratio), int(np.ceil(n_in * ratio))


@functools.lru_cache(maxsize=8)
def _tables(sr_orig: int, sr_new: int, n_out: int, device_index: int):
    ratio = float(sr_new) / sr_orig
    half, num_table = kaiser_best_filter()
    win = half * ratio if ratio < 1 else half.copy()                       # resampy/core.py: interp_win *= sample_ratio when < 1
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    treg = np.zeros(max(n_out, 1), np.float64)                             # time_register: 0, += 1/ratio each output sample,
    if n_out > 1:                                                          # accumulated sequentially (np.cumsum adds in order)
        treg[1:] = np.cumsum(np.full(n_out - 1, 1.0 / ratio))
    dev = torch.device('cuda', device_index)
    return tuple(torch.from_numpy(a).to(dev) for a in (win, delta, treg)) + (num_table, ratio)


def resample(x: torch.Tensor, sr_orig: int, sr_new: int) -> torch.Tensor:
    """x: float32 CUDA [..., n] -> float32 [..., ceil(n * sr_new / sr_orig)], librosa.resample(x, sr_orig, sr_new) of every row
    (res_type='kaiser_best', fix=True), on the current stream."""
    if not (x.is_cuda and x.dtype == torch.float32):
        raise ValueError('resample: expected a float32 CUDA tensor')
    if sr_orig == sr_new:
        return x
    x = x.contiguous()
    n_in = x.shape[-1]
    rows = x.numel() // max(n_in, 1)
    n_out, n_fix = output_lengths(n_in, sr_orig, sr_new)
    y = torch.empty(x.shape[:-1] + (n_fix,), dtype=torch.float32, device=x.device)
    if n_in == 0 or n_fix == 0:
        return y.zero_()
    win, delta, treg, num_table, ratio = _tables(sr_orig, sr_new, n_out, x.device.index or 0)
    L = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    for r0 in range(0, rows, 65535):
        r1 = min(rows, r0 + 65535)
        rc = L.salsa_resample_batch(C.c_void_p(x.data_ptr() + 4 * r0 * n_in), r1 - r0, n_in, C.c_void_p(y.data_ptr() + 4 * r0 * n_fix),
                                    n_out, n_fix, ratio, C.c_void_p(win.data_ptr()), C.c_void_p(delta.data_ptr()), win.numel(),
                                    num_table, C.c_void_p(treg.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError('salsa_resample_batch failed (%d): %s' % (rc, _lib.last_error()))
    return y


def resample_host_array(a: np.ndarray, sr_orig: int, sr_new: int, device=None) -> np.ndarray:
    """(n_channels, n) float32 host array -> resampled host array, through the device (the loader's use: io.load_audio)."""
    if not torch.cuda.is_available():
        raise RuntimeError('resampling a {} Hz file to {} Hz runs on the GPU (salsa_resample_batch); no GPU is visible'.format(sr_orig, sr_new))
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(dev)):    # loader threads: the caller's GPU, a stream of their own
        y = resample(torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev), sr_orig, sr_new)
        out = y.cpu().numpy()
    return out
