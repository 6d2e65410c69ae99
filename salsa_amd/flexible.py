"""On-the-fly SALSA / SALSA-Lite for MI355X behind the call surface of the reference's contrib/salsa_flexible.py.

    sf = SalsaFeatures(fs=24000, stft_winsize=512, hop_length=300, fmin_doa=50, fmax_doa=2000, fmax_spec=9000)
    s = sf(wav, clip_freqs=True, clip_spatial_alias=False, ew_thresh=5.0, covmat_avg_neighbours=3,
           is_tracking=True, floor_mask_ratio=1.5)            # (C + C-1, freqbins, time) float64, like :237-265

Same constructor arguments, call arguments, defaults, assertions and output layout as ``SalsaFeatures`` (:271-367)
and ``SalsaLiteFeatures`` (:373-400); the arithmetic runs in libsalsa_hip.so (salsa_extract_batch with
SALSA_FLAG_FLEX + salsa_to_freq_major -- include/salsa_hip.h lists what that flag changes relative to the dataset
scripts).  ``extract_batch`` is the same computation for a device-resident batch of clips (the Dataset-fused form SURVEY
config 4 wants).  Any microphone count from 2 to 16 (include/salsa_hip.h SALSA_MAX_MICS), like the reference's "arbitrary
channels": 2 or 3 microphones are padded with silent channels to the 4-channel kernels (closed-form 4 x 4 eigen-solver), 5 - 16
to an even count for salsa_extract_multichannel (N x N Hermitian eigenproblem by cyclic Jacobi, one lane per gated TF bin).  A silent channel
leaves the coherence gate and the principal eigenvector unchanged (the covariance only gains a zero eigenvalue); its
output planes are dropped.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .extractor import SalsaExtractor, _raise

MAX_MICS = 16   # include/salsa_hip.h SALSA_MAX_MICS


class SpatialFeaturesAbstract:
    """Common part of both representations (contrib/salsa_flexible.py:149-265)."""

    SOUND_SPEED = 343  # m/s
    F_DTYPE = np.float32
    _FEATURE = None

    def __init__(self, fs=24000, stft_winsize=512, hop_length=300, fmin_doa=50, fmax_doa=2000, fmax_spec=9000,
                 device=None):
        n_bins = stft_winsize // 2 + 1
        lower_bin = max(1, int(np.floor(fmin_doa * stft_winsize / float(fs))))
        upper_bin = int(np.floor(fmax_doa * stft_winsize / float(fs)))
        cutoff_bin = int(np.floor(fmax_spec * stft_winsize / float(fs)))
        assert upper_bin <= cutoff_bin, "Upper bin for spatial feature is " + \
            "higher than cutoff bin for spectrogram!"
        self.delta = 2 * np.pi * fs / (stft_winsize * self.SOUND_SPEED)
        self.norm_freq = np.arange(n_bins, dtype=self.F_DTYPE)[:, None]
        self.norm_freq[0, 0] = 1
        self.norm_freq *= self.delta
        self.fs, self.stft_winsize, self.hop_length, self.n_bins = fs, stft_winsize, hop_length, n_bins
        self.fmin_doa, self.fmax_doa, self.fmax_spec = fmin_doa, fmax_doa, fmax_spec
        self.lobin, self.upbin, self.cutbin = lower_bin, upper_bin, cutoff_bin
        self.device = device
        self._plans = {}

    def _plan(self, clip_freqs, clip_spatial_alias, ew_thresh=5.0, covmat_avg_neighbours=3, is_tracking=True,
              floor_mask_ratio=1.5):
        key = (bool(clip_freqs), bool(clip_spatial_alias), float(ew_thresh), int(covmat_avg_neighbours),
               bool(is_tracking), float(floor_mask_ratio))
        if key not in self._plans:
            flags = _lib.FLAG_FLEX | (0 if clip_freqs else _lib.FLAG_NO_CLIP_FREQS) | \
                (_lib.FLAG_CLIP_SPATIAL_ALIAS if clip_spatial_alias else 0)
            self._plans[key] = SalsaExtractor(
                fs=self.fs, n_fft=self.stft_winsize, hop_len=self.hop_length, win_len=self.stft_winsize,
                fmin_doa=self.fmin_doa, fmax_doa=self.fmax_doa, cond_num=ew_thresh, n_hopframes=covmat_avg_neighbours,
                is_tracking=is_tracking, is_compress_high_freq=False, audio_format='mic', feature_type=self._FEATURE,
                device=self.device, flags=flags, floor_mask_ratio=floor_mask_ratio, fmax_spec=self.fmax_spec)
        return self._plans[key]

    def extract_batch(self, audio: torch.Tensor, clip_freqs=True, clip_spatial_alias=False, **feat_kwargs) -> torch.Tensor:
        """audio float32 CUDA [B, C, N], 2 <= C <= 16 -> float64 CUDA [B, 2C-1, F, T] (freq-major like the reference)."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 3
        B, n_ch, N = audio.shape
        if not 2 <= n_ch <= MAX_MICS:
            raise ValueError('the MI355X kernels take 2 to %d microphones, got %d' % (MAX_MICS, n_ch))
        ex = self._plan(clip_freqs, clip_spatial_alias, **feat_kwargs)
        if n_ch > 4:
            n_pad = n_ch + (n_ch & 1)                                              # even: 6 .. 16
            if n_pad != n_ch:
                audio = torch.cat([audio, audio.new_zeros((B, 1, N))], dim=1)
            out = to_freq_major(ex.extract_multichannel(audio.contiguous()))        # [B, 2*n_pad-1, F, T] float64
            if n_pad != n_ch:
                out = torch.cat([out[:, :n_ch], out[:, n_pad:n_pad + n_ch - 1]], dim=1)
            return out
        if n_ch < 4:
            audio = torch.cat([audio, audio.new_zeros((B, 4 - n_ch, N))], dim=1)
        feat = ex.extract(audio.contiguous())                                  # [B, 7, T, F] float32
        out = to_freq_major(feat)                                              # [B, 7, F, T] float64
        if n_ch < 4:
            out = torch.cat([out[:, :n_ch], out[:, 4:4 + n_ch - 1]], dim=1)
        return out

    def __call__(self, wavchans, clip_freqs, clip_spatial_alias, **feat_kwargs):
        """(channels, samples) float32 numpy -> (channels + channels-1, freqbins, time) float64 numpy (:237-265)."""
        _, _ = wavchans.shape  # rank 2, like the reference
        assert wavchans.dtype == self.F_DTYPE, f"{self.F_DTYPE} expected!"
        dev = self.device if self.device is not None else 'cuda:%d' % torch.cuda.current_device()
        audio = torch.from_numpy(np.ascontiguousarray(wavchans)).to(dev)[None]
        return self.extract_batch(audio, clip_freqs, clip_spatial_alias, **feat_kwargs)[0].cpu().numpy()


class SalsaFeatures(SpatialFeaturesAbstract):
    """contrib/salsa_flexible.py:271-367.  Call kwargs: ew_thresh=5.0, covmat_avg_neighbours=3, is_tracking=True,
    floor_mask_ratio=1.5."""
    _FEATURE = 'salsa'


class SalsaLiteFeatures(SpatialFeaturesAbstract):
    """contrib/salsa_flexible.py:373-400 (no feature kwargs)."""
    _FEATURE = 'salsa_lite'

    def _plan(self, clip_freqs, clip_spatial_alias):
        return super()._plan(clip_freqs, clip_spatial_alias)


def to_freq_major(feat: torch.Tensor) -> torch.Tensor:
    """[..., T, F] float32 CUDA (time-major) -> [..., F, T] float64 (salsa_to_freq_major)."""
    assert feat.is_cuda and feat.dtype == torch.float32 and feat.is_contiguous() and feat.dim() >= 2
    T, F = feat.shape[-2:]
    rows = feat.numel() // (T * F)
    out = torch.empty(feat.shape[:-2] + (F, T), dtype=torch.float64, device=feat.device)
    with torch.cuda.device(feat.device):
        rc = _lib.load().salsa_to_freq_major(C.c_void_p(feat.data_ptr()), rows, T, F, C.c_void_p(out.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        _raise(rc)
    return out
