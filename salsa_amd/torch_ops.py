"""``torch.ops.salsa.extract``: the batched extractor as a registered PyTorch operator ("thin torch extension
registers torch.ops.salsa.extract(audio: Tensor[B,4,N] f32 cuda, ...) -> Tensor[B,7,T,F]").  Pure plumbing over
salsa_amd.extractor.SalsaExtractor / the C ABI: plans are cached per (device, parameters); a fake (meta) implementation
gives the output shape, so the op composes with FakeTensor-based tooling.  Importing this module registers the op."""
import torch

from . import _lib

_PLANS = {}


def _plan(device, audio_format, feature_type, fs, n_fft, hop_len, fmin_doa, fmax_doa, cond_num, n_hopframes, is_tracking,
          is_compress_high_freq):
    from .extractor import SalsaExtractor
    key = (str(device), audio_format, feature_type, fs, n_fft, hop_len, fmin_doa, fmax_doa, cond_num, n_hopframes,
           is_tracking, is_compress_high_freq)
    if key not in _PLANS:
        _PLANS[key] = SalsaExtractor(fs=fs, n_fft=n_fft, hop_len=hop_len, fmin_doa=fmin_doa, fmax_doa=fmax_doa,
                                     cond_num=cond_num, n_hopframes=n_hopframes, is_tracking=is_tracking,
                                     is_compress_high_freq=is_compress_high_freq, audio_format=audio_format,
                                     feature_type=feature_type, device=device)
    return _PLANS[key]


def output_shape(n_samples, audio_format='foa', feature_type='salsa', fs=24000, n_fft=512, hop_len=300, fmin_doa=50,
                 fmax_doa=9000, is_compress_high_freq=True):
    """(7, T, F) of the feature array, host arithmetic only (salsa_feature_extraction.py:298-313, lite :50-59)."""
    if n_fft not in (256, 512):
        raise AssertionError('only 256 or 512 fft is supported')
    T = 1 + n_samples // hop_len
    if feature_type == 'salsa':
        F = ((200 if n_fft == 512 else 100) if is_compress_high_freq else n_fft // 2)
    else:
        lower = max(1, int(fmin_doa * n_fft // fs))
        cutoff = min(int(9000 * n_fft // fs), n_fft // 2 + 1)
        F = cutoff - lower
    return 7, T, F


@torch.library.custom_op('salsa::extract', mutates_args=())
def extract(audio: torch.Tensor, audio_format: str = 'foa', feature_type: str = 'salsa', fs: int = 24000,
            n_fft: int = 512, hop_len: int = 300, fmin_doa: int = 50, fmax_doa: int = 9000, cond_num: float = 5.0,
            n_hopframes: int = 3, is_tracking: bool = True, is_compress_high_freq: bool = True) -> torch.Tensor:
    """audio float32 CUDA [B,4,N] (planar) -> features float32 [B,7,T,F]."""
    if not audio.is_cuda:
        raise RuntimeError('salsa::extract runs on an MI355X only (libsalsa_hip.so has no CPU path)')
    ex = _plan(audio.device, audio_format, feature_type, fs, n_fft, hop_len, fmin_doa, fmax_doa, cond_num, n_hopframes,
               is_tracking, is_compress_high_freq)
    return ex.extract(audio.contiguous())


@extract.register_fake
def _(audio, audio_format='foa', feature_type='salsa', fs=24000, n_fft=512, hop_len=300, fmin_doa=50, fmax_doa=9000,
      cond_num=5.0, n_hopframes=3, is_tracking=True, is_compress_high_freq=True):
    if audio_format not in _lib.FORMAT:
        raise ValueError('Unknown audio format {}'.format(audio_format))
    c, t, f = output_shape(audio.shape[2], audio_format, feature_type, fs, n_fft, hop_len, fmin_doa, fmax_doa,
                           is_compress_high_freq)
    return audio.new_empty((audio.shape[0], c, t, f))
