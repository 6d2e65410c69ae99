"""Import harness for the UPSTREAM reference (/root/reference, read-only) -- used ONLY by tools/make_golden.py in the
build container to produce tests/golden/*.npz.  Nothing under tests/, bench.py or the product imports this file, and
the reference itself never travels to the GPU box.

The reference's hot path imports three packages that are not installed here (librosa 0.8.0, h5py, fire).  Their
arithmetic is not under /root/reference, so it is restated below from the pinned version's published behaviour
(requirements.yml:101 librosa==0.8.0; SURVEY.md section 8c lists the call sites):

* ``librosa.stft(y, n_fft, hop_length, win_length=None, window='hann', center=True, pad_mode='reflect')``
  (librosa 0.8.0 core/spectrum.py): periodic Hann from scipy.signal.get_window(fftbins=True) in float64, zero-padded
  centred to n_fft; ``np.pad(y, n_fft//2, mode='reflect')``; frames y[t*hop : t*hop+n_fft], T = 1 + len(y)//hop;
  ``np.fft.rfft(window * frames, axis=0)`` evaluated in float64 and STORED into a complex64 matrix (dtype follows the
  float32 input) -- i.e. values are float64-accurate then rounded to float32.
* ``librosa.power_to_db(S, ref=1.0, amin=1e-10, top_db=None)`` = 10*log10(max(amin,S)) - 10*log10(max(amin,ref)),
  evaluated in the dtype of S (float32 on this path).
* ``librosa.load(path, sr, mono=False, dtype=float32)`` on a native-rate file: samples as (C, N) float32; the shim
  serves arrays registered in-memory (or .npy payloads) instead of decoding WAV.
* ``h5py.File(path,'w').create_dataset(name, data, dtype)`` / ``hf[name][:]``: an in-memory capture.
* ``fire.Fire``: no-op.
"""
import os
import sys
import types

import numpy as np
import scipy.signal

REF_ROOT = '/root/reference'

# ------------------------------------------------------------------------------------------------ librosa shim
_AUDIO = {}


def register_audio(path: str, audio: np.ndarray) -> None:
    _AUDIO[os.path.abspath(path)] = np.ascontiguousarray(audio, dtype=np.float32)


def _load(path, sr=None, mono=False, dtype=np.float32, **kw):
    a = _AUDIO[os.path.abspath(path)]
    return a.astype(dtype, copy=True), sr


def _pad_center(w, size):
    lpad = (size - len(w)) // 2
    return np.pad(w, (lpad, size - len(w) - lpad), mode='constant')


def _stft(y, n_fft=2048, hop_length=None, win_length=None, window='hann', center=True, dtype=None,
          pad_mode='reflect'):
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = win_length // 4
    fft_window = scipy.signal.get_window(window, win_length, fftbins=True)
    fft_window = _pad_center(fft_window, n_fft).reshape((-1, 1))
    y = np.asarray(y)
    if center:
        y = np.pad(y, int(n_fft // 2), mode=pad_mode)
    n_frames = 1 + (len(y) - n_fft) // hop_length
    frames = np.lib.stride_tricks.as_strided(y, shape=(n_fft, n_frames),
                                             strides=(y.itemsize, hop_length * y.itemsize))
    if dtype is None:
        dtype = np.complex64 if y.dtype == np.float32 else np.complex128
    out = np.empty((1 + n_fft // 2, n_frames), dtype=dtype, order='F')
    blk = 4096
    for s in range(0, n_frames, blk):
        out[:, s:s + blk] = np.fft.rfft(fft_window * frames[:, s:s + blk], axis=0)
    return out


def _power_to_db(S, ref=1.0, amin=1e-10, top_db=80.0):
    S = np.asarray(S)
    magnitude = np.abs(S) if np.issubdtype(S.dtype, np.complexfloating) else S
    ref_value = np.abs(ref)
    log_spec = 10.0 * np.log10(np.maximum(amin, magnitude))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


# ------------------------------------------------------------------------------------------------ h5py shim
H5_STORE = {}


class _H5File:
    def __init__(self, path, mode='r'):
        self.path = os.path.abspath(path)
        self.mode = mode
        if 'w' in mode:
            H5_STORE[self.path] = {}
            open(self.path, 'wb').close()      # compute_scaler() discovers feature files with os.listdir

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def create_dataset(self, name, data=None, dtype=None):
        H5_STORE[self.path][name] = np.array(data, dtype=dtype)

    def __getitem__(self, name):
        return H5_STORE[self.path][name]


def install():
    """Insert the stub modules and numpy-1.19 aliases, put the reference on sys.path."""
    lib = types.ModuleType('librosa')
    lib.load, lib.stft, lib.power_to_db = _load, _stft, _power_to_db
    h5 = types.ModuleType('h5py')
    h5.File = _H5File
    fire = types.ModuleType('fire')
    fire.Fire = lambda *a, **k: None
    sys.modules.setdefault('librosa', lib)
    sys.modules.setdefault('h5py', h5)
    sys.modules.setdefault('fire', fire)
    if not hasattr(np, 'int'):
        np.int = int          # removed in numpy >= 1.24; used at salsa_feature_extraction.py:302-303
    if not hasattr(np, 'float'):
        np.float = float
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
