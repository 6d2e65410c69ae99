"""File I/O either side of the hot path: WAV in (the reference uses librosa.load, salsa_feature_extraction.py:353),
feature / scaler files out (h5py 'feature', 'mean', 'std' datasets, :380-382, :253-256).

h5py is not installed in the build image or on the GPU box.  When it is importable the files are real HDF5 with the
reference's dataset names, so the reference's Database (dataset/database.py:87-96, :193-195) reads them unchanged;
otherwise the same arrays go to ``<name>.npz`` next to where the ``.h5`` would be (same keys), and load_* reads either.
"""
import os

import numpy as np

try:
    import h5py  # noqa: F401
    HAVE_H5PY = True
except Exception:  # pragma: no cover - depends on the image
    HAVE_H5PY = False


def load_audio(path: str, sr: int) -> np.ndarray:
    """-> (n_channels, n_samples) float32 in [-1, 1), like librosa.load(path, sr=sr, mono=False, dtype=float32) on a
    file whose native rate is ``sr`` (the TNSSE2021 clips are 24 kHz).  librosa.load would RESAMPLE a file of another
    rate (salsa_feature_extraction.py:353); this loader raises instead -- resample such files beforehand."""
    if path.endswith('.npy'):
        a = np.load(path)
        return np.ascontiguousarray(a, dtype=np.float32)
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    if rate != sr:
        raise ValueError('{}: sample rate {} != configured fs {} (resampling is not supported)'.format(path, rate, sr))
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    return np.ascontiguousarray(x.T)


def _alt(path):
    return os.path.splitext(path)[0] + '.npz'


def save_arrays(path_h5: str, **arrays) -> str:
    """Write float32 datasets to ``path_h5`` (HDF5 when h5py exists, else the .npz twin).  Returns the path written."""
    if HAVE_H5PY:
        import h5py
        with h5py.File(path_h5, 'w') as hf:
            for k, v in arrays.items():
                hf.create_dataset(k, data=v, dtype=np.float32)
        return path_h5
    out = _alt(path_h5)
    np.savez(out, **{k: np.asarray(v, np.float32) for k, v in arrays.items()})
    return out


def load_arrays(path: str) -> dict:
    """Read a feature / scaler file by the container its EXTENSION names: ``.npz`` -> numpy, ``.h5`` -> h5py; a ``.h5``
    name whose file is absent (or unreadable without h5py) falls back to its ``.npz`` twin."""
    if path.endswith('.npz'):
        z = np.load(path)
        return {k: z[k] for k in z.files}
    if os.path.exists(path) and HAVE_H5PY:
        import h5py
        with h5py.File(path, 'r') as hf:
            return {k: hf[k][:] for k in hf.keys()}
    if os.path.exists(path) and not os.path.exists(_alt(path)):
        raise RuntimeError('{} is HDF5 and h5py is not installed here (no .npz twin next to it)'.format(path))
    z = np.load(_alt(path))
    return {k: z[k] for k in z.files}


def feature_files(feature_dir: str):
    """Sorted feature files of a split directory, ONE name per clip: when both containers of a clip are present the
    ``.h5`` is listed if h5py can read it, else the ``.npz``."""
    by_stem = {}
    for f in os.listdir(feature_dir):
        stem, ext = os.path.splitext(f)
        if ext not in ('.h5', '.npz'):
            continue
        cur = by_stem.get(stem)
        prefer_h5 = HAVE_H5PY
        if cur is None or (ext == '.h5') == prefer_h5:
            by_stem[stem] = f
    return sorted(by_stem.values())
