"""File I/O either side of the hot path: WAV in (the reference uses librosa.load, salsa_feature_extraction.py:353),
feature / scaler files out (h5py 'feature', 'mean', 'std' datasets, :380-382, :253-256).

The files are real HDF5 with the reference's dataset names -- so the reference's Database (dataset/database.py:87-96, :193-195)
reads them unchanged -- whenever an HDF5 library can be reached: ``h5py`` if it is importable in this interpreter, else ``libhdf5``
itself through ctypes (salsa_amd/_hdf5.py; the ROCm image carries HDF5 1.10.6 under /opt/conda/lib, and tests/test_host_logic_cpu.py
reads the files back with that conda's h5py 3.3.0 -- the reader the reference uses).  With neither, a clip's ``'feature'`` array goes to ``<name>.npy`` next to where the ``.h5`` would be (round 6: one header + one raw
write of the pinned buffer -- the ``.npz`` twin of rounds 1 - 5 spent 27 of its 39 ms per clip in the zip layer's CRC-32) and
multi-array files (the scaler's ``'mean'`` / ``'std'``) to ``<name>.npz``; load_* reads all three.
"""
import os

import numpy as np

try:
    import h5py  # noqa: F401
    HAVE_H5PY = True
except Exception:  # pragma: no cover - depends on the image
    HAVE_H5PY = False
from . import _hdf5   # noqa: E402  (round 6: the HDF5 C library itself through ctypes, when h5py is not importable here but libhdf5 exists)

HAVE_HDF5 = HAVE_H5PY or _hdf5.available()    # real .h5 files are written / read; False: the numpy twins


def load_audio(path: str, sr: int, device=None) -> np.ndarray:
    """-> (n_channels, n_samples) float32 in [-1, 1), like librosa.load(path, sr=sr, mono=False, dtype=float32)
    (salsa_feature_extraction.py:353).  A WAV file whose native rate is not ``sr`` is RESAMPLED as librosa does ('kaiser_best',
    result fixed to ceil(n * sr / native) samples) -- on the device (salsa_amd/resample.py -> salsa_resample_batch; round 6); the
    TNSSE2021 clips are native 24 kHz and never take that branch.  ``.npy`` clips carry no rate and are taken as ``sr``."""
    if path.endswith('.npy'):
        a = np.load(path)
        return np.ascontiguousarray(a, dtype=np.float32)
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
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
    x = np.ascontiguousarray(x.T)
    if rate != sr:
        from . import resample
        x = resample.resample_host_array(x, int(rate), int(sr), device=device)
    return x


def audio_shape(path: str, sr: int):
    """(n_channels, n_samples) of a clip AS load_audio(path, sr) WILL RETURN IT, from its header alone (the file pipeline groups
    clips by length before any sample is read): a file of another rate counts with its resampled length."""
    if path.endswith('.npy'):
        with open(path, 'rb') as f:
            shape, _, _ = _npy_header(f)
        assert len(shape) == 2, '{}: expected a (n_channels, n_samples) array'.format(path)
        return int(shape[0]), int(shape[1])
    from scipy.io import wavfile
    rate, data = wavfile.read(path, mmap=True)
    n = int(data.shape[0])
    if rate != sr:
        n = int(np.ceil(n * (float(sr) / rate)))                      # librosa.resample: n_samples = ceil(n * ratio)
    return (1, n) if data.ndim == 1 else (int(data.shape[1]), n)


PCM_CODES = {'int16': 1, 'int32': 2, 'uint8': 3, 'float32': 4}      # SALSA_PCM_* of include/salsa_hip.h


def wav_pcm_layout(path: str, sr: int):
    """-> (sample_format, n_channels, n_frames, byte offset of the samples) of a WAV file whose data chunk the DEVICE can turn into what
    load_audio(path, sr) returns (salsa_pcm_to_planar: plain little-endian 8 / 16 / 32-bit PCM or float32 at the configured rate), else None
    (another rate: resampling; 24-bit containers, big-endian, anything scipy cannot memory-map: the host decoder of load_audio)."""
    if not path.endswith('.wav'):
        return None
    try:
        from scipy.io import wavfile
        rate, data = wavfile.read(path, mmap=True)
    except Exception:
        return None
    if rate != sr or not isinstance(data, np.memmap) or data.dtype.name not in PCM_CODES or data.dtype.byteorder == '>' \
            or not data.flags.c_contiguous or data.shape[0] == 0:
        return None
    return PCM_CODES[data.dtype.name], (1 if data.ndim == 1 else int(data.shape[1])), int(data.shape[0]), int(data.offset)


def read_raw_into(path: str, offset: int, dst_bytes: np.ndarray) -> None:
    """len(dst_bytes) bytes of the file from ``offset`` straight into a (pinned) uint8 buffer"""
    mv = memoryview(dst_bytes).cast('B')
    with open(path, 'rb') as f:
        f.seek(offset)
        got = 0
        while got < len(mv):
            k = f.readinto(mv[got:])
            if not k:
                raise IOError('{}: truncated data chunk'.format(path))
            got += k


def _npy_header(f):
    """-> (shape, fortran_order, dtype) of an open .npy file, positioned at the data"""
    major, _ = np.lib.format.read_magic(f)
    return np.lib.format.read_array_header_1_0(f) if major == 1 else np.lib.format.read_array_header_2_0(f)


def load_audio_into(path: str, sr: int, dst: np.ndarray, planar: bool = True, device=None) -> None:
    """load_audio(path, sr) written INTO ``dst`` -- (n_channels, n_samples) float32 when ``planar``, (n_samples, n_channels) otherwise
    -- without an intermediate array where the file already holds that layout: a float32 C-order .npy is read straight into the
    (pinned) destination with one readinto()."""
    if path.endswith('.npy') and planar and dst.flags.c_contiguous and dst.dtype == np.float32:
        with open(path, 'rb') as f:
            shape, fortran, dtype = _npy_header(f)
            if tuple(shape) == dst.shape and not fortran and dtype == np.float32:
                mv = memoryview(dst).cast('B')
                got = 0
                while got < len(mv):                    # (readinto may return short counts on some file systems)
                    k = f.readinto(mv[got:])
                    if not k:
                        raise IOError('{}: truncated .npy payload'.format(path))
                    got += k
                return
    a = load_audio(path, sr, device=device)
    dst[...] = a if planar else a.T


def _alt(path):
    return os.path.splitext(path)[0] + '.npz'


def _alt_npy(path):
    return os.path.splitext(path)[0] + '.npy'


def _write_npy(path: str, a: np.ndarray) -> None:
    """np.save(path, a) for a C-contiguous array as ONE header + ONE write of its buffer (np.save of an array that lives in
    pinned memory goes through tobytes(): a second copy of 27 MB per clip)"""
    a = np.ascontiguousarray(a)
    with open(path, 'wb') as f:
        np.lib.format.write_array_header_1_0(f, np.lib.format.header_data_from_array_1_0(a))
        f.write(memoryview(a).cast('B'))


def save_arrays(path_h5: str, **arrays) -> str:
    """Write float32 datasets to ``path_h5``: HDF5 when h5py exists; else a lone 'feature' array as the raw ``.npy`` twin and
    anything else as the ``.npz`` twin.  Returns the path written."""
    if HAVE_H5PY:
        import h5py
        with h5py.File(path_h5, 'w') as hf:
            for k, v in arrays.items():
                hf.create_dataset(k, data=v, dtype=np.float32)
        return path_h5
    if _hdf5.available():
        _hdf5.write(path_h5, **arrays)
        return path_h5
    if set(arrays) == {'feature'}:
        out = _alt_npy(path_h5)
        _write_npy(out, np.asarray(arrays['feature'], np.float32))
        return out
    out = _alt(path_h5)
    np.savez(out, **{k: np.asarray(v, np.float32) for k, v in arrays.items()})
    return out


def load_arrays(path: str) -> dict:
    """Read a feature / scaler file by the container its EXTENSION names: ``.npz`` / ``.npy`` -> numpy (a ``.npy`` file is a
    clip's 'feature' array), ``.h5`` -> h5py; a ``.h5`` name whose file is absent (or unreadable without h5py) falls back to its
    ``.npy`` / ``.npz`` twin."""
    if path.endswith('.npz'):
        z = np.load(path)
        return {k: z[k] for k in z.files}
    if path.endswith('.npy'):
        return {'feature': np.load(path)}
    if os.path.exists(path) and HAVE_H5PY:
        import h5py
        with h5py.File(path, 'r') as hf:
            return {k: hf[k][:] for k in hf.keys()}
    if os.path.exists(path) and _hdf5.available():
        try:
            return _hdf5.read(path)
        except IOError:
            if not (os.path.exists(_alt_npy(path)) or os.path.exists(_alt(path))):
                raise
    if os.path.exists(_alt_npy(path)):
        return {'feature': np.load(_alt_npy(path))}
    if os.path.exists(path) and not os.path.exists(_alt(path)):
        raise RuntimeError('{} is HDF5 and neither h5py nor libhdf5 can be loaded here (no .npy / .npz twin next to it)'.format(path))
    z = np.load(_alt(path))
    return {k: z[k] for k in z.files}


def feature_files(feature_dir: str):
    """Sorted feature files of a split directory, ONE name per clip: when several containers of a clip are present the
    ``.h5`` is listed if h5py can read it, else the ``.npy``, else the ``.npz``."""
    rank = {'.h5': 0 if HAVE_HDF5 else 3, '.npy': 1, '.npz': 2}
    by_stem = {}
    for f in os.listdir(feature_dir):
        stem, ext = os.path.splitext(f)
        if ext not in rank:
            continue
        cur = by_stem.get(stem)
        if cur is None or rank[ext] < rank[os.path.splitext(cur)[1]]:
            by_stem[stem] = f
    return sorted(by_stem.values())
