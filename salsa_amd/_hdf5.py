"""A few calls of the HDF5 C library through ctypes -- enough to write and read the reference's feature / scaler files (float32
datasets 'feature' (7,T,F) and 'mean' / 'std' (4,1,F) in the root group: dataset/salsa_feature_extraction.py:380-382, :253-256;
read by dataset/database.py:87-96, :193-195 with h5py) when ``h5py`` is not importable in THIS interpreter but ``libhdf5`` is on
the machine (the ROCm image ships HDF5 1.10.6 under /opt/conda/lib together with a Python 3.9 that has h5py: the tests read our files
back with that h5py, the reader the reference uses).  File I/O either side of the hot path, not the product's arithmetic.

``available()`` is False when no library can be loaded; salsa_amd/io.py then falls back to its numpy containers."""
import ctypes as C
import ctypes.util
import os

import numpy as np

_CANDIDATES = [os.environ.get('SALSA_HDF5_LIB'), ctypes.util.find_library('hdf5'), '/opt/conda/lib/libhdf5.so',
               '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so', '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so', '/usr/lib64/libhdf5.so']
_L = None
_tried = False
hid_t = C.c_int64          # HDF5 >= 1.10
H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 0, 2, 0, 0


def _lib():
    global _L, _tried
    if _tried:
        return _L
    _tried = True
    if os.environ.get('SALSA_HDF5', '1') == '0':
        return None
    for cand in _CANDIDATES:
        if not cand:
            continue
        try:
            L = C.CDLL(cand)
            if L.H5open() < 0:
                continue
            maj, mi, rel = C.c_uint(), C.c_uint(), C.c_uint()
            L.H5get_libversion(C.byref(maj), C.byref(mi), C.byref(rel))
            if (maj.value, mi.value) < (1, 10):
                continue                       # hid_t was a 32-bit int before 1.10: not bound here
            L.H5Fcreate.restype = L.H5Fopen.restype = L.H5Screate_simple.restype = L.H5Dcreate2.restype = hid_t
            L.H5Dopen2.restype = L.H5Dget_space.restype = L.H5Dget_type.restype = hid_t
            L.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid_t, hid_t]
            L.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid_t]
            L.H5Fclose.argtypes = L.H5Dclose.argtypes = L.H5Sclose.argtypes = L.H5Tclose.argtypes = [hid_t]
            L.H5Screate_simple.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            L.H5Dcreate2.argtypes = [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]
            L.H5Dwrite.argtypes = L.H5Dread.argtypes = [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]
            L.H5Dopen2.argtypes = [hid_t, C.c_char_p, hid_t]
            L.H5Dget_space.argtypes = L.H5Dget_type.argtypes = [hid_t]
            L.H5Sget_simple_extent_ndims.argtypes = [hid_t]
            L.H5Sget_simple_extent_dims.argtypes = [hid_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            L.H5Tget_class.argtypes = [hid_t]
            L.H5Tget_size.argtypes = [hid_t]
            L.H5Tget_size.restype = C.c_size_t
            L.H5Gget_num_objs.argtypes = [hid_t, C.POINTER(C.c_uint64)]
            L.H5Gget_objname_by_idx.argtypes = [hid_t, C.c_uint64, C.c_char_p, C.c_size_t]
            L.H5Gget_objname_by_idx.restype = C.c_ssize_t
            L.H5Pcreate.restype = hid_t
            L.H5Pcreate.argtypes = [hid_t]
            L.H5Pclose.argtypes = [hid_t]
            L.H5Pset_alloc_time.argtypes = L.H5Pset_fill_time.argtypes = [hid_t, C.c_int]
            L.H5Dget_offset.restype = C.c_uint64
            L.H5Dget_offset.argtypes = [hid_t]
            ts = C.c_bool(False)
            L.threadsafe = L.H5is_library_threadsafe(C.byref(ts)) >= 0 and bool(ts.value)
            L.DATASET_CREATE = hid_t.in_dll(L, 'H5P_CLS_DATASET_CREATE_ID_g').value
            L.H5Eset_auto2.argtypes = [hid_t, C.c_void_p, C.c_void_p]
            L.H5Eset_auto2(0, None, None)      # errors come back as negative return values; no stack dump on stderr
            L.version = '%d.%d.%d' % (maj.value, mi.value, rel.value)
            L.F32LE = hid_t.in_dll(L, 'H5T_IEEE_F32LE_g').value
            L.NATIVE_FLOAT = hid_t.in_dll(L, 'H5T_NATIVE_FLOAT_g').value
            L.NATIVE_DOUBLE = hid_t.in_dll(L, 'H5T_NATIVE_DOUBLE_g').value
            _L = L
            return _L
        except (OSError, AttributeError, ValueError):
            continue
    return None


def available() -> bool:
    return _lib() is not None


def version() -> str:
    return _lib().version if available() else ''


_LOCK = None
BULK_BYTES = 1 << 20      # arrays at least this large are written around the library (see write())


def _guard():
    """The library's calls from several threads: a thread-safe build serialises them itself; any other build gets a process lock."""
    global _LOCK
    if _LOCK is None:
        import threading
        _LOCK = threading.Lock()
    import contextlib
    return contextlib.nullcontext() if _lib().threadsafe else _LOCK


def write(path: str, **arrays) -> None:
    """h5py.File(path, 'w').create_dataset(name, data=a, dtype=np.float32) for every array: contiguous IEEE float32 little-endian
    datasets in the root group.  A thread-safe libhdf5 holds ONE global lock around every call, so eight writer threads pushing
    27-MB H5Dwrite()s through it would write one file at a time: large arrays are therefore only ALLOCATED through the library
    (allocation time early, fill time never), the dataset's byte offset in the file is asked for (H5Dget_offset) and the payload
    goes there with a plain pwrite() after the file is closed -- a contiguous dataset is nothing but its bytes at that offset."""
    L = _lib()
    late = []                                              # (offset, array) written around the library
    with _guard():
        fid = L.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        if fid < 0:
            raise IOError('H5Fcreate failed: ' + path)
        try:
            for name, a in arrays.items():
                a = np.ascontiguousarray(a, dtype='<f4')
                bulk = a.nbytes >= BULK_BYTES
                dims = (C.c_uint64 * max(1, a.ndim))(*a.shape)
                sid = L.H5Screate_simple(a.ndim, dims, None)
                dcpl = L.H5Pcreate(L.DATASET_CREATE) if bulk else H5P_DEFAULT
                if bulk:
                    L.H5Pset_alloc_time(dcpl, 1)           # H5D_ALLOC_TIME_EARLY
                    L.H5Pset_fill_time(dcpl, 1)            # H5D_FILL_TIME_NEVER
                did = L.H5Dcreate2(fid, name.encode(), L.F32LE, sid, H5P_DEFAULT, dcpl, H5P_DEFAULT) if sid >= 0 else -1
                rc = -1
                if did >= 0 and bulk:
                    off = L.H5Dget_offset(did)
                    rc = 0 if off != 0xFFFFFFFFFFFFFFFF else -1   # HADDR_UNDEF
                    late.append((off, a))
                elif did >= 0:
                    rc = L.H5Dwrite(did, L.NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p))
                if did >= 0:
                    L.H5Dclose(did)
                if bulk and dcpl >= 0:
                    L.H5Pclose(dcpl)
                if sid >= 0:
                    L.H5Sclose(sid)
                if rc < 0:
                    raise IOError('writing dataset %r of %s failed' % (name, path))
        finally:
            if L.H5Fclose(fid) < 0:
                raise IOError('H5Fclose failed: ' + path)
    if late:
        fd = os.open(path, os.O_WRONLY)
        try:
            for off, a in late:
                mv, done = memoryview(a).cast('B'), 0
                while done < len(mv):
                    done += os.pwrite(fd, mv[done:], off + done)
        finally:
            os.close(fd)


def read(path: str) -> dict:
    """{name: ndarray} of every dataset in the root group (``hf[name][:]``): float32 datasets as float32, float64 as float64,
    anything else converted to float64 by the library."""
    L = _lib()
    out = {}
    with _guard():
        fid = L.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
        if fid < 0:
            raise IOError('not an HDF5 file (or unreadable): ' + path)
        try:
            _read_into(L, fid, path, out)
        finally:
            L.H5Fclose(fid)
    return out


def _read_into(L, fid, path, out):
    if True:
        n = C.c_uint64()
        if L.H5Gget_num_objs(fid, C.byref(n)) < 0:
            raise IOError('cannot list ' + path)
        for i in range(n.value):
            buf = C.create_string_buffer(1024)
            if L.H5Gget_objname_by_idx(fid, i, buf, 1024) < 0:
                raise IOError('cannot name object %d of %s' % (i, path))
            did = L.H5Dopen2(fid, buf.value, H5P_DEFAULT)
            if did < 0:
                continue                                   # a group, not a dataset
            sid, tid = L.H5Dget_space(did), L.H5Dget_type(did)
            nd = L.H5Sget_simple_extent_ndims(sid)
            dims = (C.c_uint64 * max(1, nd))()
            L.H5Sget_simple_extent_dims(sid, dims, None)
            is_f32 = L.H5Tget_class(tid) == 1 and L.H5Tget_size(tid) == 4          # H5T_FLOAT = 1
            a = np.empty(tuple(int(d) for d in dims[:nd]), dtype=np.float32 if is_f32 else np.float64)
            rc = L.H5Dread(did, L.NATIVE_FLOAT if is_f32 else L.NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p))
            L.H5Tclose(tid), L.H5Sclose(sid), L.H5Dclose(did)
            if rc < 0:
                raise IOError('reading dataset %r of %s failed' % (buf.value.decode(), path))
            out[buf.value.decode()] = a
