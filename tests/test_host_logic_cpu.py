"""CPU tests of the host-side logic around the hot path (file naming, WAV reading, feature containers, CLI flags,
error behaviour without a GPU).  No compute calls."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from salsa_amd import io as sio


def test_feature_name_follows_reference_replace_quirk():
    from salsa_amd.features import feature_name
    assert feature_name('fold1_room1_mix001.wav') == 'fold1_room1_mix001.h5'
    assert feature_name('wavy_clip.wav') == 'h5y_clip.h5'          # reference: audio_fn.replace('wav','h5') (:379)
    assert feature_name('clip.npy') == 'clip.h5'


def test_wav_reader_matches_float_and_int16_conventions(tmp_path):
    from scipy.io import wavfile
    rng = np.random.RandomState(0)
    x = (rng.uniform(-1, 1, (4, 1000)) * 0.5).astype(np.float32)
    wavfile.write(tmp_path / 'f32.wav', 24000, x.T)
    assert np.array_equal(sio.load_audio(str(tmp_path / 'f32.wav'), 24000), x)
    i16 = (x * 32767).astype(np.int16)
    wavfile.write(tmp_path / 'i16.wav', 24000, i16.T)
    np.testing.assert_array_equal(sio.load_audio(str(tmp_path / 'i16.wav'), 24000), i16.astype(np.float32) / 32768.0)
    # a file of another rate is resampled as librosa.load does -- on the device (tests/test_resample.py); the header says how long it will be
    assert sio.audio_shape(str(tmp_path / 'f32.wav'), 48000) == (4, 2000)


def test_feature_container_roundtrip(tmp_path):
    f = np.random.RandomState(1).randn(7, 5, 200).astype(np.float32)
    written = sio.save_arrays(str(tmp_path / 'a.h5'), feature=f)
    assert os.path.exists(written)
    assert np.array_equal(sio.load_arrays(str(tmp_path / 'a.h5'))['feature'], f)
    assert sio.feature_files(str(tmp_path)) == [os.path.basename(written)]


def test_cli_flag_parsing():
    from salsa_amd.features import _cli
    got = {}
    _cli(lambda **kw: got.update(kw), ['--data_config=x.yml', '--cond_num=5', '--is_tracking=False', '--task=feature'])
    assert got == {'data_config': 'x.yml', 'cond_num': 5, 'is_tracking': False, 'task': 'feature'}


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from salsa_amd.features import extract_normalized_eigenvector
    with pytest.raises(RuntimeError):
        extract_normalized_eigenvector(np.zeros((2, 8, 4), np.complex64))
    with pytest.raises(ValueError):
        extract_normalized_eigenvector(np.zeros((2, 8, 4), np.complex64), audio_format='xyz')


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: with libsalsa_hip.so absent the binding raises and names the build command."""
    from salsa_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'libsalsa_hip.so'))
    with pytest.raises(RuntimeError, match='hipcc'):
        _lib.load()


def test_torch_op_is_registered_with_a_shape_function():
    """torch.ops.salsa.extract: registered on import, FakeTensor shapes follow the reference's (7,T,F), and a
    CPU tensor is refused (no CPU path)."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    import salsa_amd.torch_ops  # noqa: F401
    with FakeTensorMode():
        a = torch.empty(3, 4, 48000)
        assert tuple(torch.ops.salsa.extract(a).shape) == (3, 7, 161, 200)
        assert tuple(torch.ops.salsa.extract(a, 'mic', 'salsa_lite', 24000, 512, 300, 50, 2000).shape) == (3, 7, 161, 191)
        assert tuple(torch.ops.salsa.extract(a, 'foa', 'salsa', 24000, 256, 150, 50, 9000, 5.0, 3, True, False).shape) == (3, 7, 321, 128)
    with pytest.raises(RuntimeError):
        torch.ops.salsa.extract(torch.zeros(1, 4, 4000))


def test_bench_self_spawn_becomes_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (N ranks on loopback);
    with WORLD_SIZE already set, or N = 1, it stays in-process."""
    import bench_crnn
    calls = []
    monkeypatch.setattr(os, 'execv', lambda exe, argv: calls.append((exe, list(argv))))
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    bench_crnn.self_spawn(1, os.path.join(ROOT, 'bench.py'))
    assert calls == []
    bench_crnn.self_spawn(4, os.path.join(ROOT, 'bench.py'))
    (exe, argv), = calls
    assert exe == sys.executable and argv[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert argv[argv.index('--nproc-per-node') + 1] == '4' and argv[argv.index('--master-addr') + 1] == '127.0.0.1'
    assert 0 < int(argv[argv.index('--master-port') + 1]) < 65536
    assert argv[-5:] == [os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '7']
    monkeypatch.setenv('WORLD_SIZE', '4')
    bench_crnn.self_spawn(4, os.path.join(ROOT, 'bench.py'))
    assert len(calls) == 1                                            # already a rank of a launched job


def test_feature_files_dispatch_on_extension_and_one_name_per_clip(tmp_path):
    from salsa_amd import io as sio
    a = np.arange(12, dtype=np.float32).reshape(1, 3, 4)
    np.savez(tmp_path / 'clip_b.npz', feature=a)
    np.savez(tmp_path / 'clip_a.npz', feature=a + 1)
    (tmp_path / 'clip_a.h5').write_bytes(b'not really hdf5')              # both containers of one clip present
    (tmp_path / 'notes.txt').write_text('x')
    names = sio.feature_files(str(tmp_path))
    assert [os.path.splitext(n)[0] for n in names] == ['clip_a', 'clip_b']            # one entry per clip, sorted
    if not sio.HAVE_HDF5:
        assert names == ['clip_a.npz', 'clip_b.npz']
    else:                                                                  # an HDF5 library is reachable: the .h5 name is the clip's entry ...
        assert names == ['clip_a.h5', 'clip_b.npz']
        assert np.array_equal(sio.load_arrays(str(tmp_path / 'clip_a.h5'))['feature'], a + 1)   # ... and a broken .h5 falls back to its twin
    got = sio.load_arrays(str(tmp_path / 'clip_b.npz'))                  # an .npz name is read as .npz whatever is installed
    assert np.array_equal(got['feature'], a)
    assert np.array_equal(sio.load_arrays(str(tmp_path / 'clip_b.h5'))['feature'], a)  # .h5 name, only the twin exists


def test_tracker_countdown_equals_three_step_above_history():
    """The tracker kernel keeps no per-bin countdown (round 3): 'countdown < 1 before this step's decrement' -- the slow-rise
    test of dataset/salsa_feature_extraction.py:68-69, with the countdown starting at 3, reset to 3 by every frame that is not
    `above` and decremented by every frame that is (:30, :67-79) -- holds exactly when the three previous frames were all
    `above`.  Simulated both ways on random and adversarial `above` sequences."""
    rng = np.random.RandomState(0)
    seqs = [rng.rand(4000) < p for p in (0.05, 0.3, 0.5, 0.8, 0.97)]
    seqs += [np.ones(50, bool), np.zeros(50, bool), np.array(([True] * 3 + [False]) * 20), np.array(([True] * 4 + [False]) * 20)]
    for above in seqs:
        countdown, h = 3, [False, False, False]
        for a in above:
            slow_ref = countdown < 1                       # evaluated BEFORE the update, like tracker_step
            slow_new = h[0] and h[1] and h[2]
            assert slow_ref == slow_new
            countdown = countdown - 1 if a else 3
            h = [bool(a), h[0], h[1]]


def test_three_instruction_division_by_three_is_ieee():
    """salsa_kernels.hip div3_exact (the tracker producers' x / 3): q = RN(x y), r = fma(-3, q, x), RN(q + r y) with y = RN(1/3)
    against IEEE division in exact rational arithmetic (Markstein's theorem; the long run is tools/probes/div3_check.py)."""
    from fractions import Fraction as Fr
    y = 1.0 / 3.0

    def div3(a):
        q = a * y
        r = float(Fr(a) - 3 * Fr(q))                      # one FMA = one correctly rounded operation
        return float(Fr(q) + Fr(r) * Fr(y))

    rng = np.random.RandomState(1)
    vals = [0.0, 5e-324, 1e-310, 2.2250738585072014e-308, 1.0, 3.0, 1e300]
    vals += list(np.ldexp(rng.rand(3000) + 1.0, rng.randint(-200, 200, 3000)))
    vals += list((rng.randint(1, 2 ** 52, 2000).astype(np.float64)) * 3.0)
    for a in vals:
        a = float(a)
        assert div3(a) == a / 3.0, a.hex()


def test_bench_termination_hook_fires_while_the_main_thread_is_blocked(tmp_path):
    """bench._arm_termination: SIGTERM (what torch.distributed.run sends the surviving ranks when one rank fails) must reach the
    bail-out callback even while the main thread sits in a blocking native wait -- a Python signal handler alone would never
    run there.  The callback prints the partial line and leaves with exit code 3."""
    import signal
    import subprocess
    import sys
    import time
    from conftest import ROOT
    prog = tmp_path / 'blocked.py'
    prog.write_text(
        "import os, sys, threading\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "def cb():\n"
        "    sys.stdout.write('{\"status\": \"partial\"}\\n'); sys.stdout.flush(); os._exit(3)\n"
        "bench._arm_termination(cb)\n"
        "print('armed', flush=True)\n"
        "threading.Event().wait()\n" % ROOT)
    p = subprocess.Popen([sys.executable, str(prog)], stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == 'armed'
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=30)
    assert p.returncode == 3 and out.strip().splitlines()[-1] == '{"status": "partial"}'


def test_power_sampler_degrades_to_none_without_hwmon_files(tmp_path):
    """bench.py's power garnish (bench_crnn.PowerSampler) must never break a bench run: without the amdgpu hwmon files (this container)
    it reports None; its helper script is valid Python and stops when its parent is gone."""
    import sys
    import time
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench_crnn
    ps = bench_crnn.PowerSampler()
    try:
        assert ps.stats(time.time() - 1.0, time.time()) is None or isinstance(ps.stats(time.time() - 1.0, time.time()), dict)
    finally:
        ps.close()
    compile(bench_crnn.PowerSampler._SRC, 'power_helper', 'exec')
    # the parser on a synthetic log: two cards, the second one busy
    ps2 = bench_crnn.PowerSampler.__new__(bench_crnn.PowerSampler)
    ps2.proc, ps2.slot = object(), ''
    ps2.path = str(tmp_path / 'log.txt')
    t0 = 1000.0
    with open(ps2.path, 'w') as f:
        f.write('# 1400000000 1400000000\n')
        for i in range(10):
            f.write('%.3f 250000000 150000000 1390000000 2000000000\n' % (t0 + 0.01 * i))
    st = ps2.stats(t0, t0 + 1.0)
    assert st['mean_w'] == 1390.0 and st['sclk_mhz_mean'] == 2000.0 and st['cap_w'] == 1400.0 and st['samples'] == 10


def test_clip_readers_shape_from_header_and_read_into_place(tmp_path):
    """the file pipeline's reader primitives (round 6): audio_shape reads only the header; load_audio_into fills a caller's
    (pinned) slot -- straight from the file for a float32 .npy in the slot's layout, through load_audio otherwise -- and the raw
    .npy feature twin round-trips (one header + one write of the buffer)."""
    from scipy.io import wavfile
    from salsa_amd import io as sio
    rng = np.random.RandomState(2)
    a = (rng.uniform(-1, 1, (4, 5000)) * 0.5).astype(np.float32)
    np.save(tmp_path / 'a.npy', a)
    np.save(tmp_path / 'a64.npy', a.astype(np.float64))
    wavfile.write(tmp_path / 'c.wav', 24000, a.T)
    for name in ('a.npy', 'a64.npy', 'c.wav'):
        assert sio.audio_shape(str(tmp_path / name), 24000) == (4, 5000)
        dst = np.zeros((4, 5000), np.float32)
        sio.load_audio_into(str(tmp_path / name), 24000, dst)
        assert np.array_equal(dst, a), name
        dst_t = np.zeros((5000, 4), np.float32)
        sio.load_audio_into(str(tmp_path / name), 24000, dst_t, planar=False)
        assert np.array_equal(dst_t, a.T), name
    assert sio.audio_shape(str(tmp_path / 'c.wav'), 48000) == (4, 10000)      # another rate: the length librosa.load's resampling will give
    (tmp_path / 'short.npy').write_bytes((tmp_path / 'a.npy').read_bytes()[:-100])
    with pytest.raises(IOError):
        sio.load_audio_into(str(tmp_path / 'short.npy'), 24000, np.zeros((4, 5000), np.float32))
    f = rng.randn(7, 50, 200).astype(np.float32)
    w = sio.save_arrays(str(tmp_path / 'x.h5'), feature=f)
    if not sio.HAVE_HDF5:
        assert w.endswith('x.npy') and np.array_equal(np.load(w), f)
    assert np.array_equal(sio.load_arrays(str(tmp_path / 'x.h5'))['feature'], f) and np.array_equal(sio.load_arrays(w)['feature'], f)
    w2 = sio.save_arrays(str(tmp_path / 'foa_feature_scaler.h5'), mean=f[:4, :1], std=f[:4, :1] + 1)
    got = sio.load_arrays(str(tmp_path / 'foa_feature_scaler.h5'))
    assert set(got) == {'mean', 'std'} and (sio.HAVE_HDF5 or w2.endswith('.npz'))


CONDA_PY = '/opt/conda/bin/python3.9'     # the ROCm image's second interpreter: Python 3.9 with h5py 3.3.0 (the reference's reader library)


def _conda_h5py():
    import subprocess
    if not os.path.exists(CONDA_PY):
        return False
    return subprocess.run([CONDA_PY, '-c', 'import h5py, numpy'], capture_output=True).returncode == 0


def test_hdf5_files_through_libhdf5_round_trip_and_threads(tmp_path):
    """salsa_amd/_hdf5.py (the HDF5 C library through ctypes; round 6: the feature-file format row was "HDF5 branch never executed"):
    feature and scaler files round-trip, the bulk payload written AROUND the library at H5Dget_offset is what the library reads back,
    eight threads write at once, a float64 dataset written by someone else reads as float64, junk is rejected."""
    from concurrent.futures import ThreadPoolExecutor
    from salsa_amd import _hdf5, io as sio
    if not _hdf5.available():
        pytest.skip('no libhdf5 on this machine')
    rng = np.random.RandomState(3)
    feats = [rng.randn(7, 301, 200).astype(np.float32) for _ in range(4)]            # 1.7 MB each: the bulk path (>= 1 MB)
    assert feats[0].nbytes >= _hdf5.BULK_BYTES
    with ThreadPoolExecutor(8) as pool:
        list(pool.map(lambda i: _hdf5.write(str(tmp_path / ('c%02d.h5' % i)), feature=feats[i % 4]), range(24)))
    for i in range(24):
        got = _hdf5.read(str(tmp_path / ('c%02d.h5' % i)))
        assert list(got) == ['feature'] and got['feature'].dtype == np.float32 and np.array_equal(got['feature'], feats[i % 4])
    small = rng.randn(4, 1, 200).astype(np.float32)                                     # the library's own H5Dwrite path
    _hdf5.write(str(tmp_path / 's.h5'), mean=small, std=small + 1)
    got = _hdf5.read(str(tmp_path / 's.h5'))
    assert set(got) == {'mean', 'std'} and np.array_equal(got['std'], small + 1)
    assert open(tmp_path / 's.h5', 'rb').read(8) == b'\x89HDF\r\n\x1a\n'              # the HDF5 signature
    (tmp_path / 'junk.h5').write_bytes(b'not hdf5 at all')
    with pytest.raises(IOError):
        _hdf5.read(str(tmp_path / 'junk.h5'))
    if not sio.HAVE_H5PY:                                                               # io.py routes .h5 through it
        w = sio.save_arrays(str(tmp_path / 'via_io.h5'), feature=feats[0])
        assert w.endswith('via_io.h5') and np.array_equal(sio.load_arrays(w)['feature'], feats[0])
        assert sio.feature_files(str(tmp_path))[0] == 'c00.h5'


@pytest.mark.skipif(not _conda_h5py(), reason='needs the image\'s conda Python with h5py')
def test_feature_and_scaler_files_are_read_by_h5py_as_the_reference_reads_them(tmp_path):
    """The on-disk contract (SURVEY section 8b ii, 8f row 3): the reference's Database opens `<clip>.h5` with h5py and takes
    hf['feature'][:] (dataset/database.py:193-195), the scaler with hf['mean'][:] / hf['std'][:] (:91-94).  Our files -- written by
    salsa_amd.io.save_arrays through libhdf5 -- are read with h5py 3.3.0 in the image's conda interpreter: names, shapes, float32
    dtype and every byte; and a file h5py writes the way the reference does (:380-382) is read by ours."""
    import hashlib
    import json
    import subprocess
    from salsa_amd import _hdf5, io as sio
    if not _hdf5.available() and not sio.HAVE_H5PY:
        pytest.skip('no HDF5 library for this interpreter')
    rng = np.random.RandomState(4)
    f = rng.randn(7, 801, 200).astype(np.float32)
    mean, std = rng.randn(4, 1, 200).astype(np.float32), rng.rand(4, 1, 200).astype(np.float32) + 0.5
    pf = sio.save_arrays(str(tmp_path / 'fold1_room1_mix001.h5'), feature=f)
    ps = sio.save_arrays(str(tmp_path / 'foa_feature_scaler.h5'), mean=mean, std=std)
    assert pf.endswith('.h5') and ps.endswith('.h5')
    code = ("import h5py, numpy as np, sys, json, hashlib\n"
            "out = {}\n"
            "with h5py.File(sys.argv[1], 'r') as hf:\n"
            "    a = hf['feature'][:]\n"
            "    out['feature'] = [list(a.shape), str(a.dtype), hashlib.sha256(a.tobytes()).hexdigest(), sorted(hf.keys())]\n"
            "with h5py.File(sys.argv[2], 'r') as hf:\n"
            "    out['mean'] = [list(hf['mean'][:].shape), str(hf['mean'].dtype), hashlib.sha256(hf['mean'][:].tobytes()).hexdigest()]\n"
            "    out['std'] = [list(hf['std'][:].shape), str(hf['std'].dtype), hashlib.sha256(hf['std'][:].tobytes()).hexdigest()]\n"
            "x = (np.arange(7 * 5 * 200, dtype=np.float64).reshape(7, 5, 200) / 7.0)\n"
            "with h5py.File(sys.argv[3], 'w') as hf:\n"
            "    hf.create_dataset('feature', data=x, dtype=np.float32)\n"
            "print(json.dumps(out))\n")
    r = subprocess.run([CONDA_PY, '-c', code, pf, ps, str(tmp_path / 'from_h5py.h5')], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()    # noqa: E731
    assert out['feature'] == [[7, 801, 200], 'float32', sha(f), ['feature']]
    assert out['mean'] == [[4, 1, 200], 'float32', sha(mean)] and out['std'] == [[4, 1, 200], 'float32', sha(std)]
    back = sio.load_arrays(str(tmp_path / 'from_h5py.h5'))['feature']
    assert back.dtype == np.float32 and np.array_equal(back, (np.arange(7 * 5 * 200, dtype=np.float64).reshape(7, 5, 200) / 7.0).astype(np.float32))
