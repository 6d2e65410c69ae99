"""Drop-in Python surface of the reference's full-SALSA module (dataset/salsa_feature_extraction.py): same function
names, arguments, defaults, exceptions and file layout, with the arithmetic running in libsalsa_hip.so on an MI355X.

  reference                                             here
  extract_normalized_eigenvector(X, ...)   :17-129      extract_normalized_eigenvector  (numpy in / numpy out)
  MagStftExtractor(...).W / .extract       :132-201     MagStftExtractor
  compute_scaler(feature_dir, audio_format) :204-262    compute_scaler
  extract_features(data_config, ...)       :265-391     extract_features  (+ ``python -m salsa_amd.features --flag=...``)

There is no CPU path: without a GPU / without the built library these functions raise.
"""
import os
import shutil
import sys
from timeit import default_timer as timer

import numpy as np
import yaml

from . import io as sio

import logging  # noqa: E402
_log = logging.getLogger('salsa_amd.features')


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('salsa_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback')
    return torch


def extract_normalized_eigenvector(X, condition_number: float = 5.0, n_hopframes: int = 3, is_tracking: bool = True,
                                   audio_format: str = 'foa', fs: int = None, n_fft: int = None,
                                   lower_bin: int = None):
    """X <np.ndarray (n_bins, n_frames, n_chans=4)> clipped spectrogram -> (3, n_bins, n_frames) float64.
    Mirrors dataset/salsa_feature_extraction.py:17 (values are taken at complex64 precision, which is what the
    reference's STFT holds)."""
    if audio_format not in ('foa', 'mic'):
        raise ValueError('audio format {} is not valid'.format(audio_format))
    torch = _torch()
    from .extractor import SalsaExtractor
    X = np.asarray(X)
    assert X.ndim == 3 and X.shape[2] == 4, 'X must be (n_bins, n_frames, 4)'
    # (the plan's own DOA band is irrelevant here: the band is whatever rows X holds)
    ex = SalsaExtractor(fs=fs or 24000, n_fft=n_fft or 512, hop_len=300, fmin_doa=0, fmax_doa=(fs or 24000) // 8,
                        cond_num=condition_number, n_hopframes=n_hopframes, is_tracking=is_tracking,
                        audio_format=audio_format)
    Xd = torch.from_numpy(np.ascontiguousarray(X.astype(np.complex64))[None]).cuda()
    out = ex.eigvec(Xd, lower_bin=int(lower_bin or 0))
    return out[0].cpu().numpy()


class MagStftExtractor:
    """Log-linear spectrograms (n_channels, n_timesteps, 200|100|n_fft/2).  Mirrors :132-201."""

    def __init__(self, n_fft: int, hop_length: int, win_length: int = None, window: str = 'hann',
                 is_compress_high_freq: bool = True):
        from .extractor import compress_matrix
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.window = window
        self.win_length = self.n_fft if win_length is None else win_length
        assert self.win_length <= self.n_fft, 'Windown length is greater than nfft!'
        assert n_fft == 512 or n_fft == 256, 'nfft is not 512 or 256'
        if window != 'hann':
            raise ValueError('only the hann window of the reference configs is implemented')
        self.is_compress_high_freq = is_compress_high_freq
        self.W = compress_matrix(n_fft, is_compress_high_freq)
        self._ex = None

    def extract(self, audio_input: np.ndarray) -> np.ndarray:
        torch = _torch()
        from .extractor import SalsaExtractor
        if self._ex is None:
            self._ex = SalsaExtractor(n_fft=self.n_fft, hop_len=self.hop_length, win_len=self.win_length,
                                      is_compress_high_freq=self.is_compress_high_freq)
        a = np.ascontiguousarray(audio_input, np.float32)
        n_ch = a.shape[0]
        outs = []
        for c0 in range(0, n_ch, 4):                       # the kernel works on groups of 4 channels
            blk = a[c0:c0 + 4]
            if blk.shape[0] < 4:
                blk = np.concatenate([blk, np.zeros((4 - blk.shape[0], a.shape[1]), np.float32)])
            o = self._ex.logspec(torch.from_numpy(np.ascontiguousarray(blk[None])).cuda())[0].cpu().numpy()
            outs.append(o[:min(4, n_ch - c0)])
        return np.concatenate(outs, axis=0)


def compute_scaler(feature_dir: str, audio_format: str) -> None:
    """Mean / std of the 4 spectrogram channels over ALL files of <audio_format>_dev -> <audio_format>_feature_scaler
    (datasets 'mean', 'std', shape (4,1,F) float32).  Mirrors :204-262 (StandardScaler.partial_fit: population std)."""
    _log.info('scaler: streaming mean / std over %s_dev', audio_format)
    start_time = timer()
    train_feature_dir = os.path.join(feature_dir, audio_format + '_dev')
    feature_fn_list = sio.feature_files(train_feature_dir)
    afeature = sio.load_arrays(os.path.join(train_feature_dir, feature_fn_list[0]))['feature']
    n_channels = afeature.shape[0]
    assert n_channels == 7, 'only support n_channels = 7, got {}'.format(n_channels)
    n_feature_channels = 4
    shift = afeature[:n_feature_channels].astype(np.float64).mean(axis=1)     # conditioning of the one-pass variance
    n, s, ss = 0, np.zeros_like(shift), np.zeros_like(shift)
    for feature_fn in feature_fn_list:
        afeature = sio.load_arrays(os.path.join(train_feature_dir, feature_fn))['feature']
        d = afeature[:n_feature_channels].astype(np.float64) - shift[:, None, :]
        s += d.sum(axis=1)
        ss += (d * d).sum(axis=1)
        n += afeature.shape[1]
    feature_mean = (shift + s / n)[:, None, :]
    feature_std = np.sqrt(np.maximum(ss / n - (s / n) ** 2, 0.0))[:, None, :]
    scaler_path = os.path.join(feature_dir, audio_format + '_feature_scaler.h5')
    written = sio.save_arrays(scaler_path, mean=feature_mean, std=feature_std)
    _log.info('scaler: %d files x %s -> %s', len(feature_fn_list), afeature.shape, written)
    _log.info('scaler: %.3f s', timer() - start_time)


def write_scaler_from_sums(feature_dir: str, audio_format: str, sums, n_frames: int) -> str:
    """compute_scaler's result from statistics accumulated on the device while the dev split was extracted: mean = sum / n,
    std = sqrt(sum of squares / n - mean^2) (population, like StandardScaler.var_), both (4,1,F) float32 -> the scaler file."""
    h = sums.detach().cpu().numpy()
    mean = h[0] / n_frames
    std = np.sqrt(np.maximum(h[1] / n_frames - mean * mean, 0.0))
    scaler_path = os.path.join(feature_dir, audio_format + '_feature_scaler.h5')
    written = sio.save_arrays(scaler_path, mean=mean[:, None, :].astype(np.float32), std=std[:, None, :].astype(np.float32))
    _log.info('scaler: from the device statistics of %d frames -> %s', n_frames, written)
    return written


def _parse(data_config):
    with open(data_config, 'r') as stream:
        cfg = yaml.safe_load(stream)
    d = cfg['data']
    return cfg, d['format'], d['fs'], d['n_fft'], d['hop_len'], d['win_len'], d['fmin_doa'], d['fmax_doa']


def feature_name(audio_fn: str) -> str:
    """The reference names the feature file ``audio_fn.replace('wav', 'h5')`` (:379, every occurrence)."""
    out = audio_fn.replace('wav', 'h5')
    return out if out != audio_fn else os.path.splitext(audio_fn)[0] + '.h5'


class _FilePipeline:
    """Files in, feature files out, every stage overlapped (round-3 review: the harness ran read -> pageable copy -> extract ->
    copy back -> write serially).  ``depth`` slots, each with pinned host staging in both directions, device buffers, its own
    plan and its own copy-in / compute / copy-out streams:

      reader thread   reads each clip's HEADER, assigns it a place in a free slot's pinned input (one batch = clips of one length, in
                      file order) and hands the decode to a pool of reader threads, which read STRAIGHT INTO that place
      caller's thread queues host->device copy, the three kernels, device->host copy on the slot's streams (nothing blocks)
      writer thread   waits for the slot's event, writes the batch's feature files from the pinned output with a small pool of
                      file-writer threads, hands the slot back

    so the decode of batch i+2, the PCIe transfers and kernels of batch i+1 and the file writes of batch i run at the same time.
    An exception in any thread stops the run and is re-raised by ``run``."""

    def __init__(self, ex, depth=4, writers=None, readers=None):
        torch = _torch()
        from .extractor import SalsaExtractor
        try:
            ncpu = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            ncpu = os.cpu_count() or 1
        # round 6: ONE reader thread decoded and copied every clip (11 ms per 60-s clip = 5.3 k audio-s/s, the harness' whole
        # bound); readers and writers are pools now, sized to the CPUs this process may use
        self.readers = int(readers or max(1, min(8, ncpu // 2)))
        writers = int(writers or max(1, min(8, ncpu // 2)))
        self.torch, self.depth, self.writers = torch, depth, writers
        self.exs = [ex] + [SalsaExtractor(**ex.kwargs()) for _ in range(depth - 1)]
        dev = ex.device
        self.slots = [dict(idx=i, ex=self.exs[i], n=None, s_in=torch.cuda.Stream(device=dev), s_run=torch.cuda.Stream(device=dev),
                           s_out=torch.cuda.Stream(device=dev)) for i in range(depth)]
        self.error = None
        import threading
        self.lock = threading.Lock()                 # (the pipeline of a parameter set is shared by every call in the process: one run at a time)

    def _buffers(self, sl, batch, n_samples):
        """Views of the slot's staging for (batch, n_samples); the flat pinned / device arrays behind them only ever GROW (a tree of
        ragged clips used to re-pin its slots at every new length)."""
        if sl['n'] == (batch, n_samples):
            return
        torch, ex = self.torch, sl['ex']
        shape = (batch, 4, n_samples) if ex.audio_layout == 'planar' else (batch, n_samples, 4)
        oshape = (batch,) + tuple(ex.output_shape(n_samples))
        n_in, n_out = int(np.prod(shape)), int(np.prod(oshape))
        if sl.get('cap_in', 0) < n_in:
            sl['f_h_in'] = torch.empty(n_in, dtype=torch.float32, pin_memory=True)
            sl['f_d_in'] = torch.empty(n_in, dtype=torch.float32, device=ex.device)
            sl['cap_in'] = n_in
        if sl.get('cap_out', 0) < n_out:
            sl['f_d_out'] = torch.empty(n_out, dtype=torch.float32, device=ex.device)
            sl['f_h_out'] = torch.empty(n_out, dtype=torch.float32, pin_memory=True)
            sl['cap_out'] = n_out
        sl['h_in'], sl['d_in'] = sl['f_h_in'][:n_in].view(shape), sl['f_d_in'][:n_in].view(shape)
        sl['d_out'], sl['h_out'] = sl['f_d_out'][:n_out].view(oshape), sl['f_h_out'][:n_out].view(oshape)
        sl['n'] = (batch, n_samples)

    def _fail(self, e):
        """Record the first failure of any thread and tell every other thread to stop."""
        self.error = self.error or e
        self.stop.set()

    def _take(self, q):
        """q.get() that gives up (returns _STOP) once any thread has failed: no stage may block forever on a queue whose
        producer is gone (round-4 advice: a failed extract() or writer left the reader in free.get() and run() in join())."""
        import queue
        while True:
            try:
                return q.get(timeout=0.05)
            except queue.Empty:
                if self.stop.is_set():
                    return _STOP

    def run(self, todo, audio_dir, feature_dir, fs, batch_size, stats=None, scaler=None):
        """todo: [(count, file name)] in order.  Writes <feature_dir>/<feature_name(fn)> for every clip.
        scaler: a dict that receives 'sums' (float64 [2][4][F] device tensor: sum / sum of squares of the spectrogram channels over
        every frame extracted) and 'n' (frames) -- compute_scaler's statistics (:204-262) taken from the features while they are still
        on the device (salsa_scaler_accumulate on the slot's own stream into the slot's own sums, added at the end), instead of reading
        every feature file back."""
        import queue
        import threading
        import time
        from concurrent.futures import ThreadPoolExecutor
        torch = self.torch
        self.free, self.filled, self.inflight = queue.Queue(), queue.Queue(), queue.Queue()
        self.stop, self.error = threading.Event(), None
        # the clones were built from the constructor arguments only: carry over the plan state attached since (round-4
        # advice: with a scaler on the caller's extractor, two of every three batches came out un-normalised)
        for other in self.exs[1:]:
            self.exs[0].copy_plan_state_to(other)
        for sl in self.slots:
            self.free.put(sl)
        t_read, t_write = [0.0], [0.0]
        t_wait = {'main_for_reader': 0.0, 'writer_for_device': 0.0, 'reader_for_slot': 0.0, 'wall': time.perf_counter()}

        cap = max(1, min(batch_size, SLOT_CLIPS, len(todo)))   # clips per slot (batch_size is the caller's upper bound per device call)

        def reader():
            pool = ThreadPoolExecutor(self.readers)
            pending = []                                    # closed slots whose clips are still being read, in closing order

            def read_one(path, dst, planar):
                t0 = time.perf_counter()
                sio.load_audio_into(path, fs, dst, planar, device=self.exs[0].device)
                return time.perf_counter() - t0

            def read_raw(path, offset, dst_bytes):
                t0 = time.perf_counter()
                sio.read_raw_into(path, offset, dst_bytes)
                return time.perf_counter() - t0
            planar_plan = self.exs[0].audio_layout == 'planar'

            def publish(keep):
                """hand the oldest closed slots to the device stage once their reads have finished (never more than `keep` waiting)"""
                while pending and (len(pending) > keep or all(f.done() for f in pending[0][1])):
                    sl, futs = pending.pop(0)
                    for f in futs:
                        t_read[0] += f.result()             # (re-raises a failed read here, in the reader thread)
                    self.filled.put(sl)
            try:
                torch.cuda.set_device(self.exs[0].device)   # (a new thread starts on device 0)
                open_ = {}                                  # n_samples -> (slot, [(count, fn)], [futures]) being filled

                def close(n):
                    sl, items, futs = open_.pop(n)
                    sl['items'] = items
                    pending.append((sl, futs))
                    publish(1)                              # the next slot's reads are issued while this one's finish
                for count, fn in todo:
                    if self.stop.is_set():
                        return
                    path = os.path.join(audio_dir, fn)
                    # a plain PCM / float WAV at the configured rate goes to the device AS IT IS ON DISK (round 6: decoding 16-bit clips
                    # on the host -- astype, scale, transpose -- was the harness' bound on the dataset's own format: 14 k audio-s/s)
                    lay = sio.wav_pcm_layout(path, fs) if (RAW_PCM and planar_plan) else None
                    n_ch, n = (lay[1], lay[2]) if lay else sio.audio_shape(path, fs)     # header only
                    assert n_ch == 4, '{}: expected a 4-channel clip'.format(fn)
                    if n not in open_:
                        if len(open_) >= self.depth - 2:    # never hold every slot half-filled: flush the fullest bucket
                            close(max(open_, key=lambda k: len(open_[k][1])))
                        t0w = time.perf_counter()
                        sl = self._take(self.free)
                        t_wait['reader_for_slot'] += time.perf_counter() - t0w
                        if sl is _STOP:
                            return
                        self._buffers(sl, cap, n)
                        sl['raw'] = [0] * cap               # per clip: 0 = float32 samples in the slot, else its SALSA_PCM_* code
                        open_[n] = (sl, [], [])
                    sl, items, futs = open_[n]
                    place = sl['h_in'].numpy()[len(items)]
                    if lay:
                        sl['raw'][len(items)] = lay[0]
                        nbytes = n * 4 * {1: 2, 2: 4, 3: 1, 4: 4}[lay[0]]
                        futs.append(pool.submit(read_raw, path, lay[3], place.reshape(-1).view(np.uint8)[:nbytes]))
                    else:
                        futs.append(pool.submit(read_one, path, place, planar_plan))
                    items.append((count, fn))
                    if len(items) == cap:
                        close(n)
                for n in sorted(open_, key=lambda k: open_[k][1][0][0]):
                    close(n)
                publish(0)
            except BaseException as e:                      # noqa: BLE001 - re-raised by run()
                self._fail(e)
            finally:
                pool.shutdown(wait=True)
                self.filled.put(None)

        def writer():
            pool = ThreadPoolExecutor(self.writers)
            try:
                torch.cuda.set_device(self.exs[0].device)
                while not self.stop.is_set():
                    sl = self._take(self.inflight)
                    if sl is None or sl is _STOP:
                        return
                    t0w = time.perf_counter()
                    sl['done'].synchronize()
                    t_wait['writer_for_device'] += time.perf_counter() - t0w
                    t0 = time.perf_counter()
                    feats = sl['h_out'].numpy()
                    def save(k):
                        count, fn = sl['items'][k]
                        sio.save_arrays(os.path.join(feature_dir, feature_name(fn)), feature=feats[k])
                        _log.debug('clip %d %s -> %s', count, fn, feats[k].shape)
                    list(pool.map(save, range(len(sl['items']))))
                    t_write[0] += time.perf_counter() - t0
                    self.free.put(sl)
            except BaseException as e:                      # noqa: BLE001
                self._fail(e)
            finally:
                pool.shutdown(wait=True)

        tr, tw = threading.Thread(target=reader, daemon=True), threading.Thread(target=writer, daemon=True)
        tr.start(), tw.start()
        n_batches = 0
        try:
            while not self.stop.is_set():                   # (a failed stage: launch nothing more, not even what is already queued)
                t0w = time.perf_counter()
                sl = self._take(self.filled)
                t_wait['main_for_reader'] += time.perf_counter() - t0w
                if sl is None or sl is _STOP:
                    break
                b = len(sl['items'])
                with torch.cuda.stream(sl['s_in']):
                    sl['d_in'][:b].copy_(sl['h_in'][:b], non_blocking=True)
                sl['s_run'].wait_stream(sl['s_in'])
                with torch.cuda.stream(sl['s_run']):
                    d_audio = sl['d_in']
                    if any(sl['raw'][:b]):                  # raw file bytes in the slot: convert + de-interleave on the device
                        if sl.get('f_d_conv') is None or sl['f_d_conv'].numel() < sl['cap_in']:
                            sl['f_d_conv'] = torch.empty(sl['cap_in'], dtype=torch.float32, device=sl['ex'].device)
                        d_audio = sl['f_d_conv'][:sl['d_in'].numel()].view(sl['d_in'].shape)
                        for k in range(b):
                            if sl['raw'][k]:
                                pcm_to_planar(sl['d_in'][k], sl['raw'][k], d_audio[k])
                            else:
                                d_audio[k].copy_(sl['d_in'][k], non_blocking=True)
                    sl['ex'].extract(d_audio[:b], out=sl['d_out'][:b])
                    if scaler is not None:
                        from .extractor import scaler_accumulate
                        F_ = sl['d_out'].shape[3]
                        if sl.get('sums') is None or sl['sums'].shape[2] != F_:
                            sl['sums'] = torch.zeros((2, 4, F_), dtype=torch.float64, device=sl['ex'].device)
                        scaler_accumulate(sl['d_out'][:b], sl['sums'])
                        scaler['n'] = scaler.get('n', 0) + b * sl['d_out'].shape[2]
                sl['s_out'].wait_stream(sl['s_run'])
                with torch.cuda.stream(sl['s_out']):
                    sl['h_out'][:b].copy_(sl['d_out'][:b], non_blocking=True)
                    sl['done'] = torch.cuda.Event()
                    sl['done'].record(sl['s_out'])
                self.inflight.put(sl)
                n_batches += 1
        except BaseException as e:                          # noqa: BLE001 - a failed copy / extract in THIS thread stops the others too
            self._fail(e)
        finally:
            self.inflight.put(None)                         # (the writer finishes what is in flight, or leaves at once after a failure)
            tw.join()
            tr.join()
            if self.error:                                  # in-flight device work may still touch the pinned slots
                torch.cuda.synchronize(self.exs[0].device)
        if self.error:
            e, self.error = self.error, None
            raise e
        if scaler is not None:
            torch.cuda.synchronize(self.exs[0].device)
            for sl in self.slots:
                if sl.get('sums') is not None:
                    scaler['sums'] = sl['sums'].clone() if scaler.get('sums') is None else scaler['sums'] + sl['sums']
                    sl['sums'].zero_()
        if stats is not None:
            t_wait['wall'] = time.perf_counter() - t_wait['wall']
            stats.update(batches=n_batches, read_s=t_read[0], write_s=t_write[0], **t_wait)


def pcm_to_planar(d_raw, code, d_out):
    """d_raw: a clip's place in a device slot holding the WAV file's data chunk ([N][4] samples of SALSA_PCM_* ``code``) -> d_out float32
    [4][N], on the current stream (salsa_pcm_to_planar)."""
    import ctypes as C
    from . import _lib
    torch = _torch()
    rc = _lib.load().salsa_pcm_to_planar(C.c_void_p(d_raw.data_ptr()), int(code), int(d_out.shape[0]), int(d_out.shape[1]),
                                         C.c_void_p(d_out.data_ptr()), C.c_void_p(torch.cuda.current_stream(d_out.device).cuda_stream))
    if rc != 0:
        raise RuntimeError('salsa_pcm_to_planar failed (%d): %s' % (rc, _lib.last_error()))


_STOP = object()
_PIPELINES = {}


def _pipeline_for(ex):
    """The file pipeline of extractors with these parameters on this device, built once per process: its plans, streams and --
    above all -- its pinned host slots outlive an extract_features() call (round 6: every call built a new extractor and with it
    a new pipeline, and pinning 6 GB of slots cost more than extracting a 64-clip tree).  The caller's extractor becomes slot 0's
    plan holder; post-construction state (scaler, schedule) is re-applied to every slot by run()."""
    key = (str(ex.device),) + tuple(sorted((k, str(v)) for k, v in ex.kwargs().items()))
    pipe = _PIPELINES.get(key)
    if pipe is None:
        pipe = _PIPELINES[key] = _FilePipeline(ex)
    else:
        pipe.exs[0] = ex                                    # state source for copy_plan_state_to (same parameters by construction)
        pipe.slots[0]['ex'] = ex
    return pipe


def release_file_pipelines():
    """Drop the cached pipelines (pinned host slots, device buffers, plans)."""
    _PIPELINES.clear()


SLOT_CLIPS = 8     # clips per pipeline slot: small slots overlap read / copy / extract / copy / write sooner and pin 4x less host memory
                   # than 32-clip ones; the device does 8 x 60-s clips in ~0.45 ms, far below a slot's 8 ms of PCIe time
USE_FILE_PIPELINE = os.environ.get('SALSA_FILE_PIPELINE', '1') != '0'
RAW_PCM = os.environ.get('SALSA_RAW_PCM', '1') != '0'     # plain PCM WAV clips: upload the file's bytes, convert on the device (0: decode on the host)
FUSED_SCALER = os.environ.get('SALSA_FUSED_SCALER', '1') != '0'   # task='feature_scaler': the scaler from device statistics (0: re-read the files)


def _rmtree_parallel(path, threads: int = 8):
    """shutil.rmtree(path, ignore_errors=True) with the directory's own files unlinked from a few threads first: emptying last run's
    split folder (:344) is freeing 27 MB of page cache / tmpfs per clip, 2.5 ms each from one thread -- twice what extracting the
    clip takes (tools/probes/harness_profile.py)."""
    try:
        files = [e.path for e in os.scandir(path) if e.is_file(follow_symlinks=False)]
    except OSError:
        files = []
    if len(files) > 2 * threads:
        from concurrent.futures import ThreadPoolExecutor

        def rm(p):
            try:
                os.unlink(p)
            except OSError:
                pass
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(rm, files))
    shutil.rmtree(path, ignore_errors=True)


def _extract_split(ex, audio_dir, feature_dir, fs, batch_size, shard=None, clear=True, stats=None, scaler=None):
    """Extract the clips of one split directory, batching clips of equal length (one device round trip per batch, overlapped
    with its neighbours' and with the file reads / writes: _FilePipeline).
    shard = (rank, world): this process takes a contiguous range of the sorted file list (salsa_amd.distributed)."""
    torch = _torch()
    if clear:
        _rmtree_parallel(feature_dir)                       # the reference empties the split's folder first (:344)
    os.makedirs(feature_dir, exist_ok=True)
    audio_fn_list = sorted(os.listdir(audio_dir))
    todo = list(enumerate(audio_fn_list))
    if shard is not None:
        from .distributed import shard_range
        lo, hi = shard_range(len(todo), *shard)
        todo = todo[lo:hi]
    if USE_FILE_PIPELINE and todo:
        pipe = _pipeline_for(ex)
        with pipe.lock, torch.cuda.device(ex.device):
            pipe.run(todo, audio_dir, feature_dir, fs, batch_size, stats, scaler)
        return
    if scaler is not None:
        scaler['unavailable'] = True                       # (serial loop / empty split: compute_scaler reads the files, as the reference does)
    pending = {}                                            # n_samples -> [(count, fn, audio)]

    def flush(items):
        batch = np.stack([a for _, _, a in items])
        feats = ex.extract(torch.from_numpy(batch).cuda()).cpu().numpy()
        for (count, fn, _), f in zip(items, feats):
            sio.save_arrays(os.path.join(feature_dir, feature_name(fn)), feature=f)
            _log.debug('clip %d %s -> %s', count, fn, f.shape)

    for count, audio_fn in todo:
        audio_input = sio.load_audio(os.path.join(audio_dir, audio_fn), sr=fs, device=ex.device)
        assert audio_input.shape[0] == 4, '{}: expected a 4-channel clip'.format(audio_fn)
        lst = pending.setdefault(audio_input.shape[1], [])
        lst.append((count, audio_fn, audio_input))
        if len(lst) == batch_size:
            flush(lst)
            lst.clear()
    for lst in pending.values():
        if lst:
            flush(lst)


def extract_features(data_config: str = 'configs/tnsse2021_salsa_feature_config.yml',
                     cond_num: float = 5,
                     n_hopframes: int = 3,
                     is_tracking: bool = True,
                     is_compress_high_freq: bool = True,
                     task: str = 'feature_scaler',
                     batch_size: int = 32) -> None:
    """Extract salsa features (log-linear spectrogram + normalized eigenvector) for every clip of <format>_dev and
    <format>_eval, then the scaler.  Mirrors :265-391 (same directory naming, split iteration, sorted file order,
    (7,T,F) float32 'feature' per clip).  ``batch_size`` (clips per device call) is the only extra argument."""
    cfg, audio_format, fs, n_fft, hop_length, win_length, fmin_doa, fmax_doa = _parse(data_config)
    feature_type = 'salsa'
    fmax_doa = int(np.min((fmax_doa, fs // 2)))
    assert n_fft == 512 or n_fft == 256, 'only 256 or 512 fft is supported'
    feature_description = '{}fs_{}nfft_{}nhop_{}cond_{}fmaxdoa'.format(fs, n_fft, hop_length, int(cond_num), int(fmax_doa))
    if not is_tracking:
        feature_description = feature_description + '_notracking'
    if not is_compress_high_freq:
        feature_description = feature_description + '_nocompress'
    if audio_format == 'foa':
        splits = ['foa_dev', 'foa_eval']
    elif audio_format == 'mic':
        splits = ['mic_dev', 'mic_eval']
    else:
        raise ValueError('Unknown audio format {}'.format(audio_format))
    _log.info('feature directory name: %s', feature_description)
    if task in ['feature_scaler', 'feature']:
        from .extractor import SalsaExtractor
        ex = SalsaExtractor(fs=fs, n_fft=n_fft, hop_len=hop_length, win_len=win_length, fmin_doa=fmin_doa,
                            fmax_doa=fmax_doa, cond_num=cond_num, n_hopframes=n_hopframes, is_tracking=is_tracking,
                            is_compress_high_freq=is_compress_high_freq, audio_format=audio_format)
        scaler = {} if (task == 'feature_scaler' and FUSED_SCALER) else None
        for split in splits:
            _log.info('split %s: extracting on %s', split, ex.device)
            start_time = timer()
            audio_dir = os.path.join(cfg['data_dir'], split)
            feature_dir = os.path.join(cfg['feature_dir'], feature_type, audio_format, feature_description, split)
            # the scaler is over the files of <format>_dev (:214-215): its statistics are taken while that split is extracted
            _extract_split(ex, audio_dir, feature_dir, fs, batch_size, scaler=scaler if split.endswith('_dev') else None)
            _log.info('split %s: done in %.3f s', split, timer() - start_time)
        if scaler is not None and scaler.get('sums') is not None and not scaler.get('unavailable'):
            feature_dir = os.path.join(cfg['feature_dir'], feature_type, audio_format, feature_description)
            write_scaler_from_sums(feature_dir, audio_format, scaler['sums'], scaler['n'])
            return
    if task in ['feature_scaler', 'scaler']:
        feature_dir = os.path.join(cfg['feature_dir'], feature_type, audio_format, feature_description)
        compute_scaler(feature_dir=feature_dir, audio_format=audio_format)


def _cli(fn, argv):
    """python-fire style ``--name=value`` flags -> keyword arguments (the reference wraps the function in fire.Fire)."""
    kw = {}
    for a in argv:
        assert a.startswith('--') and '=' in a, 'use --name=value'
        k, v = a[2:].split('=', 1)
        kw[k] = yaml.safe_load(v)
    fn(**kw)


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO, format='%(message)s')      # progress lines on the console, as a CLI should
    _cli(extract_features, sys.argv[1:])
