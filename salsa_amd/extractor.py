"""Device-side front end: a plan over libsalsa_hip.so operating on torch CUDA(HIP) tensors.

torch is plumbing only (device memory, streams): every FLOP of the feature path runs in the hand-written HIP kernels
of salsa_amd/csrc/salsa_kernels.hip behind the C ABI of include/salsa_hip.h."""
import contextlib
import ctypes as C

import numpy as np
import torch

from . import _lib


def _raise(rc):
    msg = _lib.last_error()
    if rc == _lib.E_NFFT:
        raise AssertionError(msg)            # the reference asserts n_fft in (256, 512)
    if rc == _lib.E_FORMAT:
        if 'only for MIC' in msg:
            raise AssertionError(msg)        # salsa_lite_feature_extraction.py:72
        raise ValueError(msg)                # salsa_feature_extraction.py:125, :332
    if rc == _lib.E_BINS:
        raise AssertionError(msg)            # lite :59
    if rc == _lib.E_INVAL:
        raise ValueError(msg)
    raise RuntimeError('libsalsa_hip: %s (code %d)' % (msg, rc))


def bin_limits(fs, n_fft, fmin_doa, fmax_doa):
    """(lower_bin, upper_bin, cutoff_bin) -- salsa_feature_extraction.py:298-304, lite :57-58 (host integer math)."""
    lo, up, cut = C.c_int(), C.c_int(), C.c_int()
    rc = _lib.load().salsa_bin_limits(int(fs), int(n_fft), int(fmin_doa), int(fmax_doa), C.byref(lo), C.byref(up),
                                      C.byref(cut))
    if rc:
        _raise(rc)
    return lo.value, up.value, cut.value


def compress_matrix(n_fft, is_compress_high_freq=True):
    """MagStftExtractor.W (salsa_feature_extraction.py:152-175) as the kernels apply it."""
    if n_fft not in (256, 512):
        raise AssertionError('nfft is not 512 or 256')
    F = (200 if n_fft == 512 else 100) if is_compress_high_freq else n_fft // 2
    W = np.zeros((F, n_fft // 2 + 1), np.float32)
    rc = _lib.load().salsa_compress_matrix(n_fft, int(is_compress_high_freq), W.ctypes.data_as(C.POINTER(C.c_float)))
    if rc:
        _raise(rc)
    return W


class SalsaExtractor:
    """Batched SALSA / SALSA-Lite / SALSA-IPD feature extraction on one MI355X.

    Keyword names follow the reference's extract_features() and its YAML ``data`` block."""

    def __init__(self, fs=24000, n_fft=512, hop_len=300, win_len=None, fmin_doa=50, fmax_doa=9000, cond_num=5.0,
                 n_hopframes=3, is_tracking=True, is_compress_high_freq=True, audio_format='foa',
                 feature_type='salsa', audio_layout='planar', device=None, flags=0, floor_mask_ratio=0.0, fmax_spec=0):
        if audio_format not in _lib.FORMAT:
            raise ValueError('Unknown audio format {}'.format(audio_format))
        assert feature_type in _lib.FEATURE, 'Invalid feature type {}'.format(feature_type)
        self.L = _lib.load()
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self.params = _lib.SalsaParams(
            fs=int(fs), n_fft=int(n_fft), hop_len=int(hop_len), win_len=int(win_len or n_fft), fmin_doa=int(fmin_doa),
            fmax_doa=int(fmax_doa), cond_num=float(cond_num), n_hopframes=int(n_hopframes),
            is_tracking=int(bool(is_tracking)), is_compress_high_freq=int(bool(is_compress_high_freq)),
            audio_format=_lib.FORMAT[audio_format], feature_type=_lib.FEATURE[feature_type],
            audio_layout=_lib.LAYOUT[audio_layout], flags=int(flags), floor_mask_ratio=float(floor_mask_ratio),
            fmax_spec=int(fmax_spec), reserved=0)
        self.audio_layout = audio_layout
        self.feature_type = feature_type
        self._plan = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.L.salsa_plan_create(C.byref(self.params), C.byref(self._plan))
        if rc:
            self._plan = None
            _raise(rc)
        self._ws = None
        self._prefix_mode = False
        self._scaler = None
        self._pipe = None                            # (n_groups, split_pairs, graph) once set_pipeline / set_groups was called
        self._stats = None                           # solver counters (set_stats)
        self._fused = 0                              # schedule of extract() (set_fused)

    def copy_plan_state_to(self, other: 'SalsaExtractor'):
        """Re-apply to ``other`` (built from kwargs()) the plan state attached AFTER construction: the fused scaler and the
        pipelined schedule.  (Timing / prefix-issue modes are measurement state and are not carried over.)"""
        if self._scaler is not None:
            other.set_scaler(*self._scaler)
        else:
            other.set_scaler(None)
        if self._pipe is not None:
            other.set_pipeline(*self._pipe)
        if self._fused != other._fused:              # (the clones of a file pipeline must run the caller's schedule)
            other.set_fused(self._fused)

    def kwargs(self):
        """the constructor arguments of this extractor (to build further plans with the same parameters: one per pipeline slot)"""
        p = self.params
        inv = lambda d, v: [k for k, x in d.items() if x == v][0]
        return dict(fs=p.fs, n_fft=p.n_fft, hop_len=p.hop_len, win_len=p.win_len, fmin_doa=p.fmin_doa, fmax_doa=p.fmax_doa,
                    cond_num=p.cond_num, n_hopframes=p.n_hopframes, is_tracking=bool(p.is_tracking),
                    is_compress_high_freq=bool(p.is_compress_high_freq), audio_format=inv(_lib.FORMAT, p.audio_format),
                    feature_type=self.feature_type, audio_layout=self.audio_layout, device=self.device, flags=p.flags,
                    floor_mask_ratio=p.floor_mask_ratio, fmax_spec=p.fmax_spec)

    def __del__(self):
        if getattr(self, '_plan', None):
            self.L.salsa_plan_destroy(self._plan)
            self._plan = None

    # ------------------------------------------------------------------------------------------------ shapes
    def output_shape(self, n_samples):
        c, t, f = C.c_int(), C.c_int64(), C.c_int()
        rc = self.L.salsa_output_shape(self._plan, int(n_samples), C.byref(c), C.byref(t), C.byref(f))
        if rc:
            _raise(rc)
        return c.value, t.value, f.value

    def workspace_bytes(self, batch, n_samples):
        return int(self.L.salsa_workspace_bytes(self._plan, int(batch), int(n_samples)))

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return self._ws

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------------------------------------ hot path
    def extract(self, audio: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """audio float32 CUDA [B,4,N] (planar) or [B,N,4] (interleaved) -> features float32 [B,7,T,F]."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 3 and audio.is_contiguous()
        if self.audio_layout == 'planar':
            B, ch, N = audio.shape
        else:
            B, N, ch = audio.shape
        assert ch == 4, 'SALSA features are defined for 4-channel clips'
        if audio.device != self.device:
            raise ValueError('plan is bound to %s, audio is on %s' % (self.device, audio.device))
        Cn, T, F = self.output_shape(N)
        if out is None:
            out = torch.empty((B, Cn, T, F), dtype=torch.float32, device=audio.device)
        else:
            assert out.shape == (B, Cn, T, F) and out.dtype == torch.float32 and out.is_contiguous() and out.is_cuda
        nws = self.workspace_bytes(B, N)
        ws = self._workspace(nws)
        with torch.cuda.device(self.device):        # the plan's device must be current for the launches
            rc = self.L.salsa_extract_batch(self._plan, C.c_void_p(audio.data_ptr()), B, N, C.c_void_p(out.data_ptr()),
                                            C.c_void_p(ws.data_ptr()), ws.numel(), self._stream())
        if rc == _lib.PARTIAL and self._prefix_mode:
            return out                               # inside issue_prefix(): an incomplete issue the caller asked for
        if rc:
            _raise(rc)
        return out

    __call__ = extract

    def extract_multichannel(self, audio: torch.Tensor) -> torch.Tensor:
        """contrib surface, 6 - 16 microphones, even (salsa_extract_multichannel): audio float32 CUDA [B, C, N] planar ->
        [B, 2C-1, T, F] float32 (C log-spectrograms, then C-1 spatial planes).  The plan must carry FLAG_FLEX."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 3 and audio.is_contiguous()
        B, ch, N = audio.shape
        if ch not in (6, 8, 10, 12, 14, 16):
            raise ValueError('extract_multichannel takes an even number of channels from 6 to 16, got %d' % ch)
        if audio.device != self.device:
            raise ValueError('plan is bound to %s, audio is on %s' % (self.device, audio.device))
        _, T, F = self.output_shape(N)
        out = torch.empty((B, 2 * ch - 1, T, F), dtype=torch.float32, device=audio.device)
        ws = self._workspace(int(self.L.salsa_multichannel_workspace_bytes(self._plan, ch, B, N)))
        with torch.cuda.device(self.device):
            rc = self.L.salsa_extract_multichannel(self._plan, C.c_void_p(audio.data_ptr()), ch, B, N, C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(ws.data_ptr()), ws.numel(), self._stream())
        if rc:
            _raise(rc)
        return out

    def logspec(self, audio: torch.Tensor) -> torch.Tensor:
        """MagStftExtractor.extract on device: audio [B,4,N] planar -> [B,4,T,F]."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 3 and audio.is_contiguous()
        B, ch, N = audio.shape
        assert ch == 4
        n_fft = self.params.n_fft
        F = ((200 if n_fft == 512 else 100) if self.params.is_compress_high_freq else n_fft // 2)
        T = 1 + N // self.params.hop_len
        out = torch.empty((B, 4, T, F), dtype=torch.float32, device=audio.device)
        with torch.cuda.device(self.device):
            rc = self.L.salsa_logspec_batch(self._plan, C.c_void_p(audio.data_ptr()), B, 4, N,
                                            C.c_void_p(out.data_ptr()), self._stream())
        if rc:
            _raise(rc)
        return out

    def eigvec(self, X: torch.Tensor, lower_bin: int, return_gate: bool = False):
        """extract_normalized_eigenvector on device: X complex64 [B,n_bins,n_frames,4] -> float64 [B,3,n_bins,n_frames]."""
        assert X.is_cuda and X.dtype == torch.complex64 and X.dim() == 4 and X.shape[3] == 4 and X.is_contiguous()
        B, nb, nt, _ = X.shape
        out = torch.empty((B, 3, nb, nt), dtype=torch.float64, device=X.device)
        gate = torch.empty((B, nb, nt), dtype=torch.uint8, device=X.device)
        nws = int(self.L.salsa_eigvec_workspace_bytes(self._plan, B, nb, nt))
        ws = self._workspace(nws)
        with torch.cuda.device(self.device):
            rc = self.L.salsa_eigvec_batch(self._plan, C.c_void_p(X.data_ptr()), B, nb, nt, int(lower_bin),
                                           C.c_void_p(out.data_ptr()), C.c_void_p(gate.data_ptr()),
                                           C.c_void_p(ws.data_ptr()), ws.numel(), self._stream())
        if rc:
            _raise(rc)
        return (out, gate) if return_gate else out

    def eigvec_features(self, X: torch.Tensor, lower_bin: int) -> torch.Tensor:
        """extract_normalized_eigenvector through the PRODUCTION feature kernel (salsa_eigvec_feature_batch: the packed-float32
        pair solve with its float64 cold list, or the float64 instantiation under FLAG_FORCE_F64): X complex64
        [B,n_bins,n_frames,4] -> float32 [B,3,n_frames,n_bins] (time-major, as extract() writes channels 4-6)."""
        assert X.is_cuda and X.dtype == torch.complex64 and X.dim() == 4 and X.shape[3] == 4 and X.is_contiguous()
        B, nb, nt, _ = X.shape
        feat = torch.zeros((B, 7, nt, nb), dtype=torch.float32, device=X.device)
        ws = self._workspace(int(self.L.salsa_eigvec_workspace_bytes(self._plan, B, nb, nt)))
        with torch.cuda.device(self.device):
            rc = self.L.salsa_eigvec_feature_batch(self._plan, C.c_void_p(X.data_ptr()), B, nb, nt, int(lower_bin),
                                                   C.c_void_p(feat.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                                   self._stream())
        if rc:
            _raise(rc)
        return feat[:, 4:]

    def set_fused(self, mode: int):
        """Schedule of extract() (include/salsa_hip.h salsa_plan_set_fused): 0 three kernels, 1 the fused STFT + covariance / eigen
        kernel behind the tracker."""
        rc = self.L.salsa_plan_set_fused(self._plan, int(mode))
        if rc:
            _raise(rc)
        self._fused = int(mode)

    def set_stats(self, on: bool = True):
        """Attach (or detach) the solver counters (salsa_plan_set_stats); read_stats() returns and clears them."""
        if not on:
            torch.cuda.synchronize(self.device)          # launches in flight still add to the old counters
        self._stats = torch.zeros(4, dtype=torch.int64, device=self.device) if on else None
        rc = self.L.salsa_plan_set_stats(self._plan, C.c_void_p(self._stats.data_ptr()) if on else None)
        if rc:
            _raise(rc)

    def read_stats(self):
        """{'items', 'gated_frames', 'cold_frames', 'tiles'} accumulated by the covariance / eigen launches since the last read"""
        if self._stats is None:
            raise RuntimeError('read_stats(): no counters attached -- call set_stats(True) first')
        v = self._stats.cpu().tolist()
        self._stats.zero_()
        return dict(items=v[0], gated_frames=v[1], cold_frames=v[2], tiles=v[3])

    # ------------------------------------------------------------------------------------------------ timing
    def set_timing(self, enable):
        """False / 0: off.  True / 1: an event pair around every launch.  K > 1: every kernel of a call is launched K times back
        to back between one event pair and read_timing() reports elapsed / K (no event between launches)."""
        if int(enable) < 0:
            raise ValueError('set_timing takes 0, 1 or a repeat count; the prefix-issue measurement mode is issue_prefix()')
        self.L.salsa_plan_set_timing(self._plan, int(enable))

    @contextlib.contextmanager
    def issue_prefix(self, kernels: int):
        """MEASUREMENT ONLY.  Inside this context extract() issues only the first ``kernels`` launches of the path (1: the STFT,
        2: STFT + tracker) on the buffers an earlier full call left behind, so a bench can attribute the step to its kernels by
        wall-clock differences; the output it returns is NOT a feature array.  The library reports such a call with the distinct
        code SALSA_PARTIAL, which extract() accepts only here; the plan is put back to plain issue on exit, whatever happens."""
        assert kernels in (1, 2)
        self.L.salsa_plan_set_timing(self._plan, -kernels)
        self._prefix_mode = True
        try:
            yield self
        finally:
            self._prefix_mode = False
            self.L.salsa_plan_set_timing(self._plan, 0)

    def set_scaler(self, mean=None, std=None):
        """Attach (or detach with None) the feature scaler: extract() then returns the spectrogram channels already
        normalised, (x - mean) / std per frequency (dataset/database.py:197-202), fused into the STFT kernel's stores."""
        if mean is None:
            self._scaler = None
            rc = self.L.salsa_plan_set_scaler(self._plan, None, None)
        else:
            F = self.output_shape(self.params.n_fft)[2]
            m = torch.as_tensor(mean, dtype=torch.float32).reshape(4, F).contiguous().to(self.device)
            s = torch.as_tensor(std, dtype=torch.float32).reshape(4, F).contiguous().to(self.device)
            self._scaler = (m, s)                                   # keep the device arrays alive
            rc = self.L.salsa_plan_set_scaler(self._plan, C.c_void_p(m.data_ptr()), C.c_void_p(s.data_ptr()))
        if rc:
            _raise(rc)

    def set_groups(self, n_groups: int):
        """Clip-group pipelining depth of extract() (1 = single stream)."""
        with torch.cuda.device(self.device):
            rc = self.L.salsa_plan_set_groups(self._plan, int(n_groups))
        if rc:
            _raise(rc)
        self._pipe = (int(n_groups),) + (self._pipe[1:] if self._pipe else (False, False))

    def set_pipeline(self, n_groups: int = 1, split_pairs: bool = False, graph: bool = False):
        """Pipelined schedule of extract() (include/salsa_hip.h salsa_plan_set_pipeline): clip groups on plan-owned streams,
        optionally the STFT split by channel pair so the tracker starts early, optionally replayed as one hipGraph."""
        flags = (_lib.PIPE_SPLIT_PAIRS if split_pairs else 0) | (_lib.PIPE_GRAPH if graph else 0)
        with torch.cuda.device(self.device):
            rc = self.L.salsa_plan_set_pipeline(self._plan, int(n_groups), flags)
        if rc:
            _raise(rc)
        self._pipe = (int(n_groups), bool(split_pairs), bool(graph))

    def read_timing(self):
        """[(kernel name, milliseconds)] of the last extract() call (HIP events on its stream)."""
        ms = (C.c_float * _lib.MAX_KERNELS)()
        names = (C.c_char_p * _lib.MAX_KERNELS)()
        n = C.c_int()
        rc = self.L.salsa_plan_read_timing(self._plan, ms, names, C.byref(n))
        if rc:
            _raise(rc)
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]


def scaler_accumulate(feat: torch.Tensor, sums: torch.Tensor = None, n_scaler_channels: int = 4) -> torch.Tensor:
    """Device-side compute_scaler accumulation: feat float32 CUDA [B,C,T,F] -> sums float64 [2,n_sc,F] (+= in place)."""
    assert feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 4 and feat.is_contiguous()
    B, Cn, T, F = feat.shape
    if sums is None:
        sums = torch.zeros((2, n_scaler_channels, F), dtype=torch.float64, device=feat.device)
    with torch.cuda.device(feat.device):         # launch on the GPU that owns the pointers, whatever device is current
        rc = _lib.load().salsa_scaler_accumulate(C.c_void_p(feat.data_ptr()), B, Cn, T, F, n_scaler_channels,
                                                 C.c_void_p(sums.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream(feat.device).cuda_stream))
    if rc:
        _raise(rc)
    return sums


def scaler_finish(sums: torch.Tensor, n_frames_total: int):
    """-> (mean, std) float32 CUDA tensors of shape (n_sc, 1, F), population std (sklearn StandardScaler.var_)."""
    mean = sums[0] / n_frames_total
    var = torch.clamp(sums[1] / n_frames_total - mean * mean, min=0.0)
    return mean[:, None, :].float(), torch.sqrt(var)[:, None, :].float()


def normalize_(feat: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """In-place normalise-on-load of the first mean.shape[0] channels (dataset/database.py:197-202)."""
    assert feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 4 and feat.is_contiguous()
    B, Cn, T, F = feat.shape
    m = mean.reshape(-1, F).contiguous().float().to(feat.device)
    s = std.reshape(-1, F).contiguous().float().to(feat.device)
    with torch.cuda.device(feat.device):
        rc = _lib.load().salsa_normalize_batch(C.c_void_p(feat.data_ptr()), B, Cn, T, F, m.shape[0], C.c_void_p(m.data_ptr()),
                                               C.c_void_p(s.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream(feat.device).cuda_stream))
    if rc:
        _raise(rc)
    return feat


class StreamedExtractor:
    """Bulk extraction of many independent batches: ``n_streams`` plans on their own HIP streams, fed round-robin, so
    the latency-bound noise-floor tracker of one batch overlaps the STFT / eigen kernels of its neighbours (+9 % on
    MI355X with 2 streams).  ``extract_many`` yields the feature tensors in input order.  A yielded tensor is valid ONLY
    UNTIL THE GENERATOR IS ADVANCED AGAIN: the next pull launches a new batch into the same buffer.  Clone it (or finish
    with it) before calling next()."""

    def __init__(self, n_streams: int = 2, **extractor_kwargs):
        self.exs = [SalsaExtractor(**extractor_kwargs) for _ in range(n_streams)]
        self.streams = [torch.cuda.Stream(device=self.exs[0].device) for _ in range(n_streams)]
        self.outs = [None] * n_streams

    def extract_many(self, batches):
        n = len(self.exs)
        pending = []
        for i, audio in enumerate(batches):
            j = i % n
            self.streams[j].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.streams[j]):
                want = (audio.shape[0],) + tuple(self.exs[j].output_shape(
                    audio.shape[2] if self.exs[j].audio_layout == 'planar' else audio.shape[1]))
                if self.outs[j] is None or tuple(self.outs[j].shape) != want:
                    self.outs[j] = torch.empty(want, dtype=torch.float32, device=audio.device)
                self.exs[j].extract(audio, out=self.outs[j])
                ev = torch.cuda.Event()
                ev.record(self.streams[j])
            pending.append((ev, self.outs[j]))
            if len(pending) == n:
                e, o = pending.pop(0)
                e.synchronize()
                yield o
        for e, o in pending:
            e.synchronize()
            yield o


class HostPipeline:
    """Host buffers in, host buffers out (what the reference-surface harness needs): ``depth`` slots, each with pinned
    staging, device buffers, its own plan and its own copy-in / compute / copy-out streams, so the PCIe transfer of one batch
    in each direction overlaps the kernels of another (MI355X moves data in both directions at once).  ``run`` takes an
    iterable of float32 host arrays ``(B, 4, N)`` (or ``(B, N, 4)`` for interleaved plans) of ONE shape and yields float32
    host feature arrays ``(B, 7, T, F)`` in order.  A yielded array is a view of a pinned slot and is valid ONLY UNTIL THE
    GENERATOR IS ADVANCED AGAIN: the next pull starts the following batch's round trip through that very slot (its
    device-to-host copy will overwrite the view).  Copy it or finish with it before calling next()."""

    def __init__(self, depth: int = 3, **extractor_kwargs):
        self.depth = depth
        self.exs = [SalsaExtractor(**extractor_kwargs) for _ in range(depth)]
        dev = self.exs[0].device
        self.s_in = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.s_run = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.s_out = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.slots = None

    def _alloc(self, shape):
        ex = self.exs[0]
        n = shape[2] if ex.audio_layout == 'planar' else shape[1]
        oshape = (shape[0],) + tuple(ex.output_shape(n))
        dev = ex.device
        self.slots = [dict(h_in=torch.empty(shape, dtype=torch.float32, pin_memory=True),
                           d_in=torch.empty(shape, dtype=torch.float32, device=dev),
                           d_out=torch.empty(oshape, dtype=torch.float32, device=dev),
                           h_out=torch.empty(oshape, dtype=torch.float32, pin_memory=True),
                           done=None) for _ in range(self.depth)]
        self.shape = tuple(shape)

    def run(self, batches=None, fill=None, n_batches=None, shape=None):
        """Either ``batches``: an iterable of host arrays (each is copied into a pinned slot: one extra host copy), or
        ``fill(buf, i)``: a callback that writes batch i straight into the pinned numpy array ``buf`` of ``shape`` (what a file
        reader should do), called for i = 0 .. n_batches-1."""
        if fill is not None:
            if self.slots is None or tuple(shape) != self.shape:
                self._alloc(tuple(shape))
            source = range(n_batches)
        else:
            source = batches
        pending = []
        for i, a in enumerate(source):
            if fill is None:
                a = torch.as_tensor(a)
                if self.slots is None or tuple(a.shape) != self.shape:
                    assert not pending, 'all batches of one run must have the same shape'
                    self._alloc(tuple(a.shape))
            j = i % self.depth
            sl = self.slots[j]
            if sl['done'] is not None:
                sl['done'].synchronize()                          # the slot's previous round trip has finished
            if fill is None:
                sl['h_in'].copy_(a)                               # host -> pinned staging
            else:
                fill(sl['h_in'].numpy(), i)
            with torch.cuda.stream(self.s_in[j]):
                sl['d_in'].copy_(sl['h_in'], non_blocking=True)
            self.s_run[j].wait_stream(self.s_in[j])
            with torch.cuda.stream(self.s_run[j]):
                self.exs[j].extract(sl['d_in'], out=sl['d_out'])
            self.s_out[j].wait_stream(self.s_run[j])
            with torch.cuda.stream(self.s_out[j]):
                sl['h_out'].copy_(sl['d_out'], non_blocking=True)
                sl['done'] = torch.cuda.Event()
                sl['done'].record(self.s_out[j])
            pending.append(j)
            if len(pending) == self.depth:
                k = pending.pop(0)
                self.slots[k]['done'].synchronize()
                yield self.slots[k]['h_out'].numpy()
        for k in pending:
            self.slots[k]['done'].synchronize()
            yield self.slots[k]['h_out'].numpy()
