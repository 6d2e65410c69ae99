"""Hand-written MI355X kernels for the memory-bound layers of the SELD CRNN (include/salsa_nn.h), as autograd functions.
The CPU / non-channels-last / odd-channel cases fall through to the torch operator of the same name -- the CRNN is the
consumer of the feature path, its CPU tests run the model on CPU tensors."""
import ctypes as C
import os

import torch
import torch.nn.functional as F

from .. import _lib

_DT = {torch.float32: (0, 4), torch.bfloat16: (1, 8)}
USE_HIP_POOL = os.environ.get('SALSA_HIP_POOL', '1') != '0'
USE_HIP_BN = os.environ.get('SALSA_HIP_BN', '1') != '0'
USE_HIP_CONV = os.environ.get('SALSA_HIP_CONV', '1') != '0'
USE_HIP_CONV_WIDE = os.environ.get('SALSA_HIP_CONV_WIDE', '1') != '0'   # the 128 / 256 / 512-channel 3x3 layers (conv_wide.hip)


_DET_WS = {}                    # device -> workspace tensor (registered with the library for THAT device)
DET_WS_BYTES = 160 << 20        # include/salsa_nn.h: SALSA_NN_DET_WS_BYTES
# Deterministic weight gradients are the DEFAULT (SALSA_DETERMINISTIC=0 opts out): measured at +0.04 - 0.09 ms of a 10.8-ms
# training step (profiles/r4_ab_notes.txt) for bit-reproducible gradients.
_DET_ENV = os.environ.get('SALSA_DETERMINISTIC', '1') != '0'


def set_deterministic(on: bool, device=None) -> None:
    """Bit-reproducible weight gradients (include/salsa_nn.h: salsa_nn_set_deterministic): every weight-gradient kernel and the GRU
    bias column sums write per-workgroup partial slabs into a device workspace and a reduction launch adds them in slab order,
    instead of float atomics in arrival order.  On by default: the first differentiable forward of a model on a CUDA device
    switches it on for that device (``SALSA_DETERMINISTIC=0``, or ``set_deterministic(False)`` after it, selects the atomics).
    One 160-MB workspace PER DEVICE, looked up by the library from the device that is current at each launch (so a backward on
    cuda:0 never writes through cuda:1's workspace); on one device it is shared by all calls: one stream at a time, as the trainer
    uses it -- two concurrent backward passes on different streams of one device need ``set_deterministic(False)``."""
    L = _lib.load()
    if not on:
        L.salsa_nn_set_deterministic(None, 0)
        _DET_WS.clear()
        _DET_USER[0] = False
        return
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    if dev not in _DET_WS:
        ws = torch.empty(DET_WS_BYTES, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = L.salsa_nn_set_deterministic(C.c_void_p(ws.data_ptr()), DET_WS_BYTES)
        if rc:
            raise RuntimeError('salsa_nn_set_deterministic failed (%d) on %s' % (rc, dev))
        _DET_WS[dev] = ws
    _DET_USER[0] = True


_DET_USER = [None]              # None: never set by hand (the default applies); True / False: the caller's choice


def is_deterministic() -> bool:
    return bool(_lib.load().salsa_nn_get_deterministic())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def conv_backend_note() -> str:
    """Which convolution layers run on the hand-written MFMA kernels (for the bench lines)."""
    if not USE_HIP_CONV:
        return 'all on MIOpen through torch (SALSA_HIP_CONV=0)'
    if USE_HIP_CONV_WIDE:
        wrw = ('every 3x3 layer hand-written' if os.environ.get('SALSA_HIP_CONV_WIDE_WRW', '1') != '0'
               else '64->64 hand-written, the rest MIOpen')
        return ('every 3x3 layer forward + data gradient on hand-written MFMA kernels (stem 7->64 and 64->64: conv_mfma.hip; 128/256/512 '
                'channels: conv_wide.hip); weight gradients: %s; 1x1 shortcuts: conv_1x1.hip (MIOpen: nothing)' % wrw)
    return 'stem 7->64 and the five 64->64 3x3 layers on the hand-written MFMA kernels (conv_mfma.hip), the 128/256/512-channel layers on MIOpen through torch'


class _AvgPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, Cn, H, W = x.shape
        y = torch.empty((N, Cn, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            rc = _lib.load().salsa_nn_avgpool2x2_fwd(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), _DT[x.dtype][0],
                                                     N, H, W, Cn, _stream(x))
        if rc:
            raise RuntimeError('salsa_nn_avgpool2x2_fwd failed (%d)' % rc)
        ctx.shape = (N, Cn, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, Cn, H, W = ctx.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty((N, Cn, H, W), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
        with torch.cuda.device(gy.device):
            rc = _lib.load().salsa_nn_avgpool2x2_bwd(C.c_void_p(gy.data_ptr()), C.c_void_p(gx.data_ptr()), _DT[gy.dtype][0],
                                                     N, H, W, Cn, _stream(gy))
        if rc:
            raise RuntimeError('salsa_nn_avgpool2x2_bwd failed (%d)' % rc)
        return gx


def avg_pool2x2(x: torch.Tensor) -> torch.Tensor:
    """F.avg_pool2d(x, 2) (models/model_utils.py:224 'avg'); channels-last CUDA bf16 / float32 tensors take the HIP kernel."""
    if (USE_HIP_POOL and x.is_cuda and x.dim() == 4 and x.dtype in _DT and x.shape[1] % _DT[x.dtype][1] == 0
            and x.shape[2] >= 2 and x.shape[3] >= 2 and x.is_contiguous(memory_format=torch.channels_last)):
        return _AvgPool2x2.apply(x)
    return F.avg_pool2d(x, 2)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _BnAct(torch.autograd.Function):
    """y = [dropout]( [relu]( batch_norm(x) [+ residual] ) ) in training mode (salsa_nn_bn_train_fwd / salsa_nn_bn_bwd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, momentum, eps, relu, drop_p=0.0, batches_tracked=None,
                stats_part=None):
        N, Cn, H, W = x.shape
        M = N * H * W
        y = torch.empty_like(x, memory_format=torch.channels_last)
        save = torch.empty((2, Cn), dtype=torch.float32, device=x.device)
        ws = torch.empty(_lib.load().salsa_nn_bn_workspace_bytes(_DT[x.dtype][0], M, Cn) // 8 + 1, dtype=torch.float64, device=x.device)
        # the mask seed comes from torch's CPU generator (torch.manual_seed makes it reproducible; no device sync)
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if drop_p > 0 else 0
        # Where something is added before the ReLU the backward cannot recompute the mask from x: it used to read the stored y in
        # both of its passes just to test `> 0`; the forward now leaves the tests as a bit plane (one byte per 16-byte vector)
        bits = None
        if relu and residual is not None and USE_BN_RELU_BITS:
            bits = torch.empty(x.numel() // _DT[x.dtype][1], dtype=torch.uint8, device=x.device)
        args = (_ptr(x), _ptr(y), _ptr(residual), _DT[x.dtype][0], M, Cn, _ptr(weight), _ptr(bias), float(eps), float(momentum),
                _ptr(running_mean), _ptr(running_var), _ptr(save[0]), _ptr(save[1]), _ptr(ws), int(relu), float(drop_p), seed,
                _ptr(batches_tracked), _ptr(stats_part), 0 if stats_part is None else stats_part.numel() // (2 * Cn))
        with torch.cuda.device(x.device):
            if bits is not None:
                rc = _lib.load().salsa_nn_bn_train_fwd_bits(*args, _ptr(bits), _stream(x))
            else:
                rc = _lib.load().salsa_nn_bn_train_fwd(*args, _stream(x))
        if rc:
            raise RuntimeError('salsa_nn_bn_train_fwd failed (%d)' % rc)
        # the ReLU mask comes from the bit plane (or, with SALSA_BN_RELU_BITS=0, from y) only when something was added before the
        # ReLU; otherwise the backward recomputes it from x
        ctx.save_for_backward(x, bits if bits is not None else (y if (relu and residual is not None) else None), weight, bias, save)
        ctx.has_residual, ctx.relu, ctx.drop = residual is not None, (2 if bits is not None else int(bool(relu))), (float(drop_p), seed)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight, bias, save = ctx.saved_tensors
        N, Cn, H, W = x.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dres = torch.empty_like(x, memory_format=torch.channels_last) if ctx.has_residual else None
        dwb = torch.empty((2, Cn), dtype=torch.float32, device=x.device)
        ws = torch.empty(_lib.load().salsa_nn_bn_workspace_bytes(_DT[x.dtype][0], N * H * W, Cn) // 8 + 1, dtype=torch.float64, device=x.device)
        coef = torch.empty(7 * Cn, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.load().salsa_nn_bn_bwd(_ptr(gy), _ptr(y), _ptr(x), _ptr(dx), _ptr(dres), _DT[x.dtype][0], N * H * W, Cn,
                                             _ptr(weight), _ptr(bias), _ptr(save[0]), _ptr(save[1]), int(ctx.relu), _ptr(dwb[0]),
                                             _ptr(dwb[1]), _ptr(ws), _ptr(coef), ctx.drop[0], ctx.drop[1], _stream(x))
        if rc:
            raise RuntimeError('salsa_nn_bn_bwd failed (%d)' % rc)
        return dx, dwb[0].to(weight.dtype), dwb[1].to(weight.dtype), None, None, dres, None, None, None, None, None, None


class _BnReluPool(torch.autograd.Function):
    """y = avg_pool2x2(relu(batch_norm(x))) in training mode (salsa_nn_bn_train_fwd_pool / salsa_nn_bn_bwd_pool): the stem's
    tail; the full-resolution activation is neither written in the forward nor its gradient in the backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, batches_tracked=None, stats_part=None, residual=None):
        N, Cn, H, W = x.shape
        y = torch.empty((N, Cn, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        save = torch.empty((2, Cn), dtype=torch.float32, device=x.device)
        ws = torch.empty(_lib.load().salsa_nn_bn_workspace_bytes(_DT[x.dtype][0], N * H * W, Cn) // 8 + 1, dtype=torch.float64, device=x.device)
        bits = None                                  # (with a residual the backward recomputed the mask from x AND the residual)
        if residual is not None and USE_BN_RELU_BITS:
            bits = torch.empty(x.numel() // _DT[x.dtype][1], dtype=torch.uint8, device=x.device)
        args = (_ptr(x), _ptr(y), _ptr(residual), _DT[x.dtype][0], N, H, W, Cn, _ptr(weight), _ptr(bias), float(eps), float(momentum),
                _ptr(running_mean), _ptr(running_var), _ptr(save[0]), _ptr(save[1]), _ptr(ws), _ptr(batches_tracked), _ptr(stats_part),
                0 if stats_part is None else stats_part.numel() // (2 * Cn))
        with torch.cuda.device(x.device):
            if bits is not None:
                rc = _lib.load().salsa_nn_bn_train_fwd_pool_bits(*args, _ptr(bits), _stream(x))
            else:
                rc = _lib.load().salsa_nn_bn_train_fwd_pool(*args, _stream(x))
        if rc:
            raise RuntimeError('salsa_nn_bn_train_fwd_pool failed (%d)' % rc)
        ctx.save_for_backward(x, weight, bias, save, bits if bits is not None else residual)
        ctx.bits, ctx.has_residual = bits is not None, residual is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, save, residual = ctx.saved_tensors
        N, Cn, H, W = x.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dres = torch.empty_like(x, memory_format=torch.channels_last) if ctx.has_residual else None
        dwb = torch.empty((2, Cn), dtype=torch.float32, device=x.device)
        ws = torch.empty(_lib.load().salsa_nn_bn_workspace_bytes(_DT[x.dtype][0], N * H * W, Cn) // 8 + 1, dtype=torch.float64, device=x.device)
        coef = torch.empty(7 * Cn, dtype=torch.float32, device=x.device)
        fn = _lib.load().salsa_nn_bn_bwd_pool_bits if ctx.bits else _lib.load().salsa_nn_bn_bwd_pool   # (residual = the bit plane)
        with torch.cuda.device(x.device):
            rc = fn(_ptr(gy), _ptr(x), _ptr(residual), _ptr(dx), _ptr(dres), _DT[x.dtype][0], N, H, W, Cn, _ptr(weight), _ptr(bias),
                    _ptr(save[0]), _ptr(save[1]), _ptr(dwb[0]), _ptr(dwb[1]), _ptr(ws), _ptr(coef), _stream(x))
        if rc:
            raise RuntimeError('salsa_nn_bn_bwd_pool failed (%d)' % rc)
        return dx, dwb[0].to(weight.dtype), dwb[1].to(weight.dtype), None, None, None, None, None, None, dres


USE_HIP_BN_POOL = os.environ.get('SALSA_HIP_BN_POOL', '1') != '0'
USE_BN_RELU_BITS = os.environ.get('SALSA_BN_RELU_BITS', '1') != '0'   # ReLU masks of the residual BatchNorms as bit planes (round 4)
USE_HIP_BN_RES_POOL = os.environ.get('SALSA_HIP_BN_RES_POOL', '1') != '0'   # ... with a residual: the blocks before a stride-2 block


class BatchNormAct2d(torch.nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state_dict keys) whose forward can take the residual add and the ReLU
    that follow it in the upstream blocks, and in training the dropout behind that ReLU: ``bn(x, residual=None, relu=False,
    dropout_p=0.0)``.  Channels-last CUDA bf16 / float32 inputs run the fused HIP kernels; anything else runs torch's
    batch_norm + add + relu + dropout.  ``relu_pool(x)`` = avg_pool2x2(relu(bn(x))), one kernel pass in training."""

    def relu_pool(self, x, stats_part=None, residual=None):
        """avg_pool2x2(relu(bn(x) [+ residual])).  stats_part: per-workgroup [2][C] float64 partial sums of x left by the
        convolution that produced it (training)"""
        if (USE_HIP_BN and USE_HIP_BN_POOL and self.training and x.is_cuda and x.dim() == 4 and x.dtype in _DT and self.affine
                and self.track_running_stats and self.momentum is not None and x.shape[2] >= 2 and x.shape[3] >= 2
                and x.is_contiguous(memory_format=torch.channels_last)
                and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape
                                          and residual.is_contiguous(memory_format=torch.channels_last)))
                and _lib.load().salsa_nn_bn_supported(_DT[x.dtype][0], x.shape[0] * x.shape[2] * x.shape[3], x.shape[1])):
            self._stats_serial = getattr(self, '_stats_serial', 0) + 1
            return _BnReluPool.apply(x, self.weight.float(), self.bias.float(), self.running_mean, self.running_var, self.momentum,
                                     self.eps, self.num_batches_tracked, stats_part, residual)   # (the kernel counts the batch)
        return avg_pool2x2(self.forward(x, residual=residual, relu=True))

    def forward(self, x, residual=None, relu=False, dropout_p=0.0, stats_part=None):
        if not self.training:
            dropout_p = 0.0
        fused = (USE_HIP_BN and x.is_cuda and x.dim() == 4 and x.dtype in _DT and self.affine and self.track_running_stats
                 and self.momentum is not None and x.is_contiguous(memory_format=torch.channels_last)
                 and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape
                                           and residual.is_contiguous(memory_format=torch.channels_last)))
                 and _lib.load().salsa_nn_bn_supported(_DT[x.dtype][0], x.shape[0] * x.shape[2] * x.shape[3], x.shape[1])
                 and (self.training or not torch.is_grad_enabled() or not x.requires_grad))
        if not fused:
            y = super().forward(x)
            if residual is not None:
                y = y + residual
            y = F.relu(y, inplace=True) if relu else y
            return F.dropout(y, p=dropout_p, training=True) if dropout_p > 0 else y
        w, b = self.weight.float(), self.bias.float()
        if self.training:
            self._stats_serial = getattr(self, '_stats_serial', 0) + 1
            return _BnAct.apply(x, w, b, self.running_mean, self.running_var, residual, self.momentum, self.eps, relu, float(dropout_p),
                                self.num_batches_tracked, stats_part)      # (the kernel counts the batch: no add_ launch)
        N, Cn, H, W = x.shape
        y = torch.empty_like(x, memory_format=torch.channels_last)
        invstd = torch.rsqrt(self.running_var + self.eps)
        with torch.cuda.device(x.device):
            rc = _lib.load().salsa_nn_bn_eval_fwd(_ptr(x), _ptr(y), _ptr(residual), _DT[x.dtype][0], N * H * W, Cn, _ptr(w),
                                                  _ptr(b), _ptr(self.running_mean), _ptr(invstd), int(relu), _stream(x))
        if rc:
            raise RuntimeError('salsa_nn_bn_eval_fwd failed (%d)' % rc)
        return y


class _StackGroups(torch.autograd.Function):
    """``[torch.stack(g) for g in groups]`` for consecutive groups of ``n`` same-shaped tensors with ONE multi-tensor copy
    (``torch._foreach_copy_``) instead of one concatenation kernel per group; the backward hands out views of the incoming
    gradients (no kernel).  The decoder stacks the two directions of every GRU parameter and the four heads' weights every step."""

    @staticmethod
    def forward(ctx, n, *tensors):
        ctx.n = n
        outs = [torch.empty((n,) + tuple(tensors[i].shape), dtype=tensors[i].dtype, device=tensors[i].device)
                for i in range(0, len(tensors), n)]
        torch._foreach_copy_([o[j] for o in outs for j in range(n)], [t.detach() for t in tensors])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        n = ctx.n
        return (None,) + tuple(None if g is None else g[j] for g in gouts for j in range(n))


class _StackViews(torch.autograd.Function):
    """``torch.stack(g)`` for a group whose tensors already LIE stacked (pack_stacked_parameters): the result is a view of their
    common storage -- no kernel at all; the backward hands out views of the incoming gradient like _StackGroups."""

    @staticmethod
    def forward(ctx, *tensors):
        t0 = tensors[0].detach()
        return t0.as_strided((len(tensors),) + tuple(t0.shape), (t0.numel(),) + tuple(t0.stride()), t0.storage_offset())

    @staticmethod
    def backward(ctx, g):
        return tuple(g[j] for j in range(g.shape[0]))


def _lie_stacked(group):
    """the group's tensors are contiguous, equally shaped slices of ONE storage, one after the other"""
    t0 = group[0]
    if not t0.is_contiguous():
        return False
    n, base, sp = t0.numel(), t0.storage_offset(), t0.untyped_storage().data_ptr()
    return all(t.is_contiguous() and t.shape == t0.shape and t.dtype == t0.dtype and t.untyped_storage().data_ptr() == sp
               and t.storage_offset() == base + j * n for j, t in enumerate(group))


def pack_stacked_parameters(groups):
    """Re-home every group of same-shaped parameters in ONE buffer, one after the other (their ``.data`` become slices of it; the
    Parameter objects, their values and everything that refers to them -- optimizer, state dict, hooks -- are untouched), so
    that ``stack_groups`` finds them already stacked and returns a view: the two directions of every GRU parameter and the four
    heads' weights used to be copied into a stack every step (three multi-tensor copies, ~0.06 ms).  Moving the module to another
    device or dtype afterwards un-packs them again (``stack_groups`` then copies, as before): call this last."""
    with torch.no_grad():
        for g in groups:
            if _lie_stacked(g):
                continue
            buf = torch.stack([t.detach() for t in g]).contiguous()
            for j, t in enumerate(g):
                t.data = buf[j]


def stack_groups(groups):
    """groups: lists of equally many same-shaped tensors -> the stacked tensor of every group: a view where the group already lies
    stacked in memory (pack_stacked_parameters), one multi-tensor copy for the rest (see _StackGroups)."""
    n = len(groups[0])
    assert all(len(g) == n for g in groups)
    if USE_STACK_GROUPS and USE_STACK_VIEWS and all(_lie_stacked(g) for g in groups):
        return [_StackViews.apply(*g) for g in groups]
    flat = [t for g in groups for t in g]
    if not (USE_STACK_GROUPS and flat[0].is_cuda and hasattr(torch, '_foreach_copy_')):
        return [torch.stack(list(g)) for g in groups]
    return list(_StackGroups.apply(n, *flat))


USE_STACK_GROUPS = os.environ.get('SALSA_STACK_GROUPS', '1') != '0'
USE_STACK_VIEWS = os.environ.get('SALSA_STACK_VIEWS', '1') != '0'   # stacked parameters as views of packed storage (pack_stacked_parameters)
USE_HIP_FREQ_MEAN = os.environ.get('SALSA_HIP_FREQ_MEAN', '1') != '0'


class _FreqMean(torch.autograd.Function):
    """salsa_nn_freq_mean_fwd / _bwd: (N, C, T, F) bf16 channels-last -> float32 (T, N, C) time-major mean over F."""

    @staticmethod
    def forward(ctx, x):
        N, Cn, H, W = x.shape
        y = torch.empty((H, N, Cn), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.load().salsa_nn_freq_mean_fwd(_ptr(x), _ptr(y), N, H, W, Cn, 1, _stream(x))
        if rc:
            raise RuntimeError('salsa_nn_freq_mean_fwd failed (%d)' % rc)
        ctx.shape = (N, Cn, H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        N, Cn, H, W = ctx.shape
        g = g.contiguous().float()
        dx = torch.empty((N, Cn, H, W), dtype=torch.bfloat16, device=g.device, memory_format=torch.channels_last)
        with torch.cuda.device(g.device):
            rc = _lib.load().salsa_nn_freq_mean_bwd(_ptr(g), _ptr(dx), N, H, W, Cn, 1, _stream(g))
        if rc:
            raise RuntimeError('salsa_nn_freq_mean_bwd failed (%d)' % rc)
        return dx


def freq_mean_sequence(feat):
    """(B, C, T, F) encoder output -> (B, T, C) float32: the mean over frequency and the transpose of models/decoders.py in one
    pass for bf16 channels-last CUDA maps (the result is a view of a TIME-major buffer, the order the GRU scans want);
    ``feat.mean(dim=3).transpose(1, 2)`` otherwise."""
    if (USE_HIP_FREQ_MEAN and feat.is_cuda and feat.dtype == torch.bfloat16 and feat.dim() == 4 and feat.shape[1] % 8 == 0
            and feat.is_contiguous(memory_format=torch.channels_last)):
        return _FreqMean.apply(feat).transpose(0, 1)
    return feat.mean(dim=3).transpose(1, 2)


class _GradZeros:
    """Zero-initialised float32 weight-gradient buffers for the accumulating weight-gradient kernels, ONE allocation and ONE fill
    per backward pass instead of one ``torch.zeros`` per layer (25 fills of ~4 us in the training step).

    A *generation* is one forward/backward pass: the encoder announces it at every forward that can be differentiated
    (``new_generation``).  The plan -- how many buffers of which shape a pass asks for -- is LEARNED only during a learning
    generation (the first announced one, and the one after a generation that met a shape the plan has never seen: a second
    model), while every request falls back to ``torch.zeros``; it is installed when that generation is closed by the next
    announcement and then frozen.  Later generations carve their requests out of one flat zero tensor allocated at the pass's
    first request.  A slice is handed out once and never written again by this class (the flat tensor of a generation is
    dropped, not reused: the gradients that view it keep it alive), so gradient accumulation, retained graphs and
    ``param.grad`` stealing see ordinary fresh tensors.  Requests beyond the plan -- a second backward through the same forward,
    a backward that no forward announced (a direct call of an op, a stand-alone fused GRU) -- are served by ``torch.zeros`` and
    NEVER extend the plan, so the per-pass allocation cannot grow with the number of such passes (round-3 advice: 50 eval-mode
    backward passes used to reserve 459 MiB for a 9-MiB gradient)."""

    def __init__(self):
        self.plan = {}          # (shape, device) -> buffers per generation (frozen outside learning generations)
        self.taken = {}         # this generation's requests so far
        self.offsets = None     # (shape, device, i) -> offset into the flat tensor, floats
        self.total = {}         # device -> floats
        self.flat = {}          # device -> this generation's flat zero tensor
        self.learning = None    # None: no generation announced yet; True: this generation's counts become the plan; False: frozen
        self.relearn = False    # an announced generation asked for a shape the plan does not know

    def new_generation(self):
        if self.learning:                                   # close the learning generation: its counts are the plan
            if self.taken:                                  # (a forward that was never differentiated teaches nothing)
                self.plan = dict(self.taken)
                self.offsets = None
                self.learning = False
        elif self.learning is None or self.relearn:
            self.learning = True
        self.relearn = False
        self.taken = {}
        self.flat = {}

    def take(self, shape, device):
        key = (tuple(shape), device)
        i = self.taken.get(key, 0)
        self.taken[key] = i + 1
        if self.learning is not False or i >= self.plan.get(key, 0):
            if self.learning is False and key not in self.plan:
                self.relearn = True
            return torch.zeros(shape, dtype=torch.float32, device=device)
        if self.offsets is None:
            self.offsets, self.total = {}, {}
            for (shp, dev), cnt in self.plan.items():
                n = 1
                for d in shp:
                    n *= d
                for j in range(cnt):
                    self.offsets[(shp, dev, j)] = self.total.get(dev, 0)
                    self.total[dev] = self.total.get(dev, 0) + (n + 63) // 64 * 64      # 256-byte aligned slices
            self.flat = {}
        if device not in self.flat:
            self.flat[device] = torch.zeros(self.total[device], dtype=torch.float32, device=device)
        n = 1
        for d in shape:
            n *= d
        off = self.offsets[(key[0], device, i)]
        return self.flat[device][off:off + n].view(shape)


_GRAD_ZEROS = _GradZeros()
USE_GRAD_ARENA = os.environ.get('SALSA_GRAD_ARENA', '1') != '0'


def new_backward_generation(device=None):
    """Announce a new forward/backward pass to the weight-gradient buffer pool (the encoder calls it per differentiable forward);
    on a CUDA device this is also where the deterministic-gradient workspace of THAT device is (created and) selected."""
    _GRAD_ZEROS.new_generation()
    if device is not None and device.type == 'cuda' and (_DET_USER[0] is True or (_DET_USER[0] is None and _DET_ENV)) \
            and device not in _DET_WS:
        user = _DET_USER[0]
        set_deterministic(True, device)
        _DET_USER[0] = user


def _grad_zeros(shape, device):
    if not USE_GRAD_ARENA:
        return torch.zeros(shape, dtype=torch.float32, device=device)
    return _GRAD_ZEROS.take(shape, device)


_ZERO_SHIFT = {}


def _zero_shift(n, device):
    """n float32 zeros on `device` (the `shift` operand of the convolution epilogues when only the residual add is wanted)"""
    key = (n, device)
    if key not in _ZERO_SHIFT:
        _ZERO_SHIFT[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return _ZERO_SHIFT[key]


def _conv64(x, w, add=None, stats_part=None):
    """salsa_nn_conv3x3_c64: x (N,64,H,W) bf16 channels-last, w (64,64,3,3) bf16 channels-last -> (N,64,H,W); ``add`` (same
    shape, bf16 channels-last) is added in the kernel's epilogue before the single rounding; ``stats_part`` (float64,
    salsa_nn_conv3x3_c64_stats_blocks x 128) receives the output's per-channel partial sums for the BatchNorm that follows."""
    N, _, H, W = x.shape
    y = torch.empty_like(x, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        if stats_part is not None:
            rc = _lib.load().salsa_nn_conv3x3_c64_stats(_ptr(x), _ptr(w), _ptr(y), _ptr(stats_part), N, H, W, _stream(x))
        elif add is None:
            rc = _lib.load().salsa_nn_conv3x3_c64(_ptr(x), _ptr(w), _ptr(y), N, H, W, _stream(x))
        else:
            rc = _lib.load().salsa_nn_conv3x3_c64_bias_act(_ptr(x), _ptr(w), _ptr(_zero_shift(64, x.device)), _ptr(add), _ptr(y), 0,
                                                           N, H, W, _stream(x))
    if rc:
        raise RuntimeError('salsa_nn_conv3x3_c64 failed (%d)' % rc)
    return y


class _Conv3x3C64(torch.autograd.Function):
    """64 -> 64 channel 3x3 convolution on the matrix cores (salsa_amd/csrc/conv_mfma.hip): forward, data gradient (the same
    kernel with the flipped / transposed filter) and weight gradient (float32, straight into the float32 parameter's grad)."""

    @staticmethod
    def forward(ctx, x, weight, wb=None, wbt=None, skip=False, stats_part=None):
        # wb / wbt: the bf16 filter and its flipped / transposed twin from the model's ConvFilterBank, when there is one.
        # stats_part: a float64 buffer the kernel fills with its output's per-channel partial sums (not differentiated).
        # skip: also return x itself (an alias) for the residual branch, so that the branch's gradient comes back INTO this
        # node and is added in the data-gradient kernel's epilogue instead of by a separate autograd add over the tensor
        if wb is None:
            wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ctx.save_for_backward(x, wb, wbt)
        y = _conv64(x, wb, stats_part=stats_part)
        return (y, x) if skip else y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        x, wb, wbt = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if gskip is not None:
                gskip = gskip.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            gx = _conv64(gy, wbt if wbt is not None else wb.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last),
                         add=gskip)
        if ctx.needs_input_grad[1]:
            N, _, H, W = x.shape
            # (Cout, Cin, 3, 3) with channels-last memory, zeroed in place: zeros(...).contiguous(channels_last) is a fill + a copy
            gw = _grad_zeros((64, 3, 3, 64), x.device).permute(0, 3, 1, 2)
            with torch.cuda.device(x.device):
                rc = _lib.load().salsa_nn_conv3x3_c64_wrw(_ptr(x), _ptr(gy), _ptr(gw), N, H, W, _stream(x))
            if rc:
                raise RuntimeError('salsa_nn_conv3x3_c64_wrw failed (%d)' % rc)
        return gx, gw, None, None, None, None


def _conv_wide(x, w, add=None, stats_part=None):
    """salsa_nn_conv3x3_wide: x (N,Cin,H,W) bf16 channels-last, w (Cout,Cin,3,3) bf16 channels-last -> (N,Cout,H,W); ``add``
    (the output's shape, bf16 channels-last) is added in the kernel's epilogue before the single rounding; ``stats_part``
    (float64, salsa_nn_conv3x3_wide_stats_blocks x 2 x Cout) receives the output's per-channel partial sums (training)."""
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = torch.empty((N, Cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        if stats_part is not None:
            assert add is None
            rc = _lib.load().salsa_nn_conv3x3_wide_stats(_ptr(x), _ptr(w), _ptr(y), _ptr(stats_part), N, H, W, Cin, Cout, _stream(x))
        elif add is None:
            rc = _lib.load().salsa_nn_conv3x3_wide(_ptr(x), _ptr(w), _ptr(y), N, H, W, Cin, Cout, _stream(x))
        else:
            rc = _lib.load().salsa_nn_conv3x3_wide_bias_act(_ptr(x), _ptr(w), _ptr(_zero_shift(Cout, x.device)), _ptr(add), _ptr(y), 0,
                                                            N, H, W, Cin, Cout, _stream(x))
    if rc:
        raise RuntimeError('salsa_nn_conv3x3_wide failed (%d)' % rc)
    return y


import collections  # noqa: E402
_WIDE_TABLES = collections.OrderedDict()     # LRU, at most _WIDE_TABLES_MAX map shapes (ragged last batches / variable chunk
_WIDE_TABLES_MAX = 12                        # lengths would otherwise grow device memory without bound); 3 shapes per model
USE_HIP_CONV_WIDE_WRW = os.environ.get('SALSA_HIP_CONV_WIDE_WRW', '1') != '0'


def _wide_tables(N, H, W, device):
    """(vpos, inv, tile bounds) int32 device tensors of a map shape for salsa_nn_conv3x3_wide_wrw, built once on the host and cached."""
    key = (int(N), int(H), int(W), str(device))
    if key not in _WIDE_TABLES:
        import numpy as np
        L = _lib.load()
        vpos = np.empty(N * H * W, np.int32)
        inv = np.empty(int(L.salsa_nn_conv3x3_wide_table_len(N, H, W)), np.int32)
        tb = np.empty(2 * int(L.salsa_nn_conv3x3_wide_tile_count(N, H, W)), np.int32)
        rc = L.salsa_nn_conv3x3_wide_tables(N, H, W, vpos.ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p),
                                            tb.ctypes.data_as(C.c_void_p))
        if rc:
            raise RuntimeError('salsa_nn_conv3x3_wide_tables failed (%d)' % rc)
        _WIDE_TABLES[key] = tuple(torch.from_numpy(a).to(device) for a in (vpos, inv, tb))
        while len(_WIDE_TABLES) > _WIDE_TABLES_MAX:
            _WIDE_TABLES.popitem(last=False)
    _WIDE_TABLES.move_to_end(key)
    return _WIDE_TABLES[key]


def _conv_wide_wrw(x, gy):
    """salsa_nn_conv3x3_wide_wrw: x (N,Cin,H,W), gy (N,Cout,H,W) bf16 channels-last -> dW (Cout,Cin,3,3) float32."""
    N, Cin, H, W = x.shape
    Cout = gy.shape[1]
    vpos, inv, tb = _wide_tables(N, H, W, x.device)
    gw = _grad_zeros((Cout, 3, 3, Cin), x.device).permute(0, 3, 1, 2)   # channels-last memory
    with torch.cuda.device(x.device):
        rc = _lib.load().salsa_nn_conv3x3_wide_wrw(_ptr(x), _ptr(gy), _ptr(gw), _ptr(vpos), _ptr(inv), _ptr(tb), N, H, W, Cin, Cout,
                                                   _stream(x))
    if rc:
        raise RuntimeError('salsa_nn_conv3x3_wide_wrw failed (%d)' % rc)
    return gw


class _Conv3x3Wide(torch.autograd.Function):
    """The wide 3x3 convolutions (Cin, Cout in 64..512, not 64 -> 64) on the flattened-pixel implicit-GEMM MFMA kernels
    (salsa_amd/csrc/conv_wide.hip): forward, data gradient (the same kernel with the flipped / transposed filter) and weight
    gradient (transposing LDS reads, float32 result); shapes a kernel does not take fall back to torch / MIOpen."""

    @staticmethod
    def forward(ctx, x, weight, wb=None, wbt=None, skip=False, stats_part=None):
        if wb is None:
            wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ctx.save_for_backward(x, wb, wbt)
        # stats_part: a float64 buffer the kernel fills with its output's per-channel partial sums (not differentiated)
        y = _conv_wide(x, wb, stats_part=stats_part)
        return (y, x) if skip else y                                         # (skip: see _Conv3x3C64)

    @staticmethod
    def backward(ctx, gy, gskip=None):
        x, wb, wbt = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = None
        N, Cin, H, W = x.shape
        # the data gradient is a Cout -> Cin convolution: the kernel takes it when Cin is a multiple of 64 (always in the CRNN)
        own_dgrad = ctx.needs_input_grad[0] and bool(_lib.load().salsa_nn_conv3x3_wide_supported(N, H, W, wb.shape[0], Cin))
        if own_dgrad:
            if gskip is not None:
                gskip = gskip.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            gx = _conv_wide(gy, wbt if wbt is not None else wb.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last),
                            add=gskip)
        own_wrw = (ctx.needs_input_grad[1] and USE_HIP_CONV_WIDE_WRW
                   and bool(_lib.load().salsa_nn_conv3x3_wide_wrw_supported(N, H, W, Cin, wb.shape[0])))
        if own_wrw:
            gw = _conv_wide_wrw(x, gy)
        need = [ctx.needs_input_grad[0] and not own_dgrad, ctx.needs_input_grad[1] and not own_wrw, False]
        if need[0] or need[1]:
            r = torch.ops.aten.convolution_backward(gy, x, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, need)
            gx = r[0] if need[0] else gx
            gw = r[1].float() if need[1] else gw
            if need[0] and gskip is not None:
                gx = gx + gskip
        return gx, gw, None, None, None, None


def _planar_rows(x):
    """(N, C, H, W) view whose rows are contiguous and whose planes / batches do not overlap (e.g. a time crop of NCHW)."""
    N, Cn, H, W = x.shape
    return (x.stride(3) == 1 and x.stride(2) == W and x.stride(1) >= H * W and x.stride(0) >= Cn * x.stride(1)) or x.numel() == 0


def _stem_filter(weight, scale=None):
    """(64, Cin <= 8, 3, 3) -> the stem kernel's bf16 [64][10 taps][8 channels] (zero-padded), optionally scaled per output
    channel (folded BatchNorm)."""
    w = weight.detach().float()
    if scale is not None:
        w = w * scale[:, None, None, None]
    wq = torch.zeros((64, 10, 8), dtype=torch.float32, device=w.device)
    wq[:, :9, :w.shape[1]] = w.permute(0, 2, 3, 1).reshape(64, 9, w.shape[1])
    return wq.to(torch.bfloat16).contiguous()


def _conv_stem(x, wq, shift=None, relu=False):
    """salsa_nn_conv3x3_stem: x (N, Cin, H, W) float32, planar with contiguous rows -> (N, 64, H, W) bf16 channels-last."""
    N, Cin, H, W = x.shape
    y = torch.empty((N, 64, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().salsa_nn_conv3x3_stem(_ptr(x), x.stride(0), x.stride(1), _ptr(wq), _ptr(shift), _ptr(y), int(relu), N, Cin,
                                               H, W, _stream(x))
    if rc:
        raise RuntimeError('salsa_nn_conv3x3_stem failed (%d)' % rc)
    return y


USE_HIP_STEM_WRW = os.environ.get('SALSA_HIP_STEM_WRW', '1') != '0'


class _Conv3x3Stem(torch.autograd.Function):
    """The first layer (7 -> 64) on the stem kernels: forward, and the weight gradient straight from the float32 planar input
    (salsa_nn_conv3x3_stem_wrw; 8 input channels: MIOpen).  The input needs no gradient."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return _conv_stem(x, _stem_filter(weight))

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            raise RuntimeError('the stem convolution does not differentiate its input')
        if ctx.needs_input_grad[1] and USE_HIP_STEM_WRW and x.shape[1] <= 7:
            N, Cin, H, W = x.shape
            gy = gy.contiguous(memory_format=torch.channels_last)
            gw = _grad_zeros((64, Cin, 3, 3), x.device)
            with torch.cuda.device(x.device):
                rc = _lib.load().salsa_nn_conv3x3_stem_wrw(_ptr(x), x.stride(0), x.stride(1), _ptr(gy), _ptr(gw), N, Cin, H, W, _stream(x))
            if rc:
                raise RuntimeError('salsa_nn_conv3x3_stem_wrw failed (%d)' % rc)
        elif ctx.needs_input_grad[1]:
            xb = x.to(dtype=torch.bfloat16, memory_format=torch.channels_last)   # cast + layout in one pass
            wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            gw = torch.ops.aten.convolution_backward(gy.contiguous(memory_format=torch.channels_last), xb, wb, None, [1, 1], [1, 1],
                                                     [1, 1], False, [0, 0], 1, [False, True, False])[1].float()
        return gx, gw


USE_HIP_CONV_1X1 = os.environ.get('SALSA_HIP_CONV_1X1', '1') != '0'


def _conv1x1_hip(x, w):
    """salsa_nn_conv1x1: x (N,Cin,H,W) bf16 channels-last, w (Cout,Cin,1,1) bf16 -> (N,Cout,H,W) bf16 channels-last."""
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = torch.empty((N, Cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().salsa_nn_conv1x1(_ptr(x), _ptr(w), _ptr(y), N * H * W, Cin, Cout, _stream(x))
    if rc:
        raise RuntimeError('salsa_nn_conv1x1 failed (%d)' % rc)
    return y


class _Conv1x1(torch.autograd.Function):
    """The residual shortcuts' 1x1 convolutions (salsa_amd/csrc/conv_1x1.hip): forward, data gradient (the same kernel with
    the transposed filter) and weight gradient (float32); a shape a kernel does not take goes to torch / MIOpen."""

    @staticmethod
    def forward(ctx, x, weight, wb=None, wbt=None):
        if wb is None:
            wb = weight.detach().to(torch.bfloat16).contiguous()
        ctx.save_for_backward(x, wb, wbt)
        return _conv1x1_hip(x, wb)

    @staticmethod
    def backward(ctx, gy):
        x, wb, wbt = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        N, Cin, H, W = x.shape
        Cout, M = wb.shape[0], N * H * W
        L = _lib.load()
        gx = gw = None
        own_dgrad = ctx.needs_input_grad[0] and bool(L.salsa_nn_conv1x1_supported(M, Cout, Cin))
        if own_dgrad:
            gx = _conv1x1_hip(gy, wbt if wbt is not None else wb.transpose(0, 1).contiguous())
        own_wrw = ctx.needs_input_grad[1] and bool(L.salsa_nn_conv1x1_wrw_supported(M, Cin, Cout))
        if own_wrw:
            # (channels-last strides, like the trainer's weights: literally different strides for the same bytes of a 1 x 1 filter
            # send DistributedDataParallel's bucket copy down its slow path -- 'Grad strides do not match bucket view strides')
            gw = _grad_zeros((Cout, Cin, 1, 1), x.device).as_strided((Cout, Cin, 1, 1), (Cin, 1, Cin, Cin))
            with torch.cuda.device(x.device):
                rc = L.salsa_nn_conv1x1_wrw(_ptr(x), _ptr(gy), _ptr(gw), M, Cin, Cout, _stream(x))
            if rc:
                raise RuntimeError('salsa_nn_conv1x1_wrw failed (%d)' % rc)
        need = [ctx.needs_input_grad[0] and not own_dgrad, ctx.needs_input_grad[1] and not own_wrw, False]
        if need[0] or need[1]:
            r = torch.ops.aten.convolution_backward(gy, x, wb, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, need)
            gx = r[0] if need[0] else gx
            gw = r[1].float() if need[1] else gw
        return gx, gw, None, None


class Conv1x1(torch.nn.Conv2d):
    """nn.Conv2d(cin, cout, 1, bias=False) -- same parameters and state_dict keys -- whose bf16 channels-last CUDA forward
    (the trainer's autocast) runs the hand-written GEMM kernels; everything else is F.conv2d."""

    def _hip_eligible(self, x):
        bf16 = x.dtype == torch.bfloat16 or (x.is_cuda and torch.is_autocast_enabled('cuda') and
                                             torch.get_autocast_dtype('cuda') == torch.bfloat16)
        return (USE_HIP_CONV and USE_HIP_CONV_1X1 and x.is_cuda and bf16 and x.dim() == 4 and self.kernel_size == (1, 1)
                and self.stride == (1, 1) and self.padding == (0, 0) and self.bias is None and self.dilation == (1, 1)
                and self.groups == 1 and self.weight.dtype == torch.float32
                and _lib.load().salsa_nn_conv1x1_supported(x.shape[0] * x.shape[2] * x.shape[3], self.in_channels, self.out_channels))

    def forward(self, x):
        if self._hip_eligible(x):
            with torch.autocast('cuda', enabled=False):
                return _Conv1x1.apply(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last), self.weight,
                                      *_bank_filters(self))
        return super().forward(x)


USE_STEM_FUSED_BWD = os.environ.get('SALSA_STEM_FUSED_BWD', '1') != '0'
# ... and the BatchNorm backward's reduction in the same pass (round 5: salsa_nn_conv3x3_stem_wrw_bnf); 0: reduce launch + coefficient table
USE_STEM_BN_REDUCE_FUSED = os.environ.get('SALSA_STEM_BN_REDUCE_FUSED', '1') != '0'


class _StemConvBnRelu(torch.autograd.Function):
    """relu(batch_norm(conv7x64(x))) of the network's first layer in training, as ONE autograd node, so that its backward can
    skip the BatchNorm backward's apply pass: dx (the gradient of the convolution's output) has a single reader, this layer's
    weight gradient, which forms it on the fly from (g, conv output, coefficients) while staging its tiles
    (salsa_nn_conv3x3_stem_wrw_bn) -- the 524-MB dx is neither written nor read."""

    @staticmethod
    def forward(ctx, x, conv_w, bn_w, bn_b, running_mean, running_var, momentum, eps, batches_tracked):
        L = _lib.load()
        N, Cin, H, W = x.shape
        nb = L.salsa_nn_conv3x3_stem_stats_blocks(N, H, W) if USE_CONV_STATS else 0
        part = None
        if nb > 0:  # the persistent launch that also leaves the BatchNorm's statistics (no statistics pass over the 524 MB)
            x1 = torch.empty((N, 64, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
            part = torch.empty(nb * 128, dtype=torch.float64, device=x.device)
            with torch.cuda.device(x.device):
                rc = L.salsa_nn_conv3x3_stem_stats(_ptr(x), x.stride(0), x.stride(1), _ptr(_stem_filter(conv_w)), _ptr(x1), _ptr(part),
                                                   N, Cin, H, W, _stream(x))
            if rc:
                raise RuntimeError('salsa_nn_conv3x3_stem_stats failed (%d)' % rc)
        else:
            x1 = _conv_stem(x, _stem_filter(conv_w))
        N, Cn, H, W = x1.shape
        M = N * H * W
        y = torch.empty_like(x1, memory_format=torch.channels_last)
        save = torch.empty((2, Cn), dtype=torch.float32, device=x.device)
        ws = torch.empty(L.salsa_nn_bn_workspace_bytes(1, M, Cn) // 8 + 1, dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = L.salsa_nn_bn_train_fwd(_ptr(x1), _ptr(y), None, 1, M, Cn, _ptr(bn_w), _ptr(bn_b), float(eps), float(momentum),
                                         _ptr(running_mean), _ptr(running_var), _ptr(save[0]), _ptr(save[1]), _ptr(ws), 1, 0.0, 0,
                                         _ptr(batches_tracked), _ptr(part), nb if part is not None else 0, _stream(x))
        if rc:
            raise RuntimeError('salsa_nn_bn_train_fwd failed (%d)' % rc)
        ctx.save_for_backward(x, x1, bn_w, bn_b, save)
        return y

    @staticmethod
    def backward(ctx, g):
        x, x1, bn_w, bn_b, save = ctx.saved_tensors
        L = _lib.load()
        N, Cn, H, W = x1.shape
        M, Cin = N * H * W, x.shape[1]
        g = g.contiguous(memory_format=torch.channels_last)
        dwb = torch.empty((2, Cn), dtype=torch.float32, device=x.device)
        ws = torch.empty(L.salsa_nn_bn_workspace_bytes(1, M, Cn) // 8 + 1, dtype=torch.float64, device=x.device)
        coef = torch.empty(7 * Cn, dtype=torch.float32, device=x.device)
        gw = _grad_zeros((64, Cin, 3, 3), x.device)
        if USE_STEM_BN_REDUCE_FUSED:   # the BatchNorm backward's reduction inside the weight-gradient pass as well (one read of g, x1)
            nbytes = L.salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes(N, H, W)
            slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                rc = L.salsa_nn_conv3x3_stem_wrw_bnf(_ptr(x), x.stride(0), x.stride(1), _ptr(g), _ptr(x1), _ptr(save[0]), _ptr(save[1]),
                                                     _ptr(bn_w), _ptr(bn_b), 1, _ptr(gw), _ptr(dwb[0]), _ptr(dwb[1]), _ptr(slabs), nbytes,
                                                     N, Cin, H, W, _stream(x))
            if rc:
                raise RuntimeError('salsa_nn_conv3x3_stem_wrw_bnf failed (%d)' % rc)
            return None, gw, dwb[0], dwb[1], None, None, None, None, None
        with torch.cuda.device(x.device):
            rc = L.salsa_nn_bn_bwd(_ptr(g), None, _ptr(x1), None, None, 1, M, Cn, _ptr(bn_w), _ptr(bn_b), _ptr(save[0]), _ptr(save[1]),
                                   1, _ptr(dwb[0]), _ptr(dwb[1]), _ptr(ws), _ptr(coef), 0.0, 0, _stream(x))
            if rc:
                raise RuntimeError('salsa_nn_bn_bwd (coefficients only) failed (%d)' % rc)
            rc = L.salsa_nn_conv3x3_stem_wrw_bn(_ptr(x), x.stride(0), x.stride(1), _ptr(g), _ptr(x1), _ptr(coef), 1, _ptr(gw), N, Cin,
                                                H, W, _stream(x))
        if rc:
            raise RuntimeError('salsa_nn_conv3x3_stem_wrw_bn failed (%d)' % rc)
        return None, gw, dwb[0], dwb[1], None, None, None, None, None


USE_FILTER_BANK = os.environ.get('SALSA_FILTER_BANK', '1') != '0'

# Cached derivatives of parameters (the filter bank, the folded inference filters) must notice every parameter update.  The
# tensors' version counters are NOT enough: the fused optimizers (torch.optim.Adam(fused=True), which the trainer uses) update
# parameters in place without bumping them.  A global optimizer hook counts the steps of ANY optimizer; the caches compare
# this epoch (and the version counters, for updates made without an optimizer, and the storage pointers).
_PARAM_EPOCH = [0]


def _bump_param_epoch(*_):
    _PARAM_EPOCH[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402
_register_step_hook(_bump_param_epoch)   # (an import side effect: a process-global hook on EVERY optimizer's step())


def invalidate_conv_caches(model=None):
    """Declare every cached derivative of the convolution / BatchNorm parameters stale (the bf16 filter bank, the folded
    inference filters): the next forward rebuilds them.  The caches notice optimizer steps (any torch optimizer),
    ``load_state_dict`` (a post-hook on the encoder), in-place tensor operations that bump version counters and moved storages
    by themselves.  What they CANNOT see is an update that goes through none of those: ``p.data.copy_(...)`` / ``p.data.mul_``
    (``.data`` has its own version counter), an EMA weight swap written that way, or a raw-pointer / custom-kernel updater.
    Call this after such an update.  ``model`` is accepted for symmetry with torch APIs; the epoch is process-global."""
    _bump_param_epoch()


class ConvFilterBank:
    """bf16 working copies of the hand-written 3x3 convolutions' float32 master filters -- the forward kernels' channels-last
    layout and the data-gradient kernels' flipped / transposed one -- produced for ALL layers by one launch
    (salsa_nn_conv_filter_bank) instead of four small torch kernels per layer per step (autocast's cast, the channels-last
    copy, flip, transpose copy).  A layer asking for its filters compares its weight's version counter (every in-place update,
    bumps it) and the global count of optimizer steps (``_PARAM_EPOCH``: fused optimizers do not bump version counters) with
    the ones recorded at the last refresh; the first stale layer refreshes the whole bank.  The views handed out alias the bank: a refresh between a forward and its backward (a weight modified in place
    mid-graph, which autograd itself rejects) would change what that backward reads."""

    def __init__(self, convs):
        self.convs = [c for c in convs if ((isinstance(c, Conv3x3) and c.kernel_size == (3, 3)) or
                                           (isinstance(c, Conv1x1) and c.kernel_size == (1, 1)))
                      and c.in_channels % 32 == 0 and c.out_channels % 32 == 0 and c.groups == 1]
        for i, c in enumerate(self.convs):
            c._bank = (self, i)
        self._ptrs, self._epoch = None, -1

    def _build(self):
        dev = self.convs[0].weight.device
        sizes = [c.weight.numel() for c in self.convs]
        self._flat = torch.empty(2 * sum(sizes), dtype=torch.bfloat16, device=dev)
        rows, off, blk, self._fwd, self._bwd = [], 0, 0, [], []
        for c, n in zip(self.convs, sizes):
            co, ci, k = c.out_channels, c.in_channels, c.kernel_size[0]
            f, b = self._flat[off:off + n], self._flat[off + n:off + 2 * n]
            self._fwd.append(f.view(co, k, k, ci).permute(0, 3, 1, 2))          # (Cout, Cin, k, k), channels-last memory
            self._bwd.append(b.view(ci, k, k, co).permute(0, 3, 1, 2))          # (Cin, Cout, k, k), taps flipped
            rows.append([c.weight.data_ptr(), f.data_ptr(), b.data_ptr(), co, ci, *c.weight.stride(), blk, k * k])
            off, blk = off + 2 * n, blk + (co // 32) * (ci // 32)
        self._desc = torch.tensor(rows, dtype=torch.int64).to(dev)
        self._blocks = blk
        self._ptrs = [c.weight.data_ptr() for c in self.convs]
        self._versions = [-1] * len(self.convs)

    def refresh(self):
        if self._ptrs is None or any(c.weight.data_ptr() != p for c, p in zip(self.convs, self._ptrs)):
            self._build()                                            # first use, or the parameters moved (.to(), load)
        w0 = self.convs[0].weight
        with torch.cuda.device(w0.device):
            rc = _lib.load().salsa_nn_conv_filter_bank(_ptr(self._desc), len(self.convs), self._blocks, _stream(w0))
        if rc:
            raise RuntimeError('salsa_nn_conv_filter_bank failed (%d)' % rc)
        self._versions = [c.weight._version for c in self.convs]
        self._epoch = _PARAM_EPOCH[0]

    def mark_stale(self):
        """Force a refresh at the next request (the encoder calls this at the start of every TRAINING forward: one ~10-us launch
        per step buys independence from any invalidation protocol where it matters most -- a bank that silently stayed on old
        weights trains on them with no error)."""
        self._epoch = -1

    def filters(self, i):
        """(forward filter, data-gradient filter) of layer i as bf16 channels-last views, refreshed if its weight changed."""
        w = self.convs[i].weight
        if (self._ptrs is None or self._epoch != _PARAM_EPOCH[0] or self._versions[i] != w._version
                or self._ptrs[i] != w.data_ptr()):
            self.refresh()
        return self._fwd[i], self._bwd[i]


def _bank_filters(conv):
    """the layer's filters from its model's ConvFilterBank, or (None, None): float32 CUDA master weights only"""
    bank = getattr(conv, '_bank', None)
    w = conv.weight
    if bank is None or not USE_FILTER_BANK or not w.is_cuda or w.dtype != torch.float32:
        return None, None
    return bank[0].filters(bank[1])


USE_FUSED_SKIP = os.environ.get('SALSA_FUSED_SKIP', '1') != '0'
USE_CONV_STATS = os.environ.get('SALSA_CONV_STATS', '1') != '0'   # BatchNorm statistics from the 64 -> 64 convolution's epilogue
C64_STATS_MIN_PIX = int(os.environ.get('SALSA_C64_STATS_MIN_PIX', '0'))   # ... only for maps of at least this many pixels (N H W)
USE_WIDE_CONV_STATS = os.environ.get('SALSA_WIDE_CONV_STATS', '1') != '0'   # the same in the wide kernels' epilogue


class Conv3x3(torch.nn.Conv2d):
    """nn.Conv2d(cin, cout, 3, padding=1, bias=False) whose 64 -> 64 instances run the MFMA kernel for bf16 channels-last
    CUDA inputs (i.e. under the trainer's autocast) and whose (Cin <= 8) -> 64 instance -- the network's first layer -- runs
    the stem kernel on float32 planar inputs under bf16 autocast; everything else is F.conv2d."""

    def _stem_eligible(self, x):
        return (USE_HIP_CONV and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _planar_rows(x)
                and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
                and self.in_channels <= 8 and self.out_channels == 64 and self.kernel_size == (3, 3) and self.stride == (1, 1)
                and self.padding == (1, 1) and self.bias is None and self.dilation == (1, 1) and self.groups == 1
                and not x.requires_grad and x.shape[0] * x.shape[2] * x.shape[3] < 2 ** 31 // 64)

    def _hip_eligible(self, x):
        bf16 = x.dtype == torch.bfloat16 or (x.is_cuda and torch.is_autocast_enabled('cuda') and
                                             torch.get_autocast_dtype('cuda') == torch.bfloat16)
        return (USE_HIP_CONV and x.is_cuda and bf16 and self.in_channels == 64 and self.out_channels == 64 and x.dim() == 4
                and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1) and self.bias is None
                and self.dilation == (1, 1) and self.groups == 1 and x.shape[0] * x.shape[2] * x.shape[3] < 2 ** 31 // 64)

    def _wide_eligible(self, x):
        bf16 = x.dtype == torch.bfloat16 or (x.is_cuda and torch.is_autocast_enabled('cuda') and
                                             torch.get_autocast_dtype('cuda') == torch.bfloat16)
        return (USE_HIP_CONV and USE_HIP_CONV_WIDE and x.is_cuda and bf16 and x.dim() == 4 and self.kernel_size == (3, 3)
                and self.stride == (1, 1) and self.padding == (1, 1) and self.bias is None and self.dilation == (1, 1)
                and self.groups == 1 and not (self.in_channels == 64 and self.out_channels == 64)
                and _lib.load().salsa_nn_conv3x3_wide_supported(x.shape[0], x.shape[2], x.shape[3], self.in_channels,
                                                                self.out_channels))

    def stats_buffer(self, x):
        """A float64 buffer for the output's per-channel partial sums when this call will run the 64 -> 64 or the wide MFMA kernel
        in training (salsa_nn_conv3x3_c64_stats / _wide_stats), else None: pass it to forward / forward_skip and on to the BatchNorm."""
        if USE_CONV_STATS and USE_WIDE_CONV_STATS and torch.is_grad_enabled() and not self._hip_eligible(x) and self._wide_eligible(x):
            nb = _lib.load().salsa_nn_conv3x3_wide_stats_blocks(x.shape[0], x.shape[2], x.shape[3], self.in_channels, self.out_channels)
            return torch.empty(nb * 2 * self.out_channels, dtype=torch.float64, device=x.device) if nb > 0 else None
        if not (USE_CONV_STATS and torch.is_grad_enabled() and self._hip_eligible(x)):
            return None
        if x.shape[0] * x.shape[2] * x.shape[3] < C64_STATS_MIN_PIX:
            return None                                   # (the BatchNorm takes its own statistics pass)
        nb = _lib.load().salsa_nn_conv3x3_c64_stats_blocks(x.shape[0], x.shape[2], x.shape[3])
        return torch.empty(nb * 128, dtype=torch.float64, device=x.device) if nb > 0 else None

    def forward_skip(self, x, stats_part=None):
        """(conv(x), x'): x' is x routed through the convolution's autograd node, for a residual / shortcut branch that forks
        off x -- the branch's gradient is then added inside the data-gradient kernel (training, MFMA kernels only; otherwise
        x' is x and autograd adds as usual)."""
        if USE_FUSED_SKIP and torch.is_grad_enabled() and x.requires_grad and (self._hip_eligible(x) or self._wide_eligible(x)):
            with torch.autocast('cuda', enabled=False):
                xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                if self._hip_eligible(x):
                    return _Conv3x3C64.apply(xb, self.weight, *_bank_filters(self), True, stats_part)
                return _Conv3x3Wide.apply(xb, self.weight, *_bank_filters(self), True, stats_part)
        return self.forward(x, stats_part), x

    def forward(self, x, stats_part=None):
        if self._hip_eligible(x):
            with torch.autocast('cuda', enabled=False):
                return _Conv3x3C64.apply(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last), self.weight,
                                         *_bank_filters(self), False, stats_part)
        if self._wide_eligible(x):
            with torch.autocast('cuda', enabled=False):
                return _Conv3x3Wide.apply(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last), self.weight,
                                          *_bank_filters(self), False, stats_part)
        if self._stem_eligible(x):
            with torch.autocast('cuda', enabled=False):
                return _Conv3x3Stem.apply(x, self.weight)
        return super().forward(x)


USE_GEMM_1X1 = os.environ.get('SALSA_GEMM_1X1', '1') != '0'


def conv1x1(conv, x):
    """A 1x1 / stride 1 convolution (the residual shortcuts, models/model_utils.py:340-349) of a channels-last CUDA tensor as
    what it is -- a plain GEMM over the flattened pixels, [N*H*W, Cin] x [Cin, Cout] -- through torch's linear (hipBLASLt;
    bf16 under autocast) instead of MIOpen's convolution path.  Inference only: in training the GEMM backward (a reduction over
    all N*H*W pixels for the weight gradient) measured 4 % SLOWER per step than MIOpen's (2152 vs 2240 chunks/s)."""
    if (USE_GEMM_1X1 and not (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)) and x.is_cuda and x.dim() == 4 and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.groups == 1 and conv.bias is None and x.is_contiguous(memory_format=torch.channels_last)):
        N, Cin, H, W = x.shape
        y = F.linear(x.permute(0, 2, 3, 1).reshape(N * H * W, Cin), conv.weight.reshape(conv.out_channels, Cin))
        return y.view(N, H, W, conv.out_channels).permute(0, 3, 1, 2)            # (N, Cout, H, W), channels-last strides
    return conv(x)


def _folded(conv, bn, stem=False):
    """(filter, shift) of an eval-mode conv + BatchNorm pair folded for the kernels' epilogue: filter = w * gamma / sigma in
    bf16 (channels-last, or the stem kernel's layout), shift = beta - mean * gamma / sigma in float32.  Cached on the conv
    module and keyed on the version counters of the five tensors involved, the global optimizer-step count (fused optimizers
    do not bump version counters) and the BatchNorm's statistics serial, so repeated inference calls do not re-run the seven
    small kernels per layer."""
    srcs = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    # (_stats_serial: the fused training kernels update the running statistics through raw pointers, invisible to torch's counters)
    key = tuple((t.data_ptr(), t._version) for t in srcs) + (float(bn.eps), stem, getattr(bn, '_stats_serial', 0), _PARAM_EPOCH[0])
    hit = getattr(conv, '_fold_cache', None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    with torch.no_grad():
        scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).float()
        shift = (bn.bias - bn.running_mean * scale).float().contiguous()
        if stem:
            wf = _stem_filter(conv.weight, scale)
        else:
            wf = (conv.weight.float() * scale[:, None, None, None]).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    conv._fold_cache = (key, wf, shift)
    return wf, shift


FOLD_SHORTCUT = os.environ.get('SALSA_FOLD_SHORTCUT', '1') != '0'


def folded_shortcut(conv, bn, x):
    """Eval-mode 1x1 shortcut + BatchNorm (models/model_utils.py:340-349) as ONE GEMM: the BatchNorm's scale goes into the 1x1
    filter and its shift is RETURNED, to be added by the block's last convolution epilogue together with this residual
    (conv_bn_act(..., residual=sc, residual_shift=shift)) -- no normalisation pass over the shortcut tensor.  None when the
    layer does not qualify (the caller then runs bn(conv1x1(x)))."""
    if not (FOLD_SHORTCUT and not bn.training and not torch.is_grad_enabled() and bn.track_running_stats and bn.affine
            and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None
            and x.is_contiguous(memory_format=torch.channels_last)):
        return None
    wf, shift = _folded(conv, bn)
    N, Cin, H, W = x.shape
    with torch.autocast('cuda', enabled=False):
        y = F.linear(x.permute(0, 2, 3, 1).reshape(N * H * W, Cin), wf.reshape(conv.out_channels, Cin))
    return y.view(N, H, W, conv.out_channels).permute(0, 3, 1, 2), shift    # (N, Cout, H, W) with channels-last strides


def conv_bn_act(conv, bn, x, residual=None, relu=True, dropout_p=0.0, pool=False, skip=False, residual_shift=None):
    """``_conv_bn_act`` below; with ``skip=True`` returns (result, x'): x' is x for a branch that forks off x (residual,
    shortcut) -- in training on the MFMA kernels routed through the convolution's autograd node (Conv3x3.forward_skip)."""
    if not skip:
        return _conv_bn_act(conv, bn, x, residual, relu, dropout_p, pool, residual_shift)
    assert residual_shift is None
    if isinstance(conv, Conv3x3) and bn.training and torch.is_grad_enabled() and not pool:
        part = conv.stats_buffer(x) if isinstance(bn, BatchNormAct2d) else None
        c, xs = conv.forward_skip(x, part)
        return (bn(c, residual=residual, relu=relu, dropout_p=dropout_p, stats_part=part) if part is not None
                else bn(c, residual=residual, relu=relu, dropout_p=dropout_p)), xs
    return _conv_bn_act(conv, bn, x, residual, relu, dropout_p, pool), x


def _conv_bn_act(conv, bn, x, residual=None, relu=True, dropout_p=0.0, pool=False, residual_shift=None):
    """dropout(relu(bn(conv(x)) + residual)) of the reference blocks (dropout in training only).  In eval mode, for the 64 -> 64 3x3 convolutions under bf16
    autocast, the BatchNorm is folded into the filter (scale) and a per-channel shift that the MFMA kernel applies -- with the
    residual add and the ReLU -- before its single rounding: the normalised activation never makes a round trip to HBM.
    ``pool=True`` appends the 2x2 average pool of the upstream stem; on that eval path it is taken inside the kernel, so the
    full-resolution activation is not written at all.  Every other case is conv -> BatchNormAct2d (-> avg_pool2x2)."""
    if (isinstance(conv, Conv3x3) and conv._hip_eligible(x) and not bn.training and not torch.is_grad_enabled()
            and bn.track_running_stats and bn.affine
            and (residual is None or (residual.dtype == torch.bfloat16 and residual.shape[1] == 64
                                      and residual.is_contiguous(memory_format=torch.channels_last)))):
        wf, shift = _folded(conv, bn)
        if residual_shift is not None:
            shift = shift + residual_shift               # the folded shortcut BatchNorm's shift rides in the same epilogue add
        xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        N, _, H, W = xb.shape
        fuse_pool = pool and H % 2 == 0 and W % 2 == 0
        y = (torch.empty((N, 64, H // 2, W // 2), dtype=torch.bfloat16, device=xb.device, memory_format=torch.channels_last)
             if fuse_pool else torch.empty_like(xb, memory_format=torch.channels_last))
        fn = _lib.load().salsa_nn_conv3x3_c64_bias_act_pool if fuse_pool else _lib.load().salsa_nn_conv3x3_c64_bias_act
        with torch.cuda.device(x.device):
            rc = fn(_ptr(xb), _ptr(wf), _ptr(shift), _ptr(residual), _ptr(y), int(relu), N, H, W, _stream(xb))
        if rc:
            raise RuntimeError('salsa_nn_conv3x3_c64_bias_act%s failed (%d)' % ('_pool' if fuse_pool else '', rc))
        return y if fuse_pool or not pool else avg_pool2x2(y)
    if (isinstance(conv, Conv3x3) and conv._wide_eligible(x) and not bn.training and not torch.is_grad_enabled()
            and bn.track_running_stats and bn.affine
            and (residual is None or (residual.dtype == torch.bfloat16 and residual.shape[1] == conv.out_channels
                                      and residual.is_contiguous(memory_format=torch.channels_last)))):
        # the wide layers at inference: the same folding on conv_wide.hip's epilogue
        wf, shift = _folded(conv, bn)
        if residual_shift is not None:
            shift = shift + residual_shift
        xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        N, Cin, H, W = xb.shape
        y = torch.empty((N, conv.out_channels, H, W), dtype=torch.bfloat16, device=xb.device, memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            rc = _lib.load().salsa_nn_conv3x3_wide_bias_act(_ptr(xb), _ptr(wf), _ptr(shift), _ptr(residual), _ptr(y), int(relu), N, H, W,
                                                            Cin, conv.out_channels, _stream(xb))
        if rc:
            raise RuntimeError('salsa_nn_conv3x3_wide_bias_act failed (%d)' % rc)
        return avg_pool2x2(y) if pool else y      # (pool: the last layer of a block before a stride-2 block)
    if residual_shift is not None:                       # no folding epilogue on this path: the shift joins the residual here
        residual = (residual.float() + residual_shift.view(1, -1, 1, 1)).to(residual.dtype)
    if (isinstance(conv, Conv3x3) and conv._stem_eligible(x) and residual is None and not bn.training
            and not torch.is_grad_enabled() and bn.track_running_stats and bn.affine):
        wq, shift = _folded(conv, bn, stem=True)
        y = _conv_stem(x, wq, shift, relu)
        return avg_pool2x2(y) if pool else y
    if (USE_STEM_FUSED_BWD and USE_HIP_STEM_WRW and USE_HIP_BN and isinstance(conv, Conv3x3) and isinstance(bn, BatchNormAct2d)
            and bn.training and torch.is_grad_enabled() and relu and residual is None and dropout_p == 0.0 and not pool
            and conv._stem_eligible(x) and x.shape[1] <= 7 and bn.affine and bn.track_running_stats and bn.momentum is not None
            and conv.weight.requires_grad and _lib.load().salsa_nn_bn_supported(1, x.shape[0] * x.shape[2] * x.shape[3], 64)):
        bn._stats_serial = getattr(bn, '_stats_serial', 0) + 1
        with torch.autocast('cuda', enabled=False):
            return _StemConvBnRelu.apply(x, conv.weight, bn.weight.float(), bn.bias.float(), bn.running_mean, bn.running_var,
                                         bn.momentum, bn.eps, bn.num_batches_tracked)
    part = (conv.stats_buffer(x) if isinstance(conv, Conv3x3) and isinstance(bn, BatchNormAct2d) and bn.training else None)
    if pool and relu and dropout_p == 0.0 and isinstance(bn, BatchNormAct2d) and (residual is None or USE_HIP_BN_RES_POOL):
        return (bn.relu_pool(conv(x, part), part, residual) if part is not None else bn.relu_pool(conv(x), None, residual))
    if part is not None:
        y = bn(conv(x, part), residual=residual, relu=relu, dropout_p=dropout_p, stats_part=part)
    else:
        y = bn(conv(x), residual=residual, relu=relu, dropout_p=dropout_p)
    return avg_pool2x2(y) if pool else y
