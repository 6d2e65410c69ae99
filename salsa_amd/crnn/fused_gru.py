"""Bidirectional GRU layer whose time recurrence is ONE hand-written HIP launch (salsa_amd/csrc/gru_scan.hip, C ABI in
include/salsa_gru.h) instead of ~500 tiny library kernels per layer per step.  The dense parts stay GEMMs: the input
projection W_ih x + b_ih for all timesteps before the scan, dW_hh / dW_ih after the backward scan.  Same float32
arithmetic as torch.nn.GRU (gate order r, z, n); parameters are read from an nn.GRU so state dicts are unchanged.
Without gradients (inference) the scan keeps W_hh in registers as float16 instead of streaming it from L2 every step."""
import ctypes as C
import os

import torch

from .. import _lib


def _stream(t):
    """the current stream of the tensor's device (launches are also wrapped in torch.cuda.device(t.device): the kernels
    must run on the GPU that owns the pointers even when another device is current)"""
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


REGISTER_WEIGHTS = os.environ.get('SALSA_GRU_REGW', '1') != '0'


def _scan_inference(gi, whh, bhh):
    """salsa_gru_scan_fwd_regw: no-grad scan with the recurrent weights held in registers (float16), H = 256."""
    T, B, D, H3 = gi.shape
    H = H3 // 3
    hs = torch.empty((T, B, D, H), dtype=torch.float32, device=gi.device)
    with torch.cuda.device(gi.device):
        rc = _lib.load().salsa_gru_scan_fwd_regw(C.c_void_p(gi.data_ptr()), C.c_void_p(whh.detach().contiguous().data_ptr()),
                                                 C.c_void_p(bhh.detach().contiguous().data_ptr()), C.c_void_p(hs.data_ptr()),
                                                 None, T, B, D, H, _stream(gi))
    if rc:
        raise RuntimeError('salsa_gru_scan_fwd_regw failed (%d)' % rc)
    return hs


_ONES = {}
LEAN = os.environ.get('SALSA_GRU_LEAN', '1') != '0'   # GEMM / GEMV forms of the projection and bias gradients (0: einsum + reductions)
HIP_COLSUM = os.environ.get('SALSA_HIP_COLSUM', '1') != '0'   # bias gradients by salsa_nn_colsum2 (0: the ones-vector GEMV)


def _bias_grad(g2, shape):
    """sum over the rows of g2 (T*B, D*3H) float32 contiguous -> `shape`: salsa_nn_colsum2 into a zero slice of the backward
    pass's gradient pool (the ones-vector GEMV took 17 us per call, four calls per step)."""
    if not (HIP_COLSUM and g2.is_cuda and g2.dtype == torch.float32 and g2.is_contiguous()):
        return torch.mv(g2.t(), _ones(g2.shape[0], g2.device)).view(shape)
    from .nn_ops import _grad_zeros
    out = _grad_zeros(tuple(shape), g2.device)
    with torch.cuda.device(g2.device):
        rc = _lib.load().salsa_nn_colsum2(C.c_void_p(g2.data_ptr()), None, C.c_void_p(out.data_ptr()), None, g2.shape[0], g2.shape[1],
                                          _stream(g2))
    if rc:
        raise RuntimeError('salsa_nn_colsum2 failed (%d)' % rc)
    return out


def _ones(n, device):
    key = (n, device)
    if key not in _ONES:
        if len(_ONES) >= 16:                 # bounded: variable batch sizes / sequence lengths must not accumulate forever
            _ONES.clear()
        _ONES[key] = torch.ones(n, dtype=torch.float32, device=device)
    return _ONES[key]


class _InputProjection(torch.autograd.Function):
    """gi[t, b, d] = W_ih[d] x[t, b] + b_ih[d] for both directions (D, 3H, In): the einsum + bias of the scan's input, with
    the bias gradient as a ones-row GEMV instead of autograd's reduction over (t, b).  TIME-MAJOR in and out (x (T, B, In) ->
    gi (T, B, D, 3H)): the GEMM's rows are already in the scan's order, so neither gi nor its gradient is transposed (two 7.8-MB
    copies per layer per step when x was batch-major)."""

    @staticmethod
    def forward(ctx, x, wih, bih):
        ctx.save_for_backward(x, wih)
        T, B, In = x.shape
        D, G, _ = wih.shape
        gi = torch.addmm(bih.reshape(1, D * G), x.reshape(T * B, In), wih.reshape(D * G, In).t())      # (T*B, D*3H)
        return gi.view(T, B, D, G)

    @staticmethod
    def backward(ctx, dgi):
        x, wih = ctx.saved_tensors
        T, B, In = x.shape
        D, G, _ = wih.shape
        g2 = dgi.reshape(T * B, D * G)                                                                 # rows as in x
        dx = dwih = dbih = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(g2, wih.reshape(D * G, In)).view(T, B, In)
        if ctx.needs_input_grad[1]:
            dwih = torch.mm(g2.t(), x.reshape(T * B, In)).view(D, G, In)
        if ctx.needs_input_grad[2]:
            dbih = _bias_grad(g2, (D, G))
        return dx, dwih, dbih


class _GruScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gi, whh, bhh, half_weights=False):
        """gi (T,B,D,3H) float32 contiguous; whh (D,3H,H); bhh (D,3H) -> hs (T,B,D,H).  half_weights: the register-resident
        kernels (W_hh rounded to float16, forward and backward consistently) instead of the float32 streaming ones."""
        T, B, D, H3 = gi.shape
        H = H3 // 3
        L = _lib.load()
        whh = whh.contiguous()
        hs = torch.empty((T, B, D, H), dtype=torch.float32, device=gi.device)
        need_grad = gi.requires_grad or whh.requires_grad or bhh.requires_grad
        saved = torch.empty((T, B, D, 4 * H), dtype=torch.float32, device=gi.device) if need_grad else None
        sv = C.c_void_p(saved.data_ptr() if saved is not None else 0)
        with torch.cuda.device(gi.device):
            if half_weights:
                rc = L.salsa_gru_scan_fwd_regw(C.c_void_p(gi.data_ptr()), C.c_void_p(whh.data_ptr()), C.c_void_p(bhh.contiguous().data_ptr()),
                                               C.c_void_p(hs.data_ptr()), sv, T, B, D, H, _stream(gi))
            else:
                whh_t = whh.transpose(1, 2).contiguous()
                rc = L.salsa_gru_scan_fwd(C.c_void_p(gi.data_ptr()), C.c_void_p(whh_t.data_ptr()), C.c_void_p(bhh.contiguous().data_ptr()),
                                          C.c_void_p(hs.data_ptr()), sv, T, B, D, H, _stream(gi))
        if rc:
            raise RuntimeError('salsa_gru_scan_fwd failed (%d)' % rc)
        if need_grad:
            ctx.save_for_backward(whh, hs, saved)
        ctx.half_weights = bool(half_weights)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        whh, hs, saved = ctx.saved_tensors
        T, B, D, H = hs.shape
        L = _lib.load()
        dhs = dhs.contiguous()
        dgi = torch.empty((T, B, D, 3 * H), dtype=torch.float32, device=hs.device)
        dgh = torch.empty_like(dgi)
        scan = L.salsa_gru_scan_bwd_regw if ctx.half_weights else L.salsa_gru_scan_bwd
        with torch.cuda.device(hs.device):
            rc = scan(C.c_void_p(dhs.data_ptr()), C.c_void_p(whh.data_ptr()), C.c_void_p(hs.data_ptr()),
                      C.c_void_p(saved.data_ptr()), C.c_void_p(dgi.data_ptr()), C.c_void_p(dgh.data_ptr()), T, B, D, H, _stream(hs))
        if rc:
            raise RuntimeError('salsa_gru_scan_bwd failed (%d)' % rc)
        # dW_hh[d] = sum_t dgh[t, :, d]^T h_before[t, :, d]: h before step t is hs[t - 1] in the forward direction and hs[t + 1] in
        # the reverse one (zero at the scan's first step), so each direction is ONE GEMM on shifted, strided VIEWS of dgh and hs --
        # no zero-filled shifted copy of hs; the bias gradient is a ones-row GEMV (torch's reduce over the two leading axes of
        # a 4-D tensor took 21 us per call, the GEMV takes a few)
        if not LEAN:
            hprev = torch.zeros_like(hs)                   # h before each step, per direction's scan order
            hprev[1:, :, 0] = hs[:-1, :, 0]
            if D > 1:
                hprev[:-1, :, 1] = hs[1:, :, 1]
            return dgi, torch.einsum('tbdr,tbdk->drk', dgh, hprev), dgh.sum(dim=(0, 1)), None
        dwhh = torch.empty((D, 3 * H, H), dtype=torch.float32, device=hs.device)
        if T > 1:
            torch.mm(dgh[1:, :, 0].reshape(-1, 3 * H).t(), hs[:-1, :, 0].reshape(-1, H), out=dwhh[0])
            if D > 1:
                torch.mm(dgh[:-1, :, 1].reshape(-1, 3 * H).t(), hs[1:, :, 1].reshape(-1, H), out=dwhh[1])
        else:
            dwhh.zero_()
        dbhh = _bias_grad(dgh.view(T * B, D * 3 * H), (D, 3 * H))
        return dgi, dwhh, dbhh, None


def bigru_forward(gru: torch.nn.GRU, x: torch.Tensor, training: bool, half_weights: bool = False) -> torch.Tensor:
    """x (B,T,In) float32 CUDA -> (B,T,2H); equivalent to ``gru(x)[0]`` for a batch_first bidirectional nn.GRU.
    half_weights (the caller is under bf16 autocast): run the register-resident kernels, i.e. with W_hh rounded to float16
    in the recurrence -- finer than the bf16 autocast would give nn.GRU -- for training and for the no-gradient forward
    alike.  A float32 caller (half_weights=False) gets the float32 streaming kernels in both modes, so pure-float32
    evaluation matches float32 training and the reference's nn.GRU."""
    assert gru.batch_first and gru.bidirectional and gru.bias
    out = x.transpose(0, 1).contiguous()        # time-major between the layers: the scans' order (one copy in, one view out)
    for layer in range(gru.num_layers):
        names = ['_l%d' % layer, '_l%d_reverse' % layer]
        from .nn_ops import stack_groups
        wih, whh, bih, bhh = stack_groups([[getattr(gru, kind + n) for n in names]   # (D,3H,In), (D,3H,H), (D,3H), (D,3H):
                                           for kind in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')])   # one copy launch
        if layer > 0 and training and gru.dropout > 0:
            out = torch.nn.functional.dropout(out, p=gru.dropout, training=True)
        gi = _InputProjection.apply(out, wih, bih) if (out.is_cuda and LEAN) else \
            (torch.einsum('tbi,dgi->tbdg', out, wih) + bih).contiguous()           # (T,B,D,3H)
        no_grad = not (torch.is_grad_enabled() and (gi.requires_grad or whh.requires_grad))
        if REGISTER_WEIGHTS and half_weights and whh.shape[2] == 256 and no_grad:
            hs = _scan_inference(gi, whh, bhh)                                      # W_hh (float16) resident in registers
        else:
            hs = _GruScan.apply(gi, whh, bhh, bool(half_weights and REGISTER_WEIGHTS and whh.shape[2] == 256))   # (T,B,D,H)
        out = hs.view(hs.shape[0], hs.shape[1], -1)                                 # (T, B, D*H)
    return out.transpose(0, 1)                                                      # (B, T, 2H), a view
