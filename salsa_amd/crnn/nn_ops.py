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


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


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
