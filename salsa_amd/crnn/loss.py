"""SELD training loss (models/interfaces.py:304-355): 0.3 * BCE-with-logits(SED) + 0.7 * sum over x,y,z of the
activity-masked mean absolute error."""
import os

import torch
import torch.nn.functional as F

FUSED_LOSS = os.environ.get('SALSA_FUSED_LOSS', '1') == '1'   # one HIP launch for the loss and its gradients (CUDA tensors)


def masked_mae(pred, target, mask):
    n = min(pred.shape[1], target.shape[1])
    pred, target, mask = pred[:, :n], target[:, :n], mask[:, :n]
    return torch.sum(torch.abs(pred - target) * mask) / torch.sum(mask)


def seld_loss(pred, sed_gt, doa_gt, loss_weight=(0.3, 0.7)):
    """pred: dict from SeldCRNN; sed_gt (B,T,12); doa_gt (B,T,36) -> (loss, sed_loss, doa_loss)."""
    nc = sed_gt.shape[-1]
    logit, doa = pred['event_frame_logit'].float(), pred['doa_frame_output'].float()
    if (FUSED_LOSS and logit.is_cuda and sed_gt.dtype == torch.float32 and doa_gt.dtype == torch.float32
            and logit.shape == sed_gt.shape and doa.shape == doa_gt.shape and doa.shape[-1] == 3 * nc):
        out = _SeldLoss.apply(logit, doa, sed_gt, doa_gt, float(loss_weight[0]), float(loss_weight[1]))
        return out[0], out[1], out[2]
    sed = F.binary_cross_entropy_with_logits(logit, sed_gt)
    d = sum(masked_mae(doa[..., i * nc:(i + 1) * nc], doa_gt[..., i * nc:(i + 1) * nc], sed_gt) for i in range(3))
    return loss_weight[0] * sed + loss_weight[1] * d, sed, d


class _SeldLoss(torch.autograd.Function):
    """salsa_nn_seld_loss: the loss above and its gradients in ONE launch (the eager version is ~35 kernels of a few
    microseconds on 30 k-element tensors), one more launch in the backward to scale them by the incoming gradients."""

    @staticmethod
    def forward(ctx, logit, doa, sed_gt, doa_gt, w_sed, w_doa):
        from .. import _lib
        from .nn_ops import _ptr, _stream
        logit, doa, sed_gt, doa_gt = logit.contiguous(), doa.contiguous(), sed_gt.contiguous(), doa_gt.contiguous()
        nc = sed_gt.shape[-1]
        rows = sed_gt.numel() // nc
        out = torch.empty(3, dtype=torch.float32, device=logit.device)
        ws = torch.empty(192, dtype=torch.float64, device=logit.device)       # include/salsa_nn.h SALSA_SELD_LOSS_WS
        ga, gb = torch.empty_like(logit), torch.empty_like(doa)
        with torch.cuda.device(logit.device):
            rc = _lib.load().salsa_nn_seld_loss(_ptr(logit), _ptr(doa), _ptr(sed_gt), _ptr(doa_gt), rows, nc, w_sed, w_doa, _ptr(out),
                                                _ptr(ga), _ptr(gb), _ptr(ws), _stream(logit))
        if rc:
            raise RuntimeError('salsa_nn_seld_loss failed (%d)' % rc)
        ctx.set_materialize_grads(False)          # unused outputs (the two detached parts) arrive as None, not as zero fills
        ctx.save_for_backward(ga, gb)
        ctx.w = (w_sed, w_doa)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_loss, g_sed, g_doa):
        from .. import _lib
        from .nn_ops import _ptr, _stream
        ga, gb = ctx.saved_tensors
        if g_loss is None and g_sed is None and g_doa is None:
            return None, None, None, None, None, None
        oa, ob = torch.empty_like(ga), torch.empty_like(gb)

        def scalar(g):
            return None if g is None else g.to(device=ga.device, dtype=torch.float32).contiguous()
        g_loss, g_sed, g_doa = scalar(g_loss), scalar(g_sed), scalar(g_doa)
        with torch.cuda.device(ga.device):
            rc = _lib.load().salsa_nn_seld_loss_bwd(_ptr(ga), ga.numel(), _ptr(gb), gb.numel(), _ptr(g_loss), _ptr(g_sed), _ptr(g_doa),
                                                    ctx.w[0], ctx.w[1], _ptr(oa), _ptr(ob), _stream(ga))
        if rc:
            raise RuntimeError('salsa_nn_seld_loss_bwd failed (%d)' % rc)
        return oa, ob, None, None, None, None
