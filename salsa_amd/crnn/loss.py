"""SELD training loss (models/interfaces.py:304-355): 0.3 * BCE-with-logits(SED) + 0.7 * sum over x,y,z of the
activity-masked mean absolute error."""
import torch
import torch.nn.functional as F


def masked_mae(pred, target, mask):
    n = min(pred.shape[1], target.shape[1])
    pred, target, mask = pred[:, :n], target[:, :n], mask[:, :n]
    return torch.sum(torch.abs(pred - target) * mask) / torch.sum(mask)


def seld_loss(pred, sed_gt, doa_gt, loss_weight=(0.3, 0.7)):
    """pred: dict from SeldCRNN; sed_gt (B,T,12); doa_gt (B,T,36) -> (loss, sed_loss, doa_loss)."""
    nc = sed_gt.shape[-1]
    logit, doa = pred['event_frame_logit'].float(), pred['doa_frame_output'].float()
    sed = F.binary_cross_entropy_with_logits(logit, sed_gt)
    d = sum(masked_mae(doa[..., i * nc:(i + 1) * nc], doa_gt[..., i * nc:(i + 1) * nc], sed_gt) for i in range(3))
    return loss_weight[0] * sed + loss_weight[1] * d, sed, d
