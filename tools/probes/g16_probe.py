import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from conftest import load_golden
from salsa_amd.crnn import nn_ops, model as M
from salsa_amd.crnn.loss import seld_loss
from salsa_amd.crnn.testing import dropout_off, g16_batch, seeded_fill
from salsa_amd.crnn.train import Trainer
meta, a = load_golden('g16_crnn_train')
for on in (True, False):
    nn_ops.USE_HIP_POOL = nn_ops.USE_HIP_BN = nn_ops.USE_HIP_CONV = on
    M.FUSED_GRU = on
    for amp in (torch.bfloat16, None):
        tr = Trainer('cuda:0', total_steps=10, amp_dtype=amp)
        seeded_fill(tr.raw_model, meta['weight_seed']); nn_ops.invalidate_conv_caches(tr.raw_model)
        x, sed, doa = (t.cuda() for t in g16_batch(meta))
        tr.model.train()
        with dropout_off(tr.raw_model):
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp is not None):
                pred = tr.model(tr._input_layout(x))
            loss, s, d = seld_loss(pred, sed, doa); loss.backward()
        params = dict(tr.raw_model.named_parameters()); out = []
        for k, st in meta['grad_strides'].items():
            got = params[k].grad.float().reshape(-1)[::st].cpu().numpy().astype(np.float64); ref = a['grad:' + k].astype(np.float64)
            out.append('%s cos %.4f ratio %.3f' % (k.split('.', 1)[1][:22], got @ ref / np.linalg.norm(got) / np.linalg.norm(ref), np.linalg.norm(got) / np.linalg.norm(ref)))
        print('hip' if on else 'torch', 'bf16' if amp else 'fp32', 'loss %.5f vs %.5f |' % (float(loss), a['loss'][0]), ' | '.join(out))
