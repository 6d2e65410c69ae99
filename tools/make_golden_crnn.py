#!/usr/bin/env python
"""Golden vector for the CRNN consumer: fills salsa_amd.crnn.SeldCRNN with seeded random weights (and BN statistics),
copies them into the REFERENCE model (models/encoders.py PannResNet22 + models/decoders.py SeldDecoder, imported from
/root/reference) through an explicit name map, runs the reference forward + interpolate in eval mode and stores the
outputs.  tests/test_crnn_cpu.py regenerates the same weights from the seed and must reproduce these outputs.
Build-container only (needs /root/reference)."""
import json
import logging
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import types  # noqa: E402

import ref_shims  # noqa: E402

ref_shims.install()                                   # h5py / librosa / fire stand-ins + reference on sys.path
pl = types.ModuleType('pytorch_lightning')            # models/__init__ pulls in the Lightning interface; only the
pl.LightningModule = torch.nn.Module                  # nn.Module classes of encoders.py / decoders.py are used here
sys.modules.setdefault('pytorch_lightning', pl)
ipy = types.ModuleType('IPython')                     # metrics/SELD2020_evaluation_metrics.py:16 imports embed
ipy.embed = lambda *a, **k: None
sys.modules.setdefault('IPython', ipy)

from models.decoders import SeldDecoder  # noqa: E402  (reference)
from models.encoders import PannResNet22  # noqa: E402  (reference)
from models.model_utils import interpolate_tensor as ref_interp  # noqa: E402  (reference)

from salsa_amd.crnn.model import SeldCRNN  # noqa: E402
from salsa_amd.crnn.testing import name_map, seeded_fill  # noqa: E402

logging.getLogger('lightning').setLevel(logging.ERROR)
mine = SeldCRNN()
seeded_fill(mine, seed=7)
enc = PannResNet22(n_input_channels=7)
dec = SeldDecoder(n_output_channels=512, n_classes=12, output_format='reg_xyz', decoder_type='bigru', freq_pool='avg',
                  decoder_size=256)
ref_sd = {}
for k, v in mine.state_dict().items():
    ref_sd[name_map(k)] = v
missing = enc.load_state_dict({k[len('encoder.'):]: v for k, v in ref_sd.items() if k.startswith('encoder.')}, strict=True)
dec.load_state_dict({k[len('decoder.'):]: v for k, v in ref_sd.items() if k.startswith('decoder.')}, strict=True)
enc.eval(), dec.eval()
g = torch.Generator().manual_seed(11)
x = torch.randn(2, 7, 64, 200, generator=g)
with torch.no_grad():
    out = dec(enc(x))
    ev = ref_interp(out['event_frame_logit'], ratio=16 * 10 / 80)
    doa = ref_interp(out['doa_frame_output'], ratio=16 * 10 / 80)
path = os.path.join(ROOT, 'tests', 'golden', 'g9_crnn.npz')
np.savez_compressed(path, meta=np.array(json.dumps({'weight_seed': 7, 'input_seed': 11, 'input_shape': [2, 7, 64, 200]})),
                    event_frame_logit=ev.numpy(), doa_frame_output=doa.numpy())
print(path, ev.shape, doa.shape, float(ev.abs().mean()), float(doa.abs().mean()))

# the reference model's own state-dict keys and shapes (names only -- data, not source): pins salsa_amd/crnn/checkpoint.py's
# key map to what a reference checkpoint (experiments/inference.py:115-116, checkpoint['state_dict']) really contains
ref_keys = {}
for prefix, mod in (('encoder.', enc), ('decoder.', dec)):
    for k, v in mod.state_dict().items():
        ref_keys[prefix + k] = list(v.shape)
kpath = os.path.join(ROOT, 'tests', 'golden', 'g14_ref_state_dict_keys.npz')
np.savez_compressed(kpath, meta=np.array(json.dumps({'keys': ref_keys, 'source': 'PannResNet22(n_input_channels=7) + SeldDecoder(512, 12, reg_xyz, bigru, avg, 256) state_dict() under the SeldModel attribute names encoder / decoder'})))
print(kpath, len(ref_keys), 'keys')

# ---------------------------------------------------------------------------------------------------------------- g16
# TRAINING semantics of the consumer pinned to the reference modules (round-3 review, "missing 2"): the reference encoder
# (models/model_utils.py:345-367 basic blocks: batch-statistic BatchNorm, residual adds) + decoder in train() mode, the
# reference loss (models/interfaces.py:304-355 compute_classwise_clareg_loss / compute_masked_reg_loss, called unmodified on
# a stand-in `self` that only carries n_classes and loss_weight -- BaseModel's constructor wants a dataset on disk), backward,
# with every dropout switched off (F.dropout -> identity, nn.Dropout p = 0, GRU dropout 0) so the pass is deterministic.
# Stored: the three loss values, the gradients of seven named parameters (big ones strided), the running statistics two
# BatchNorm layers are left with, and which parameters the reference INITIALISES to exactly zero (zero_init_residual).
import torch.nn.functional as F  # noqa: E402

from models.interfaces import BaseModel  # noqa: E402  (reference)

fresh = PannResNet22(n_input_channels=7)
zero_init = sorted('encoder.' + k for k, v in fresh.state_dict().items()
                   if v.dtype.is_floating_point and v.numel() > 1 and not k.endswith(('running_mean', 'bias')) and float(v.abs().max()) == 0.0)

enc.train(), dec.train()
for m in list(enc.modules()) + list(dec.modules()):
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
    if isinstance(m, torch.nn.GRU):
        m.dropout = 0.0
_real_dropout = F.dropout
F.dropout = lambda x, p=0.5, training=True, inplace=False: x
try:
    g = torch.Generator().manual_seed(16)
    xb = torch.randn(4, 7, 128, 200, generator=g)
    sed = (torch.rand(4, 16, 12, generator=g) < 0.2).float()
    v = torch.randn(4, 16, 3, 12, generator=g)
    v = v / v.norm(dim=2, keepdim=True)
    doa_gt = (v * sed[:, :, None, :]).reshape(4, 16, 36)
    out = dec(enc(xb))
    pred = {'event_frame_logit': ref_interp(out['event_frame_logit'], ratio=16 * 10 / 80),
            'doa_frame_output': ref_interp(out['doa_frame_output'], ratio=16 * 10 / 80)}
    stand_in = types.SimpleNamespace(n_classes=12, loss_weight=(0.3, 0.7))
    stand_in.compute_masked_reg_loss = BaseModel.compute_masked_reg_loss
    stand_in.compute_doa_reg_loss = types.MethodType(BaseModel.compute_doa_reg_loss, stand_in)
    loss, sed_loss, doa_loss = BaseModel.compute_classwise_clareg_loss(stand_in, {'event_frame_gt': sed, 'doa_frame_gt': doa_gt}, pred)
    loss.backward()
finally:
    F.dropout = _real_dropout

inv = {name_map(k): k for k in mine.state_dict().keys()}                     # reference key -> our key
ref_params = dict([('encoder.' + k, p) for k, p in enc.named_parameters()] + [('decoder.' + k, p) for k, p in dec.named_parameters()])
ref_buffers = dict([('encoder.' + k, b) for k, b in enc.named_buffers()] + [('decoder.' + k, b) for k, b in dec.named_buffers()])
want = ['encoder.stem.conv1.weight', 'encoder.stages.0.conv1.weight', 'encoder.stages.2.short_conv.weight', 'encoder.stages.7.bn2.weight',
        'encoder.stages.5.bn1.bias', 'decoder.gru.weight_hh_l1_reverse', 'decoder.event.fc2.weight']
grads, strides = {}, {}
for ours in want:
    p = ref_params[name_map(ours)]
    flat = p.grad.detach().reshape(-1)
    st = max(1, flat.numel() // 4096)
    grads[ours], strides[ours] = flat[::st].numpy().copy(), st
stats = {}
for ours in ('encoder.stem.bn1', 'encoder.stages.7.bn2', 'encoder.stages.2.short_bn'):
    for suffix in ('running_mean', 'running_var', 'num_batches_tracked'):
        stats[ours + '.' + suffix] = ref_buffers[name_map(ours + '.' + suffix)].detach().numpy().copy()
tpath = os.path.join(ROOT, 'tests', 'golden', 'g16_crnn_train.npz')
np.savez_compressed(
    tpath, meta=np.array(json.dumps({'weight_seed': 7, 'input_seed': 16, 'input_shape': [4, 7, 128, 200], 'grad_strides': strides,
                                     'zero_init': [inv.get(k, k) for k in zero_init],
                                     'source': 'reference PannResNet22 + SeldDecoder in train(), BaseModel.compute_classwise_clareg_loss, dropout off'})),
    loss=np.array([float(loss), float(sed_loss), float(doa_loss)]), **{'grad:' + k: v for k, v in grads.items()},
    **{'stat:' + k: v for k, v in stats.items()})
print(tpath, float(loss), float(sed_loss), float(doa_loss), {k: float(np.abs(v).mean()) for k, v in grads.items()}, len(zero_init), 'zero-initialised')
