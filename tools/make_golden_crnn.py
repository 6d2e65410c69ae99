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
