"""Reference checkpoints <-> salsa_amd.crnn.SeldCRNN.

The reference saves Lightning checkpoints and loads them with ``model.load_state_dict(checkpoint['state_dict'])``
(experiments/inference.py:115-116); the keys are those of its ``SeldModel`` (models/seld_models.py:29-30: ``encoder.*`` =
models/encoders.py PannResNet22 -> ``conv_block1`` + ``resnet.layer{1..4}.{0,1}``, ``decoder.*`` = models/decoders.py SeldDecoder
-> ``gru`` + ``event_fc_{1,2}`` / ``{x,y,z}_fc_{1,2}``).  This module holds the one-to-one key map between that naming and
ours, so weights trained with the reference run on the MI355X model (config 5) and vice versa.  Tensors are identical in shape
and meaning (same architecture, same parameterisation): the map renames, nothing is transposed or re-scaled."""
import re
from collections import OrderedDict

import torch


def to_reference_key(k: str) -> str:
    """salsa_amd.crnn.SeldCRNN state-dict key -> the reference SeldModel's key."""
    m = re.match(r'encoder\.stem\.(conv1|bn1|conv2|bn2)\.(.*)', k)
    if m:
        return 'encoder.conv_block1.%s.%s' % m.groups()
    m = re.match(r'encoder\.stages\.(\d+)\.(\w+)\.(.*)', k)
    if m:
        i, part, rest = int(m.group(1)), m.group(2), m.group(3)
        layer, blk = i // 2 + 1, i % 2
        if part in ('short_conv', 'short_bn'):
            strided = layer > 1                                   # stride-2 shortcut = [AvgPool2d, conv1x1, BN]
            idx = (1 if strided else 0) + (0 if part == 'short_conv' else 1)
            return 'encoder.resnet.layer%d.%d.downsample.%d.%s' % (layer, blk, idx, rest)
        return 'encoder.resnet.layer%d.%d.%s.%s' % (layer, blk, part, rest)
    m = re.match(r'decoder\.(event|x|y|z)\.fc(\d)\.(.*)', k)
    if m:
        return 'decoder.%s_fc_%s.%s' % m.groups()
    return k                                                       # decoder.gru.*


def reference_state_dict(model) -> 'OrderedDict[str, torch.Tensor]':
    """The model's state dict under the reference's key names (what ``torch.save({'state_dict': ...})`` should hold for the
    reference's inference script to load it)."""
    return OrderedDict((to_reference_key(k), v) for k, v in model.state_dict().items())


def _unwrap(sd):
    if isinstance(sd, dict) and 'state_dict' in sd and not torch.is_tensor(sd['state_dict']):
        sd = sd['state_dict']                                      # a whole Lightning checkpoint
    for prefix in ('model.', 'module.'):                           # wrapped modules (DDP, a parent LightningModule)
        if sd and all(k.startswith(prefix) for k in sd):
            sd = OrderedDict((k[len(prefix):], v) for k, v in sd.items())
    return sd


def load_reference_state_dict(model, sd, strict: bool = True):
    """Load a reference checkpoint (its ``state_dict`` or the whole ``torch.load`` result) into a SeldCRNN.  Strict both
    ways by default: every reference key must land on a parameter / buffer of ours and every one of ours must be covered (shapes
    included); with ``strict=False`` returns the (missing, unexpected) key lists instead of raising.  Invalidates the cached
    bf16 / folded convolution filters."""
    sd = _unwrap(sd)
    ours = model.state_dict()
    inverse = {to_reference_key(k): k for k in ours}
    assert len(inverse) == len(ours), 'key map is not one-to-one'
    mapped, unexpected = OrderedDict(), []
    for k, v in sd.items():
        if k in inverse:
            mapped[inverse[k]] = v
        else:
            unexpected.append(k)
    missing = [to_reference_key(k) for k in ours if k not in mapped]
    bad_shape = ['%s: %s vs %s' % (to_reference_key(k), tuple(v.shape), tuple(ours[k].shape))
                 for k, v in mapped.items() if tuple(v.shape) != tuple(ours[k].shape)]
    if strict and (missing or unexpected or bad_shape):
        raise RuntimeError('reference checkpoint does not match SeldCRNN: missing %s; unexpected %s; shape mismatches %s'
                           % (missing[:8], unexpected[:8], bad_shape[:8]))
    model.load_state_dict(mapped, strict=strict and not bad_shape)
    from .nn_ops import invalidate_conv_caches
    invalidate_conv_caches(model)
    return missing, unexpected
