"""Helpers shared by tools/make_golden_crnn.py and the CRNN tests: deterministic weights and the parameter-name map
between salsa_amd.crnn.SeldCRNN and the reference's PannResNet22 + SeldDecoder state dicts."""
import re

import torch


def seeded_fill(model, seed: int):
    """Overwrite every parameter and BN statistic with seeded values (also the zero-initialised last BN of each
    residual block, so every layer contributes to the output)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, t in model.state_dict().items():
            if t.dtype not in (torch.float32, torch.float64):
                continue
            if name.endswith('running_var'):
                t.copy_(0.5 + torch.rand(t.shape, generator=g))
            elif name.endswith('running_mean') or name.endswith('bias'):
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif t.dim() == 1:                                   # BN weight
                t.copy_(0.5 + torch.rand(t.shape, generator=g))
            else:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * (1.0 / fan_in) ** 0.5)


def name_map(k: str) -> str:
    """salsa_amd.crnn name -> reference name (encoder.* / decoder.* prefixes kept)."""
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
