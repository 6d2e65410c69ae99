"""SELD CRNN with the reference's architecture and initialisation (models/encoders.py:26-56 PannResNet22,
models/model_utils.py:187-228 ConvBlock, :312-367 basic block, :429-500 ResNet, models/decoders.py:13-154 SeldDecoder
with decoder_type='bigru', freq_pool='avg', decoder_size=256 -- experiments/configs/seld.yml:25-32).

Shapes for an 8-s training chunk (7,640,200): stem (64,320,100) -> stage1 (64,320,100) -> stage2 (128,160,50) ->
stage3 (256,80,25) -> stage4 (512,40,12) -> mean over frequency (40,512) -> BiGRU (40,512) -> SED logits (40,12) and
xyz (40,36).  14.11 M parameters (11.21 M encoder + 2.90 M decoder)."""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .nn_ops import (BatchNormAct2d, Conv1x1, Conv3x3, ConvFilterBank, avg_pool2x2, conv1x1, conv_bn_act, folded_shortcut, freq_mean_sequence,
                     invalidate_conv_caches, new_backward_generation, stack_groups)


GRU_FP32 = os.environ.get('SALSA_GRU_FP32', '1') == '1'
FUSED_GRU = os.environ.get('SALSA_FUSED_GRU', '1') == '1'
BATCHED_HEADS = os.environ.get('SALSA_BATCHED_HEADS', '1') == '1'   # the four FC heads as one batch of four (Decoder._heads_batched)


def _xavier(layer):
    nn.init.xavier_uniform_(layer.weight)
    if getattr(layer, 'bias', None) is not None:
        layer.bias.data.zero_()
    return layer


def _conv(cin, cout, k):
    cls = Conv3x3 if k == 3 else Conv1x1 if k == 1 else nn.Conv2d                     # same parameters / state_dict; 64 -> 64 runs on the MFMA kernel
    return _xavier(cls(cin, cout, kernel_size=k, stride=1, padding=k // 2, bias=False))


class Stem(nn.Module):
    """two 3x3 conv + BN + ReLU, then 2x2 average pool (model_utils.py:187-228 with pool_type='avg')"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, cout, 3), BatchNormAct2d(cout)
        self.conv2, self.bn2 = _conv(cout, cout, 3), BatchNormAct2d(cout)

    def forward(self, x):
        x = conv_bn_act(self.conv1, self.bn1, x)
        return conv_bn_act(self.conv2, self.bn2, x, pool=True)                # ... -> 2x2 average pool


class ResBlock(nn.Module):
    """basic residual block; stride 2 is an average pool followed by stride-1 convs; dropout 0.1 after the first ReLU;
    the last BN starts at zero (zero_init_residual) -- model_utils.py:312-367."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.stride, self.dropout_p = stride, 0.1
        # set by the Encoder: this block's output is only ever read by the NEXT block's stride-2 average pool, so the pool is
        # taken here, fused into the last BatchNorm + add + ReLU (pool_out), and the next block skips its own (pre_pooled)
        self.pool_out, self.pre_pooled = False, False
        self.conv1, self.bn1 = _conv(cin, cout, 3), BatchNormAct2d(cout)
        self.conv2, self.bn2 = _conv(cout, cout, 3), BatchNormAct2d(cout)
        nn.init.zeros_(self.bn2.weight)
        self.short_conv, self.short_bn = None, None
        if stride != 1 or cin != cout:                      # 1x1 conv + BN shortcut (after the pool when strided)
            self.short_conv, self.short_bn = _conv(cin, cout, 1), BatchNormAct2d(cout)

    def forward(self, x):
        y = avg_pool2x2(x) if (self.stride == 2 and not self.pre_pooled) else x
        # dropout(relu(bn1(conv1(y)))), training only; ys is y for the shortcut branch (in training routed through conv1's
        # autograd node: the branch's gradient is added inside conv1's data-gradient kernel, not by a separate pass)
        out, ys = conv_bn_act(self.conv1, self.bn1, y, dropout_p=self.dropout_p, skip=True)
        if self.short_conv is not None:
            folded = folded_shortcut(self.short_conv, self.short_bn, ys)      # eval: BatchNorm folded into the 1x1 GEMM, its shift
            if folded is not None:                                            # added by conv2's epilogue with the residual
                return conv_bn_act(self.conv2, self.bn2, out, residual=folded[0], pool=self.pool_out, residual_shift=folded[1])
        sc = self.short_bn(conv1x1(self.short_conv, ys)) if self.short_conv is not None else ys
        return conv_bn_act(self.conv2, self.bn2, out, residual=sc, pool=self.pool_out)   # relu(bn2(conv2(out)) + shortcut) [-> pool]


class Encoder(nn.Module):
    """PANN ResNet22-style: stem + 4 stages x 2 blocks (64/128/256/512); time and frequency /16 (encoders.py:26-56)."""
    time_downsample_ratio = 16
    n_output_channels = 512

    def __init__(self, n_input_channels=7, p_dropout=0.0):
        super().__init__()
        self.p_dropout = p_dropout
        self.stem = Stem(n_input_channels, 64)
        blocks, cin = [], 64
        for cout, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
            blocks += [ResBlock(cin, cout, stride), ResBlock(cout, cout, 1)]
            cin = cout
        for prev, nxt in zip(blocks[:-1], blocks[1:]):
            if nxt.stride == 2:
                prev.pool_out, nxt.pre_pooled = True, True
        self.stages = nn.Sequential(*blocks)
        # one kernel per step makes the bf16 working copies of all 3x3 filters (not a module, parameter or buffer: no state)
        self._filter_bank = ConvFilterBank([m for m in self.modules() if isinstance(m, (Conv3x3, Conv1x1))])
        # load_state_dict copies into the parameters under no_grad: the cached bf16 / folded filters must not survive it
        self.register_load_state_dict_post_hook(lambda module, incompatible: invalidate_conv_caches())

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            self._filter_bank.mark_stale()                     # training: the bank is rebuilt every forward, unconditionally
        if torch.is_grad_enabled():
            new_backward_generation(x.device)                  # weight-gradient buffers: one allocation + one fill per pass (any
                                                               # differentiable forward, eval-mode fine-tuning / saliency included)
        x = self.stem(x)
        x = F.dropout(x, p=self.p_dropout, training=self.training)
        return self.stages(x)


def _init_gru(rnn):
    """reference init (model_utils.py:156-184): per-gate uniform(+-sqrt(3/fan_in)) for W_ih and the r,z blocks of W_hh,
    orthogonal for the n block of W_hh, zero biases."""
    def uni(t):
        b = math.sqrt(3.0 / t.shape[1])
        nn.init.uniform_(t, -b, b)

    for name, p in rnn.named_parameters():
        if 'bias' in name:
            nn.init.zeros_(p)
            continue
        h = p.shape[0] // 3
        with torch.no_grad():
            for g in range(3):
                blk = p[g * h:(g + 1) * h]
                if 'weight_hh' in name and g == 2:
                    nn.init.orthogonal_(blk)
                else:
                    uni(blk)


class Head(nn.Module):
    def __init__(self, d, n_out):
        super().__init__()
        self.fc1, self.fc2 = _xavier(nn.Linear(d, d // 2)), _xavier(nn.Linear(d // 2, n_out))
        self.drop1, self.drop2 = nn.Dropout(0.2), nn.Dropout(0.2)

    def forward(self, x):
        return self.fc2(self.drop2(F.relu(self.fc1(self.drop1(x)), inplace=True)))


class Decoder(nn.Module):
    """mean over frequency -> 2-layer BiGRU(256) with dropout 0.3 -> SED head + x/y/z heads (decoders.py:13-154)."""

    def __init__(self, n_in=512, n_classes=12, size=256):
        super().__init__()
        self.n_classes = n_classes
        self.gru = nn.GRU(n_in, size, num_layers=2, batch_first=True, bidirectional=True, dropout=0.3)
        _init_gru(self.gru)
        self.event = Head(2 * size, n_classes)
        self.x, self.y, self.z = Head(2 * size, n_classes), Head(2 * size, n_classes), Head(2 * size, n_classes)

    def forward(self, feat):
        seq = freq_mean_sequence(feat)                          # (B, T', 512): mean over frequency
        if seq.is_cuda and GRU_FP32:
            # 0.4 % of the FLOPs but, under bf16 autocast, ~6000 per-timestep cell kernels per forward (torch's native
            # fallback); in float32 the whole sequence goes through MIOpen's fused RNN
            amp = torch.is_autocast_enabled('cuda')
            with torch.autocast(device_type='cuda', enabled=False):
                if FUSED_GRU:
                    from .fused_gru import bigru_forward
                    seq = bigru_forward(self.gru, seq.float(), self.training, half_weights=amp)   # one HIP launch per layer
                else:
                    seq, _ = self.gru(seq.float())
            # the heads too (0.02 % of the FLOPs): under autocast their eight small linears cost 32 cast kernels per step
            # (weights and biases to bf16, their gradients back) around GEMMs of a few microseconds
            with torch.autocast(device_type='cuda', enabled=False):
                return self._heads(seq.float())
        else:
            seq, _ = self.gru(seq)
        return self._heads(seq)

    def _heads(self, seq):
        if BATCHED_HEADS and seq.dim() == 3:
            return self._heads_batched(seq)
        doa = torch.cat([torch.tanh(self.x(seq)), torch.tanh(self.y(seq)), torch.tanh(self.z(seq))], dim=-1)
        return {'event_frame_logit': self.event(seq), 'doa_frame_output': doa}

    def _heads_batched(self, seq):
        """The four heads (same shapes, same input, independent dropout masks) as ONE batch of four: two batched GEMMs, two
        dropout launches and one tanh instead of eight small GEMMs, eight dropouts, three tanh and a concatenation -- with their
        backward kernels the heads were ~0.45 ms of the 12-ms training step for 0.02 % of its FLOPs.  The parameters stay where
        the reference's state dict has them (event / x / y / z . fc1 / fc2); they are stacked per call (four small copies)."""
        heads = (self.event, self.x, self.y, self.z)
        B, T, D = seq.shape
        w1, b1, w2, b2 = stack_groups([[h.fc1.weight for h in heads], [h.fc1.bias for h in heads],      # (4, D/2, D), (4, D/2),
                                       [h.fc2.weight for h in heads], [h.fc2.bias for h in heads]])     # (4, n_classes, D/2), ...
        b1, b2 = b1.unsqueeze(1), b2.unsqueeze(1)
        x = seq.reshape(1, B * T, D).expand(4, B * T, D)
        x = F.dropout(x, 0.2, self.training)                                # four independent masks (drop1 of each head)
        h = F.relu(torch.baddbmm(b1, x, w1.transpose(1, 2)), inplace=True)
        h = F.dropout(h, 0.2, self.training)
        o = torch.baddbmm(b2, h, w2.transpose(1, 2))                        # (4, B*T, n_classes)
        doa = torch.tanh(o[1:]).permute(1, 0, 2).reshape(B, T, 3 * o.shape[2])   # [x | y | z] along the last axis
        return {'event_frame_logit': o[0].view(B, T, -1), 'doa_frame_output': doa}


_INTERP_IDX = {}


def interpolate_tensor(t, ratio: float):
    """nearest-index resampling along time (model_utils.py:57-75): out[j] = in[floor(j / ratio)]."""
    n_out = int(round(t.shape[1] * float(ratio)))
    r = int(round(float(ratio)))
    if r >= 1 and float(ratio) == r and n_out == r * t.shape[1]:
        # integer ratio (the SELD configuration: 2): floor(j / r) = j // r, i.e. every frame r times -- a broadcast view and one
        # copy, whose backward is one sum, instead of arange / div / floor / index and an index_add into a zero tensor
        return t.unsqueeze(2).expand(t.shape[0], t.shape[1], r, *t.shape[2:]).reshape(t.shape[0], n_out, *t.shape[2:])
    key = (n_out, float(ratio), t.device)
    idx = _INTERP_IDX.get(key)
    if idx is None:
        if len(_INTERP_IDX) > 16:
            _INTERP_IDX.clear()
        idx = _INTERP_IDX[key] = torch.floor(torch.arange(n_out, device=t.device) / float(ratio)).long()
    return t[:, idx]


class SeldCRNN(nn.Module):
    """forward(x (B,7,T,200)) -> dict at the LABEL rate (seld_models.py:39-66: encoder, decoder, then
    interpolate by time_downsample_ratio * label_rate / feature_rate = 16 * 10 / 80 = 2)."""

    def __init__(self, n_input_channels=7, n_classes=12, label_rate=10, feature_rate=80):
        super().__init__()
        self.encoder = Encoder(n_input_channels)
        self.decoder = Decoder(self.encoder.n_output_channels, n_classes)
        self.ratio = self.encoder.time_downsample_ratio * label_rate / feature_rate

    def forward(self, x):
        out = self.decoder(self.encoder(x))
        return {k: interpolate_tensor(v, self.ratio) for k, v in out.items()}

    def pack_parameters(self):
        """Lay the parameters the decoder stacks every step (the two directions of each GRU parameter, the four heads' weights) out
        stacked in memory, so that the stack is a view instead of a copy (nn_ops.pack_stacked_parameters).  Values, Parameter
        objects and state-dict keys are unchanged.  Call it after the last ``.to(device)`` (the Trainer does); returns self."""
        from .nn_ops import pack_stacked_parameters
        gru, dec = self.decoder.gru, self.decoder
        groups = [[getattr(gru, '%s_l%d%s' % (kind, layer, rev)) for rev in ('', '_reverse')]
                  for layer in range(gru.num_layers) for kind in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
        heads = (dec.event, dec.x, dec.y, dec.z)
        groups += [[h.fc1.weight for h in heads], [h.fc1.bias for h in heads], [h.fc2.weight for h in heads], [h.fc2.bias for h in heads]]
        pack_stacked_parameters(groups)
        if hasattr(gru, '_flat_weights'):      # nn.GRU caches references to its weights for the (unused here) MIOpen path
            gru._flat_weights = [getattr(gru, n) if hasattr(gru, n) else None for n in gru._flat_weights_names]
        return self

    def load_reference_state_dict(self, sd, strict: bool = True):
        """Load weights trained with the reference (``torch.load(ckpt)`` or its ``['state_dict']``, experiments/inference.py:
        115-116): see salsa_amd/crnn/checkpoint.py."""
        from .checkpoint import load_reference_state_dict
        return load_reference_state_dict(self, sd, strict)

    def reference_state_dict(self):
        """This model's state dict under the reference's key names."""
        from .checkpoint import reference_state_dict
        return reference_state_dict(self)
