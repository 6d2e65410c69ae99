"""Helpers shared by tools/make_golden_crnn.py and the CRNN tests: deterministic weights (the parameter-name map between
salsa_amd.crnn.SeldCRNN and the reference's state dicts is product code: salsa_amd/crnn/checkpoint.py)."""

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


from .checkpoint import to_reference_key as name_map  # noqa: E402,F401  (the map lives in the product: crnn/checkpoint.py)


import contextlib  # noqa: E402


@contextlib.contextmanager
def dropout_off(model):
    """Every dropout of ``model`` switched off for the duration (F.dropout -> identity, nn.Dropout p = 0, GRU inter-layer
    dropout 0, the residual blocks' fused dropout 0): the deterministic training pass the g16 fixture was made with."""
    import torch.nn.functional as F
    saved = []
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            saved.append((m, 'p', m.p))
            m.p = 0.0
        if isinstance(m, torch.nn.GRU):
            saved.append((m, 'dropout', m.dropout))
            m.dropout = 0.0
        if hasattr(m, 'dropout_p'):
            saved.append((m, 'dropout_p', m.dropout_p))
            m.dropout_p = 0.0
    real = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x
    try:
        yield
    finally:
        F.dropout = real
        for m, k, v in saved:
            setattr(m, k, v)


def g16_batch(meta):
    """the seeded (4, 7, 128, 200) batch + labels of tests/golden/g16_crnn_train.npz (tools/make_golden_crnn.py draws them so)"""
    g = torch.Generator().manual_seed(meta['input_seed'])
    x = torch.randn(*meta['input_shape'], generator=g)
    sed = (torch.rand(4, 16, 12, generator=g) < 0.2).float()
    v = torch.randn(4, 16, 3, 12, generator=g)
    v = v / v.norm(dim=2, keepdim=True)
    return x, sed, (v * sed[:, :, None, :]).reshape(4, 16, 36)
