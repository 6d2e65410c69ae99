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
