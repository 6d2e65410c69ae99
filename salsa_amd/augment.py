"""Training augmentations of the reference's FOA SALSA recipe, batched on the device (utilities/transforms.py):
  * TfmapRandomSwapChannelFoa (:365-437): with p = 0.5, draw m in {0,1}^4 and swap x<->y / negate x, y, z in BOTH the
    feature channels (W Y Z X | Y Z X: rows 1/3 and 4/6 swap, rows 6, 4, 5 negate) and the xyz regression targets;
  * RandomShiftUpDownNp (:286-320): with p = 0.5, shift every channel 1..range-1 bins up or down in frequency with
    reflect padding.
The deterministic cores take the random draw as an argument (tests pin them against numpy restatements); the ``random_*``
wrappers draw per sample from a torch.Generator.  Pure index / sign operations: torch is used as plumbing only."""
import torch


def swap_channels_foa(x, y_doa, m, n_classes: int = 12):
    """x (B,7,T,F), y_doa (B,T_lab,36), m (B,4) in {0,1} -> transformed copies."""
    x_new, y_new = x.clone(), y_doa.clone()
    m = m.to(torch.bool)
    s = m[:, 0]
    if s.any():                                                  # swap x, y
        x_new[s, 1], x_new[s, 3] = x[s, 3], x[s, 1]
        x_new[s, 4], x_new[s, 6] = x[s, 6], x[s, 4]
        y_new[s, :, :n_classes], y_new[s, :, n_classes:2 * n_classes] = y_doa[s, :, n_classes:2 * n_classes], y_doa[s, :, :n_classes]
    for bit, xrow, lo in ((1, 6, 0), (2, 4, n_classes), (3, 5, 2 * n_classes)):   # negate x (row -1), y (row -3), z (row -2)
        s = m[:, bit]
        if s.any():
            x_new[s, xrow] = -x_new[s, xrow]
            y_new[s, :, lo:lo + n_classes] = -y_new[s, :, lo:lo + n_classes]
    return x_new, y_new


def shift_up_down(x, shift_len, up):
    """x (B,C,T,F); shift_len (B,) ints >= 0 (0 = untouched); up (B,) bool.  np.pad(mode='reflect') semantics."""
    B, C, T, F = x.shape
    f = torch.arange(F, device=x.device)[None, :]
    s = shift_len.to(x.device)[:, None]
    src_up = (f - s).abs()                                       # pad s at the front, keep the first F: reflect at bin 0
    src_dn = f + s
    src_dn = torch.where(src_dn > F - 1, 2 * (F - 1) - src_dn, src_dn)   # pad s at the back, drop the first s
    src = torch.where(up.to(x.device)[:, None], src_up, src_dn)          # (B,F)
    return torch.gather(x, 3, src[:, None, None, :].expand(B, C, T, F))


def random_swap_channels_foa(x, y_sed, y_doa, gen=None, p: float = 0.5, n_classes: int = 12):
    B = x.shape[0]
    apply = torch.rand(B, generator=gen) < p
    m = torch.randint(0, 2, (B, 4), generator=gen) * apply[:, None]
    xn, yn = swap_channels_foa(x, y_doa, m.to(x.device), n_classes)
    return xn, y_sed, yn


def random_shift_up_down(x, gen=None, p: float = 0.5, freq_shift_range: int = 10):
    B = x.shape[0]
    apply = torch.rand(B, generator=gen) < p
    shift = torch.randint(1, freq_shift_range, (B,), generator=gen) * apply
    up = torch.randint(0, 2, (B,), generator=gen).bool()
    return shift_up_down(x, shift, up)
