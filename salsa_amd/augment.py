"""Training augmentations of the reference's FOA SALSA recipe, batched on the device (utilities/transforms.py):
  * TfmapRandomSwapChannelFoa (:365-437): with p = 0.5, draw m in {0,1}^4 and swap x<->y / negate x, y, z in BOTH the
    feature channels (W Y Z X | Y Z X: rows 1/3 and 4/6 swap, rows 6, 4, 5 negate) and the xyz regression targets;
  * RandomShiftUpDownNp (:286-320): with p = 0.5, shift every channel 1..range-1 bins up or down in frequency with
    reflect padding.
  * TfmapRandomSwapChannelMic (:440-523): m in {0,1}^3 -- swap M2/M3, swap M1/M4, swap (M1,M2)/(M3,M4) -- which permutes
    the four spectrogram rows AND re-references the three phase rows (p12 p13 p14) by differences, plus the matching
    swap / negation of the xyz targets;
  * CompositeCutout (:257-283) = one of RandomCutoutNp (:58-121), SpecAugmentNp (:124-194), RandomCutoutHoleNp (:197-254),
    all of which fill rectangles with a value drawn between the sample's min and max (zeros in the last
    ``n_zero_channels`` spatial rows): one core, ``fill_rects``.
The deterministic cores take the random draw as an argument; ``draw_*`` reproduce the reference's draws IN ITS CALL ORDER
from a numpy RandomState-like source, so ``reference_train_transform`` under ``np.random.seed(s)`` equals the
reference's SeldDataset augmentation under the same seed (golden g11); the ``random_*`` wrappers are the batched
on-device forms drawing per sample from a torch.Generator.  Pure index / sign / difference operations: torch is used
as plumbing only."""
import numpy as np
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


# ------------------------------------------------------------------------------------------------------- MIC swap
def swap_channels_mic(x, y_doa, m, n_classes: int = 12):
    """x (B,7,T,F) rows M1 M2 M3 M4 p12 p13 p14, y_doa (B,T_lab,36), m (B,3) in {0,1} (transforms.py:469-523)."""
    m = m.to(torch.bool)
    x_new, y_new = x.clone(), y_doa.clone()
    nc = n_classes
    s = m[:, 0]
    if s.any():                                                  # swap M2 and M3 -> swap x and y
        x_new[s, 1], x_new[s, 2] = x[s, 2], x[s, 1]
        x_new[s, 4], x_new[s, 5] = x[s, 5], x[s, 4]
        y_new[s, :, :nc], y_new[s, :, nc:2 * nc] = y_doa[s, :, nc:2 * nc], y_doa[s, :, :nc]
    s = m[:, 1]
    if s.any():                                                  # swap M1 and M4 -> swap x and y, negate both
        c, yc = x_new.clone(), y_new.clone()
        x_new[s, 0], x_new[s, 3] = c[s, 3], c[s, 0]
        x_new[s, 6] = -c[s, 6]
        x_new[s, 5] = c[s, 5] - c[s, 6]
        x_new[s, 4] = c[s, 4] - c[s, 6]
        y_new[s, :, :nc], y_new[s, :, nc:2 * nc] = -yc[s, :, nc:2 * nc], -yc[s, :, :nc]
    s = m[:, 2]
    if s.any():                                                  # swap M1/M2 and M3/M4 -> negate y and z
        c = x_new.clone()
        x_new[s, 0], x_new[s, 1], x_new[s, 2], x_new[s, 3] = c[s, 1], c[s, 0], c[s, 3], c[s, 2]
        x_new[s, 4] = -c[s, 4]
        x_new[s, 5] = c[s, 6] - c[s, 4]
        x_new[s, 6] = c[s, 5] - c[s, 4]
        y_new[s, :, nc:2 * nc] = -y_new[s, :, nc:2 * nc]
        y_new[s, :, 2 * nc:] = -y_new[s, :, 2 * nc:]
    return x_new, y_new


def random_swap_channels_mic(x, y_sed, y_doa, gen=None, p: float = 0.5, n_classes: int = 12):
    B = x.shape[0]
    apply = torch.rand(B, generator=gen) < p
    m = torch.randint(0, 2, (B, 3), generator=gen) * apply[:, None]
    xn, yn = swap_channels_mic(x, y_doa, m.to(x.device), n_classes)
    return xn, y_sed, yn


# ------------------------------------------------------------------------------------------------------- cutouts
def fill_rects(x, top, h, left, w, value, n_zero_channels=None, is_filled_last_channels: bool = True):
    """x (B,C,T,F); top/h/left/w (B,R) integer rectangles (h or w = 0: none), value (B,R): applied in order r = 0..R-1.
    The first C - n_zero_channels rows get ``value``; the last n_zero_channels rows get 0 (or are left alone)."""
    B, C, T, F = x.shape
    out = x.clone()
    dev = x.device
    t = torch.arange(T, device=dev)[None, :, None]
    f = torch.arange(F, device=dev)[None, None, :]
    top, h, left, w = (a.to(dev) for a in (top, h, left, w))
    value = value.to(dev, x.dtype)
    nz = 0 if n_zero_channels is None else int(n_zero_channels)
    for r in range(top.shape[1]):
        mask = ((t >= top[:, r, None, None]) & (t < (top[:, r] + h[:, r])[:, None, None]) &
                (f >= left[:, r, None, None]) & (f < (left[:, r] + w[:, r])[:, None, None]))[:, None]   # (B,1,T,F)
        out[:, :C - nz] = torch.where(mask, value[:, r, None, None, None], out[:, :C - nz])
        if nz and is_filled_last_channels:
            out[:, C - nz:] = torch.where(mask, torch.zeros((), dtype=x.dtype, device=dev), out[:, C - nz:])
    return out


def _cutout_ratio_bounds(image_aspect_ratio):
    r_1, r_2 = 0.3, 1 / 0.3                                      # transforms.py:78-85
    if image_aspect_ratio > 1:
        r_1 = r_1 * image_aspect_ratio
    elif image_aspect_ratio < 1:
        r_2 = r_2 * image_aspect_ratio
    return r_1, r_2


def draw_random_cutout(rng, T, F, vmin, vmax, image_aspect_ratio=1.0, random_value=None):
    """RandomCutoutNp.apply's draws in order (transforms.py:99-110) -> [(top, h, left, w, value)]."""
    r_1, r_2 = _cutout_ratio_bounds(image_aspect_ratio)
    s = rng.uniform(0.02, 0.3) * T * F
    r = rng.uniform(r_1, r_2)
    w = min(int(np.sqrt(s / r)), F - 1)
    h = min(int(np.sqrt(s * r)), T - 1)
    left = rng.randint(0, F - w)
    top = rng.randint(0, T - h)
    c = rng.uniform(vmin, vmax) if random_value is None else random_value
    return [(top, h, left, w, c)]


def draw_spec_augment(rng, T, F, vmin, vmax, time_max_width=None, freq_max_width=None, n_time_stripes=1,
                      n_freq_stripes=1):
    """SpecAugmentNp.apply's draws in order (transforms.py:154-192): full-width time stripes, then full-height frequency
    stripes."""
    tmw = max(1, int(0.15 * T) if time_max_width is None else time_max_width)
    fmw = max(1, int(0.2 * F) if freq_max_width is None else freq_max_width)
    rects = []
    for _ in range(n_time_stripes):
        dur = rng.randint(1, tmw, 1)[0]
        start = rng.randint(0, T - dur, 1)[0]
        rects.append((start, dur, 0, F, rng.uniform(vmin, vmax, 1)[0]))
    for _ in range(n_freq_stripes):
        dur = rng.randint(1, fmw, 1)[0]
        start = rng.randint(0, F - dur, 1)[0]
        rects.append((0, T, start, dur, rng.uniform(vmin, vmax, 1)[0]))
    return rects


def draw_cutout_holes(rng, T, F, vmin, vmax, n_max_holes=8, max_h_size=8, max_w_size=8, filled_value=None):
    """RandomCutoutHoleNp.apply's draws in order (transforms.py:233-246): n_max_holes holes of the maximum size."""
    h, w = max(max_h_size, 5), max(max_w_size, 5)
    rects = []
    for _ in range(n_max_holes):
        left = rng.randint(0, F - w)
        top = rng.randint(0, T - h)
        rects.append((top, h, left, w, rng.uniform(vmin, vmax) if filled_value is None else filled_value))
    return rects


def draw_composite_cutout(rng, T, F, vmin, vmax, image_aspect_ratio=1.0):
    """CompositeCutout.apply (transforms.py:276-283)."""
    choice = rng.randint(0, 3, 1)[0]
    if choice == 0:
        return draw_random_cutout(rng, T, F, vmin, vmax, image_aspect_ratio)
    if choice == 1:
        return draw_spec_augment(rng, T, F, vmin, vmax)
    return draw_cutout_holes(rng, T, F, vmin, vmax)


def _rects_to_tensors(rect_lists, n_rects=8):
    B = len(rect_lists)
    geo = torch.zeros((4, B, n_rects), dtype=torch.long)
    val = torch.zeros((B, n_rects), dtype=torch.float64)
    for b, rects in enumerate(rect_lists):
        for r, (top, h, left, w, c) in enumerate(rects):
            geo[0, b, r], geo[1, b, r], geo[2, b, r], geo[3, b, r], val[b, r] = int(top), int(h), int(left), int(w), float(c)
    return geo[0], geo[1], geo[2], geo[3], val


def reference_train_transform(x, y_sed, y_doa, audio_format='foa', rng=np.random, image_aspect_ratio=None,
                              n_classes: int = 12, freq_shift_range: int = 10, p: float = 0.5):
    """One sample through the reference's training augmentation for SALSA features (datamodule.py:45-52 FOA, :73-82 MIC;
    applied joint-then-plain, dataloader.py:56-60), drawing from ``rng`` exactly as the reference draws from np.random.
    x (7,T,F), y_sed (T_lab,n_classes), y_doa (T_lab,3*n_classes) tensors (any device) -> same."""
    xb, yb = x[None], y_doa[None]
    if rng.rand() < p:                                           # MapDataAugmentBase.__call__ :346-353
        if audio_format == 'foa':
            m = rng.randint(2, size=(4,))                        # :409
            xb, yb = swap_channels_foa(xb, yb, torch.as_tensor(m)[None], n_classes)
        else:
            m = rng.randint(2, size=(3,))                        # :483
            xb, yb = swap_channels_mic(xb, yb, torch.as_tensor(m)[None], n_classes)
    if rng.rand() < p:                                           # RandomShiftUpDownNp :298-320
        shift = rng.randint(1, freq_shift_range, 1)[0]
        up = rng.choice(['up', 'down'], 1)[0] == 'up'
        xb = shift_up_down(xb, torch.as_tensor([shift]), torch.as_tensor([bool(up)]))
    if audio_format == 'mic' and rng.rand() < p:                 # CompositeCutout(n_zero_channels=3) is in the MIC recipe only
        T, F = xb.shape[-2:]
        ratio = T / 200 if image_aspect_ratio is None else image_aspect_ratio
        rects = draw_composite_cutout(rng, T, F, float(xb.min()), float(xb.max()), ratio)
        xb = fill_rects(xb, *_rects_to_tensors([rects]), n_zero_channels=3)
    return xb[0], y_sed, yb[0]


def reference_draws(rng, T, F, audio_format='foa', minmax_after=None, image_aspect_ratio=None, freq_shift_range: int = 10,
                    p: float = 0.5):
    """The draws reference_train_transform makes, IN THE SAME ORDER from the same numpy source, as a parameter dict for ONE
    sample in apply_augment_hip's format -- so that the reference's own augmented samples (golden g11, np.random.seed(s)) can be
    demanded from the HIP kernel itself.  The cutout's fill value is drawn between the min and max of the sample AFTER swap and
    shift (transforms.py:99-110 reads the array it is handed): ``minmax_after(d)`` is called with the swap / shift draws and must
    return that pair (the test gets it from a first pass of the kernel without the cutout).  The drawn VALUES are passed as
    ``u`` with ``minmax`` = (0, 1): the kernel's lo + (hi - lo) * u is then the value itself, bit for bit."""
    d = dict(m=torch.zeros((1, 4), dtype=torch.long), shift=torch.zeros(1, dtype=torch.long), up=torch.zeros(1, dtype=torch.bool),
             top=torch.zeros((1, 8), dtype=torch.long), h=torch.zeros((1, 8), dtype=torch.long),
             left=torch.zeros((1, 8), dtype=torch.long), w=torch.zeros((1, 8), dtype=torch.long),
             u=torch.zeros((1, 8), dtype=torch.float32), minmax=torch.tensor([[0.0, 1.0]]))
    if rng.rand() < p:
        nbits = 4 if audio_format == 'foa' else 3
        d['m'][0, :nbits] = torch.as_tensor(rng.randint(2, size=(nbits,)))
    if rng.rand() < p:
        d['shift'][0] = int(rng.randint(1, freq_shift_range, 1)[0])
        d['up'][0] = bool(rng.choice(['up', 'down'], 1)[0] == 'up')
    if audio_format == 'mic' and rng.rand() < p:
        vmin, vmax = minmax_after(d)
        ratio = T / 200 if image_aspect_ratio is None else image_aspect_ratio
        rects = draw_composite_cutout(rng, T, F, float(vmin), float(vmax), ratio)
        top, h, left, w, val = _rects_to_tensors([rects])
        d['top'], d['h'], d['left'], d['w'] = top, h, left, w
        d['u'] = val.to(torch.float32)                           # (the float64 draw lands in a float32 array in the reference too)
    return d


def draw_composite_cutout_batch(B, T, F, gen=None, p: float = 0.5, image_aspect_ratio: float = 1.0):
    """Per-sample CompositeCutout draws for a batch (host, torch.Generator): with probability p one of the three cutouts
    with the reference's size distributions.  -> top, h, left, w (B,8) int64 (h = w = 0: no rectangle) and u (B,8) in
    [0,1): the fill value of rectangle r is min + (max - min) * u[r] of the sample."""
    u = lambda *shape: torch.rand(*shape, generator=gen, dtype=torch.float64)     # noqa: E731
    ri = lambda hi, n: torch.floor(u(n) * hi.double()).long()                   # noqa: E731  randint(0, hi) per element
    apply = u(B) < p
    choice = torch.randint(0, 3, (B,), generator=gen)
    r_1, r_2 = _cutout_ratio_bounds(image_aspect_ratio)
    top, h, left, w = (torch.zeros((B, 8), dtype=torch.long) for _ in range(4))
    # choice 0: one rectangle of random area and aspect
    area = (0.02 + 0.28 * u(B)) * T * F
    ratio = r_1 + (r_2 - r_1) * u(B)
    w0 = torch.clamp(torch.sqrt(area / ratio).long(), max=F - 1)
    h0 = torch.clamp(torch.sqrt(area * ratio).long(), max=T - 1)
    c0 = choice == 0
    w[c0, 0], h[c0, 0] = w0[c0], h0[c0]
    left[c0, 0], top[c0, 0] = ri(F - w0, B)[c0], ri(T - h0, B)[c0]
    # choice 1: a time stripe and a frequency stripe
    tmw, fmw = max(1, int(0.15 * T)), max(1, int(0.2 * F))
    c1 = choice == 1
    td = 1 + ri(torch.full((B,), max(1, tmw - 1)), B)
    fd = 1 + ri(torch.full((B,), max(1, fmw - 1)), B)
    h[c1, 0], w[c1, 0], top[c1, 0] = td[c1], F, ri(T - td, B)[c1]
    h[c1, 1], w[c1, 1], left[c1, 1] = T, fd[c1], ri(F - fd, B)[c1]
    # choice 2: eight 8 x 8 holes
    c2 = choice == 2
    h[c2], w[c2] = 8, 8
    left[c2] = torch.floor(u(B, 8) * (F - 8)).long()[c2]
    top[c2] = torch.floor(u(B, 8) * (T - 8)).long()[c2]
    h[~apply], w[~apply] = 0, 0
    return top, h, left, w, u(B, 8).float()


def sample_minmax(x):
    """(min, max) of every sample of x (B, C, T, F) -- the cutout's fill range -- in ONE pass over x: aminmax along the
    contiguous frequency rows, then the small (B, C, T) results.  (x.amin(dim=(1, 2, 3)) and x.amax(...) on the time-cropped view
    of the extractor's output were two passes with 32 outputs each: 0.3 ms apiece of the 12-ms config-4 step.)"""
    lo, hi = torch.aminmax(x, dim=3)
    return lo.amin(dim=(1, 2)), hi.amax(dim=(1, 2))


def random_composite_cutout(x, gen=None, p: float = 0.5, image_aspect_ratio: float = 1.0, n_zero_channels=3):
    """Batched CompositeCutout with torch operators; fill values uniform between the sample's min and max."""
    B, C, T, F = x.shape
    top, h, left, w, u = draw_composite_cutout_batch(B, T, F, gen, p, image_aspect_ratio)
    lo, hi = sample_minmax(x)
    value = lo[:, None] + (hi - lo)[:, None] * u.to(x.device, x.dtype)
    return fill_rects(x, top, h, left, w, value, n_zero_channels=n_zero_channels)


def swap_targets(y_doa, m, audio_format='foa', n_classes: int = 12):
    """The xyz-target half of the channel swaps, branch-free (torch.where on the per-sample bits: no host
    synchronisation when m lives on the device).  y_doa (B,T_lab,3*nc), m (B,>=3|4) -> new tensor; equals the y that
    swap_channels_foa / swap_channels_mic return."""
    nc = n_classes
    b = m.to(device=y_doa.device, dtype=torch.bool)[:, :, None, None]            # (B, bits, 1, 1)
    x, y, z = y_doa[:, :, :nc], y_doa[:, :, nc:2 * nc], y_doa[:, :, 2 * nc:]
    x, y = torch.where(b[:, 0], y, x), torch.where(b[:, 0], x, y)                # bit 0: swap x and y (both formats)
    if audio_format == 'foa':                                                     # bits 1..3: negate x, y, z
        x, y, z = torch.where(b[:, 1], -x, x), torch.where(b[:, 2], -y, y), torch.where(b[:, 3], -z, z)
    else:
        x, y = torch.where(b[:, 1], -y, x), torch.where(b[:, 1], -x, y)          # bit 1: swap x and y, negate both
        y, z = torch.where(b[:, 2], -y, y), torch.where(b[:, 2], -z, z)          # bit 2: negate y and z
    return torch.cat([x, y, z], dim=2)


def draw_augment(B, T, F, audio_format='foa', gen=None, p: float = 0.5, freq_shift_range: int = 10):
    """All per-sample draws of the SALSA training recipe (datamodule.py:45-52 FOA, :73-82 MIC) for a batch, on the host:
    dict(m (B,4) swap bits -- MIC uses the first three --, shift (B,) 0 = none, up (B,), top/h/left/w (B,8), u (B,8))."""
    nbits = 4 if audio_format == 'foa' else 3
    m = torch.zeros((B, 4), dtype=torch.long)
    m[:, :nbits] = torch.randint(0, 2, (B, nbits), generator=gen) * (torch.rand(B, generator=gen) < p)[:, None]
    shift = torch.randint(1, freq_shift_range, (B,), generator=gen) * (torch.rand(B, generator=gen) < p)
    up = torch.randint(0, 2, (B,), generator=gen).bool()
    if audio_format == 'mic':
        top, h, left, w, u = draw_composite_cutout_batch(B, T, F, gen, p, T / 200)
    else:
        top, h, left, w, u = (torch.zeros((B, 8), dtype=torch.long),) * 4 + (torch.zeros((B, 8)),)
    return dict(m=m, shift=shift, up=up, top=top, h=h, left=left, w=w, u=u)


def apply_augment_torch(x, y_doa, d, audio_format='foa', n_classes: int = 12):
    """The drawn augmentation with torch operators (any device): swap -> shift -> cutout; the cutout's fill range is the
    min / max of the sample BEFORE augmentation.  Returns (x', y_doa')."""
    lo, hi = sample_minmax(x)
    if audio_format == 'foa':
        xn, yn = swap_channels_foa(x, y_doa, d['m'].to(x.device), n_classes)
    else:
        xn, yn = swap_channels_mic(x, y_doa, d['m'][:, :3].to(x.device), n_classes)
    xn = shift_up_down(xn, d['shift'], d['up'])
    if audio_format == 'mic':
        value = lo[:, None] + (hi - lo)[:, None] * d['u'].to(x.device, x.dtype)
        xn = fill_rects(xn, d['top'], d['h'], d['left'], d['w'], value, n_zero_channels=3)
    return xn, yn


def _rows_contiguous(x):
    """[B,7,T,F] whose (T,F) blocks are dense: contiguous tensors and time-cropped views of them."""
    return x.stride(3) == 1 and x.stride(2) == x.shape[3] and x.stride(1) >= x.shape[2] * x.shape[3] and x.stride(0) >= 7 * x.stride(1)


def apply_augment_hip(x, d, audio_format='foa'):
    """The drawn augmentation of the FEATURES in one pass of libsalsa_hip.so (salsa_augment_batch): x float32 CUDA
    [B,7,T,F] contiguous -> new tensor.  (The targets are a (B, T_lab, 36) sign / swap: apply_augment_torch's y.)"""
    import ctypes as C
    from . import _lib
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 7 and _rows_contiguous(x)
    B, _, T, F = x.shape
    par = torch.zeros((B, 40), dtype=torch.int32)
    par[:, 0:4] = d['m']
    par[:, 4], par[:, 5] = d['shift'], d['up']
    par[:, 8:16], par[:, 16:24], par[:, 24:32], par[:, 32:40] = d['top'], d['h'], d['left'], d['w']
    # pinned staging: a pageable host-to-device copy would make the host wait for the stream and serialise the step
    par = par.pin_memory().to(x.device, non_blocking=True)
    u = d['u'].float().contiguous().pin_memory().to(x.device, non_blocking=True)
    if 'minmax' in d:                                            # explicit fill range (reference_draws: (0, 1) with u = the values)
        minmax = d['minmax'].float().contiguous().to(x.device)
    else:
        minmax = torch.stack(sample_minmax(x), dim=1).contiguous()
    assert minmax.shape == (B, 2)
    out = torch.empty((B, 7, T, F), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().salsa_augment_batch(C.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), C.c_void_p(out.data_ptr()), B, T, F,
                                             _lib.FORMAT[audio_format], 3 if audio_format == 'mic' else 0,
                                             C.c_void_p(par.data_ptr()), C.c_void_p(u.data_ptr()), C.c_void_p(minmax.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc:
        raise RuntimeError('salsa_augment_batch failed: ' + _lib.last_error())
    return out


def augment_batch(x, y_sed, y_doa, audio_format='foa', gen=None, n_classes: int = 12):
    """The reference's SALSA training recipe on a batch: channel swap (format-specific, changes the targets), frequency
    shift, and -- MIC only -- CompositeCutout(n_zero_channels=3) (datamodule.py:45-52, :73-82).  CUDA float32 features go
    through the one-pass HIP kernel; the (tiny) target transform and everything on CPU use the torch operators."""
    B, _, T, F = x.shape
    d = draw_augment(B, T, F, audio_format, gen)
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _rows_contiguous(x):
        m_dev = d['m'].pin_memory().to(x.device, non_blocking=True)
        return apply_augment_hip(x, d, audio_format), y_sed, swap_targets(y_doa, m_dev, audio_format, n_classes)
    xn, yn = apply_augment_torch(x, y_doa, d, audio_format, n_classes)
    return xn, y_sed, yn
