"""Device augmentations against numpy restatements of utilities/transforms.py:286-320 and :365-437 (CPU tensors)."""
import numpy as np
import torch


def _ref_swap(x, y_doa, m, nc=12):
    x_new, y = x.copy(), y_doa.copy()
    if m[0]:
        x_new[1], x_new[3] = x[3], x[1]
        x_new[-3], x_new[-1] = x[-1], x[-3]
        y[:, :nc], y[:, nc:2 * nc] = y_doa[:, nc:2 * nc], y_doa[:, :nc]
    if m[1]:
        x_new[-1] = -x_new[-1]
        y[:, :nc] = -y[:, :nc]
    if m[2]:
        x_new[-3] = -x_new[-3]
        y[:, nc:2 * nc] = -y[:, nc:2 * nc]
    if m[3]:
        x_new[-2] = -x_new[-2]
        y[:, 2 * nc:] = -y[:, 2 * nc:]
    return x_new, y


def test_swap_channels_matches_reference_for_all_16_draws():
    from salsa_amd.augment import swap_channels_foa
    rng = np.random.RandomState(0)
    x = rng.randn(16, 7, 9, 11).astype(np.float32)
    y = rng.randn(16, 5, 36).astype(np.float32)
    m = np.array([[(i >> b) & 1 for b in range(4)] for i in range(16)])
    xn, yn = swap_channels_foa(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(m))
    for i in range(16):
        rx, ry = _ref_swap(x[i], y[i], m[i])
        assert np.array_equal(xn[i].numpy(), rx) and np.array_equal(yn[i].numpy(), ry)


def test_frequency_shift_matches_numpy_reflect_pad():
    from salsa_amd.augment import shift_up_down
    rng = np.random.RandomState(1)
    x = rng.randn(6, 7, 4, 200).astype(np.float32)
    shift = np.array([0, 1, 9, 3, 9, 5])
    up = np.array([True, True, True, False, False, False])
    out = shift_up_down(torch.from_numpy(x), torch.from_numpy(shift), torch.from_numpy(up)).numpy()
    for i in range(6):
        s = int(shift[i])
        if s == 0:
            ref = x[i]
        elif up[i]:
            ref = np.pad(x[i], ((0, 0), (0, 0), (s, 0)), mode='reflect')[:, :, :200]
        else:
            ref = np.pad(x[i], ((0, 0), (0, 0), (0, s)), mode='reflect')[:, :, s:]
        assert np.array_equal(out[i], ref), i


def test_random_wrappers_shapes_and_rates():
    from salsa_amd.augment import random_shift_up_down, random_swap_channels_foa
    g = torch.Generator().manual_seed(0)
    x, sed, doa = torch.randn(64, 7, 8, 200), torch.zeros(64, 4, 12), torch.randn(64, 4, 36)
    xn, s2, dn = random_swap_channels_foa(x, sed, doa, gen=g)
    assert xn.shape == x.shape and s2 is sed and dn.shape == doa.shape
    assert torch.equal(xn[:, 0], x[:, 0]) and torch.equal(xn[:, 2], x[:, 2])          # W rows never move
    xs = random_shift_up_down(x, gen=g)
    changed = (xs != x).flatten(1).any(dim=1).float().mean()
    assert 0.2 < float(changed) < 0.8


def test_reference_train_transform_reproduces_golden_under_the_same_seed():
    """Same np.random seed -> same augmented sample as the reference's SeldDataset (golden g11, 48 seeds x 2 formats)."""
    import hashlib
    from conftest import load_golden
    from salsa_amd.augment import reference_train_transform
    meta, a = load_golden('g11_augment')
    x, y_sed, y_doa = (torch.from_numpy(a[k]) for k in ('x', 'y_sed', 'y_doa'))
    for fmt in ('foa', 'mic'):
        changed = 0
        for s, (hx, hd) in zip(meta['seeds'], meta['sha'][fmt]):
            np.random.seed(s)
            xo, so, do = reference_train_transform(x, y_sed, y_doa, audio_format=fmt, rng=np.random,
                                                   image_aspect_ratio=meta['image_aspect_ratio'])
            xo, do = xo.numpy(), do.numpy()
            if ('%s_x_%d' % (fmt, s)) in a:
                assert np.array_equal(xo, a['%s_x_%d' % (fmt, s)]) and np.array_equal(do, a['%s_doa_%d' % (fmt, s)])
            assert hashlib.sha256(np.ascontiguousarray(xo).tobytes()).hexdigest() == hx, (fmt, s)
            assert hashlib.sha256(np.ascontiguousarray(do).tobytes()).hexdigest() == hd, (fmt, s)
            assert so is y_sed
            changed += not np.array_equal(xo, a['x'])
        assert changed > len(meta['seeds']) // 2


def test_batched_mic_swap_and_cutout():
    from salsa_amd.augment import fill_rects, random_composite_cutout, random_swap_channels_mic, swap_channels_mic
    g = torch.Generator().manual_seed(3)
    x, sed, doa = torch.randn(64, 7, 40, 200), torch.zeros(64, 4, 12), torch.randn(64, 4, 36)
    xn, s2, dn = random_swap_channels_mic(x, sed, doa, gen=g)
    assert s2 is sed and not torch.equal(xn, x)
    # every draw is an involution on the microphone permutation: applying the same single swap twice restores x
    for bit in range(3):
        m = torch.zeros(64, 3, dtype=torch.long)
        m[:, bit] = 1
        x1, d1 = swap_channels_mic(x, doa, m)
        x2, d2 = swap_channels_mic(x1, d1, m)
        assert torch.allclose(x2, x, atol=1e-6) and torch.equal(d2, doa)
    out = random_composite_cutout(x, gen=g, p=1.0, image_aspect_ratio=40 / 200)
    diff = out != x
    assert diff.any(dim=(1, 2, 3)).all()                                   # p = 1: every sample is cut
    assert (out[:, 4:][diff[:, 4:]] == 0).all()                            # spatial rows are zero-filled
    frac = diff[:, 0].float().mean(dim=(1, 2))
    assert frac.max() <= 0.5 and frac.min() > 0
    same = fill_rects(x, *(torch.zeros(64, 2, dtype=torch.long),) * 4, torch.zeros(64, 2))
    assert torch.equal(same, x)


def test_branch_free_target_swap_equals_the_swap_cores():
    from salsa_amd.augment import swap_channels_foa, swap_channels_mic, swap_targets
    rng = np.random.RandomState(3)
    y = torch.from_numpy(rng.randn(16, 5, 36).astype(np.float32))
    x = torch.zeros(16, 7, 1, 1)
    m4 = torch.tensor([[(i >> b) & 1 for b in range(4)] for i in range(16)])
    assert torch.equal(swap_targets(y, m4, 'foa'), swap_channels_foa(x, y, m4)[1])
    m3 = m4[:, :3]
    assert torch.equal(swap_targets(y, m3, 'mic'), swap_channels_mic(x, y, m3)[1])
