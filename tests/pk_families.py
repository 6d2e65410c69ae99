"""Adversarial spectrum families for the packed-float32 covariance / eigen solver (tests/test_gpu_pk_stress.py; round-5 review,
weak 1: the solver's hand-back thresholds had only met the bench's synthetic clips).  TEST INFRASTRUCTURE.

Every family returns a complex64 block X [B][n_bins][n_frames][4] for salsa_eigvec_feature_batch (the reference's
extract_normalized_eigenvector input, dataset/salsa_feature_extraction.py:17-129, per clip).  Two constructions:

* DESIGNED windows.  x_t = env_t * sum_k s_k(t) exp(2 pi i k t / 7) v_k with V = [v_0..v_3] unitary per (clip, bin): over ANY seven
  consecutive frames the cross terms sum_t exp(2 pi i (k - l) t / 7) vanish, so the 7-frame covariance (:99-100) is
  sum_k s_k^2 v_k v_k^H -- eigenvalues s_k^2 and eigenvectors v_k chosen at will, up to the float32 rounding of X and the slow drift
  of s_k(t), which is what sweeps a quantity THROUGH a threshold: lambda_1 / lambda_2 ramps across cond_num in steps far below
  float32 resolution, |u_0|^2 sweeps four decades, inter-channel phases sit within 1e-3 of the +-pi/2 and +-pi singular lines.
  env_t is a 48-loud / 8-quiet frame pattern (quiet = 1e-4 of the loud amplitude): the noise-floor tracker (:63-87) keeps its floor
  at the quiet level, so ~86 % of all frames are gated IN and reach the solver; windows that straddle a quiet stretch are
  partial sums (other, rank-deficient covariances).
* AUDIO-derived spectra (torch.stft of 4-channel signals): int16-quantised, hard-clipped, DC-offset and keyed-tone signals.

Names follow the reference: lambda = singular values of R (:103), cond = condition_number (:106), u_0 = u[0, 0] (:118)."""
import math

import torch

LOUD, QUIET, QUIET_AMP = 48, 8, 1e-4


def envelope(n_frames, device, dtype=torch.float64):
    t = torch.arange(n_frames, device=device)
    return torch.where((t % (LOUD + QUIET)) < LOUD, 1.0, QUIET_AMP).to(dtype)


def _unitary(gen, B, nb, device, first_col=None):
    """random unitary [B][nb][4][4] (complex128); first_col [B][nb][4] complex128 -> column 0 is that direction (unit-normalised)"""
    z = torch.randn((B, nb, 4, 4, 2), generator=gen, device=device, dtype=torch.float64)
    m = torch.complex(z[..., 0], z[..., 1])
    if first_col is not None:
        m[..., :, 0] = first_col
    q, r = torch.linalg.qr(m)
    if first_col is not None:
        # QR leaves column 0 = first_col / (|first_col| e^{i arg r00}): put the requested phases back (relative phases matter)
        d = r[..., 0, 0] / r[..., 0, 0].abs()
        q = q.clone()
        q[..., :, 0] = q[..., :, 0] * d[..., None]
    return q


def designed(gen, B, nb, nt, device, lam, V, level=1.0):
    """lam: callable(t [nt] float64 in [0,1]) -> eigenvalues [B][nb][nt][4] (float64, any scale); V unitary [B][nb][4][4]."""
    t = torch.arange(nt, device=device, dtype=torch.float64)
    s = torch.sqrt(lam(t / max(1, nt - 1)))                                           # [B][nb][nt][4]
    k = torch.arange(4, device=device, dtype=torch.float64)
    ph = 2.0 * math.pi * ((t[:, None] * k[None, :]) % 7.0) / 7.0                     # exact integer phase steps
    coef = s * torch.polar(torch.ones_like(ph), ph)[None, None]                        # [B][nb][nt][4] over k
    x = torch.einsum('bftk,bfck->bftc', coef.to(torch.complex128), V)                 # sum_k coef_k v_k[c]
    x = x * (level * envelope(nt, device))[None, None, :, None]
    return x.to(torch.complex64).contiguous()


def _rand(gen, shape, device):
    return torch.rand(shape, generator=gen, device=device, dtype=torch.float64)


def family(name, seed, B, nb, nt, device, cond=5.0):
    """-> X complex64 [B][nb][nt][4]"""
    gen = torch.Generator(device=device).manual_seed(seed)
    one = torch.ones((B, nb, 1), device=device, dtype=torch.float64)
    if name in ('ratio_sweep', 'level_1e-4', 'level_1e+2'):
        # lambda_1 / lambda_2 ramps over cond (1 -+ 1e-4) along time, with a per-bin offset so that every bin crosses the
        # coherence threshold (:106) at another frame; lambda_3,4 random below lambda_2
        off = (_rand(gen, (B, nb, 1), device) - 0.5) * 1e-4
        l34 = _rand(gen, (B, nb, 1, 2), device)

        def lam(tt):
            eps = (2.0 * tt[None, None, :] - 1.0) * 1e-4 + off
            l2 = 1.0 / (cond * (1.0 + eps))
            return torch.stack([one.expand_as(l2), l2, l2 * l34[..., 0], l2 * l34[..., 1] * l34[..., 0]], dim=-1)
        level = {'ratio_sweep': 1.0, 'level_1e-4': 1e-4, 'level_1e+2': 1e2}[name]
        return designed(gen, B, nb, nt, device, lam, _unitary(gen, B, nb, device), level)
    if name == 'u0_sweep':
        # |u_0|^2 of the principal eigenvector from 1e-4 to 1 across bins and (slowly, by a factor 10) along time is not
        # possible with a fixed V: the bins carry the sweep, time carries a comfortable ratio drift 8..40
        a0 = torch.sqrt(10.0 ** (-4.0 * _rand(gen, (B, nb), device)))                # |u_0| in [1e-2, 1]
        rest = torch.randn((B, nb, 3, 2), generator=gen, device=device, dtype=torch.float64)
        rest = torch.complex(rest[..., 0], rest[..., 1])
        rest = rest / rest.abs().pow(2).sum(-1, keepdim=True).sqrt() * torch.sqrt(1.0 - a0 ** 2)[..., None]
        v1 = torch.cat([a0[..., None].to(torch.complex128), rest], dim=-1)
        l34 = _rand(gen, (B, nb, 1, 2), device)

        def lam(tt):
            l2 = 1.0 / (8.0 + 32.0 * tt[None, None, :] * one)
            return torch.stack([one.expand_as(l2), l2, l2 * l34[..., 0], l2 * l34[..., 1]], dim=-1)
        return designed(gen, B, nb, nt, device, lam, _unitary(gen, B, nb, device, v1))
    if name == 'phase_lines':
        # principal eigenvector entries at relative phase within 1e-3 of +-pi/2 (Re(u_i / u_0) -> 0: FOA conditioning, :118) and
        # +-pi (the branch cut of np.angle(u_i conj(u_0)), :122)
        mag = 0.2 + _rand(gen, (B, nb, 4), device)
        base = torch.tensor([0.5 * math.pi, -0.5 * math.pi, math.pi, -math.pi], device=device, dtype=torch.float64)
        pick = torch.randint(0, 4, (B, nb, 3), generator=gen, device=device)
        phi = base[pick] + (_rand(gen, (B, nb, 3), device) - 0.5) * 2e-3
        v1 = torch.polar(mag, torch.cat([torch.zeros((B, nb, 1), device=device, dtype=torch.float64), phi], dim=-1))
        l34 = _rand(gen, (B, nb, 1, 2), device)

        def lam(tt):
            l2 = 1.0 / (6.0 + 20.0 * tt[None, None, :] * one)
            return torch.stack([one.expand_as(l2), l2, l2 * l34[..., 0], l2 * l34[..., 1]], dim=-1)
        return designed(gen, B, nb, nt, device, lam, _unitary(gen, B, nb, device, v1))
    if name == 'degenerate_tail':
        # lambda_2 ~ lambda_3 ~ lambda_4 (equal to 1e-6 relative: a triple root of the quartic at the threshold) with
        # lambda_1 / lambda_2 ramping 4.9 .. 5.1 (x cond / 5)
        jit = 1.0 + (_rand(gen, (B, nb, 1, 2), device) - 0.5) * 2e-6

        def lam(tt):
            l2 = 1.0 / (cond * (0.98 + 0.04 * tt[None, None, :] * one))
            return torch.stack([one.expand_as(l2), l2, l2 * jit[..., 0], l2 * jit[..., 1]], dim=-1)
        return designed(gen, B, nb, nt, device, lam, _unitary(gen, B, nb, device))
    if name == 'rank1_plus_floor':
        # a dominant source over a diffuse floor 30 .. 90 dB down (lambda_2,3,4 -> 0: the scaled matrix is numerically rank 1)
        fl = 10.0 ** (-3.0 - 6.0 * _rand(gen, (B, nb, 1), device))

        def lam(tt):
            l2 = fl * (1.0 + tt[None, None, :])
            return torch.stack([one.expand_as(l2), l2, 0.5 * l2, 0.25 * l2], dim=-1)
        return designed(gen, B, nb, nt, device, lam, _unitary(gen, B, nb, device))
    if name == 'random_mix':
        # random complex Gaussian spectra with two rank-1 sources per clip, random SNR: the generic case (gate margins anywhere)
        z = torch.randn((B, nb, nt, 4, 2), generator=gen, device=device, dtype=torch.float32)
        x = 0.05 * torch.complex(z[..., 0], z[..., 1])
        for _ in range(2):
            st = torch.randn((B, nb, 1, 4, 2), generator=gen, device=device, dtype=torch.float32)
            st = torch.complex(st[..., 0], st[..., 1])
            sg = torch.randn((B, nb, nt, 1, 2), generator=gen, device=device, dtype=torch.float32)
            amp = 10.0 ** (-1.5 + 2.0 * torch.rand((B, nb, 1, 1), generator=gen, device=device, dtype=torch.float32))
            x = x + amp * torch.complex(sg[..., 0], sg[..., 1]) * st
        return (x * envelope(nt, device, torch.float32)[None, None, :, None]).contiguous()
    raise KeyError(name)


def audio_family(name, seed, B, n_samples, device, nb):
    """4-channel audio with a defect -> complex64 spectra [B][nb][T][4] of bins 1..nb (torch.stft, n_fft 512, hop 300, periodic
    Hann, centre / reflect: librosa's framing).  The float32 STFT differs from the extractor's float64 one in the last bits, which is
    irrelevant here: both solver instantiations read the same X."""
    from salsa_amd.synth import synth_clips_device
    y = synth_clips_device(seed, B, n_samples, device=device)
    gen = torch.Generator(device=device).manual_seed(seed)
    if name == 'int16_quantised':
        y = torch.round(y * (32768.0 / 8.0)).clamp(-32768, 32767) * (8.0 / 32768.0)       # full scale = 8 (the bursts peak near 6)
    elif name == 'int16_quiet':
        y = torch.round(y * 0.002 * 32768.0).clamp(-32768, 32767) / 32768.0               # 54 dB down: a few LSBs of noise
    elif name == 'hard_clipped':
        y = y.clamp(-0.5, 0.5)
    elif name == 'dc_offset':
        y = y + torch.tensor([0.3, -0.2, 0.25, 0.1], device=device)[None, :, None]
    elif name == 'keyed_tones':
        t = torch.arange(n_samples, device=device, dtype=torch.float64) / 24000.0
        y = 0.001 * y
        for j in range(6):
            f0 = float(200.0 + 1500.0 * j + 37.0 * (seed % 7))
            ph = 2.0 * math.pi * torch.rand((B, 4, 1), generator=gen, device=device, dtype=torch.float64)
            gain = 0.2 + torch.rand((B, 4, 1), generator=gen, device=device, dtype=torch.float64)
            key = ((t * (1.5 + 0.4 * j)) % 1.0 < 0.8).to(torch.float64)                   # on 80 % of the time, re-keyed
            y = y + (gain * torch.sin(2.0 * math.pi * f0 * t[None, None, :] + ph) * key).to(torch.float32)
    else:
        raise KeyError(name)
    win = torch.hann_window(512, periodic=True, device=device, dtype=torch.float32)
    S = torch.stft(y.reshape(B * 4, n_samples), n_fft=512, hop_length=300, win_length=512, window=win, center=True,
                   pad_mode='reflect', return_complex=True)                                # [B*4][257][T]
    S = S.reshape(B, 4, 257, -1)[:, :, 1:1 + nb]                                           # bins 1..nb
    return S.permute(0, 2, 3, 1).contiguous()                                              # [B][nb][T][4]


DESIGNED = ('ratio_sweep', 'level_1e-4', 'level_1e+2', 'u0_sweep', 'phase_lines', 'degenerate_tail', 'rank1_plus_floor', 'random_mix')
AUDIO = ('int16_quantised', 'int16_quiet', 'hard_clipped', 'dc_offset', 'keyed_tones')
