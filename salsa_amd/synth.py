"""Deterministic synthetic 4-channel clips (no dataset exists in the container or on the GPU box).

The recipe is the one SURVEY.md section 8(d) fixes for every BASELINE.json config: 0.01*N(0,1) diffuse noise plus
three AR(1)(0.9)-filtered directional bursts with per-channel gains U(-1,1) (channel 0 gain 1) and per-channel
circular delays {0,k,2k,3k} samples.  Legacy ``np.random.RandomState`` streams are frozen by numpy, so a (seed,
n_samples) pair names the same float32 clip here, on the GPU box and in the golden fixtures (which also store a
SHA-256 of the clip so a drifted generator fails loudly instead of silently).
"""
import hashlib

import numpy as np

FS = 24000


def _ar1(x: np.ndarray, a: float = 0.9) -> np.ndarray:
    """y[n] = x[n] + a*y[n-1] (float64).  Blocked closed form so 60-s clips do not need a Python loop per sample."""
    y = np.empty_like(x)
    blk = 4096
    pw = a ** np.arange(1, blk + 1)
    # lower-triangular Toeplitz apply per block, carrying the state between blocks
    state = 0.0
    k = a ** np.arange(blk)
    for s in range(0, x.shape[0], blk):
        seg = x[s:s + blk]
        n = seg.shape[0]
        conv = np.convolve(seg, k[:n])[:n]
        y[s:s + n] = conv + state * pw[:n]
        state = y[s + n - 1]
    return y


def synth_clip(seed: int, n_samples: int = 60 * FS, n_ch: int = 4, fs: int = FS) -> np.ndarray:
    """(n_ch, n_samples) float32 clip."""
    rng = np.random.RandomState(seed)
    y = 0.01 * rng.randn(n_ch, n_samples)
    burst = int(min(5 * fs, max(64, n_samples // 6)))
    for _ in range(3):
        start = int(rng.randint(0, max(1, n_samples - burst)))
        src = _ar1(rng.randn(burst))
        gains = np.concatenate(([1.0], rng.uniform(-1.0, 1.0, n_ch - 1)))
        k = int(rng.randint(0, 4))
        for c in range(n_ch):
            y[c, start:start + burst] += gains[c] * np.roll(src, c * k)
    return np.ascontiguousarray(y.astype(np.float32))


def synth_batch(seed0: int, batch: int, n_samples: int = 60 * FS) -> np.ndarray:
    """(batch, 4, n_samples) float32, clip i uses seed0+i (BASELINE config 2: seeds 2021..2052)."""
    return np.stack([synth_clip(seed0 + i, n_samples) for i in range(batch)], axis=0)


def synth_stft_block(seed: int, n_bins: int, n_frames: int, n_ch: int = 4, kind: str = 'mixed') -> np.ndarray:
    """(n_bins, n_frames, n_ch) complex64 spectrogram block for unit-testing the eigenvector stage alone."""
    rng = np.random.RandomState(seed)
    noise = 0.02 * (rng.randn(n_bins, n_frames, n_ch) + 1j * rng.randn(n_bins, n_frames, n_ch))
    X = noise
    if kind != 'noise':
        # two directional events, each a rank-1 steering vector times a smooth envelope over a time span
        for ev in range(2):
            t0 = int(rng.randint(0, max(1, n_frames // 2)))
            t1 = t0 + int(rng.randint(n_frames // 4, n_frames // 2 + 1))
            steer = rng.uniform(-1, 1, n_ch) * np.exp(1j * rng.uniform(-np.pi, np.pi, n_ch) * (0.15 if ev == 0 else 1.0))
            steer[0] = 1.0
            s = (rng.randn(n_bins, n_frames) + 1j * rng.randn(n_bins, n_frames)) * rng.uniform(0.2, 1.5, (n_bins, 1))
            env = np.zeros(n_frames)
            env[t0:t1] = 1.0
            X = X + (s * env)[:, :, None] * steer[None, None, :]
    return np.ascontiguousarray(X.astype(np.complex64))


def sha256_of(arr: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
