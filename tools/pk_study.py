"""Error study of the packed-float32 pair solve (salsa_math.h: herm4_gate_eigvec_pk) against the float64 solve on the same
float64 covariances, run on the CPU through tests/hostemu (the g++ build of the kernels' per-thread arithmetic).

  python tools/pk_study.py [--clips 4] [--seconds 20] [--random 2000000]

Windows: (a) every (bin, frame pair) of the STFT (oracle restatement of librosa's) of the bench's synthetic clips, FOA band
1..191 -- tracker-gated or not, all are solved; (b) random steering-vector windows over a wide range of SNR, |u_0| and gate
margins.  Reports, for FOA and MIC: how many frames the packed solve certifies / hands back (`unsure`), gate disagreements among
certified frames (must be 0 outside |margin| < 1e-9), and the feature error of certified frames in units of the test bar
(1e-6 + 1e-5 |ref|).  Writes profiles/r4_pk_study.json."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EMU_SRC = os.path.join(ROOT, 'tests', 'hostemu', 'hostemu.cpp')
EMU_SO = os.path.join(ROOT, 'tests', 'hostemu', 'libhostemu.so')


def emu():
    hdr = os.path.join(ROOT, 'salsa_amd', 'csrc', 'salsa_math.h')
    if not os.path.exists(EMU_SO) or os.path.getmtime(EMU_SO) < max(os.path.getmtime(EMU_SRC), os.path.getmtime(hdr)):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-o', EMU_SO, EMU_SRC])
    L = C.CDLL(EMU_SO)
    L.hostemu_pk_pairs.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_int, C.c_double, C.c_int] + [C.c_void_p] * 6
    return L


COV32 = 1


def run_pairs(L, X, cond, fmt, dk):
    """X (n, 8, 4) complex64 -> dict of per-(item, frame) arrays"""
    X = np.ascontiguousarray(X.astype(np.complex64))
    n = X.shape[0]
    out = dict(rank64=np.zeros(2 * n, np.uint8), margin=np.zeros(2 * n), e64=np.zeros((2 * n, 3)),
               pass32=np.zeros(2 * n, np.uint8), unsure=np.zeros(2 * n, np.uint8), e32=np.zeros((2 * n, 3), np.float32))
    L.hostemu_pk_pairs(X.ctypes.data, n, cond, fmt, dk, COV32, out['rank64'].ctypes.data, out['margin'].ctypes.data,
                       out['e64'].ctypes.data, out['pass32'].ctypes.data, out['unsure'].ctypes.data, out['e32'].ctypes.data)
    return out


def summarise(o):
    cert = o['unsure'] == 0
    r64, p32 = o['rank64'].astype(bool), o['pass32'].astype(bool)
    dis = cert & (r64 != p32)
    both = cert & r64 & p32
    ref = o['e64'][both]
    got = o['e32'][both].astype(np.float64)
    fin = np.isfinite(ref).all(axis=1) & np.isfinite(got).all(axis=1)
    err = np.abs(got[fin] - ref[fin].astype(np.float32).astype(np.float64)) / (1e-6 + 1e-5 * np.abs(ref[fin]))
    return {'frames': int(len(cert)), 'gate_pass_f64': int(r64.sum()), 'unsure': int((~cert).sum()),
            'unsure_frac_of_gate_pass': float((~cert & r64).sum() / max(1, r64.sum())),
            'unsure_frac_all': float((~cert).mean()),
            'gate_disagreements_certified': int(dis.sum()),
            'gate_disagreements_outside_1e-9': int((dis & (np.abs(o['margin']) >= 1e-9)).sum()),
            'certified_features': int(fin.sum()), 'nonfinite': int((~fin).sum()),
            'err_over_bar_max': float(err.max()) if err.size else 0.0,
            'err_over_bar_p999': float(np.quantile(err.max(axis=1), 0.999)) if err.size else 0.0,
            'err_over_bar_median': float(np.median(err.max(axis=1))) if err.size else 0.0}


def stft_windows(seed, seconds, fmt):
    from oracle import oracle as orc
    from salsa_amd.synth import synth_clip
    y = synth_clip(seed, int(seconds * 24000))
    S = np.stack([orc.stft(y[c]) for c in range(4)], axis=-1)            # (257, T, 4) complex
    lo, hi = (1, 192) if fmt == 'foa' else (1, 85)
    S = S[lo:hi].astype(np.complex64)
    T = S.shape[1] // 2 * 2
    idx = (np.arange(0, T, 2)[:, None] + np.arange(-3, 5)[None, :]) % S.shape[1]     # frames t-3 .. t+4 (wrap)
    W = S[:, idx, :]                                                     # (bins, pairs, 8, 4)
    k = np.repeat(np.arange(lo, hi), W.shape[1])
    return W.reshape(-1, 8, 4), k


def random_windows(rng, n):
    X = (rng.randn(n, 8, 4) + 1j * rng.randn(n, 8, 4)) * rng.uniform(0.02, 0.6, (n, 1, 1))
    steer = rng.uniform(-1, 1, (n, 4)) * np.exp(1j * rng.uniform(-np.pi, np.pi, (n, 4)))
    steer[:, 0] = rng.choice([1.0, 1.0, 0.3, 0.05, 1e-3], n)            # small |u_0| now and then
    amp = (rng.randn(n, 8, 1) + 1j * rng.randn(n, 8, 1)) * 10 ** rng.uniform(-1.0, 1.0, (n, 1, 1))
    X = X + amp * steer[:, None, :]
    return X * 10 ** rng.uniform(-4, 3, (n, 1, 1))                       # any overall level


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=4)
    ap.add_argument('--seconds', type=float, default=20.0)
    ap.add_argument('--random', type=int, default=2000000)
    ap.add_argument('--cov64', action='store_true', help='float64 covariance rounded once (the first variant) instead of the float32 one')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r4_pk_study.json'))
    a = ap.parse_args()
    COV32 = 0 if a.cov64 else 1
    L = emu()
    rep = {'tolerances': 'see salsa_math.h SALSA_PK_*', 'bar': '1e-6 + 1e-5 |ref| (tests/test_gpu_parity.py)'}
    for fmt, code in (('foa', 0), ('mic', 1)):
        parts = []
        for i in range(a.clips):
            W, k = stft_windows(2021 + i, a.seconds, fmt)
            if fmt == 'foa':
                parts.append(run_pairs(L, W, 5.0, 0, 1.0))
            else:                                                        # dk depends on the bin: run bin by bin
                for kk in np.unique(k):
                    parts.append(run_pairs(L, W[k == kk], 5.0, 1, 0.858673 * kk))
        o = {key: np.concatenate([p[key] for p in parts]) for key in parts[0]}
        rep['%s_stft_clips' % fmt] = summarise(o)
        rng = np.random.RandomState(11 + code)
        o = run_pairs(L, random_windows(rng, a.random // 2), 5.0, code, 0.858673 * 7)
        rep['%s_random' % fmt] = summarise(o)
        for key in ('%s_stft_clips' % fmt, '%s_random' % fmt):
            print(key, json.dumps(rep[key]))
    json.dump(rep, open(a.out, 'w'), indent=1)
