"""Error budget of the packed-float32 gate certificate (salsa_math.h herm4_gate_eigvec_pk, SALSA_PK_GATE_TOL), MEASURED: for every
frame of the adversarial families of tests/pk_families.py (and the bench's synthetic clips) the Taylor coefficients t0..t3 of the
characteristic quartic at mu1 / cond as the float32 pair code computes them from its float32 covariance, against the same
coefficients from the exact covariance in long double (tests/hostemu hostemu_pk_coeffs).  The certificate keeps a float32 gate
decision only when every |t_k| >= SALSA_PK_GATE_TOL, so it is sound iff max |t32_k - t64_k| < SALSA_PK_GATE_TOL.

  python tools/pk_coeff_study.py            -> profiles/r6_pk_coeff_study.json
(The host build contracts no FMAs and divides exactly where the device uses v_rcp_f32: the device's errors differ in the last bit
of each operation, not in magnitude; the GPU stress tests/test_gpu_pk_stress.py holds the device build end to end.)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
U = 2.0 ** -24


def windows(X):
    """X (nb, nt, 4) complex64 -> pair items (n, 8, 4): frames t-3 .. t+4 for even t (wrap), all bins"""
    nt = X.shape[1] // 2 * 2
    idx = (np.arange(0, nt, 2)[:, None] + np.arange(-3, 5)[None, :]) % X.shape[1]
    return np.ascontiguousarray(X[:, idx, :].reshape(-1, 8, 4))


def coeffs(L, W, cond):
    W = np.ascontiguousarray(W.astype(np.complex64))
    n = W.shape[0]
    t32, t64 = np.zeros((2 * n, 7)), np.zeros((2 * n, 7))
    L.hostemu_pk_coeffs(W.ctypes.data, n, cond, t32.ctypes.data, t64.ctypes.data)
    ok = np.isfinite(t32).all(axis=1) & np.isfinite(t64).all(axis=1)
    return t32[ok], t64[ok]


def summarise(t32, t64):
    """t0..t3: coefficient + evaluation error at the float32 c (every frame); mu1: over the frames whose Newton iteration the
    kernel's own test calls converged (|last step| <= 4e-6 mu1), relative to mu1"""
    err = np.abs(t32[:, :5] - t64[:, :5])
    conv = np.abs(t32[:, 5]) <= 4e-6 * t32[:, 4]
    err[:, 4] = np.where(conv, err[:, 4] / np.maximum(t64[:, 4], 1e-30), 0.0)
    names = ('t0', 't1', 't2', 't3', 'mu1')
    return {'frames': int(len(err)), 'newton_converged': int(conv.sum()),
            'max_abs_err': {k: float(err[:, i].max()) for i, k in enumerate(names)},
            'p9999_abs_err': {k: float(np.quantile(err[:, i], 0.9999)) for i, k in enumerate(names)},
            'rms_err': {k: float(np.sqrt((err[:, i] ** 2).mean())) for i, k in enumerate(names)}}


if __name__ == '__main__':
    import torch
    import pk_families as pf
    from tools.pk_study import emu, stft_windows
    L = emu()
    L.hostemu_pk_coeffs.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_void_p, C.c_void_p]
    rep = {'unit': 'absolute error on the trace-1..2 scale; u = 2^-24 = %.3g' % U, 'SALSA_PK_GATE_TOL': 2e-5, 'families': {}}
    worst = np.zeros(5)
    total = 0
    for cond in (5.0, 2.0):
        for name in pf.DESIGNED:
            X = pf.family(name, 99, 1, 48, 56 * 60, 'cpu', cond=cond)[0].numpy()
            t32, t64 = coeffs(L, windows(X), cond)
            r = summarise(t32, t64)
            rep['families']['%s cond %g' % (name, cond)] = r
            worst = np.maximum(worst, [r['max_abs_err'][k] for k in ('t0', 't1', 't2', 't3', 'mu1')])
            total += r['frames']
            print('%-24s %8d frames  max |err| t0 %.2e t1 %.2e t2 %.2e t3 %.2e mu1 %.2e' % ((name + ' c%g' % cond, r['frames']) + tuple(r['max_abs_err'].values())))
        for name in pf.AUDIO:
            X = pf.audio_family(name, 7, 1, 300 * (56 * 60 - 1), 'cpu', 48)[0].numpy()
            t32, t64 = coeffs(L, windows(X), cond)
            r = summarise(t32, t64)
            rep['families']['%s cond %g' % (name, cond)] = r
            worst = np.maximum(worst, [r['max_abs_err'][k] for k in ('t0', 't1', 't2', 't3', 'mu1')])
            total += r['frames']
            print('%-24s %8d frames  max |err| t0 %.2e t1 %.2e t2 %.2e t3 %.2e mu1 %.2e' % ((name + ' c%g' % cond, r['frames']) + tuple(r['max_abs_err'].values())))
    W, _ = stft_windows(2021, 20.0, 'foa')
    t32, t64 = coeffs(L, W, 5.0)
    r = summarise(t32, t64)
    rep['families']['bench clip (synth_clip 2021, 20 s, FOA band)'] = r
    worst = np.maximum(worst, [r['max_abs_err'][k] for k in ('t0', 't1', 't2', 't3', 'mu1')])
    total += r['frames']
    rep['total_frames'] = int(total)
    rep['worst_abs_err'] = dict(zip(('t0', 't1', 't2', 't3', 'mu1'), map(float, worst)))
    rep['tolerance_over_worst_coefficient_error'] = float(2e-5 / worst[:4].max())
    rep['worst_relative_mu1_error_converged'] = float(worst[4])
    print('TOTAL %d frames: worst coefficient error %.3g = %.1f u; SALSA_PK_GATE_TOL / worst = %.1f; worst relative mu1 error (converged frames) %.3g'
          % (total, worst[:4].max(), worst[:4].max() / U, 2e-5 / worst[:4].max(), worst[4]))
    json.dump(rep, open(os.path.join(ROOT, 'profiles', 'r6_pk_coeff_study.json'), 'w'), indent=1)
