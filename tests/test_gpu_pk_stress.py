"""GPU stress of the PRODUCTION covariance / eigen solver (packed-float32 pair solve + float64 cold list, salsa_math.h
herm4_gate_eigvec_pk) against the all-float64 instantiation (SALSA_FLAG_FORCE_F64) on > 1e8 gated frames drawn from adversarial
families (tests/pk_families.py): eigenvalue ratios swept through cond_num (dataset/salsa_feature_extraction.py:106), |u_0|^2 over
four decades (:118), inter-channel phases on the +-pi/2 and +-pi lines (:118, :122), a degenerate eigenvalue tail, numerically
rank-1 windows, overall levels 1e-4 and 1e+2, int16-quantised / hard-clipped / DC-offset / keyed-tone audio.  Asserted per family:
gate patterns BIT-EQUAL, every feature inside 1e-6 + 1e-5 |ref| of the float64 instantiation, and a slice of every family inside the
same bar of the CPU oracle.  Printed (and written to gpurun_out/r6_pk_stress.json): gated frames, cold-list rate, worst error / bar."""
import json
import math
import os

import numpy as np
import pytest
import torch

import pk_families as pf
from conftest import ROOT
from test_gpu_parity import _extractor

pytestmark = pytest.mark.gpu

DELTA = 2.0 * math.pi * 24000.0 / (512.0 * 343.0)        # MIC normalisation, salsa_feature_extraction.py:45 (delta * k)


def _run_family(dev, oracle, name, X, fmt, report, cond=5.0, lower_bin=1):
    from salsa_amd import _lib
    B, nb, nt, _ = X.shape
    kw = dict(audio_format=fmt, fmax_doa=9000 if fmt == 'foa' else 4000, cond_num=cond)
    ex = _extractor(**kw)
    ex.set_stats(True)
    out = ex.eigvec_features(X, lower_bin)                                # [B][3][nt][nb] float32, the solver that ships
    st = ex.read_stats()
    ref = _extractor(flags=_lib.FLAG_FORCE_F64, **kw).eigvec_features(X, lower_bin)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(ref).all()), name
    g0, g1 = (out != 0).any(dim=1), (ref != 0).any(dim=1)
    mism = int((g0 != g1).sum())
    d = (out - ref).abs()
    wrapped = 0
    if fmt == 'mic':
        # np.angle's branch cut (:122): at relative phase +-pi the two solvers may land on opposite sides; such a pair differs by
        # one full turn 2 pi / (delta k) and BOTH values sit within the bar of +-pi / (delta k) -- counted, and held to that
        turn = (2.0 * math.pi / (DELTA * (torch.arange(nb, device=dev, dtype=torch.float32) + lower_bin)))[None, None, None, :]
        w = (turn - d).abs() < d
        on_cut = ((ref.abs() - 0.5 * turn).abs() <= 1e-6 + 1e-5 * 0.5 * turn) & ((out.abs() - 0.5 * turn).abs() <= 1e-6 + 1e-5 * 0.5 * turn)
        assert not bool((w & ~on_cut).any()), name
        wrapped = int(w.sum())
        d = torch.where(w, (turn - d).abs(), d)
    rel = d / (1e-6 + 1e-5 * ref.abs())
    worst = float(rel.max())
    # a slice against the CPU oracle (the reference's algorithm in float64): clip 0, 6 bins spread over the band
    bins = [0, 1, nb // 3, nb // 2, nb - 2, nb - 1]
    worst_orc, flips = 0.0, 0
    for b in bins:
        Xb = X[0, b:b + 1].cpu().numpy()
        o, aux = oracle.extract_normalized_eigenvector(Xb, cond, 3, True, fmt, fs=24000, n_fft=512, lower_bin=lower_bin + b, return_aux=True)
        got = out[0, :, :, b].cpu().numpy().astype(np.float64)            # (3, nt)
        want = o[:, 0, :]
        dd = np.abs(got - want)
        if fmt == 'mic':
            tn = 2.0 * math.pi / (DELTA * (b + lower_bin))
            dd = np.minimum(dd, np.abs(tn - dd))
        bad = (dd > 1e-6 + 1e-5 * np.abs(want)).any(axis=0)
        # a gate flip against the ORACLE is admitted only where its threshold margin is numerically zero (as test_gpu_parity._check)
        flip = bad & ((got != 0).any(axis=0) != (want != 0).any(axis=0))
        assert np.all(np.abs(aux['margin'][0][flip]) < 1e-9), (name, fmt, b)
        flips += int(flip.sum())
        ok = ~flip
        worst_orc = max(worst_orc, float((dd[:, ok] / (1e-6 + 1e-5 * np.abs(want[:, ok]))).max()))
    row = dict(family=name, format=fmt, cond=cond, frames=B * nb * nt, gated_frames=st['gated_frames'], cold_frames=st['cold_frames'],
               cold_rate=st['cold_frames'] / max(1, st['gated_frames']), emitted=int(g0.sum()), gate_mismatches=mism,
               worst_err_over_bar=worst, worst_err_over_bar_vs_oracle_slice=worst_orc, oracle_margin_flips=flips, branch_cut_wraps=wrapped)
    report.append(row)
    print('%-18s %s cond %.0f: %9d gated, %8d cold (%.3f %%), %9d emitted, gate mismatches %d, worst |err| / bar %.3f (oracle slice %.3f), '
          'branch-cut wraps %d' % (name, fmt, cond, row['gated_frames'], row['cold_frames'], 100 * row['cold_rate'], row['emitted'], mism,
                                   worst, worst_orc, wrapped))
    assert mism == 0, '%s / %s: gate patterns of the packed and the float64 instantiation differ in %d TF bins' % (name, fmt, mism)
    assert worst <= 1.0, '%s / %s: packed-float32 feature %.3f x the bar away from the float64 instantiation' % (name, fmt, worst)
    assert worst_orc <= 1.0, '%s / %s: %.3f x the bar away from the oracle' % (name, fmt, worst_orc)
    del out, ref, d, rel
    return row


def test_packed_solver_adversarial_stress(oracle):
    assert torch.cuda.is_available()
    dev = torch.device('cuda:0')
    B, nb, nt = 4, 192, 56 * 300
    report = []
    for name in pf.DESIGNED:
        X = pf.family(name, 20210 + len(report), B, nb, nt, dev)
        for fmt in ('foa', 'mic'):
            _run_family(dev, oracle, name, X, fmt, report)
        del X
    X = pf.family('ratio_sweep', 777, B, nb, nt, dev, cond=2.0)           # the other cond_num the goldens use
    _run_family(dev, oracle, 'ratio_sweep', X, 'foa', report, cond=2.0)
    del X
    for name in pf.AUDIO:
        X = pf.audio_family(name, 4040 + len(report), B, 300 * (nt - 1), dev, nb)
        assert X.shape == (B, nb, nt, 4)
        for fmt in ('foa', 'mic'):
            _run_family(dev, oracle, name, X, fmt, report)
        del X
    gated = sum(r['gated_frames'] for r in report)
    cold = sum(r['cold_frames'] for r in report)
    worst = max(r['worst_err_over_bar'] for r in report)
    print('TOTAL: %d gated frames through both instantiations, %d to the float64 cold list (%.3f %%), 0 gate mismatches, worst |err| / bar %.3f'
          % (gated, cold, 100.0 * cold / gated, worst))
    assert gated >= 100_000_000, gated
    # the swept families must actually have reached the hand-back logic (a stress that never trips it proves nothing)
    by = {(r['family'], r['format'], r['cond']): r for r in report}
    assert by[('ratio_sweep', 'foa', 5.0)]['cold_frames'] > 1000 and by[('degenerate_tail', 'foa', 5.0)]['cold_frames'] > 0
    assert by[('u0_sweep', 'foa', 5.0)]['cold_frames'] > 1000
    assert 0.05 < by[('ratio_sweep', 'foa', 5.0)]['emitted'] / by[('ratio_sweep', 'foa', 5.0)]['gated_frames'] < 0.95   # both sides of cond_num
    out_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out_dir):
        json.dump({'bar': '|packed - float64| <= 1e-6 + 1e-5 |float64| per feature; gate patterns bit-equal', 'block': [B, nb, nt],
                   'total_gated_frames': gated, 'total_cold_frames': cold, 'worst_err_over_bar': worst, 'families': report},
                  open(os.path.join(out_dir, 'r6_pk_stress.json'), 'w'), indent=1)


def test_float64_entry_decides_degenerate_tails_like_the_oracle(oracle):
    """salsa_eigvec_batch (float64 output + gate codes, the entry that mirrors extract_normalized_eigenvector) on the
    `degenerate_tail` family -- lambda_2 ~ lambda_3 ~ lambda_4 at lambda_1 / cond, where the characteristic quartic alone loses the
    coherence test to cube-root precision: every TF bin of the block against the oracle, gate codes included."""
    dev = torch.device('cuda:0')
    nb, nt = 24, 56 * 40
    X = pf.family('degenerate_tail', 31, 2, nb, nt, dev)
    for fmt in ('foa', 'mic'):
        ex = _extractor(audio_format=fmt, fmax_doa=9000 if fmt == 'foa' else 4000)
        out, gate = ex.eigvec(X, 1, return_gate=True)
        out, gate = out.cpu().numpy(), gate.cpu().numpy()
        n_flip = 0
        for b in range(2):
            ref, aux = oracle.extract_normalized_eigenvector(X[b].cpu().numpy(), 5.0, 3, True, fmt, fs=24000, n_fft=512, lower_bin=1, return_aux=True)
            emitted_ref = aux['rank'] == 2
            emitted = gate[b] == 2
            flip = emitted != emitted_ref
            assert np.all(np.abs(aux['margin'][flip]) < 1e-9), (fmt, b, float(np.abs(aux['margin'][flip]).max()))
            n_flip += int(flip.sum())
            assert np.array_equal(gate[b] > 0, aux['sig'])                                        # the noise gate, bit for bit
            d = np.abs(out[b] - ref)
            if fmt == 'mic':
                turn = (2.0 * math.pi / (DELTA * (np.arange(nb) + 1.0)))[None, :, None]
                d = np.minimum(d, np.abs(turn - d))
            ok = ~flip[None].repeat(3, axis=0)
            assert float((d[ok] / (1e-8 + 1e-8 * np.abs(ref[ok]))).max()) <= 1.0, fmt
            assert 0.1 < emitted_ref[aux['sig']].mean() < 0.9                                     # both sides of cond_num are present
        print('degenerate_tail through salsa_eigvec_batch (%s): %d flips, all at |margin| < 1e-9' % (fmt, n_flip))
