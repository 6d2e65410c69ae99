"""SELD metrics against the reference's SELD2021 metric code (golden g12, tools/make_golden.py::g12_metrics)."""
import os

import numpy as np

from conftest import load_golden


def test_seld_metrics_match_reference(tmp_path):
    from salsa_amd.crnn.metrics import SeldMetrics, evaluate_csv_dirs, load_dcase_csv
    meta, a = load_golden('g12_metrics')
    m = SeldMetrics(n_classes=12, doa_threshold=20)
    assert list(m.scores()) == meta['no_data_scores']
    os.makedirs(tmp_path / 'gt')
    os.makedirs(tmp_path / 'pred')
    for fi in range(meta['n_files']):
        for kind in ('gt', 'pred'):
            np.savetxt(tmp_path / kind / ('f%d.csv' % fi), a['%s%d' % (kind, fi)], fmt='%d', delimiter=',')
        pred, gt = load_dcase_csv(str(tmp_path / 'pred' / ('f%d.csv' % fi))), load_dcase_csv(str(tmp_path / 'gt' / ('f%d.csv' % fi)))
        assert len(gt) == len(a['gt%d' % fi]) and gt[0][2:] == tuple(float(v) if i < 2 else int(v) for i, v in
                                                                       enumerate((a['gt%d' % fi][0][3], a['gt%d' % fi][0][4], a['gt%d' % fi][0][2])))
        m.update(pred, gt)
        ref = a['cumulative'][fi]
        counters = [m.TP, m.FP, m.FN, m.S, m.D, m.I, m.Nref, m.DE_TP, m.DE_FP, m.DE_FN]
        assert counters == [int(v) for v in ref[4:14]], fi                 # integer bookkeeping: exact
        np.testing.assert_allclose(m.total_DE, ref[14], rtol=1e-12)
        np.testing.assert_allclose(m.scores(), ref[:4], rtol=1e-12)
    ER, F, LE, LR, err = evaluate_csv_dirs(str(tmp_path / 'pred'), str(tmp_path / 'gt'), ['f%d.csv' % i for i in range(4)])
    np.testing.assert_allclose([ER, F, LE, LR], a['cumulative'][-1][:4], rtol=1e-12)
    np.testing.assert_allclose(err, (ER + 1 - F + LE / 180 + 1 - LR) / 4)


def test_submission_rows_round_trip_through_the_metric(tmp_path):
    """postprocess.to_dcase_rows -> CSV -> metrics: a perfect prediction scores ER 0, F 1, LE 0, LR 1."""
    from salsa_amd.crnn.metrics import SeldMetrics, load_dcase_csv
    from salsa_amd.crnn.postprocess import to_dcase_rows, write_dcase_csv
    rng = np.random.RandomState(0)
    prob = (rng.rand(600, 12) < 0.03).astype(np.float32)
    azi, ele = rng.randint(-179, 180, (600, 12)), rng.randint(-60, 61, (600, 12))
    xyz = np.concatenate([np.cos(np.deg2rad(azi)) * np.cos(np.deg2rad(ele)), np.sin(np.deg2rad(azi)) * np.cos(np.deg2rad(ele)),
                          np.sin(np.deg2rad(ele))], axis=1)
    rows = to_dcase_rows(prob, xyz)
    write_dcase_csv(str(tmp_path / 'p.csv'), rows)
    got = load_dcase_csv(str(tmp_path / 'p.csv'))
    assert len(got) == int(prob.sum())
    t, c = got[5][0], got[5][1]
    assert (got[5][2], got[5][3]) == (azi[t, c], ele[t, c])
    m = SeldMetrics()
    m.update(got, got)
    ER, F, LE, LR = m.scores()
    assert ER == 0 and abs(F - 1) < 1e-12 and LE < 1e-5 and abs(LR - 1) < 1e-12
