"""SELD evaluation of DCASE-format outputs: location-sensitive detection (ER, F) and class-sensitive localisation (LE, LR)
over 1-s segments, as the reference computes them (metrics/SELD2021_evaluation_metrics.py:81-195 SELDMetrics,
metrics/dcase_utils.py:8-57 load_output_format_file, :185-228 segment_labels; driven by models/interfaces.py:163-180).
Host bookkeeping on a few thousand CSV rows per clip: numpy + scipy's Hungarian solver, no device work.

A "row" is (frame, class, azimuth_deg, elevation_deg[, track]).  Within one segment and class, every frame holds a list
of DOAs; reference and predicted DOAs of a common frame are paired by minimum total great-circle distance, distances
are averaged per reference slot (its position in the frame's list -- the reference ignores track ids), and a slot counts
as a true positive when its average is within ``doa_threshold`` degrees."""
import csv

import numpy as np
from scipy.optimize import linear_sum_assignment

_EPS = np.finfo(float).eps


def load_dcase_csv(path: str, version: str = '2021'):
    """DCASE output-format CSV -> list of (frame, class, azimuth, elevation, track).  4 columns: submission rows
    (frame, class, azi, ele); 5 columns: (frame, class, track, azi, ele).  (dcase_utils.py:8-57, polar formats.)"""
    if version not in ('2020', '2021'):
        raise ValueError('version {} is not implemented'.format(version))
    rows = []
    with open(path, 'r', newline='') as f:
        for w in csv.reader(f):
            if not w:
                continue
            if len(w) == 4:
                rows.append((int(w[0]), int(w[1]), float(w[2]), float(w[3]), 0))
            elif len(w) == 5:
                rows.append((int(w[0]), int(w[1]), float(w[3]), float(w[4]), int(w[2])))
            else:
                raise ValueError('Cartesian (6-column) rows are not used on this path')
    return rows


def segment_rows(rows, max_frames: int = 600, label_rate: int = 10):
    """rows -> {segment: {class: {frame_in_segment: [(azi, ele), ...]}}} in arrival order (dcase_utils.py:185-228)."""
    n_seg = int(np.ceil(max_frames / float(label_rate)))
    seg = {s: {} for s in range(n_seg)}
    for frame, cls, azi, ele, *_ in rows:
        if 0 <= frame < n_seg * label_rate:
            seg[frame // label_rate].setdefault(cls, {}).setdefault(frame % label_rate, []).append((azi, ele))
    return seg


def angular_distance_deg(azi1, ele1, azi2, ele2):
    """Great-circle distance in degrees between polar directions given in degrees (SELD2021...:198-209)."""
    a1, e1, a2, e2 = (np.asarray(v, dtype=np.float64) * np.pi / 180. for v in (azi1, ele1, azi2, ele2))
    d = np.sin(e1) * np.sin(e2) + np.cos(e1) * np.cos(e2) * np.cos(np.abs(a1 - a2))
    return np.arccos(np.clip(d, -1, 1)) * 180 / np.pi


class SeldMetrics:
    def __init__(self, n_classes: int = 12, doa_threshold: float = 20):
        self.n_classes, self.doa_threshold = n_classes, doa_threshold
        self.TP = self.FP = self.FN = 0
        self.S = self.D = self.I = self.Nref = 0
        self.total_DE = 0.0
        self.DE_TP = self.DE_FP = self.DE_FN = 0

    def update(self, pred_rows, gt_rows, max_frames: int = 600, label_rate: int = 10):
        pred, gt = segment_rows(pred_rows, max_frames, label_rate), segment_rows(gt_rows, max_frames, label_rate)
        for s in range(len(gt)):
            seg_fn = seg_fp = 0
            for c in range(self.n_classes):
                g, p = gt[s].get(c), pred[s].get(c)
                n_g = max(len(v) for v in g.values()) if g else 0
                n_p = max(len(v) for v in p.values()) if p else 0
                self.Nref += n_g
                if g and p:
                    per_slot = {}                                          # reference slot -> distances of its pairings
                    for frame, g_doas in g.items():
                        if frame not in p:
                            continue
                        ga, pa = np.array(g_doas, dtype=np.float64), np.array(p[frame], dtype=np.float64)
                        cost = angular_distance_deg(ga[:, None, 0], ga[:, None, 1], pa[None, :, 0], pa[None, :, 1])
                        ri, ci = linear_sum_assignment(cost)
                        for r, col in zip(ri, ci):
                            per_slot.setdefault(int(r), []).append(cost[r, col])
                    if not per_slot:         # no frame in common: the reference books the PREDICTED count as misses (:141-146)
                        seg_fn += n_p
                        self.FN += n_p
                        self.DE_FN += n_p
                    else:
                        for dists in per_slot.values():
                            avg = sum(dists) / len(dists)
                            self.total_DE += avg
                            self.DE_TP += 1
                            if avg <= self.doa_threshold:
                                self.TP += 1
                            else:
                                seg_fp += 1
                                self.FP += 1
                        if n_p > n_g:
                            seg_fp += n_p - n_g
                            self.FP += n_p - n_g
                            self.DE_FP += n_p - n_g
                        elif n_p < n_g:
                            seg_fn += n_g - n_p
                            self.FN += n_g - n_p
                            self.DE_FN += n_g - n_p
                elif g:
                    seg_fn += n_g
                    self.FN += n_g
                    self.DE_FN += n_g
                elif p:
                    seg_fp += n_p
                    self.FP += n_p
                    self.DE_FP += n_p
            self.S += min(seg_fp, seg_fn)
            self.D += max(0, seg_fn - seg_fp)
            self.I += max(0, seg_fp - seg_fn)

    def scores(self):
        """-> (ER, F, LE, LR) (SELD2021...:56-78)."""
        ER = (self.S + self.D + self.I) / float(self.Nref + _EPS)
        F = self.TP / (_EPS + self.TP + 0.5 * (self.FP + self.FN))
        LE = self.total_DE / float(self.DE_TP + _EPS) if self.DE_TP else 180
        LR = self.DE_TP / (_EPS + self.DE_TP + self.DE_FN)
        return ER, F, LE, LR

    def seld_error(self):
        """Aggregate used for model selection (interfaces.py:179): mean of ER, 1-F, LE/180, 1-LR."""
        ER, F, LE, LR = self.scores()
        return (ER + (1.0 - F) + LE / 180.0 + (1.0 - LR)) / 4


def evaluate_csv_dirs(pred_dir: str, gt_dir: str, filenames, n_classes: int = 12, doa_threshold: float = 20,
                      max_frames: int = 600, label_rate: int = 10):
    """evaluate_output_prediction_csv (interfaces.py:163-180): -> (ER, F, LE, LR, seld_error) over the listed CSV files."""
    import os
    m = SeldMetrics(n_classes, doa_threshold)
    for fn in filenames:
        m.update(load_dcase_csv(os.path.join(pred_dir, fn)), load_dcase_csv(os.path.join(gt_dir, fn)), max_frames, label_rate)
    return m.scores() + (m.seld_error(),)
