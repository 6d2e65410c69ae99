"""Inference post-processing of the SELD CRNN to the DCASE submission format (models/interfaces.py:97-139 combine_chunks,
:210-258 write_classwise_output_to_file): overlapping-chunk averaging, SED threshold, xyz -> azimuth/elevation in
integer degrees, one CSV row per (frame, active class).  CPU / numpy: it is bookkeeping on (600, 12) arrays."""
import csv

import numpy as np


def combine_chunks(frame_output_pred: np.ndarray, chunk_len: int, chunk_hop_len: int, n_frames: int = 600,
                   combine_method: str = 'mean') -> np.ndarray:
    """(n_chunks, chunk_len, C) label-rate chunk outputs -> (n_frames, C) file output; overlaps are averaged pairwise
    in arrival order exactly like the reference (a running (a+b)/2, not a uniform mean)."""
    out = np.zeros((n_frames,) + frame_output_pred.shape[2:], dtype=np.float32)
    starts = np.arange(0, n_frames - chunk_len + 1, chunk_hop_len).tolist()
    if (n_frames - chunk_len) % chunk_hop_len != 0:
        starts.append(n_frames - chunk_len)
    overlap = chunk_len - chunk_hop_len
    assert abs(frame_output_pred.shape[0] - len(starts)) < 2
    for i, s in enumerate(starts):
        e = s + chunk_len
        if i == 0:
            out[s:e] = frame_output_pred[i]
            continue
        if combine_method == 'mean':
            out[s:s + overlap] = (out[s:s + overlap] + frame_output_pred[i, :overlap]) / 2
        elif combine_method == 'gmean':
            out[s:s + overlap] = np.sqrt(out[s:s + overlap] * frame_output_pred[i, :overlap])
        else:
            raise ValueError('combine method {} is unknown'.format(combine_method))
        out[s + overlap:e] = frame_output_pred[i, overlap:]
    return out


def to_dcase_rows(event_prob: np.ndarray, doa_xyz: np.ndarray, sed_threshold: float = 0.3, n_classes: int = 12,
                  max_nframes_per_file: int = 600, eval_version: str = '2021', as_array: bool = False):
    """event_prob (T, 12) sigmoid outputs, doa_xyz (T, 36) -> list of [frame, class, (0,) azimuth, elevation] rows, in the
    reference's order (frame ascending, class ascending within a frame: models/interfaces.py:232-258).  The reference walks the
    600 frames in a Python loop; here the active (frame, class) pairs come from one np.nonzero (row-major = the same order) and
    the angles are computed for those pairs only -- 1024 clips per inference step make the loop the bottleneck otherwise.
    as_array: the rows as one (n, 5 | 4) int64 array instead of a list of lists (bulk inference keeps them that way)."""
    active = event_prob >= sed_threshold
    assert active.shape[0] >= max_nframes_per_file, 'n_output_frames of sed < max_nframes_per_file'
    t, c = np.nonzero(active[:max_nframes_per_file, :n_classes])
    x, y, z = doa_xyz[t, c], doa_xyz[t, n_classes + c], doa_xyz[t, 2 * n_classes + c]
    azi = np.around(np.arctan2(y, x) * 180.0 / np.pi).astype(np.int64)
    ele = np.around(np.arctan2(z, np.sqrt(x ** 2 + y ** 2)) * 180.0 / np.pi).astype(np.int64)
    azi[azi == 180] = -180
    cols = [t, c, np.zeros_like(t), azi, ele] if eval_version == '2021' else [t, c, azi, ele]
    rows = np.stack(cols, axis=1).astype(np.int64)
    return rows if as_array else rows.tolist()


def write_dcase_csv(path: str, rows) -> None:
    with open(path, 'w', newline='') as f:
        csv.writer(f).writerows(rows)
