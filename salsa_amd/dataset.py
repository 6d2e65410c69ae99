"""On-device counterpart of the reference's data layer (dataset/database.py + dataset/dataloader.py) for training
straight from raw audio (BASELINE.json config 4) or from a tree of precomputed feature files (config 3; add_feature_files +
load_feature_scaler = Database.load_chunk_data / load_feature_scaler): clips are extracted on the GPU, kept there in the reference's
concatenated ``(C, sum T, F)`` layout (database.py:230-231), normalised on load with the scaler (first 4 channels only,
:197-202) and sliced into chunks with the reference's segment index arithmetic (:98-119).  ``__getitem__`` returns the
same 4-tuple contract as SeldDataset (dataloader.py:37-62) minus augmentation: (X (7,chunk,F), sed, doa, name).

Nothing here runs in DataLoader worker processes: HIP contexts do not survive fork, so extraction happens on the
training process' stream (SURVEY.md section 7 'hard parts')."""
import os

import numpy as np
import torch

from .extractor import SalsaExtractor, normalize_, scaler_accumulate, scaler_finish


def second2frame(second: float, fs: int, hop_len: int) -> int:
    """database.py:75-81"""
    return int(round(int(second * fs) / hop_len))


def get_segment_idxes(n_frames: int, chunk_len: int, chunk_hop_len: int, downsample_ratio: int, pointer: int):
    """database.py:98-119 verbatim semantics: chunk start indices (at the segment rate) + the advanced pointer."""
    assert n_frames % downsample_ratio == 0, 'n_features_frames is not divisible by downsample ratio'
    n_crop = n_frames // downsample_ratio
    cl, ch = chunk_len // downsample_ratio, chunk_hop_len // downsample_ratio
    assert cl <= n_crop, 'Number of cropped frame is less than chunk len'
    idxes = np.arange(pointer, pointer + n_crop - cl + 1, ch).tolist()
    if (n_crop - cl) % ch != 0:
        idxes.append(pointer + n_crop - cl)             # include the leftover of the cropped data
    return idxes, pointer + n_crop


def sort_tracks(track_number: np.ndarray) -> np.ndarray:
    """database.py:242-251: track ids from the shortest to the longest track (row counts; ids without rows count 0)."""
    n_tracks = int(np.max(track_number)) + 1
    durations = np.zeros((n_tracks,), dtype=np.int32)
    for itrack in range(n_tracks):
        durations[itrack] = np.sum(track_number == itrack)
    return np.argsort(durations)


def load_classwise_gt(gt_meta_fn, n_frames: int, n_classes: int = 12, label_upsample_ratio: int = 8,
                      output_format: str = 'reg_xyz'):
    """Database.load_classwise_gt (dataset/database.py:253-296): a DCASE2021 metadata CSV -- rows
    ``frame_number, sound_class_idx, track_number, azimuth, elevation`` at the 10-Hz label rate, no header -- to the training
    targets of a clip of ``n_frames`` FEATURE frames: sed (n_label_frames, n_classes) float32 in {0, 1} and doa (n_label_frames,
    3 * n_classes) float32 = [x | y | z] unit vectors of the active classes, zeros elsewhere.  Tracks are written from the
    shortest to the longest (:268, :275), rows of a track in file order, so where two tracks hold the same class in the same
    frame the LONGER track's direction stays (and a later row of one track overrides an earlier one).  The trigonometry runs
    in float32 on the float32 radian arrays, as the reference's numpy does (:285-291).  CPU / numpy: a clip has a few thousand
    rows."""
    import pandas as pd
    assert n_frames % label_upsample_ratio == 0, 'mismatch ground truth and feature frame rate'
    if output_format not in ('reg_xyz', 'accdoa'):
        raise ValueError('doa output format {} is not valid'.format(output_format))
    n_label_frames = n_frames // label_upsample_ratio
    df = pd.read_csv(gt_meta_fn, header=None, names=['frame_number', 'sound_class_idx', 'track_number', 'azimuth', 'elevation'])
    frame_number, sound_class_idx, track_number = df['frame_number'].values, df['sound_class_idx'].values, df['track_number'].values
    azimuth, elevation = df['azimuth'].values, df['elevation'].values
    sed = np.zeros((n_label_frames, n_classes), dtype=np.float32)
    azi = np.zeros((n_label_frames, n_classes), dtype=np.float32)
    ele = np.zeros((n_label_frames, n_classes), dtype=np.float32)
    if len(df):
        # one fancy assignment in the reference's write order (tracks shortest first, file order inside a track): numpy keeps the
        # LAST value written to a repeated index, which is what the reference's nested loops leave behind
        rank = np.empty(int(np.max(track_number)) + 1, dtype=np.int64)
        rank[sort_tracks(track_number)] = np.arange(len(rank))
        order = np.argsort(rank[track_number.astype(np.int64)], kind='stable')
        fr, cl = frame_number[order].astype(np.int64), sound_class_idx[order].astype(np.int64)
        sed[fr, cl] = 1.0
        azi[fr, cl] = azimuth[order] * np.pi / 180.0
        ele[fr, cl] = elevation[order] * np.pi / 180.0
    x, y, z = np.cos(azi) * np.cos(ele), np.sin(azi) * np.cos(ele), np.sin(ele)
    off = sed < 1
    x[off] = 0.0
    y[off] = 0.0
    z[off] = 0.0
    return sed, np.concatenate((x, y, z), axis=-1)


class GpuFeatureBank(torch.utils.data.Dataset):
    def __init__(self, extractor: SalsaExtractor = None, fs=24000, hop_len=300, label_rate=10, chunk_len_s=8.0,
                 chunk_hop_len_s=0.5, n_classes=12, max_clip_s=60, device=None):
        self.ex = extractor                                                    # None: a bank of precomputed feature files (add_feature_files)
        self.device = extractor.device if extractor is not None else torch.device(device if device is not None else 'cuda')
        self.fs, self.hop_len, self.label_rate, self.n_classes = fs, hop_len, label_rate, n_classes
        self.chunk_len = second2frame(chunk_len_s, fs, hop_len)
        self.chunk_hop_len = second2frame(chunk_hop_len_s, fs, hop_len)
        self.upsample = int((fs / hop_len) / label_rate)                       # feature frames per label frame (8)
        self.max_frames = int(max_clip_s * label_rate) * self.upsample         # "make sure we have 4800 frames" (:205-207)
        self.blocks, self.names, self.chunk_idx, self.chunk_name = [], [], [], []
        self.sed, self.doa, self.gt_idx = [], [], []
        self.pointer = self.gt_pointer = 0
        self.features = None
        self.mean = self.std = None
        self._sums, self._n = None, 0

    # -------------------------------------------------------------------------------------------- ingest
    def add_clips(self, audio, names, sed=None, doa=None, gt_meta=None):
        """audio: float32 [B,4,N] (numpy or CUDA tensor).  Labels, one of: ``gt_meta`` = per-clip paths of DCASE metadata CSVs
        (read by load_classwise_gt, as Database.load_chunk_data does, database.py:209-211); ``sed`` / ``doa`` = per-clip label
        arrays at label rate, (T_lab, n_classes) and (T_lab, 3*n_classes); none of them -> zeros (inference)."""
        assert self.ex is not None, 'this bank was built without an extractor: it takes feature files only'
        a = audio if torch.is_tensor(audio) else torch.from_numpy(np.ascontiguousarray(audio, np.float32))
        self._ingest(self.ex.extract(a.to(self.ex.device).contiguous()), names, sed, doa, gt_meta)

    def add_feature_files(self, feature_files, names=None, sed=None, doa=None, gt_meta=None):
        """PRECOMPUTED features (BASELINE config 3; what extract_features() wrote: a 'feature' dataset (7, T, F) per clip) instead of
        audio -- Database.load_chunk_data's file loop (database.py:190-207: read, trim to max_nframes_per_file * label_upsample_ratio
        frames; the normalisation of :197-202 happens in finalize(), on the device).  Clips of one shape go to the device in one copy."""
        from . import io as sio
        names = [os.path.splitext(os.path.basename(f))[0] for f in feature_files] if names is None else list(names)
        assert len(names) == len(feature_files)
        arrays = [np.ascontiguousarray(sio.load_arrays(f)['feature'], np.float32) for f in feature_files]
        i = 0
        while i < len(arrays):                                                 # runs of equal shape, file order kept (the pointers are sequential)
            j = i + 1
            while j < len(arrays) and arrays[j].shape == arrays[i].shape:
                j += 1
            feats = torch.from_numpy(np.stack(arrays[i:j])).to(self.device)
            sl = slice(i, j)
            self._ingest(feats, names[sl], None if sed is None else sed[sl], None if doa is None else doa[sl],
                         None if gt_meta is None else gt_meta[sl])
            i = j

    def load_feature_scaler(self, scaler_file):
        """<fmt>_feature_scaler.h5 -> the bank's scaler: Database.load_feature_scaler (database.py:87-96: 'mean', 'std' of shape (4, 1, F))"""
        from . import io as sio
        z = sio.load_arrays(scaler_file)
        self.set_scaler(z['mean'], z['std'])
        return self.mean, self.std

    def _ingest(self, feats, names, sed=None, doa=None, gt_meta=None):
        assert gt_meta is None or (sed is None and doa is None), 'give either metadata CSVs or label arrays'
        n_frames = min(feats.shape[2], self.max_frames)
        n_frames -= n_frames % self.upsample
        feats = feats[:, :, :n_frames].contiguous()
        self._sums = scaler_accumulate(feats, self._sums)
        self._n += feats.shape[0] * n_frames
        if gt_meta is not None:
            assert len(gt_meta) == len(names)
            labels = [load_classwise_gt(fn, n_frames, self.n_classes, self.upsample) for fn in gt_meta]
            sed, doa = [lab[0] for lab in labels], [lab[1] for lab in labels]
        for i, name in enumerate(names):
            idxes, self.pointer = get_segment_idxes(n_frames, self.chunk_len, self.chunk_hop_len, 1, self.pointer)
            gidx, self.gt_pointer = get_segment_idxes(n_frames, self.chunk_len, self.chunk_hop_len, self.upsample,
                                                      self.gt_pointer)
            assert len(idxes) == len(gidx), 'nchunks for sed and gt are different'
            self.blocks.append(feats[i])
            self.names.append(name)
            self.chunk_idx += idxes
            self.gt_idx += gidx
            self.chunk_name += [name] * len(idxes)
            n_lab = n_frames // self.upsample
            dev = feats.device
            self.sed.append(torch.zeros(n_lab, self.n_classes, device=dev) if sed is None
                            else torch.as_tensor(sed[i][:n_lab], dtype=torch.float32, device=dev))
            self.doa.append(torch.zeros(n_lab, 3 * self.n_classes, device=dev) if doa is None
                            else torch.as_tensor(doa[i][:n_lab], dtype=torch.float32, device=dev))
        self.features = None

    def fit_scaler(self):
        """scaler over everything ingested so far (the reference fits it on all dev files)."""
        self.mean, self.std = scaler_finish(self._sums, self._n)
        return self.mean, self.std

    def set_scaler(self, mean, std):
        self.mean, self.std = torch.as_tensor(np.asarray(mean, np.float32)), torch.as_tensor(np.asarray(std, np.float32))

    def finalize(self):
        """concatenate along time (database.py:230) and normalise the spectrogram channels in place."""
        assert self.mean is not None, 'call fit_scaler() or set_scaler() first'
        feats = torch.cat(self.blocks, dim=1).contiguous()                   # (7, sum T, F)
        normalize_(feats[None], self.mean, self.std)
        self.features = feats
        self.sed_all, self.doa_all = torch.cat(self.sed), torch.cat(self.doa)
        self.blocks = []
        return self

    # -------------------------------------------------------------------------------------------- Dataset
    def __len__(self):
        return len(self.chunk_idx)

    def __getitem__(self, i):
        assert self.features is not None, 'call finalize() first'
        s, g = self.chunk_idx[i], self.gt_idx[i]
        x = self.features[:, s:s + self.chunk_len]
        n_lab = self.chunk_len // self.upsample
        return x, self.sed_all[g:g + n_lab], self.doa_all[g:g + n_lab], self.chunk_name[i]

    def batch(self, indices):
        xs, ss, ds, ns = zip(*(self[i] for i in indices))
        return torch.stack(xs), torch.stack(ss), torch.stack(ds), list(ns)
