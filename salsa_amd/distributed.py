"""Multi-GPU use of the feature path: one process per GPU, clips sharded across ranks, NO collective on the data path
(clips are independent; a clip is never split in time).  The only exchange is the scaler
(dataset/salsa_feature_extraction.py:204-262 computes it over ALL dev files): each rank accumulates float64 sum /
sum-of-squares / count over its shard and one all_reduce(SUM) of a [3][4][F] tensor finishes it -- RCCL over xGMI when
the process group is ``nccl`` on GPUs, gloo in the CPU tests."""
import os

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous balanced range [lo, hi) of a sorted list for ``rank`` of ``world`` (sizes differ by at most one)."""
    assert 0 <= rank < world
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_list(items, rank: int, world: int):
    lo, hi = shard_range(len(items), rank, world)
    return list(items[lo:hi])


def scaler_partial(features, n_feature_channels: int = 4):
    """float64 partial sums over an iterable of (7,T,F) arrays -> (count, sum[4,F], sumsq[4,F]); (0, None, None) if empty."""
    n, s, ss = 0, None, None
    for f in features:
        assert f.shape[0] == 7, 'only support n_channels = 7, got {}'.format(f.shape[0])
        x = np.asarray(f[:n_feature_channels], np.float64)
        if s is None:
            s, ss = np.zeros((n_feature_channels, x.shape[2])), np.zeros((n_feature_channels, x.shape[2]))
        s += x.sum(axis=1)
        ss += (x * x).sum(axis=1)
        n += x.shape[1]
    return n, s, ss


def scaler_finish(n, s, ss):
    """-> (mean, std) each (4,1,F) float32, population std like sklearn's StandardScaler.var_."""
    mean = s / n
    var = np.maximum(ss / n - mean * mean, 0.0)
    return mean[:, None, :].astype(np.float32), np.sqrt(var)[:, None, :].astype(np.float32)


def scaler_allreduce(n, s, ss, n_freq: int, device=None):
    """All-reduce the partial sums over the default process group; every rank gets the global (mean, std)."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros((3, 4, n_freq), dtype=torch.float64, device=device)
    if n:
        buf[0] = torch.from_numpy(s).to(buf.device)
        buf[1] = torch.from_numpy(ss).to(buf.device)
        buf[2].fill_(float(n))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    h = buf.cpu().numpy()
    return scaler_finish(h[2, 0, 0], h[0], h[1])


def extract_features_sharded(data_config: str, cond_num: float = 5, n_hopframes: int = 3, is_tracking: bool = True,
                             is_compress_high_freq: bool = True, task: str = 'feature_scaler', batch_size: int = 32,
                             feature_type: str = 'salsa') -> None:
    """extract_features() of salsa_amd.features / salsa_amd.lite_features run by every rank of an initialised
    torch.distributed job (launch with ``python -m torch.distributed.run --nproc-per-node N``): rank r extracts the
    r-th contiguous range of each split's sorted file list into the shared feature tree; the scaler is all-reduced and
    written by rank 0.  Single process (no process group): identical to the plain function."""
    import torch
    import torch.distributed as dist
    from . import io as sio
    from .extractor import SalsaExtractor
    from .features import _extract_split, _parse

    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    cfg, audio_format, fs, n_fft, hop_length, win_length, fmin_doa, fmax_doa = _parse(data_config)
    fmax_doa = int(np.min((fmax_doa, fs // 2)))
    if feature_type == 'salsa':
        desc = '{}fs_{}nfft_{}nhop_{}cond_{}fmaxdoa'.format(fs, n_fft, hop_length, int(cond_num), int(fmax_doa))
        desc += '' if is_tracking else '_notracking'
        desc += '' if is_compress_high_freq else '_nocompress'
        if audio_format not in ('foa', 'mic'):
            raise ValueError('Unknown audio format {}'.format(audio_format))
    else:
        assert feature_type in ['salsa_lite', 'salsa_ipd'], 'Invalid feature type {}'.format(feature_type)
        assert audio_format == 'mic', 'SALSA-Lite and SALSA-IPD are only for MIC format!'
        desc = '{}fs_{}nfft_{}nhop_{}fmaxdoa'.format(fs, n_fft, hop_length, int(fmax_doa))
    root = os.path.join(cfg['feature_dir'], feature_type, audio_format, desc)
    if task in ['feature_scaler', 'feature']:
        ex = SalsaExtractor(fs=fs, n_fft=n_fft, hop_len=hop_length, win_len=win_length, fmin_doa=fmin_doa,
                            fmax_doa=fmax_doa, cond_num=cond_num, n_hopframes=n_hopframes, is_tracking=is_tracking,
                            is_compress_high_freq=is_compress_high_freq, audio_format=audio_format,
                            feature_type=feature_type)
        from . import features as _f
        fused = {} if (task == 'feature_scaler' and _f.FUSED_SCALER) else None   # this rank's scaler statistics, taken on its device
        for split in (audio_format + '_dev', audio_format + '_eval'):
            feature_dir = os.path.join(root, split)
            if rank == 0:
                _f._rmtree_parallel(feature_dir)
                os.makedirs(feature_dir, exist_ok=True)
            if world > 1:
                dist.barrier()
            _extract_split(ex, os.path.join(cfg['data_dir'], split), feature_dir, fs, batch_size,
                           shard=(rank, world), clear=False, scaler=fused if split.endswith('_dev') else None)
        if world > 1:
            dist.barrier()
        if fused is not None:
            # every rank must take the same branch: a rank with an empty shard (or on the serial loop) has no sums of its own
            ok = torch.tensor([0 if fused.get('unavailable') else 1], dtype=torch.int32)
            if world > 1:
                okd = ok.to(ex.device) if dist.get_backend() == 'nccl' else ok
                dist.all_reduce(okd, op=dist.ReduceOp.MIN)
                ok = okd.cpu()
            if int(ok.item()) == 1:
                F = ex.output_shape(ex.params.n_fft)[2]
                h = fused['sums'].detach().cpu().numpy() if fused.get('sums') is not None else np.zeros((2, 4, F))
                device = ex.device if (dist.is_initialized() and dist.get_backend() == 'nccl') else None
                mean, std = scaler_allreduce(fused.get('n', 0), h[0], h[1], F, device=device)
                if rank == 0:
                    sio.save_arrays(os.path.join(root, audio_format + '_feature_scaler.h5'), mean=mean, std=std)
                if world > 1:
                    dist.barrier()
                return
    if task in ['feature_scaler', 'scaler']:
        dev_dir = os.path.join(root, audio_format + '_dev')
        files = shard_list(sio.feature_files(dev_dir), rank, world)
        feats = (sio.load_arrays(os.path.join(dev_dir, f))['feature'] for f in files)
        n, s, ss = scaler_partial(feats)
        any_file = sio.load_arrays(os.path.join(dev_dir, sio.feature_files(dev_dir)[0]))['feature']
        device = torch.device('cuda', torch.cuda.current_device()) if (dist.is_initialized() and dist.get_backend() == 'nccl') else None
        mean, std = scaler_allreduce(n, s, ss, any_file.shape[2], device=device)
        if rank == 0:
            sio.save_arrays(os.path.join(root, audio_format + '_feature_scaler.h5'), mean=mean, std=std)
        if world > 1:
            dist.barrier()
