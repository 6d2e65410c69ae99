"""world_size-2 gloo tests (CPU) of the N>1 path: clip sharding (no data-path collective) and the all-reduced scaler.
The features fed to the scaler come from the CPU oracle, which is allowed in tests/ as the checker."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def test_shard_ranges_partition_the_list():
    from salsa_amd.distributed import shard_list, shard_range
    for n in (0, 1, 5, 32, 600):
        for world in (1, 2, 3, 8):
            got = []
            sizes = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                got += list(range(lo, hi))
                sizes.append(hi - lo)
            assert got == list(range(n)) and max(sizes) - min(sizes) <= 1
    assert shard_list(list('abcde'), 1, 2) == ['d', 'e']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from salsa_amd.distributed import scaler_allreduce, scaler_partial, shard_list
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    names = sorted(f for f in os.listdir(tmp) if f.endswith('.npy'))
    mine = shard_list(names, rank, world)
    n, s, ss = scaler_partial(np.load(os.path.join(tmp, f)) for f in mine)
    mean, std = scaler_allreduce(n, s, ss, 200)
    np.save(os.path.join(tmp, 'mean_rank%d' % rank), mean)
    np.save(os.path.join(tmp, 'std_rank%d' % rank), std)
    np.save(os.path.join(tmp, 'mine_rank%d' % rank), np.array(mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_clips', [3, 1])
def test_two_rank_sharded_scaler_matches_single_process(tmp_path, oracle, n_clips):
    from salsa_amd.synth import synth_clip
    feats = []
    for i in range(n_clips):                      # ragged lengths; with n_clips=1 rank 1 gets an EMPTY shard
        f = oracle.extract_salsa(synth_clip(40 + i, 9000 + 3000 * i))
        np.save(tmp_path / ('clip%02d.npy' % i), f)
        feats.append(f)
    ref_mean, ref_std = oracle.compute_scaler(feats)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    seen = []
    for r in range(2):
        np.testing.assert_allclose(np.load(tmp_path / ('mean_rank%d.npy' % r)), ref_mean, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(np.load(tmp_path / ('std_rank%d.npy' % r)), ref_std, rtol=1e-5, atol=1e-6)
        seen += list(np.load(tmp_path / ('mine_rank%d.npy' % r)))
    assert sorted(seen) == ['clip%02d.npy' % i for i in range(n_clips)]      # every clip exactly once
