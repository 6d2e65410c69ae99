"""Batched, sharded inference (BASELINE config 5; reference test_step + write_output_submission, models/seld_models.py:
110-117, models/interfaces.py:210-258): every rank takes a contiguous range of the sorted clip list, runs
features -> CRNN forward -> sigmoid / xyz -> combine_chunks -> DCASE rows on its own GPU, and the per-clip rows are gathered
with one all_gather_object (Python lists of a few hundred integers per clip: control-plane traffic, no tensor collective on
the data path -- SURVEY.md section 8e "Batched inference: shard clips, gather results").

The per-rank engine is `infer_pipelined`: sub-batches are ISSUED `depth` deep (default 2), so while the device works on
sub-batch k + 1 the host turns sub-batch k's (600, 12) / (600, 36) outputs -- copied into pinned host slots behind a HIP event,
no device-wide synchronize anywhere -- into DCASE rows.  `stamps` receives (lo, hi, t_issue, t_rows_on_host) per sub-batch: the
true per-clip latency of config 5 (issue of a clip's sub-batch -> its rows exist on the host), measured in the run that is also
timed for throughput."""
import collections
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from ..distributed import shard_list
from .postprocess import combine_chunks, to_dcase_rows


def infer_pipelined(n_items: int, featurize: Callable[[int, int], 'torch.Tensor'], forward: Callable[['torch.Tensor'], tuple],
                    sub_batch: int = 32, depth: int = 2, sed_threshold: float = 0.3, n_label_frames: int = 600,
                    as_array: bool = False, stamps: Optional[list] = None) -> list:
    """featurize(lo, hi) -> feature tensor [hi - lo, 7, T, F] of items lo..hi-1 on the model's device; forward(features) ->
    (event probabilities [b, n_label_frames, 12], xyz [b, n_label_frames, 36]).  Returns the DCASE rows of every item, in
    item order.  The device is never idle waiting for the host: up to `depth` sub-batches are in flight."""
    import torch
    assert depth >= 1 and sub_batch >= 1
    results = [None] * n_items
    slots: Dict[int, dict] = {}
    pending = collections.deque()

    def finish(k, lo, hi, t_issue):
        s = slots[k % depth]
        if s['event'] is not None:
            s['event'].synchronize()                     # this sub-batch's outputs are on the host (nothing else is waited for)
        p, d = s['p'][:hi - lo].numpy(), s['d'][:hi - lo].numpy()
        for i in range(lo, hi):
            # one chunk per file (test_chunk_len = the whole clip): combine_chunks places it, as the reference does
            fp = combine_chunks(p[i - lo][None], n_label_frames, n_label_frames, n_frames=n_label_frames)
            fd = combine_chunks(d[i - lo][None], n_label_frames, n_label_frames, n_frames=n_label_frames)
            results[i] = to_dcase_rows(fp, fd, sed_threshold=sed_threshold, max_nframes_per_file=n_label_frames, as_array=as_array)
        if stamps is not None:
            stamps.append((lo, hi, t_issue, time.perf_counter()))

    for k, lo in enumerate(range(0, n_items, sub_batch)):
        hi = min(n_items, lo + sub_batch)
        t_issue = time.perf_counter()
        prob, xyz = forward(featurize(lo, hi))
        prob, xyz = prob.detach().float(), xyz.detach().float()
        on_gpu = prob.is_cuda
        s = slots.get(k % depth)
        if s is None or s['p'].shape[0] < hi - lo:
            s = slots[k % depth] = {'p': torch.empty((hi - lo,) + tuple(prob.shape[1:]), dtype=torch.float32, pin_memory=on_gpu),
                                    'd': torch.empty((hi - lo,) + tuple(xyz.shape[1:]), dtype=torch.float32, pin_memory=on_gpu),
                                    'event': torch.cuda.Event() if on_gpu else None}
        s['p'][:hi - lo].copy_(prob, non_blocking=True)
        s['d'][:hi - lo].copy_(xyz, non_blocking=True)
        if s['event'] is not None:
            s['event'].record()
        pending.append((k, lo, hi, t_issue))
        while len(pending) >= depth:                     # slot (k + 1) % depth is free again before sub-batch k + 1 is issued
            finish(*pending.popleft())
    while pending:
        finish(*pending.popleft())
    return results


def infer_clips_sharded(names: Sequence[str], featurize: Callable[[List[str]], 'torch.Tensor'],
                        forward: Callable[['torch.Tensor'], tuple], rank: int = 0, world: int = 1, sub_batch: int = 32,
                        sed_threshold: float = 0.3, n_label_frames: int = 600, gather: bool = True, depth: int = 2,
                        stamps: Optional[list] = None) -> Dict[str, list]:
    """names: all clip names (any order; sharded over the SORTED list).  featurize(list of names) -> feature tensor
    [b, 7, T, F] on the model's device (e.g. SalsaExtractor.extract of the clips' audio with the scaler attached, cropped
    to 8 * n_label_frames frames); forward(features) -> (event probabilities [b, n_label_frames, 12], xyz [b, .., 36]), e.g.
    Trainer.infer.  Returns {clip name: DCASE rows} for ALL clips on every rank (gather=True) or for this rank's shard."""
    mine = shard_list(sorted(names), rank, world)
    rows = infer_pipelined(len(mine), lambda lo, hi: featurize(mine[lo:hi]), forward, sub_batch=sub_batch, depth=depth,
                           sed_threshold=sed_threshold, n_label_frames=n_label_frames, stamps=stamps)
    out = dict(zip(mine, rows))
    if gather and world > 1:
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, out)
        out = {k: v for part in parts for k, v in part.items()}
    return out
