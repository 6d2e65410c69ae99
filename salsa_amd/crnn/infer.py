"""Batched, sharded inference (BASELINE config 5; reference test_step + write_output_submission, models/seld_models.py:
110-117, models/interfaces.py:210-258): every rank takes a contiguous range of the sorted clip list, runs
features -> CRNN forward -> sigmoid / xyz -> combine_chunks -> DCASE rows on its own GPU, and the per-clip rows are gathered
with one all_gather_object (Python lists of a few hundred integers per clip: control-plane traffic, no tensor collective on
the data path -- SURVEY.md section 8e "Batched inference: shard clips, gather results")."""
from typing import Callable, Dict, List, Sequence

import numpy as np

from ..distributed import shard_list
from .postprocess import combine_chunks, to_dcase_rows


def infer_clips_sharded(names: Sequence[str], featurize: Callable[[List[str]], 'torch.Tensor'],
                        forward: Callable[['torch.Tensor'], tuple], rank: int = 0, world: int = 1, sub_batch: int = 32,
                        sed_threshold: float = 0.3, n_label_frames: int = 600, gather: bool = True) -> Dict[str, list]:
    """names: all clip names (any order; sharded over the SORTED list).  featurize(list of names) -> feature tensor
    [b, 7, T, F] on the model's device (e.g. SalsaExtractor.extract of the clips' audio with the scaler attached, cropped
    to 8 * n_label_frames frames); forward(features) -> (event probabilities [b, n_label_frames, 12], xyz [b, .., 36]), e.g.
    Trainer.infer.  Returns {clip name: DCASE rows} for ALL clips on every rank (gather=True) or for this rank's shard."""
    mine = shard_list(sorted(names), rank, world)
    out = {}
    for i in range(0, len(mine), sub_batch):
        group = mine[i:i + sub_batch]
        prob, xyz = forward(featurize(group))
        prob, xyz = np.asarray(prob.detach().float().cpu()), np.asarray(xyz.detach().float().cpu())
        for name, p, d in zip(group, prob, xyz):
            # one chunk per file (test_chunk_len = the whole clip): combine_chunks places it, as the reference does
            fp = combine_chunks(p[None], n_label_frames, n_label_frames, n_frames=n_label_frames)
            fd = combine_chunks(d[None], n_label_frames, n_label_frames, n_frames=n_label_frames)
            out[name] = to_dcase_rows(fp, fd, sed_threshold=sed_threshold, max_nframes_per_file=n_label_frames)
    if gather and world > 1:
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, out)
        out = {k: v for part in parts for k, v in part.items()}
    return out
