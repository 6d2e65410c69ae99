#!/usr/bin/env python
"""Two Trainers one after the other in ONE process with the data-parallel gradient path on (what bench.py does: the `crnn` leg,
then the `config4` leg): step times of each.  One rank, 1-rank RCCL group (or SALSA_TWO_TRAINERS_BACKEND=gloo)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29541')
from salsa_amd.crnn.train import Trainer, synthetic_batch  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
backend = os.environ.get('SALSA_TWO_TRAINERS_BACKEND', 'nccl')
rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))   # (torchrun with 2 ranks + gloo: both on cuda:0)
if backend == 'nccl':
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group('gloo', rank=rank, world_size=world)
x, sed, doa = synthetic_batch(32, dev, seed=1 + int(os.environ.get('RANK', 0)))
for k in range(3):
    tr = Trainer(dev, ddp=True, bf16_grad_allreduce=(backend == 'nccl'))
    for _ in range(4):
        tr.train_step(x, sed, doa)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train_step(x, sed, doa)
    torch.cuda.synchronize()
    if rank == 0:
      print('trainer %d (%s, %s): %.2f ms/step   allocated %.1f GB reserved %.1f GB' %
          (k, backend, 'buckets' if tr.grad_sync is not None else 'torch DDP', (time.perf_counter() - t0) * 100,
           torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
dist.destroy_process_group()
