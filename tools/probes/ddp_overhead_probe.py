#!/usr/bin/env python
"""What does the DistributedDataParallel wrapper cost BEFORE any communication?  One rank, a 1-rank RCCL group: the kernels of a
training step that the plain (no DDP) step does not have, and the step's wall time, for the bf16-compressed and the float32
gradient hooks.   python tools/probes/ddp_overhead_probe.py [--fp32-grads]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from torch.profiler import ProfilerActivity, profile

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29519')
from salsa_amd.crnn.train import Trainer, synthetic_batch  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
x, sed, doa = synthetic_batch(32, dev, seed=1)


def table(ddp):
    tr = Trainer(dev, ddp=ddp, bf16_grad_allreduce='--fp32-grads' not in sys.argv)
    for _ in range(6):
        tr.train_step(x, sed, doa)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            tr.train_step(x, sed, doa)
        torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        tr.train_step(x, sed, doa)
    t1.record()
    torch.cuda.synchronize()
    return {e.key: (e.count / 3, e.device_time_total / 3e3) for e in prof.key_averages() if e.device_time_total > 0}, t0.elapsed_time(t1) / 20


plain, ms_plain = table(False)
ddp, ms_ddp = table(True)
print('step: plain %.3f ms, DDP on one rank %.3f ms (+%.3f)' % (ms_plain, ms_ddp, ms_ddp - ms_plain))
print('kernel time per step: plain %.3f ms, DDP %.3f ms' % (sum(v[1] for v in plain.values()), sum(v[1] for v in ddp.values())))
rows = []
for k, (n, ms) in ddp.items():
    n0, ms0 = plain.get(k, (0, 0.0))
    if abs(ms - ms0) > 0.003 or abs(n - n0) >= 1:
        rows.append((ms - ms0, k, n0, n, ms0, ms))
for k, (n0, ms0) in plain.items():
    if k not in ddp:
        rows.append((-ms0, k, n0, 0, ms0, 0.0))
rows.sort(reverse=True)
print('%-90s %8s %8s %9s %9s' % ('kernel (differences only)', 'n plain', 'n ddp', 'ms plain', 'ms ddp'))
for d, k, n0, n, ms0, ms in rows[:40]:
    print('%-90s %8.1f %8.1f %9.3f %9.3f' % (k[:90], n0, n, ms0, ms))
dist.destroy_process_group()
