"""Training step of the SELD CRNN on PyTorch-ROCm: bf16 autocast, channels-last activations, every convolution / BatchNorm /
pool / GRU scan on the hand-written HIP kernels of salsa_amd/csrc (MIOpen only if they are switched off), Adam with the
reference's piecewise-linear learning-rate schedule (utilities/learning_utils.py:17-52, experiments/configs/seld.yml:
37-52), and data parallelism as one process per GPU: bucketed gradient all-reduce on RCCL over xGMI overlapped with the backward
pass (grad_sync.py; torch's DDP behind SALSA_GRAD_SYNC=ddp) -- the reference only has Lightning's implicit ddp_spawn,
experiments/train.py:98.
BatchNorm statistics stay per rank, as in the reference (no SyncBN)."""
import os

import numpy as np
import torch
import torch.distributed as dist

from .loss import seld_loss
from .model import SeldCRNN

MILESTONES = (0.0, 0.1, 0.7, 1.0)
LRS = (3e-4, 3e-4, 3e-4, 1e-4)


def lr_at(progress: float) -> float:
    """piecewise-linear interpolation of LRS over training progress in [0, 1]."""
    return float(np.interp(min(max(progress, 0.0), 1.0), MILESTONES, LRS))


def synthetic_batch(batch: int, device, seed: int = 0, n_frames: int = 640, n_freq: int = 200, n_classes: int = 12):
    """TNSSE2021-shaped synthetic training chunks: features N(0,1) on the 4 spectrogram
    channels and U(-1,1)*Bernoulli(0.25) on the 3 spatial channels; SED Bernoulli(0.05); DOA unit vectors where active."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.empty(batch, 7, n_frames, n_freq)
    x[:, :4] = torch.randn(batch, 4, n_frames, n_freq, generator=g)
    x[:, 4:] = (torch.rand(batch, 3, n_frames, n_freq, generator=g) * 2 - 1) * \
        (torch.rand(batch, 3, n_frames, n_freq, generator=g) < 0.25)
    n_lab = n_frames // 8                                          # 80 feature frames/s -> 10 labels/s
    sed = (torch.rand(batch, n_lab, n_classes, generator=g) < 0.05).float()
    v = torch.randn(batch, n_lab, 3, n_classes, generator=g)
    v = v / v.norm(dim=2, keepdim=True)
    doa = (v * sed[:, :, None, :]).reshape(batch, n_lab, 3 * n_classes)
    return x.to(device), sed.to(device), doa.to(device)


class Trainer:
    def __init__(self, device, amp_dtype=torch.bfloat16, ddp: bool = None, total_steps: int = 1313 * 50,
                 bf16_grad_allreduce: bool = True, seed: int = 2021):
        torch.manual_seed(seed)
        self.device = torch.device(device)
        if self.device.type == 'cuda' and os.environ.get('SALSA_MIOPEN_FIND', '0') == '1':
            torch.backends.cudnn.benchmark = True           # MIOpen solver search: measured SLOWER here (701 vs 724 chunks/s), off by default
        self.amp_dtype = amp_dtype
        self.total_steps = total_steps
        self.step_idx = 0
        model = SeldCRNN().to(self.device)
        self.channels_last = self.device.type == 'cuda' and os.environ.get('SALSA_CHANNELS_LAST', '1') == '1'
        if self.channels_last:
            model = model.to(memory_format=torch.channels_last)
        if self.device.type == 'cuda':
            model.pack_parameters()                          # (stacked GRU / head parameters as views: no per-step copies)
        self.raw_model = model
        use_ddp = ddp if ddp is not None else (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        self.grad_sync = None
        if use_ddp and os.environ.get('SALSA_GRAD_SYNC', 'buckets') != 'ddp':
            # bucketed all-reduce overlapped with the backward pass, a few multi-tensor kernels per step (grad_sync.py; torch's
            # DDP wrapper costs this model +0.5 - 1.0 ms per 11-ms step on ONE rank: 143 device copies, one per gradient)
            from .grad_sync import BucketedGradSync
            wire = torch.bfloat16 if (bf16_grad_allreduce and self.device.type == 'cuda') else torch.float32
            self.grad_sync = BucketedGradSync(list(model.parameters()), bucket_mb=25, wire_dtype=wire, module=model)
            self.grad_sync.broadcast_parameters(0)           # (SALSA_GRAD_SYNC=ddp selects torch's DistributedDataParallel instead)
        elif use_ddp:
            from torch.nn.parallel import DistributedDataParallel as DDP
            kw = dict(device_ids=[self.device.index], output_device=self.device.index) if self.device.type == 'cuda' else {}
            model = DDP(model, bucket_cap_mb=25, gradient_as_bucket_view=True, **kw)   # 56.4 MB fp32 -> 3 buckets
            if bf16_grad_allreduce and self.device.type == 'cuda':
                from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
                model.register_comm_hook(None, default_hooks.bf16_compress_hook)       # halves the bytes on xGMI
        self.model = model
        # one fused kernel per parameter group on the GPU (same update rule; SALSA_FUSED_ADAM=0 -> torch's foreach path)
        fused = self.device.type == 'cuda' and os.environ.get('SALSA_FUSED_ADAM', '1') == '1'
        if self.device.type == 'cuda' and os.environ.get('SALSA_HIP_ADAM', '1') == '1':
            from .optim import HipAdam                        # the same rule in ONE launch (torch's fused path: three, 0.17 ms)
            self.opt = HipAdam(self.model.parameters(), lr=LRS[0])
        else:
            self.opt = torch.optim.Adam(self.model.parameters(), lr=LRS[0], fused=fused)

    def _input_layout(self, x):
        """channels-last for MIOpen -- unless the first layer runs the stem kernel, which reads the extractor's planar
        float32 layout (time crops included) as it is."""
        from . import nn_ops
        if not self.channels_last:
            return x
        if (nn_ops.USE_HIP_CONV and self.amp_dtype == torch.bfloat16 and x.dtype == torch.float32 and x.dim() == 4
                and x.shape[1] <= 8 and nn_ops._planar_rows(x)):
            return x
        return x.contiguous(memory_format=torch.channels_last)

    def train_step(self, x, sed, doa):
        self.model.train()
        lr = lr_at(self.step_idx / max(1, self.total_steps))
        for gparam in self.opt.param_groups:
            gparam['lr'] = lr
        x = self._input_layout(x)
        self.opt.zero_grad(set_to_none=True)
        if self.grad_sync is not None:
            self.grad_sync.begin()
        with torch.autocast(device_type=self.device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            pred = self.model(x)
        loss, sed_l, doa_l = seld_loss(pred, sed, doa)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()                # averaged gradients are in place
        self.opt.step()
        self.step_idx += 1
        return loss.detach(), sed_l.detach(), doa_l.detach()

    @torch.no_grad()
    def infer(self, x):
        """eval forward (bf16 autocast): sigmoid SED probabilities and xyz at label rate (inference path, config 5)."""
        self.model.eval()
        x = self._input_layout(x)
        with torch.autocast(device_type=self.device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            out = self.raw_model(x)
        return torch.sigmoid(out['event_frame_logit'].float()), out['doa_frame_output'].float()
