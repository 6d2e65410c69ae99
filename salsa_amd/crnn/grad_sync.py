"""Data-parallel gradient averaging for the SELD CRNN trainer: bucketed all-reduce on RCCL over xGMI overlapped with the backward
pass -- what torch's DistributedDataParallel does for the reference's ``ddp_spawn`` (experiments/train.py:98), with the per-step
bookkeeping done by a handful of multi-tensor kernels instead of one device copy per parameter.

Why not DDP itself: on ONE rank (a 1-rank RCCL group, nothing to communicate) the DDP wrapper costs this model +0.51 ms per step
with float32 buckets and +0.98 ms with the bf16 compression hook, of an 11.2-ms step (``tools/probes/ddp_overhead_probe.py``):
143 device-to-device copies per step, one per parameter gradient into its bucket view (0.59 ms), plus the hook's casts.  That is
paid at every N > 1 before the first byte crosses a link.

Here: parameters are cut into buckets in reverse registration order (roughly the order their gradients appear); a
post-accumulate-grad hook counts a bucket's gradients in; when the last one has arrived the bucket is gathered into ONE flat
buffer by one multi-tensor copy, cast to the wire dtype (bf16 halves the bytes on xGMI; float32 selectable), and all-reduced
asynchronously (on the process group's own stream) while the backward pass goes on.  ``finish()`` (before the optimizer step) waits for the
buckets in launch order, scales by 1 / world and points every ``.grad`` at its slice of the bucket (no scatter pass).  Replicas start identical because every rank builds the model from the same seed; ``broadcast_parameters`` does it
explicitly when asked (parameters AND buffers, as DDP's constructor does).

Contract (what DDP enforces, enforced here too):
 * collectives are issued in BUCKET order on every rank -- a bucket whose gradients are all in waits for the buckets before it
   (gradient ARRIVAL order may differ between ranks whose graphs differ; launch order may not, or the ranks dead-lock);
 * one synchronised backward per ``finish()``: a second backward before ``finish()`` raises instead of silently dropping the
   accumulated gradient.  Gradient accumulation: run the first micro-batches under ``no_sync()`` (hooks count nothing, gradients
   accumulate in ``.grad`` as usual) and the last one outside it -- its buckets carry the accumulated sums;
 * BatchNorm running statistics are NOT synchronised per step (torch DDP's ``broadcast_buffers=True`` does that at every
   forward): per-rank statistics are the reference's behaviour (Lightning ddp_spawn without SyncBN keeps rank 0's at save time).
"""
import contextlib

import torch
import torch.distributed as dist


def _dense(p):
    """non-overlapping and dense (some permutation of a contiguous layout): its strides can be imposed on a flat slice"""
    if p.is_contiguous() or p.dim() != 4:
        return p.is_contiguous()
    return p.is_contiguous(memory_format=torch.channels_last)


class _Bucket:
    __slots__ = ('params', 'sizes', 'flat', 'wire', 'views', 'pending', 'work', 'event', 'ready')


class BucketedGradSync:
    def __init__(self, params, bucket_mb: float = 25.0, wire_dtype=torch.bfloat16, process_group=None, module=None):
        self.group = process_group
        self.module = module                             # optional: its BUFFERS are broadcast with the parameters
        self._sync = True                                # False inside no_sync()
        self._next = 0                                   # index of the next bucket to launch (bucket order = launch order)
        self.world = dist.get_world_size(process_group)
        params = [p for p in params if p.requires_grad]
        assert params, 'no trainable parameters'
        self.device = params[0].device
        nccl = self.device.type == 'cuda' and dist.get_backend(process_group) == 'nccl'
        self.wire_dtype = wire_dtype if nccl else torch.float32                          # (gloo: float32 on the wire)
        self.buckets, self._bucket_of = [], {}
        cap, cur, cur_bytes = int(bucket_mb * 2 ** 20), [], 0
        for p in reversed(params):                       # gradients appear roughly in reverse registration order
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= cap:
                self._add_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._add_bucket(cur)
        self._launched = []
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]

    def _add_bucket(self, plist):
        b = _Bucket()
        b.params = list(plist)
        b.sizes = [p.numel() for p in b.params]
        b.flat = torch.zeros(sum(b.sizes), dtype=torch.float32, device=self.device)
        b.wire = b.flat if self.wire_dtype == torch.float32 else torch.zeros(sum(b.sizes), dtype=self.wire_dtype, device=self.device)
        # views of the flat buffer with each PARAMETER's strides (channels-last convolution filters stay channels-last): gradients
        # come with their parameter's layout, and the multi-tensor copy only takes its one-launch path when source and
        # destination strides agree (otherwise: one copy kernel per tensor, 143 per step)
        b.views = [v.as_strided(p.shape, p.stride()) if _dense(p) else v.view(p.shape) for v, p in zip(b.flat.split(b.sizes), b.params)]
        b.pending, b.work, b.event, b.ready = len(b.params), None, None, False
        for p in b.params:
            self._bucket_of[p] = b
        self.buckets.append(b)

    def broadcast_parameters(self, src: int = 0):
        """Rank ``src``'s parameters -- and, when the module was given, its buffers (BatchNorm running statistics and counters,
        what DDP's constructor also synchronises) -- to every rank; the cached filter derivatives are declared stale."""
        with torch.no_grad():
            for b in self.buckets:
                for p in b.params:
                    dist.broadcast(p.data, src, group=self.group)
            if self.module is not None:
                for buf in self.module.buffers():
                    dist.broadcast(buf.data, src, group=self.group)
        from .nn_ops import invalidate_conv_caches
        invalidate_conv_caches(self.module)               # (`.data` writes bypass the caches' version counters)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context launch no collective and count nothing; gradients add up in
        ``.grad``.  The first backward OUTSIDE it reduces the accumulated sums (torch DDP's ``no_sync`` contract)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def begin(self):
        """Start of a training step: forget whatever an interrupted step (an exception between backward and finish) left behind."""
        if self._launched or any(b.pending != len(b.params) for b in self.buckets):
            for b in self._launched:
                if b.work is not None:
                    b.work.wait()
            for b in self.buckets:
                b.pending, b.work, b.ready = len(b.params), None, False
            self._launched = []
        self._next = 0

    # ---- backward-time half
    def _on_grad(self, p):
        if not self._sync:
            return
        b = self._bucket_of[p]
        b.pending -= 1
        if b.pending < 0:
            raise RuntimeError('BucketedGradSync: a parameter received a second gradient before finish() -- a second backward() '
                               'in one step.  For gradient accumulation run all but the last micro-batch under no_sync().')
        if b.pending == 0:
            b.ready = True
            self._launch_ready()

    def _launch_ready(self):
        """launch, in bucket order, every bucket up to the first one that is not complete yet"""
        while self._next < len(self.buckets) and self.buckets[self._next].ready:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b):
        grads = [p.grad for p in b.params]
        with torch.no_grad():
            # gather: one multi-tensor copy (strided sources are fine: each view has its parameter's shape)
            torch._foreach_copy_(b.views, grads)
            if b.wire is not b.flat:
                b.wire.copy_(b.flat)                     # float32 -> wire dtype, one kernel
        # issued from the current stream: the process group runs the collective on ITS OWN stream, which first waits for what the
        # current stream has queued so far (the gather and the cast above) -- the backward kernels that follow overlap it
        b.work = dist.all_reduce(b.wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._launched.append(b)

    # ---- before the optimizer step
    def finish(self):
        """Wait for every bucket's all-reduce and write the averaged gradients back into the parameters' ``.grad``."""
        for b in self.buckets:                           # a bucket none of whose gradients appeared is skipped; a bucket that is
            if 0 < b.pending < len(b.params):            # PARTLY filled means a parameter got no gradient this step
                missing = [i for i, p in enumerate(b.params) if p.grad is None]
                raise RuntimeError('BucketedGradSync: %d parameter(s) of a bucket received no gradient this step' % len(missing))
        while self._next < len(self.buckets):            # complete buckets held back behind one that got no gradient at all
            b = self.buckets[self._next]                 # (identical on every rank for identical graphs: the order stays fixed)
            if b.ready:
                self._launch(b)
            self._next += 1
        inv = 1.0 / self.world
        with torch.no_grad():
            for b in self._launched:
                b.work.wait()                            # (RCCL: makes the current stream wait for the collective's stream)
                if b.wire is not b.flat:
                    b.flat.copy_(b.wire)                 # wire dtype -> float32
                b.flat.mul_(inv)
                for p, v in zip(b.params, b.views):      # the averaged gradients ARE the bucket views now: no scatter pass
                    p.grad = v
                b.pending, b.work = len(b.params), None
        for b in self.buckets:
            b.pending, b.ready = len(b.params), False
        self._launched = []
        self._next = 0

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
