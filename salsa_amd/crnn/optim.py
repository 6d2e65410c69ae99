"""Adam for the SELD CRNN trainer as ONE launch per step (salsa_nn_adam_step, include/salsa_nn.h): torch.optim.Adam's update rule --
the reference's optimiser (experiments/configs/seld.yml:37-52, Lightning's configure_optimizers) -- over all parameter tensors at
once.  torch's own fused path needs three multi_tensor_apply launches for this network's 143 tensors, two of them a few dozen
workgroups of tiny BatchNorm / bias tensors: 0.165 ms per step for 0.39 GB of traffic; this one runs at the copy rate.

State layout and keys are torch.optim.Adam's (``step`` a float32 scalar tensor, ``exp_avg``, ``exp_avg_sq``), so a state dict moves
between the two.  CUDA float32 parameters only; no amsgrad / maximize / capturable (the trainer falls back to torch.optim.Adam on CPU
and when SALSA_HIP_ADAM=0).  The HIP library is required: there is no silent fallback inside this class."""
import ctypes as C

import torch

from .. import _lib

CHUNK = 8192           # include/salsa_nn.h: SALSA_NN_ADAM_CHUNK
MAX_TENSORS = 192      # SALSA_NN_ADAM_MAX_TENSORS


class _Entry(C.Structure):
    _fields_ = [('p', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_int64)]


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('HipAdam: invalid hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._plans = {}                       # per group index: (key, table tensor, chunk tensor, n_chunks) per batch of tensors

    def _plan(self, gi, ps, ms, vs):
        """device tables of one group's tensors (rebuilt only when a pointer or size changes: parameters and moments stay put)"""
        key = tuple((p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()) for p, m, v in zip(ps, ms, vs))
        plan = self._plans.get(gi)
        if plan is not None and plan[0] == key:
            return plan[1]
        dev = ps[0].device
        batches = []
        for b0 in range(0, len(ps), MAX_TENSORS):
            sl = slice(b0, b0 + MAX_TENSORS)
            ent = (_Entry * len(ps[sl]))(*[_Entry(p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()) for p, m, v in zip(ps[sl], ms[sl], vs[sl])])
            table = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).to(dev)
            chunks = [(t, c) for t, p in enumerate(ps[sl]) for c in range((p.numel() + CHUNK - 1) // CHUNK)]
            # big tensors first: the last workgroups to start are then the short ones
            chunks.sort(key=lambda tc: -min(CHUNK, ps[sl][tc[0]].numel() - tc[1] * CHUNK))
            cht = torch.tensor(chunks, dtype=torch.int32).to(dev)
            batches.append((table, cht, len(chunks), len(ps[sl])))
        self._plans[gi] = (key, batches)
        return batches

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's state loads as it is; every `step` becomes this class's form: an OWN float32 host scalar per parameter
        (torch's fused / capturable paths keep it on the device -- reading it would synchronise once per parameter per step -- and a
        state saved by the round-4 HipAdam shares one tensor object among a group's parameters)."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if 'step' in st:
                st['step'] = torch.tensor(float(st['step']), dtype=torch.float32)
        for g in self.param_groups:
            g.pop('step_tensor', None)
        self._plans = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.load()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and not p.grad.is_sparse):
                    raise RuntimeError('HipAdam: CUDA float32 dense parameters and gradients only')
                if not p.is_contiguous() and not (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last)):
                    raise RuntimeError('HipAdam: parameters must be dense in memory')
                st = self.state[p]
                if not st:
                    # torch.optim.Adam's layout: every parameter owns its `step` (a host float32 scalar).  (Round-4 advice: one
                    # tensor OBJECT shared by the group survived load_state_dict into torch's single-tensor / foreach paths, which
                    # then advanced it once per parameter; and a parameter whose first gradient came late inherited the group's
                    # count.)  143 host scalar increments per step cost nothing.
                    st['step'] = torch.zeros((), dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # the update is element-wise: any dense layout is fine as long as p, its gradient and its moments agree on it
            grads = [p.grad if p.grad.stride() == p.stride() else p.grad.contiguous(memory_format=torch.channels_last)
                     if (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last)) else p.grad.contiguous() for p in ps]
            for p, g in zip(ps, grads):
                if g.stride() != p.stride() or self.state[p]['exp_avg'].stride() != p.stride():
                    raise RuntimeError('HipAdam: gradient / moment layout differs from the parameter layout')
            beta1, beta2 = group['betas']
            stream = C.c_void_p(torch.cuda.current_stream(ps[0].device).cuda_stream)
            # one launch per DISTINCT step count (the bias correction depends on it): one in any normal run, more only when
            # parameters started receiving gradients at different times or a torch state dict with uneven counts was loaded
            by_step = {}
            for p, g in zip(ps, grads):
                by_step.setdefault(int(float(self.state[p]['step'])), []).append((p, g))
            for k, count in enumerate(sorted(by_step)):
                sub = by_step[count]
                sp = [p for p, _ in sub]
                sg = [g for _, g in sub]
                ms = [self.state[p]['exp_avg'] for p in sp]
                vs = [self.state[p]['exp_avg_sq'] for p in sp]
                b0 = 0
                with torch.cuda.device(sp[0].device):
                    for table, cht, n_chunks, nt in self._plan((gi, k), sp, ms, vs):
                        gp = (C.c_void_p * nt)(*[g.data_ptr() for g in sg[b0:b0 + nt]])
                        rc = L.salsa_nn_adam_step(C.c_void_p(table.data_ptr()), gp, nt, C.c_void_p(cht.data_ptr()), n_chunks,
                                                  float(group['lr']), float(beta1), float(beta2), float(group['eps']),
                                                  float(group['weight_decay']), count + 1, stream)
                        if rc:
                            raise RuntimeError('salsa_nn_adam_step failed (%d)' % rc)
                        b0 += nt
                for p in sp:
                    self.state[p]['step'] += 1      # (host scalars: no kernel)
        return loss
