#!/usr/bin/env python
"""BatchNorm backward alone, one launch against three (salsa_nn_set_bn_onepass), on the maps of stages 2 - 4 at batch 32:
back-to-back C-ABI calls between two events.  Needs tools/probes/bn_onepass.patch applied (git apply) and a rebuild: the one-launch
kernel was measured slower everywhere (profiles/r5_ab_notes.txt) and is not in the library."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
L = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for (N, Cn, H, W), relu, drop in (((2, 512, 4, 4), 0, 0.0), ((32, 512, 40, 12), 1, 0.1), ((32, 512, 40, 12), 2, 0.0), ((32, 512, 40, 12), 0, 0.0),
                                  ((32, 256, 80, 25), 1, 0.1), ((32, 256, 80, 25), 2, 0.0), ((32, 256, 80, 25), 0, 0.0),
                                  ((32, 128, 160, 50), 1, 0.1)):
    M = N * H * W
    mk = lambda: torch.randn((N, Cn, H, W), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    x, dy, dx, dres = mk(), mk(), mk(), mk()
    bits = torch.randint(0, 256, (M * Cn // 8,), dtype=torch.uint8, device=dev)
    ga, be, mu = torch.rand(Cn, device=dev) + 0.5, torch.randn(Cn, device=dev), torch.randn(Cn, device=dev) * 0.1
    isd = torch.rand(Cn, device=dev) + 0.5
    dwb = torch.empty((2, Cn), device=dev)
    ws = torch.empty(L.salsa_nn_bn_workspace_bytes(1, M, Cn) // 8 + 1, dtype=torch.float64, device=dev)
    coef = torch.empty(7 * Cn, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        rc = L.salsa_nn_bn_bwd(P(dy), P(bits) if relu == 2 else None, P(x), P(dx), P(dres) if relu == 2 else None, 1, M, Cn, P(ga), P(be), P(mu), P(isd),
                               relu, P(dwb[0]), P(dwb[1]), P(ws), P(coef), drop, 1234, st)
        assert rc == 0, rc
    out = []
    for on in (1, 0, 1, 0):
        L.salsa_nn_set_bn_onepass(on)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record(); torch.cuda.synchronize()
        out.append('%s %.1f us' % ('one' if on else 'three', e0.elapsed_time(e1) * 20))
    L.salsa_nn_set_bn_onepass(1)
    print('%s relu %d drop %.1f: %s' % ((N, Cn, H, W), relu, drop, ' | '.join(out)))
