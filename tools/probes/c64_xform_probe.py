#!/usr/bin/env python
"""c64_xform_probe.py -- round 4: conv -> BatchNorm -> ReLU (-> dropout) -> conv with the normalised activation formed inside
the second convolution (salsa_nn_conv3x3_c64_xform_stats / _wrw_xform) against the unfused chain (salsa_nn_bn_train_fwd's apply
pass + salsa_nn_conv3x3_c64_stats / _wrw), at the stem's and stage 1's map sizes: results and times."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from salsa_amd import _lib  # noqa: E402

L = _lib.load()
dev = torch.device('cuda:0')
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (N, H, W, p) in ((32, 640, 200, 0.0), (32, 320, 100, 0.1), (3, 37, 45, 0.1)):
    g = torch.Generator(device='cuda').manual_seed(1)
    x1 = torch.randn(N, 64, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device=dev, generator=g) / 24).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, 64, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gamma, beta = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2
    M = N * H * W
    xf = x1.float().permute(0, 2, 3, 1).reshape(M, 64)
    mean = xf.mean(0).contiguous()
    invstd = torch.rsqrt(xf.var(0, unbiased=False) + 1e-5).contiguous()
    seed = 1234
    # unfused: bn apply (training kernel with given statistics = eval entry point has no dropout; use the training one)
    a = torch.empty_like(x1)
    save = torch.empty(2, 64, device=dev)
    ws = torch.empty(L.salsa_nn_bn_workspace_bytes(1, M, 64) // 8 + 1, dtype=torch.float64, device=dev)
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)

    def bn_apply():
        rc = L.salsa_nn_bn_train_fwd(P(x1), P(a), None, 1, M, 64, P(gamma), P(beta), 1e-5, 0.1, P(rm), P(rv), P(save[0]), P(save[1]), P(ws), 1,
                                     p, seed, None, None, 0, ST())
        assert rc == 0
    bn_apply()
    nb = L.salsa_nn_conv3x3_c64_stats_blocks(N, H, W)
    y_ref, y_new = torch.empty_like(x1), torch.empty_like(x1)
    part_ref, part_new = torch.empty(nb * 128, dtype=torch.float64, device=dev), torch.empty(nb * 128, dtype=torch.float64, device=dev)
    conv_ref = lambda: L.salsa_nn_conv3x3_c64_stats(P(a), P(w), P(y_ref), P(part_ref), N, H, W, ST())
    conv_new = lambda: L.salsa_nn_conv3x3_c64_xform_stats(P(x1), P(w), P(y_new), P(part_new), P(save[0]), P(save[1]), P(gamma), P(beta), p, seed, N, H, W, ST())
    assert conv_ref() == 0 and conv_new() == 0
    torch.cuda.synchronize()
    d = (y_new.float() - y_ref.float()).abs()
    print('%d x %d x %d p=%.1f  forward: max |diff| %.4f (|y| max %.2f), differing elements %.5f %%' %
          (N, H, W, p, float(d.max()), float(y_ref.float().abs().max()), 100 * float((d > 0).float().mean())))
    sr = part_ref.view(nb, 2, 64).sum(0)
    sn = part_new.view(nb, 2, 64).sum(0)
    print('   statistics: rel diff %.2e' % float(((sr - sn).abs() / (sr.abs() + 1e-9)).max()))
    gw_ref, gw_new = torch.zeros(64, 3, 3, 64, device=dev), torch.zeros(64, 3, 3, 64, device=dev)
    wrw_ref = lambda: L.salsa_nn_conv3x3_c64_wrw(P(a), P(dy), P(gw_ref), N, H, W, ST())
    wrw_new = lambda: L.salsa_nn_conv3x3_c64_wrw_xform(P(x1), P(dy), P(gw_new), P(save[0]), P(save[1]), P(gamma), P(beta), p, seed, N, H, W, ST())
    assert wrw_ref() == 0 and wrw_new() == 0
    torch.cuda.synchronize()
    print('   weight gradient: rel diff %.2e' % float((gw_ref - gw_new).abs().max() / gw_ref.abs().max()))
    if N * H * W > 100000:
        t_bn, t_cr, t_cn, t_wr, t_wn = timed(bn_apply), timed(conv_ref), timed(conv_new), timed(wrw_ref), timed(wrw_new)
        print('   ms: bn pass (finalize + apply) %.3f | conv+stats %.3f -> xform %.3f | wrw %.3f -> xform %.3f' % (t_bn, t_cr, t_cn, t_wr, t_wn))
