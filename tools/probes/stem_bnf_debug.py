#!/usr/bin/env python
"""The slabs of salsa_nn_conv3x3_stem_wrw_bnf against float32 torch: G, Xh, S0, dbeta, dgamma and the combined dW on a small ragged shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from salsa_amd import _lib
from salsa_amd.crnn import nn_ops
dev = 'cuda:0'
L = _lib.load()
torch.manual_seed(0)
N, Cin, H, W = [int(a) for a in (sys.argv[1:5] or (2, 7, 40, 70))]
x = torch.randn((N, Cin, H, W), device=dev)
w = torch.randn(64, Cin, 3, 3, device=dev) * 0.2
x1 = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), padding=1).bfloat16().contiguous(memory_format=torch.channels_last)
g = torch.randn((N, 64, H, W), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
x1f = x1.float()
mean = x1f.mean((0, 2, 3)); var = x1f.var((0, 2, 3), unbiased=False); invstd = (var + 1e-5).rsqrt()
gamma = torch.rand(64, device=dev) + 0.5; beta = torch.randn(64, device=dev) * 0.3
xh = (x1f - mean[None, :, None, None]) * invstd[None, :, None, None]
live = (xh * gamma[None, :, None, None] + beta[None, :, None, None]) > 0
gm = g.float() * live
M = N * H * W
xb = x.bfloat16().float()
patches = F.unfold(xb, 3, padding=1).view(N, Cin * 9, H * W)            # [N][ci*9+tap][p]
G = torch.einsum('ncp,nkp->ck', gm.reshape(N, 64, -1), patches)
Xh = torch.einsum('ncp,nkp->ck', xh.bfloat16().float().reshape(N, 64, -1), patches)
S0 = patches.sum((0, 2))
dbeta = gm.sum((0, 2, 3)); dgamma = (gm * xh).sum((0, 2, 3))
a = gamma * invstd
dW_ref = (a[:, None] * (G - (dbeta / M)[:, None] * S0[None] - (dgamma / M)[:, None] * Xh)).view(64, Cin, 3, 3)
nbytes = L.salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes(N, H, W)
slabs = torch.zeros(nbytes // 4, device=dev)
gw = torch.zeros((64, Cin, 3, 3), device=dev); dwb = torch.zeros((2, 64), device=dev)
P = nn_ops._ptr
rc = L.salsa_nn_conv3x3_stem_wrw_bnf(P(x), x.stride(0), x.stride(1), P(g), P(x1), P(mean), P(invstd), P(gamma), P(beta), 1, P(gw), P(dwb[0]), P(dwb[1]),
                                     P(slabs), nbytes, N, Cin, H, W, nn_ops._stream(x))
assert rc == 0
torch.cuda.synchronize()
S = slabs.view(-1, 2 * 4096 + 192).double().sum(0)
Gk = S[:4096].view(64, 64); Xk = S[4096:8192].view(64, 64); S0k = S[8192:8256] + S[8256:8320]; dgk = S[8320:8384]
rel = lambda p, q: float((p - q).abs().max() / q.abs().max())
print('G   ', rel(Gk[:, :Cin * 9].float(), G), ' dbeta(col 63)', rel(Gk[:, 63].float(), dbeta))
print('Xh  ', rel(Xk[:, :Cin * 9].float(), Xh))
print('S0  ', rel(S0k[:Cin * 9].float(), S0))
print('dgam', rel(dgk.float(), dgamma), ' out dgamma', rel(dwb[0], dgamma), ' out dbeta', rel(dwb[1], dbeta))
print('dW  ', rel(gw, dW_ref), ' |dW| max', float(dW_ref.abs().max()))
err = ((S0k[:Cin * 9].float() - S0).abs() / S0.abs().max()).view(Cin, 9)
print('S0 error by (ci, tap):'); print((err * 1000).round().int())
perslab = slabs.view(-1, 2 * 4096 + 192)[:, 8192:8320]
tiles_w = (W + 31) // 32; tiles_h = (H + 3) // 4
# per-tile reference S0 for tile 1 (n=0, th=0, tw=1 if exists)
for t in range(min(3, perslab.shape[0])):
    tw, th, n = t % tiles_w, (t // tiles_w) % tiles_h, t // (tiles_w * tiles_h)
    pt = F.unfold(xb[n:n + 1], 3, padding=1).view(Cin * 9, H, W)[:, th * 4:th * 4 + 4, tw * 32:tw * 32 + 32].sum((1, 2))
    got = perslab[t, :64] + perslab[t, 64:]
    print('tile', t, (tw, th, n), 'max abs err', float((got[:Cin * 9] - pt).abs().max()), 'ref max', float(pt.abs().max()))
t = 0
tw, th, n = 0, 0, 0
full = F.unfold(xb[n:n + 1], 3, padding=1).view(Cin * 9, H, W)[:, 0:4, 0:32]
pt = full.sum((1, 2)); got = perslab[0, :64] + perslab[0, 64:]
torch.set_printoptions(precision=3, linewidth=200)
print('ref ', pt[:18]); print('got ', got[:18]); print('kh0 ', perslab[0, :18]); print('kh1 ', perslab[0, 64:82])
# candidates: per row sums, per (hw,kh) block sums
blk = full.view(Cin * 9, 4, 2, 2, 8).sum(4)      # [c][rr][hw][kh]
print('ref kh0', blk[:18, :, :, 0].sum((1, 2))); print('ref kh1', blk[:18, :, :, 1].sum((1, 2)))
print('ref hw1 only kh0', blk[:18, :, 1, 0].sum(1)); print('ref last row kh0', blk[:18, 3, :, 0].sum(1))
