"""GPU tests of the CRNN consumer (hand-written HIP layers behind include/salsa_nn.h / salsa_gru.h; torch supplies autograd,
the decoder GEMMs and the optimizer): forward against the reference-model golden, every kernel against float32 torch
operators incl. gradients, and bf16 training steps that must reduce the loss on a fixed batch."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_gpu_forward_matches_reference_model():
    from salsa_amd.crnn import SeldCRNN
    from salsa_amd.crnn.testing import seeded_fill
    meta, a = load_golden('g9_crnn')
    m = SeldCRNN()
    seeded_fill(m, meta['weight_seed'])
    m = m.cuda().eval()
    x = torch.randn(*meta['input_shape'], generator=torch.Generator().manual_seed(meta['input_seed'])).cuda()
    with torch.no_grad():
        out = m(x)
    np.testing.assert_allclose(out['event_frame_logit'].cpu().numpy(), a['event_frame_logit'], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(out['doa_frame_output'].cpu().numpy(), a['doa_frame_output'], rtol=2e-3, atol=2e-4)


def test_bf16_training_steps_reduce_loss():
    from salsa_amd.crnn.train import Trainer, synthetic_batch
    tr = Trainer('cuda:0', total_steps=100)
    x, sed, doa = synthetic_batch(4, 'cuda:0', seed=1)
    losses = [float(tr.train_step(x, sed, doa)[0]) for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    p, d = tr.infer(x)
    assert p.shape == (4, 80, 12) and d.shape == (4, 80, 36) and float(p.min()) >= 0 and float(p.max()) <= 1


def test_on_the_fly_features_feed_the_model():
    """config 4 plumbing: raw 8-s MIC audio -> SALSA on device -> CRNN step, no host round trip."""
    from salsa_amd.crnn.train import Trainer, synthetic_batch
    from salsa_amd.extractor import SalsaExtractor
    from salsa_amd.synth import synth_clip
    ys = np.stack([synth_clip(60 + i, 8 * 24000) for i in range(2)])
    feats = SalsaExtractor(audio_format='mic', fmax_doa=4000).extract(torch.from_numpy(ys).cuda())
    assert feats.shape == (2, 7, 641, 200)
    tr = Trainer('cuda:0', total_steps=10)
    _, sed, doa = synthetic_batch(2, 'cuda:0', seed=2)
    loss = tr.train_step(feats[:, :, :640], sed, doa)[0]
    assert np.isfinite(float(loss))


def test_fused_gru_scan_matches_torch_gru_forward_and_backward():
    """The hand-written scan (salsa_amd/csrc/gru_scan.hip) against torch.nn.GRU in float32: outputs and every gradient."""
    from salsa_amd.crnn.fused_gru import bigru_forward
    torch.manual_seed(0)
    for T, B in ((40, 5), (7, 2), (300, 3)):
        gru = torch.nn.GRU(512, 256, num_layers=2, batch_first=True, bidirectional=True, dropout=0.0).cuda()
        x = torch.randn(B, T, 512, device='cuda', requires_grad=True)
        ref, _ = gru(x)
        g = torch.randn_like(ref)
        ref.backward(g)
        ref_grads = [x.grad.clone()] + [p.grad.clone() for p in gru.parameters()]
        x.grad = None
        gru.zero_grad()
        out = bigru_forward(gru, x, training=False)
        out.backward(g)
        got_grads = [x.grad.clone()] + [p.grad.clone() for p in gru.parameters()]
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5), float((out - ref).abs().max())
        for a, b in zip(got_grads, ref_grads):
            assert torch.allclose(a, b, rtol=2e-3, atol=2e-4), float((a - b).abs().max())


def test_hip_avgpool_matches_torch_forward_and_backward():
    """salsa_nn_avgpool2x2 (channels-last, bf16 and float32, odd sizes) against F.avg_pool2d: values and gradients."""
    import torch.nn.functional as F
    from salsa_amd.crnn.nn_ops import _AvgPool2x2, avg_pool2x2
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(0)
    for dtype, (n, c, h, w) in ((torch.bfloat16, (3, 64, 40, 25)), (torch.float32, (2, 8, 7, 6)), (torch.bfloat16, (2, 128, 80, 13))):
        x = torch.randn((n, c, h, w), device=dev, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = avg_pool2x2(xa), F.avg_pool2d(xb, 2)
        assert isinstance(ya.grad_fn, _AvgPool2x2._backward_cls) and ya.shape == yb.shape
        assert torch.equal(ya, yb), dtype
        gy = torch.randn(yb.shape, device=dev, generator=g).to(dtype)
        ya.backward(gy)
        yb.backward(gy)
        assert torch.equal(xa.grad, xb.grad), dtype
    x = torch.randn(2, 7, 8, 8, device=dev)                                     # 7 channels: falls through to torch
    assert torch.equal(avg_pool2x2(x), F.avg_pool2d(x, 2))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_batchnorm_add_relu_matches_torch(dtype):
    """BatchNormAct2d (salsa_nn_bn_*) against nn.BatchNorm2d + add + ReLU: outputs, every gradient, running statistics,
    training and eval mode.  float32: 1e-5; bf16 activations: one bf16 ulp on outputs, bf16-level on gradients."""
    import torch.nn as nn
    import torch.nn.functional as F
    from salsa_amd.crnn.nn_ops import BatchNormAct2d, _BnAct
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)   # bf16: gradients are stored in bf16
    for (n, c, h, w), use_res, relu in (((4, 64, 20, 12), False, True), ((3, 128, 9, 7), True, True), ((2, 512, 5, 3), True, False),
                                        ((2, 64, 6, 4), False, False)):
        ref, fus = nn.BatchNorm2d(c).to(dev), BatchNormAct2d(c).to(dev)
        with torch.no_grad():
            ref.weight.copy_(torch.rand(c, device=dev, generator=g) + 0.5)
            ref.bias.copy_(torch.randn(c, device=dev, generator=g))
        fus.load_state_dict(ref.state_dict())
        for step in range(2):                                                    # two steps: running statistics accumulate
            x = (torch.randn((n, c, h, w), device=dev, generator=g) * 2 + 0.5).to(dtype).contiguous(memory_format=torch.channels_last)
            r = torch.randn((n, c, h, w), device=dev, generator=g).to(dtype).contiguous(memory_format=torch.channels_last) if use_res else None
            xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            ra, rb = (r.clone().requires_grad_(True), r.clone().requires_grad_(True)) if use_res else (None, None)
            # reference: torch's batch_norm + add + relu evaluated in float32 on the same (bf16-valued) inputs -- the fused
            # kernel keeps float32 until its single store, torch's bf16 chain rounds after the BN and again after the add
            xb = x.float().clone().requires_grad_(True)
            rb = r.float().clone().requires_grad_(True) if use_res else None
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
                ya = fus(xa, residual=ra, relu=relu)
            yb = ref(xb)
            yb = yb + rb if use_res else yb
            yb = F.relu(yb) if relu else yb
            assert isinstance(ya.grad_fn, _BnAct._backward_cls) and ya.dtype == dtype
            out_tol = tol if dtype == torch.float32 else dict(rtol=2.0 ** -8, atol=1e-3)   # one rounding to bf16
            torch.testing.assert_close(ya.float(), yb, **out_tol)
            gy = torch.randn(ya.shape, device=dev, generator=g).to(dtype)
            ya.backward(gy)
            yb.backward(gy.float())
            torch.testing.assert_close(xa.grad.float(), xb.grad.float(), **tol)
            torch.testing.assert_close(fus.weight.grad, ref.weight.grad, rtol=tol['rtol'], atol=tol['atol'] * (n * h * w) ** 0.5)
            torch.testing.assert_close(fus.bias.grad, ref.bias.grad, rtol=tol['rtol'], atol=tol['atol'] * (n * h * w) ** 0.5)
            if use_res:
                torch.testing.assert_close(ra.grad.float(), rb.grad.float(), **tol)
            torch.testing.assert_close(fus.running_mean, ref.running_mean, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(fus.running_var, ref.running_var, rtol=1e-4, atol=1e-5)
            assert int(fus.num_batches_tracked) == int(ref.num_batches_tracked) == step + 1
            fus.zero_grad(); ref.zero_grad()
        ref.eval(); fus.eval()
        with torch.no_grad():
            ya = fus(x, residual=r, relu=relu)
            yb = ref(x.float())
            yb = yb + r.float() if use_res else yb
            yb = F.relu(yb) if relu else yb
        torch.testing.assert_close(ya.float(), yb, **out_tol)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_batchnorm_relu_avgpool_matches_torch(dtype):
    """BatchNormAct2d.relu_pool (salsa_nn_bn_train_fwd_pool / salsa_nn_bn_bwd_pool, the stem's tail of
    models/model_utils.py:187-228) against nn.BatchNorm2d + ReLU + F.avg_pool2d in float32: outputs, every gradient, running
    statistics; even and odd extents (floor mode drops the last row / column, whose gradient is zero)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from salsa_amd.crnn.nn_ops import BatchNormAct2d, _BnReluPool
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(3)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    for n, c, h, w in ((4, 64, 40, 24), (3, 64, 9, 7), (2, 128, 5, 6), (1, 512, 2, 3)):
        ref, fus = nn.BatchNorm2d(c).to(dev), BatchNormAct2d(c).to(dev)
        with torch.no_grad():
            ref.weight.copy_(torch.rand(c, device=dev, generator=g) + 0.5)
            ref.bias.copy_(torch.randn(c, device=dev, generator=g))
        fus.load_state_dict(ref.state_dict())
        for step in range(2):
            x = (torch.randn((n, c, h, w), device=dev, generator=g) * 2 + 0.5).to(dtype).contiguous(memory_format=torch.channels_last)
            xa, xb = x.clone().requires_grad_(True), x.float().clone().requires_grad_(True)
            ya = fus.relu_pool(xa)
            yb = F.avg_pool2d(F.relu(ref(xb)), 2)
            assert isinstance(ya.grad_fn, _BnReluPool._backward_cls) and ya.dtype == dtype and ya.shape == yb.shape
            torch.testing.assert_close(ya.float(), yb, **(tol if dtype == torch.float32 else dict(rtol=2.0 ** -8, atol=1e-3)))
            gy = torch.randn(ya.shape, device=dev, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
            ya.backward(gy)
            yb.backward(gy.float())
            torch.testing.assert_close(xa.grad.float(), xb.grad, **tol)
            torch.testing.assert_close(fus.weight.grad, ref.weight.grad, rtol=tol['rtol'], atol=tol['atol'] * (n * h * w) ** 0.5)
            torch.testing.assert_close(fus.bias.grad, ref.bias.grad, rtol=tol['rtol'], atol=tol['atol'] * (n * h * w) ** 0.5)
            torch.testing.assert_close(fus.running_mean, ref.running_mean, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(fus.running_var, ref.running_var, rtol=1e-4, atol=1e-5)
            assert int(fus.num_batches_tracked) == int(ref.num_batches_tracked) == step + 1
            fus.zero_grad(); ref.zero_grad()
    fus.eval()                                                                   # eval mode: the unfused composite
    with torch.no_grad():
        torch.testing.assert_close(fus.relu_pool(x).float(), F.avg_pool2d(F.relu(ref.eval()(x.float())), 2),
                                   **(tol if dtype == torch.float32 else dict(rtol=2.0 ** -7, atol=2e-3)))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_batchnorm_relu_dropout(dtype):
    """The dropout fused behind BatchNorm + ReLU (salsa_nn_bn_train_fwd drop_p): every output is either dropped or the
    undropped output / (1 - p); the drop rate is p; the same seed gives the same mask; the backward -- which regenerates the
    mask from the seed -- matches autograd through batch_norm + relu + (that mask); eval mode drops nothing."""
    import torch.nn as nn
    import torch.nn.functional as F
    from salsa_amd.crnn.nn_ops import BatchNormAct2d
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(5)
    p = 0.1
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    for (n, c, h, w), use_res in (((8, 64, 40, 25), False), ((4, 128, 20, 13), True)):
        fus, ref = BatchNormAct2d(c).to(dev), nn.BatchNorm2d(c).to(dev)
        with torch.no_grad():
            fus.weight.copy_(torch.rand(c, device=dev, generator=g) + 0.5)
            fus.bias.copy_(torch.rand(c, device=dev, generator=g) + 0.5)          # most pre-activations positive
        ref.load_state_dict(fus.state_dict())
        x = torch.randn((n, c, h, w), device=dev, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
        r = torch.randn((n, c, h, w), device=dev, generator=g).to(dtype).contiguous(memory_format=torch.channels_last) if use_res else None
        xa = x.clone().requires_grad_(True)
        ra = r.clone().requires_grad_(True) if use_res else None
        torch.manual_seed(11)
        ya = fus(xa, residual=ra, relu=True, dropout_p=p)
        with torch.no_grad():
            stats = {k: v.clone() for k, v in fus.named_buffers()}                 # (weights are saved for the backward)
            plain = fus(x, residual=r, relu=True).float()                          # same batch statistics, no dropout
            torch.manual_seed(11)
            again = fus(x, residual=r, relu=True, dropout_p=p)
            for k, v in fus.named_buffers():
                v.copy_(stats[k])
        assert torch.equal(ya, again)                                              # the seed decides the mask
        scale = 65536.0 / (65536 - round(p * 65536))
        live = plain > 0
        kept = (ya != 0) & live
        torch.testing.assert_close(ya.float()[kept], (plain * scale)[kept], rtol=2.0 ** -7 if dtype == torch.bfloat16 else 1e-5, atol=1e-6)
        assert bool((ya[~live] == 0).all())
        rate = 1.0 - kept.sum().item() / live.sum().item()
        assert abs(rate - p) < 4 * (p * (1 - p) / live.sum().item()) ** 0.5 + 1e-4, rate
        # no structure along channels or rows: the per-channel and per-row drop rates stay near p as well
        n_live = live.sum((0, 2, 3)).float().clamp_min(1)
        per_c = 1.0 - kept.sum((0, 2, 3)).float() / n_live
        assert bool(((per_c - p).abs() < 5 * (p * (1 - p) / n_live).sqrt()).all())
        # backward against autograd with the observed mask as a constant
        mask = ((ya != 0) | ~live).float()
        xb = x.float().clone().requires_grad_(True)
        rb = r.float().clone().requires_grad_(True) if use_res else None
        yb = ref(xb)
        yb = F.relu(yb + rb if use_res else yb) * mask * scale
        gy = torch.randn(ya.shape, device=dev, generator=g).to(dtype)
        ya.backward(gy)
        yb.backward(gy.float())
        torch.testing.assert_close(xa.grad.float(), xb.grad, **tol)
        torch.testing.assert_close(fus.weight.grad, ref.weight.grad, rtol=tol['rtol'], atol=tol['atol'] * (n * h * w) ** 0.5)
        torch.testing.assert_close(fus.bias.grad, ref.bias.grad, rtol=tol['rtol'], atol=tol['atol'] * (n * h * w) ** 0.5)
        if use_res:
            torch.testing.assert_close(ra.grad.float(), rb.grad, **tol)
        fus.eval()
        with torch.no_grad():
            assert torch.equal(fus(x, residual=r, relu=True, dropout_p=p), fus(x, residual=r, relu=True))


def test_mfma_conv3x3_c64_matches_torch_forward_and_gradients():
    """salsa_nn_conv3x3_c64 (64 -> 64, bf16 in / f32 accumulate / bf16 out) against F.conv2d: output, data gradient (the same
    kernel with the flipped, transposed filter) and weight gradient (salsa_nn_conv3x3_c64_wrw, transposing LDS reads), incl.
    ragged tile edges."""
    import torch.nn.functional as F
    from salsa_amd.crnn.nn_ops import Conv3x3, _Conv3x3C64
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(2)
    conv = Conv3x3(64, 64, 3, padding=1, bias=False).to(dev)
    for (n, h, w) in ((2, 12, 40), (3, 9, 37), (1, 1, 1), (2, 70, 33)):
        x = torch.randn((n, 64, h, w), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        xa, xb = x.clone().requires_grad_(True), x.float().clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            ya = conv(xa)
        assert isinstance(ya.grad_fn, _Conv3x3C64._backward_cls) and ya.dtype == torch.bfloat16
        wref = conv.weight.detach().bfloat16().float().requires_grad_(True)        # the bf16-rounded filter, float32 math
        yb = F.conv2d(xb, wref, padding=1)
        torch.testing.assert_close(ya.float(), yb, rtol=2.0 ** -7, atol=2e-2)       # one bf16 rounding of a K=576 dot product
        gy = torch.randn(ya.shape, device=dev, generator=g).bfloat16()
        ya.backward(gy)
        yb.backward(gy.float())
        torch.testing.assert_close(xa.grad.float(), xb.grad, rtol=2.0 ** -7, atol=3e-2)
        assert conv.weight.grad.dtype == torch.float32                              # float32 sums of bf16 products: tight
        torch.testing.assert_close(conv.weight.grad, wref.grad, rtol=1e-4, atol=1e-4 * float(wref.grad.abs().max()))
        conv.zero_grad()
    y32 = conv(torch.randn(1, 64, 8, 8, device=dev))                               # float32 without autocast: torch path
    assert y32.dtype == torch.float32 and not isinstance(y32.grad_fn, _Conv3x3C64._backward_cls)


def test_mfma_conv_many_tiles_per_workgroup_and_determinism():
    """The persistent 64 -> 64 kernel rotates three LDS tiles with loads two tiles ahead: a shape that gives every workgroup ~20
    tiles (and one that gives most of them a single tile), against float32 convolution of the same bf16 operands, and bit-equal
    repeated launches (a race in the rotation would show as a sporadic difference)."""
    import torch.nn.functional as F
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(12)
    for n, h, w_ in ((16, 640, 100), (2, 40, 300), (1, 1, 1)):
        x = torch.randn((n, 64, h, w_), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((64, 64, 3, 3), device=dev, generator=g) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = nn_ops._conv64(x, w)
        for _ in range(3):
            assert torch.equal(nn_ops._conv64(x, w), y)
        ref = F.conv2d(x.float(), w.float(), padding=1)
        torch.testing.assert_close(y.float(), ref, rtol=2.0 ** -7, atol=2.0 ** -7 * float(ref.abs().max()))


def test_stem_conv_kernel_matches_torch():
    """salsa_nn_conv3x3_stem (7 -> 64 on float32 planar input): plain forward and weight gradient through Conv3x3 under bf16
    autocast against F.conv2d on the bf16-rounded operands in float32; the folded-BatchNorm + ReLU epilogue (eval) against
    conv -> BatchNorm -> ReLU; ragged sizes (tile edges, one-row / one-column images)."""
    import torch.nn.functional as F
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(9)
    for (n, cin, h, w) in ((2, 7, 40, 70), (1, 7, 9, 33), (3, 4, 8, 32), (1, 8, 1, 5), (1, 7, 17, 1)):
        conv = nn_ops.Conv3x3(cin, 64, kernel_size=3, stride=1, padding=1, bias=False).to(dev)
        bn = nn_ops.BatchNormAct2d(64).to(dev)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
            bn.running_mean.uniform_(-0.3, 0.3); bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn((n, cin, h + 1, w), device=dev, generator=g)[:, :, :h]     # a time crop: strided batches / channels
        assert not x.is_contiguous() or n * cin == 1
        xr, wr = x.to(torch.bfloat16).float(), conv.weight.detach().to(torch.bfloat16).float()
        ref = F.conv2d(xr, wr, padding=1)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            assert conv._stem_eligible(x)
            y = conv(x)
        assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
        assert isinstance(y.grad_fn, nn_ops._Conv3x3Stem._backward_cls)
        torch.testing.assert_close(y.float(), ref, rtol=2.0 ** -8, atol=2e-3)
        gy = torch.randn(y.shape, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y.backward(gy)
        wref = conv.weight.detach().clone().requires_grad_(True)
        F.conv2d(xr, wref.to(torch.bfloat16).float(), padding=1).backward(gy.float())
        wtol = 2e-3 if cin <= 7 else 2e-2          # own kernel (float32 accumulation of the same bf16 operands) | MIOpen
        torch.testing.assert_close(conv.weight.grad, wref.grad, rtol=wtol, atol=wtol * float(wref.grad.abs().max()))
        bn.eval(); conv.eval()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            fused = nn_ops.conv_bn_act(conv, bn, x)
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        wf = (conv.weight.detach() * scale[:, None, None, None]).to(torch.bfloat16).float()
        want = F.relu(F.conv2d(xr, wf, padding=1) + (bn.bias - bn.running_mean * scale)[None, :, None, None])
        torch.testing.assert_close(fused.float(), want, rtol=2.0 ** -8, atol=2e-3)


def test_whole_model_hip_layers_on_vs_off():
    """The CRNN with every hand-written layer (MFMA convolutions, pools, fused BatchNorm, GRU scan) against the same weights
    with those layers switched to torch / MIOpen: bf16-autocast eval forward and one training step's loss and gradients."""
    from salsa_amd.crnn import model as M, nn_ops
    from salsa_amd.crnn.loss import seld_loss
    from salsa_amd.crnn.train import synthetic_batch
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = M.SeldCRNN().to(dev).to(memory_format=torch.channels_last)
    # no dropout: torch's nn.GRU draws its inter-layer dropout from MIOpen's own generator, so the two paths could not see
    # the same masks
    with torch.no_grad():                                                           # zero_init_residual would switch the
        for mod in net.modules():                                                   # residual branches' gradients off
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.GRU):
            mod.dropout = 0.0
        if isinstance(mod, M.ResBlock):
            mod.dropout_p = 0.0
    real_dropout = M.F.dropout
    M.F.dropout = lambda t, p=0.5, training=True, inplace=False: t
    x, sed, doa = synthetic_batch(4, dev, seed=3)
    x = x.contiguous(memory_format=torch.channels_last)

    def run(on, amp=True):
        nn_ops.USE_HIP_POOL = nn_ops.USE_HIP_BN = nn_ops.USE_HIP_CONV = on
        M.FUSED_GRU = on
        net.eval()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            out_eval = [v.float() for _, v in sorted(net(x).items())]
        net.train()
        net.zero_grad()
        torch.manual_seed(1)                                                        # same dropout masks
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            out = net(x)
        loss = seld_loss(out, sed, doa)[0]
        loss.backward()
        g = {n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None}
        return out_eval, float(loss.detach()), g

    try:
        bufs = {n: b.clone() for n, b in net.named_buffers()}
        ev_on, loss_on, g_on = run(True)
        for n, b in net.named_buffers():
            b.copy_(bufs[n])
        ev_off, loss_off, g_off = run(False)
        for n, b in net.named_buffers():
            b.copy_(bufs[n])
        ev_ref, loss_ref, g_ref = run(False, amp=False)                             # float32 everywhere: the yardstick
    finally:
        nn_ops.USE_HIP_POOL = nn_ops.USE_HIP_BN = nn_ops.USE_HIP_CONV = True
        M.FUSED_GRU = True
        M.F.dropout = real_dropout
    print('loss hip / torch-bf16 / float32', loss_on, loss_off, loss_ref)
    for a, b in zip(ev_on, ev_off):
        torch.testing.assert_close(a, b, rtol=5e-2, atol=5e-2)                      # bf16 activations through 22 layers
    assert abs(loss_on - loss_ref) < 1e-2 * max(1.0, abs(loss_ref)) and abs(loss_off - loss_ref) < 1e-2 * max(1.0, abs(loss_ref))
    # gradients: at random initialisation EVERY bf16 implementation is tens of percent away from float32 per tensor
    # (and torch's own path is not reproducible run to run: atomics); ours must be no noisier than torch's
    e_on = sorted((g_on[n] - g_ref[n]).norm().item() / (g_ref[n].norm().item() + 1e-12) for n in g_ref)
    e_off = sorted((g_off[n] - g_ref[n]).norm().item() / (g_ref[n].norm().item() + 1e-12) for n in g_ref)
    mid = len(e_on) // 2
    print('gradient error vs float32, median / max: hip %.3f / %.3f   torch-bf16 %.3f / %.3f' % (e_on[mid], e_on[-1], e_off[mid], e_off[-1]))
    assert e_on[mid] <= 1.25 * e_off[mid] + 2e-2 and e_on[-1] <= 1.5 * e_off[-1] + 5e-2


def test_register_resident_gru_inference_scan():
    """salsa_gru_scan_fwd_regw (no-grad path, W_hh as float16 in registers) against torch.nn.GRU in float32 and against the
    float32 streaming scan: long sequences, both directions, two layers."""
    from salsa_amd.crnn import fused_gru
    dev = torch.device('cuda:0')
    torch.manual_seed(4)
    gru = torch.nn.GRU(512, 256, num_layers=2, batch_first=True, bidirectional=True, dropout=0.3).to(dev).eval()
    x = torch.randn(5, 300, 512, device=dev)
    with torch.no_grad():
        ref = gru(x)[0]
        fast = fused_gru.bigru_forward(gru, x, training=False, half_weights=True)     # what a bf16-autocast caller gets
        fp32 = fused_gru.bigru_forward(gru, x, training=False)                        # a float32 caller: float32 streaming scan
        fused_gru.REGISTER_WEIGHTS = False
        try:
            slow = fused_gru.bigru_forward(gru, x, training=False, half_weights=True)
        finally:
            fused_gru.REGISTER_WEIGHTS = True
    torch.testing.assert_close(slow, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(fast, ref, rtol=2e-3, atol=2e-3)                     # float16 weights: 2^-11 per weight
    assert not torch.equal(fast, slow)                                              # (the two kernels really are different)
    assert torch.equal(fp32, slow)            # pure-float32 evaluation never takes the float16-weight kernel


def test_eval_conv_with_folded_batchnorm_epilogue():
    """conv_bn_act in eval mode (BatchNorm folded into the MFMA convolution's filter and epilogue, residual add and ReLU before
    the single rounding) against the unfused float32 computation on the same bf16-valued inputs."""
    import torch.nn.functional as F
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(6)
    conv = nn_ops.Conv3x3(64, 64, 3, padding=1, bias=False).to(dev).eval()
    bn = nn_ops.BatchNormAct2d(64).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(64, device=dev, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(64, device=dev, generator=g))
        bn.running_mean.copy_(torch.randn(64, device=dev, generator=g))
        bn.running_var.copy_(torch.rand(64, device=dev, generator=g) + 0.5)
    for use_res, relu in ((False, True), (True, True), (True, False)):
        x = torch.randn((2, 64, 21, 45), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        r = torch.randn((2, 64, 21, 45), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last) if use_res else None
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            y = nn_ops.conv_bn_act(conv, bn, x, residual=r, relu=relu)
        assert y.dtype == torch.bfloat16
        ref = F.batch_norm(F.conv2d(x.float(), conv.weight.float(), padding=1), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                           False, 0.0, bn.eps)
        ref = ref + r.float() if use_res else ref
        ref = F.relu(ref) if relu else ref
        torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=8e-2)            # bf16 filter (scaled) and output rounding
    # pool=True: the stem's 2x2 average pool inside the kernel (even sizes; tile edges in both directions), and the unfused
    # fall-back for odd sizes
    for (h, w), use_res in (((22, 46), False), ((8, 64), True), ((4, 2), False), ((21, 45), False)):
        x = torch.randn((2, 64, h, w), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        r = torch.randn((2, 64, h, w), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last) if use_res else None
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            y = nn_ops.conv_bn_act(conv, bn, x, residual=r, relu=True, pool=True)
        assert y.shape == (2, 64, h // 2, w // 2) and y.dtype == torch.bfloat16
        ref = F.batch_norm(F.conv2d(x.float(), conv.weight.float(), padding=1), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                           False, 0.0, bn.eps)
        ref = F.avg_pool2d(F.relu(ref + r.float() if use_res else ref), 2)
        torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=8e-2)


def test_register_resident_gru_training_scan_gradients():
    """half_weights=True (what the model passes under bf16 autocast): forward + BPTT through the register-resident kernels,
    against torch.nn.GRU in float32 -- outputs and every gradient within the float16 rounding of W_hh."""
    from salsa_amd.crnn import fused_gru
    dev = torch.device('cuda:0')
    torch.manual_seed(7)
    gru = torch.nn.GRU(512, 256, num_layers=2, batch_first=True, bidirectional=True, dropout=0.0).to(dev).train()
    ref = torch.nn.GRU(512, 256, num_layers=2, batch_first=True, bidirectional=True, dropout=0.0).to(dev).train()
    ref.load_state_dict(gru.state_dict())
    x = torch.randn(6, 80, 512, device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = fused_gru.bigru_forward(gru, xa, training=True, half_weights=True)
    yb = ref(xb)[0]
    torch.testing.assert_close(ya, yb, rtol=2e-3, atol=2e-3)
    gy = torch.randn_like(yb)
    ya.backward(gy)
    yb.backward(gy)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-2, atol=2e-3)
    for (n, p), (_, q) in zip(gru.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-2, atol=1e-2 * float(q.grad.abs().max()), msg=n)


def test_wide_conv_kernel_matches_torch_forward_and_gradients():
    """salsa_nn_conv3x3_wide (flattened-pixel implicit GEMM, conv_wide.hip) against float32 F.conv2d on the bf16-rounded
    operands: every channel pairing of the residual stages, map widths 50 / 25 / 12, batches whose 512-pixel tiles straddle
    image boundaries and end ragged; forward, data gradient and the hand-written weight gradient (salsa_nn_conv3x3_wide_wrw)."""
    import torch.nn.functional as F
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(0)
    for n, cin, cout, h, w in ((3, 64, 128, 40, 50), (2, 128, 128, 37, 50), (5, 128, 256, 20, 25), (3, 256, 256, 23, 25),
                               (7, 256, 512, 40, 12), (5, 512, 512, 9, 12), (2, 128, 64, 16, 50), (1, 96, 192, 5, 7),
                               (3, 64, 128, 30, 33),    # weight gradient: the three-buffer kernel at a width without an instantiation
                               (5, 128, 128, 3, 40),    # ... and the two-buffer kernel (tiles spanning three images need more slots)
                               (1, 128, 128, 5, 7), (2, 128, 256, 9, 7)):   # fewer pixels than one 128-pixel tile / than two
        assert nn_ops._lib.load().salsa_nn_conv3x3_wide_supported(n, h, w, cin, cout)
        x = torch.randn((n, cin, h, w), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn((cout, cin, 3, 3), device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5)
        xa = x.clone().requires_grad_(True)
        wa = wt.clone().requires_grad_(True)
        y = nn_ops._Conv3x3Wide.apply(xa, wa)
        xr = x.float().requires_grad_(True)
        wr = wt.to(torch.bfloat16).float().requires_grad_(True)
        ref = F.conv2d(xr, wr, padding=1)
        scale = float(ref.abs().max())
        assert float((y.float() - ref).abs().max()) < 1e-2 * scale, (cin, cout, h, w)      # bf16 output rounding: 2^-9 relative
        gy = torch.randn(ref.shape, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y.backward(gy)
        ref.backward(gy.float())
        assert float((xa.grad.float() - xr.grad).abs().max()) < 1e-2 * float(xr.grad.abs().max()), (cin, cout, h, w)
        assert float((wa.grad - wr.grad).abs().max()) < 3e-2 * float(wr.grad.abs().max()), (cin, cout, h, w)
    assert not nn_ops._lib.load().salsa_nn_conv3x3_wide_supported(2, 8, 8, 48, 64)          # Cin not a multiple of 32
    assert not nn_ops._lib.load().salsa_nn_conv3x3_wide_supported(2, 40, 200, 128, 128)     # 200-pixel rows: chunk exceeds its LDS buffer


def test_eval_wide_conv_with_folded_batchnorm_epilogue():
    """conv_bn_act in eval mode for the wide layers (conv_wide.hip's epilogue: folded BatchNorm shift, residual add, ReLU before
    the single rounding) against the unfused float32 computation on the same bf16-valued inputs."""
    import torch.nn.functional as F
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(8)
    for cin, cout, h, w, n in ((64, 128, 20, 50, 3), (128, 128, 20, 50, 2), (256, 512, 40, 12, 3), (512, 512, 10, 12, 5)):
        conv = nn_ops.Conv3x3(cin, cout, 3, padding=1, bias=False).to(dev).eval()
        bn = nn_ops.BatchNormAct2d(cout).to(dev).eval()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(cout, device=dev, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(cout, device=dev, generator=g))
            bn.running_mean.copy_(torch.randn(cout, device=dev, generator=g) * 0.3)
            bn.running_var.copy_(torch.rand(cout, device=dev, generator=g) + 0.5)
        for use_res, relu in ((False, True), (True, True), (True, False)):
            x = torch.randn((n, cin, h, w), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
            r = torch.randn((n, cout, h, w), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last) if use_res else None
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
                y = nn_ops.conv_bn_act(conv, bn, x, residual=r, relu=relu)
            assert y.dtype == torch.bfloat16 and y.shape == (n, cout, h, w)
            ref = F.batch_norm(F.conv2d(x.float(), conv.weight.float(), padding=1), bn.running_mean, bn.running_var, bn.weight,
                               bn.bias, False, 0.0, bn.eps)
            ref = ref + r.float() if use_res else ref
            ref = F.relu(ref) if relu else ref
            torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=8e-2)        # bf16 filter (scaled) and output rounding


def test_conv_filter_bank_matches_torch_casts_and_tracks_weight_updates():
    """ConvFilterBank (salsa_nn_conv_filter_bank: one launch for all layers) against what it replaces -- torch's per-layer
    ``weight.to(bf16).contiguous(channels_last)`` and ``.flip(2, 3).transpose(0, 1).contiguous(channels_last)`` -- bit for
    bit, for contiguous and channels-last master weights; an in-place weight update is picked up at the next request."""
    from salsa_amd.crnn.nn_ops import Conv1x1, Conv3x3, ConvFilterBank
    dev = torch.device('cuda:0')
    torch.manual_seed(4)
    convs = [Conv3x3(ci, co, 3, padding=1, bias=False).to(dev) for ci, co in ((64, 64), (64, 128), (128, 96), (256, 512))]
    convs.append(Conv1x1(128, 256, 1, bias=False).to(dev))                                      # a shortcut's 1x1 filter
    convs[2].weight.data = convs[2].weight.data.contiguous(memory_format=torch.channels_last)   # another stride pattern
    extra = Conv3x3(7, 64, 3, padding=1, bias=False).to(dev)                                     # not bankable (Cin % 32)
    bank = ConvFilterBank(convs + [extra])
    assert len(bank.convs) == 5 and not hasattr(extra, '_bank')

    def check():
        for i, c in enumerate(convs):
            f, b = bank.filters(i)
            wb = c.weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            assert f.shape == wb.shape and f.is_contiguous(memory_format=torch.channels_last) and torch.equal(f, wb)
            wt = wb.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
            assert b.shape == wt.shape and b.is_contiguous(memory_format=torch.channels_last) and torch.equal(b, wt)

    check()
    launches = bank._versions[:]
    bank.filters(3)
    assert bank._versions == launches                                     # nothing changed: no refresh
    with torch.no_grad():
        for c in convs:
            c.weight.mul_(1.5).add_(0.01)                                   # what an optimizer step does: in place
    check()
    convs[1].weight.data = torch.randn_like(convs[1].weight)              # re-allocated parameter storage
    check()
    # a FUSED optimizer updates the parameters in place WITHOUT bumping their version counters (torch 2.10): the bank must
    # notice through the global optimizer-step hook -- left stale, the convolutions would train on their initial weights
    opt = torch.optim.Adam([c.weight for c in convs], lr=0.05, fused=True)
    for c in convs:
        c.weight.grad = torch.randn_like(c.weight)
    before = [c.weight.detach().clone() for c in convs]
    opt.step()
    assert all(not torch.equal(b, c.weight) for b, c in zip(before, convs))
    check()


def test_folded_filter_cache_follows_parameter_and_statistics_updates():
    """conv_bn_act's eval path caches the folded (filter, shift) pair per layer; the cache must miss after an in-place weight
    update AND after a training-mode forward (whose fused kernels update the running statistics through raw pointers)."""
    import torch.nn.functional as F
    from salsa_amd.crnn.nn_ops import BatchNormAct2d, Conv3x3, conv_bn_act
    dev = torch.device('cuda:0')
    torch.manual_seed(9)
    conv, bn = Conv3x3(64, 64, 3, padding=1, bias=False).to(dev), BatchNormAct2d(64).to(dev)
    x = torch.randn(2, 64, 24, 20, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def fused_eval():
        conv.eval(); bn.eval()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            return conv_bn_act(conv, bn, x).float()

    def reference():
        y = F.conv2d(x.float(), conv.weight.float(), padding=1)
        return F.relu(F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))

    tol = dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(fused_eval(), reference(), **tol)
    assert conv._fold_cache is not None
    first = conv._fold_cache[1]
    fused_eval()
    assert conv._fold_cache[1] is first                                      # unchanged parameters: cache hit
    conv.train(); bn.train()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        for _ in range(3):
            conv_bn_act(conv, bn, x * 3 + 1)                                 # moves the running statistics a long way
    torch.testing.assert_close(fused_eval(), reference(), **tol)
    assert conv._fold_cache[1] is not first
    with torch.no_grad():
        conv.weight.mul_(0.5)
    torch.testing.assert_close(fused_eval(), reference(), **tol)
    opt = torch.optim.Adam(list(conv.parameters()) + list(bn.parameters()), lr=0.2, fused=True)   # no version bumps (see the bank test)
    for p_ in list(conv.parameters()) + list(bn.parameters()):
        p_.grad = torch.randn_like(p_)
    opt.step()
    torch.testing.assert_close(fused_eval(), reference(), **tol)


def test_batchnorm_statistics_survive_a_large_mean():
    """Batch variance as E[x^2] - mean^2: with |mean| = 300 sigma, float32 sums of x^2 leave ~2 digits; the kernel
    accumulates across row batches (and keeps its per-block partials) in float64 and must match float64 statistics of the
    same input: the running variance to 1e-4, the normalised output to 1e-3 (x - mean itself carries 300 * 2^-24 of input
    rounding)."""
    from salsa_amd.crnn.nn_ops import BatchNormAct2d
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(12)
    x = (torch.randn((8, 64, 64, 50), device=dev, generator=g) + 300.0).contiguous(memory_format=torch.channels_last)
    fus = BatchNormAct2d(64).to(dev)
    ya = fus(x)
    x64 = x.double()
    mean64, var64 = x64.mean(dim=(0, 2, 3)), x64.var(dim=(0, 2, 3), unbiased=False)
    n = x.numel() // 64
    torch.testing.assert_close(fus.running_var.double(), 0.9 + 0.1 * var64 * n / (n - 1), rtol=1e-4, atol=0)
    torch.testing.assert_close(fus.running_mean.double(), 0.1 * mean64, rtol=1e-6, atol=0)
    y64 = (x64 - mean64[None, :, None, None]) / torch.sqrt(var64 + fus.eps)[None, :, None, None]
    torch.testing.assert_close(ya.double(), y64, rtol=1e-3, atol=1e-3)


def test_conv1x1_kernels_match_torch_forward_and_gradients():
    """Conv1x1 (salsa_amd/csrc/conv_1x1.hip: the residual shortcuts' forward, data gradient and weight gradient) against
    float32 F.conv2d on the same bf16-valued inputs; pixel counts that are not multiples of the 128- / 64-pixel tiles; with
    and without the model's filter bank."""
    import torch.nn.functional as F
    from salsa_amd.crnn.nn_ops import Conv1x1, ConvFilterBank, _Conv1x1
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(21)
    for (cin, cout), (n, h, w), banked in (((64, 128), (3, 20, 13), False), ((128, 256), (2, 9, 7), True), ((256, 512), (4, 40, 12), True),
                                           ((64, 128), (1, 1, 1), False)):
        conv = Conv1x1(cin, cout, kernel_size=1, bias=False).to(dev)
        if banked:
            ConvFilterBank([conv])
        x = torch.randn((n, cin, h, w), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        xa, xb = x.clone().requires_grad_(True), x.float().clone().requires_grad_(True)
        wb = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            ya = conv(xa)
        assert isinstance(ya.grad_fn, _Conv1x1._backward_cls) and ya.dtype == torch.bfloat16
        yb = F.conv2d(xb, wb)
        torch.testing.assert_close(ya.float(), yb, rtol=2.0 ** -7, atol=2e-3 * float(yb.abs().max()))
        gy = torch.randn(ya.shape, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ya.backward(gy)
        yb.backward(gy.float())
        torch.testing.assert_close(xa.grad.float(), xb.grad, rtol=2.0 ** -7, atol=2e-3 * float(xb.grad.abs().max()))
        assert conv.weight.grad.dtype == torch.float32
        torch.testing.assert_close(conv.weight.grad, wb.grad, rtol=1e-3, atol=1e-3 * float(wb.grad.abs().max()))


def test_training_trajectory_with_cached_filters_matches_uncached():
    """End-to-end guard for everything that caches a function of the parameters (ConvFilterBank, the 1x1 kernels' filters):
    30 fused-Adam steps on a fixed batch with the caches on must follow the same loss curve as with them off.  (A bank that
    misses the fused optimizer's updates -- they do not bump version counters -- still passes every per-kernel test while the
    convolutions silently keep their initial weights; the loss at step 30 then lags by 12 %.)"""
    from salsa_amd.crnn import nn_ops
    from salsa_amd.crnn.train import Trainer, synthetic_batch
    x, sed, doa = synthetic_batch(8, 'cuda:0', seed=1)
    losses = {}
    saved = nn_ops.USE_FILTER_BANK
    try:
        for bank in (True, False):
            nn_ops.USE_FILTER_BANK = bank
            torch.manual_seed(0)
            tr = Trainer('cuda:0')
            for _ in range(30):
                loss = tr.train_step(x, sed, doa)[0]
            losses[bank] = float(loss)
    finally:
        nn_ops.USE_FILTER_BANK = saved
    assert losses[True] < 1.08 and abs(losses[True] - losses[False]) < 0.03 * losses[False], losses


def test_conv_epilogue_statistics_feed_the_batchnorm():
    """salsa_nn_conv3x3_c64_stats: the training forward of a 64 -> 64 layer leaves the per-channel sum / sum of squares of its
    bf16-rounded output as per-workgroup partial rows; summed they must equal the sums of the stored tensor (ragged sizes:
    tiles that hang over the image must not contribute), and conv -> BatchNorm(+ReLU)(+pool) fed from them must match the same
    layers with BatchNorm's own statistics pass: outputs, running statistics, every gradient."""
    from salsa_amd import _lib
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(31)
    for n, h, w in ((2, 40, 70), (1, 9, 33), (3, 17, 5), (4, 320, 100)):
        x = torch.randn((n, 64, h, w), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wb = (torch.randn((64, 64, 3, 3), device=dev, generator=g) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        nb = _lib.load().salsa_nn_conv3x3_c64_stats_blocks(n, h, w)
        part = torch.full((nb * 128,), float('nan'), dtype=torch.float64, device=dev)
        y = nn_ops._conv64(x, wb, stats_part=part)
        assert torch.equal(y, nn_ops._conv64(x, wb))                       # the same output as the plain kernel
        tot = part.view(nb, 2, 64).sum(0)
        yf = y.double()
        torch.testing.assert_close(tot[0], yf.sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(tot[1], (yf * yf).sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3)
    saved = nn_ops.USE_CONV_STATS
    try:
        for pool in (False, True):
            res = {}
            for use in (True, False):
                nn_ops.USE_CONV_STATS = use
                torch.manual_seed(3)
                conv, bn = nn_ops.Conv3x3(64, 64, 3, padding=1, bias=False).to(dev), nn_ops.BatchNormAct2d(64).to(dev)
                xx = torch.randn((3, 64, 24, 36), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    out = nn_ops.conv_bn_act(conv, bn, xx, pool=pool)
                out.float().square().mean().backward()
                res[use] = (out.detach().float(), bn.running_mean.clone(), bn.running_var.clone(), xx.grad.float(),
                            conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
            for a, b in zip(res[True], res[False]):
                torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()) + 1e-6)
    finally:
        nn_ops.USE_CONV_STATS = saved


@pytest.mark.parametrize('reduce_fused', [True, False])
def test_first_layer_backward_without_the_batchnorm_apply_pass(reduce_fused):
    """_StemConvBnRelu: relu(bn(conv7->64(x))) as one autograd node whose backward hands (g, conv output, BatchNorm
    coefficients) to the weight-gradient kernel instead of materialising the BatchNorm's input gradient
    (salsa_nn_bn_bwd with dx = NULL + salsa_nn_conv3x3_stem_wrw_bn) -- or, reduce_fused (the default), folds the BatchNorm
    backward's reduction into that pass as well (salsa_nn_conv3x3_stem_wrw_bnf: dW = a (G - b S0 - k' Xh) from one read of g and
    x1).  Must match the two-node path (conv -> BatchNormAct2d): output, running statistics and the gradients of the filter, gamma
    and beta; ragged sizes; the reduce-fused backward twice gives the same bits."""
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    saved, saved_r = nn_ops.USE_STEM_FUSED_BWD, nn_ops.USE_STEM_BN_REDUCE_FUSED
    nn_ops.USE_STEM_BN_REDUCE_FUSED = reduce_fused
    try:
        for n, cin, h, w in ((2, 7, 40, 70), (1, 7, 9, 33), (3, 4, 17, 5), (4, 7, 64, 200)):
            res = {}
            for fused in (True, False):
                nn_ops.USE_STEM_FUSED_BWD = fused
                torch.manual_seed(5)
                conv, bn = nn_ops.Conv3x3(cin, 64, 3, padding=1, bias=False).to(dev), nn_ops.BatchNormAct2d(64).to(dev)
                with torch.no_grad():
                    bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
                x = torch.randn((n, cin, h, w), device=dev)
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    out = nn_ops.conv_bn_act(conv, bn, x)
                assert isinstance(out.grad_fn, nn_ops._StemConvBnRelu._backward_cls) == fused
                gy = torch.randn(out.shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                out.backward(gy)
                res[fused] = (out.detach().float(), bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked),
                              conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
            assert res[True][3] == res[False][3] == 1
            for a, b in zip(res[True][:3] + res[True][4:], res[False][:3] + res[False][4:]):
                torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()) + 1e-6)
            if reduce_fused:                                                     # bit-reproducible (slabs, fixed order)
                nn_ops.USE_STEM_FUSED_BWD = True
                conv.weight.grad = bn.weight.grad = bn.bias.grad = None
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    out = nn_ops.conv_bn_act(conv, bn, x)                          # (the two-node modules of the last pass: same weights)
                out.backward(gy)
                first = [t.grad.clone() for t in (conv.weight, bn.weight, bn.bias)]
                conv.weight.grad = bn.weight.grad = bn.bias.grad = None
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    out = nn_ops.conv_bn_act(conv, bn, x)
                out.backward(gy)
                assert all(torch.equal(a, t.grad) for a, t in zip(first, (conv.weight, bn.weight, bn.bias)))
    finally:
        nn_ops.USE_STEM_FUSED_BWD, nn_ops.USE_STEM_BN_REDUCE_FUSED = saved, saved_r


def test_first_layer_reduce_fused_backward_at_bench_size_with_offset_inputs():
    """Round-5 advisor: the reduce-fused first-layer backward forms dW = a (G - b S0 - k' Xh) from float32 sums over ~N H W / 512
    pixels per workgroup; with inputs far from zero mean (un-normalised dB spectrograms: -60 +- 12) G and b S0 are large and nearly
    cancel.  At the bench size (32 x 7 x 640 x 200) both backward variants -- reduction fused (MODE 2) and separate (MODE 1) --
    are held to a float64 evaluation of the same layer (bf16-rounded inputs and filter, exact BatchNorm algebra): the fused
    variant may not be further from it than the separate one's error (x2) or 2e-3 of max |dW|, for normalised AND raw-dB inputs."""
    import torch.nn.functional as F
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    saved, saved_r = nn_ops.USE_STEM_FUSED_BWD, nn_ops.USE_STEM_BN_REDUCE_FUSED
    n, cin, h, w = 32, 7, 640, 200
    g = torch.Generator(device=dev).manual_seed(23)
    try:
        for label, mean, std in (('normalised', 0.3, 1.0), ('raw dB', -60.0, 12.0)):
            x = torch.empty((n, cin, h, w), device=dev)
            x[:, :4] = torch.randn((n, 4, h, w), device=dev, generator=g) * std + mean
            x[:, 4:] = (torch.rand((n, 3, h, w), device=dev, generator=g) * 2 - 1) * (torch.rand((n, 3, h, w), device=dev, generator=g) < 0.25)
            gy = None
            grads = {}
            for fused_reduce in (True, False):
                nn_ops.USE_STEM_FUSED_BWD, nn_ops.USE_STEM_BN_REDUCE_FUSED = True, fused_reduce
                torch.manual_seed(5)
                conv, bn = nn_ops.Conv3x3(cin, 64, 3, padding=1, bias=False).to(dev), nn_ops.BatchNormAct2d(64).to(dev)
                with torch.no_grad():
                    bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    out = nn_ops.conv_bn_act(conv, bn, x)
                if gy is None:
                    gy = torch.randn(out.shape, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                out.backward(gy)
                grads[fused_reduce] = (conv.weight.grad.double().clone(), bn.weight.grad.double().clone(), bn.bias.grad.double().clone())
                wq, gam, bet, eps = conv.weight.detach().bfloat16().double(), bn.weight.detach().double(), bn.bias.detach().double(), bn.eps
                del out
            # float64 reference, clip by clip (unfold: 63 x 128000 patches per clip)
            xq = x.bfloat16()
            s1 = torch.zeros(64, dtype=torch.float64, device=dev)
            s2 = torch.zeros_like(s1)
            for i in range(n):
                z = F.conv2d(xq[i:i + 1].float(), wq.float(), padding=1).double()        # float32 conv of bf16 values: products exact, sums ~1e-7
                s1 += z.sum(dim=(0, 2, 3)); s2 += (z * z).sum(dim=(0, 2, 3))
            cnt = n * h * w
            mu = s1 / cnt
            var = s2 / cnt - mu * mu
            rstd = (var + eps).rsqrt()
            dgam = torch.zeros_like(s1); dbet = torch.zeros_like(s1)
            for i in range(n):
                z = F.conv2d(xq[i:i + 1].float(), wq.float(), padding=1).double()
                xh = (z - mu[None, :, None, None]) * rstd[None, :, None, None]
                dz = gy[i:i + 1].double() * ((gam[None, :, None, None] * xh + bet[None, :, None, None]) > 0)
                dgam += (dz * xh).sum(dim=(0, 2, 3)); dbet += dz.sum(dim=(0, 2, 3))
            dW = torch.zeros((64, cin * 9), dtype=torch.float64, device=dev)
            for i in range(n):
                z = F.conv2d(xq[i:i + 1].float(), wq.float(), padding=1).double()
                xh = (z - mu[None, :, None, None]) * rstd[None, :, None, None]
                dz = gy[i:i + 1].double() * ((gam[None, :, None, None] * xh + bet[None, :, None, None]) > 0)
                dzo = (gam * rstd)[None, :, None, None] * (dz - dbet[None, :, None, None] / cnt - xh * dgam[None, :, None, None] / cnt)
                patches = F.unfold(xq[i:i + 1].double(), 3, padding=1)                    # (1, cin * 9, h * w)
                dW += dzo.reshape(64, h * w) @ patches[0].T
            dW = dW.reshape(64, cin, 3, 3)
            scale = float(dW.abs().max())
            err = {k: float((v[0] - dW).abs().max()) / scale for k, v in grads.items()}
            eg = {k: float((v[1] - dgam).abs().max()) / float(dgam.abs().max()) for k, v in grads.items()}
            print('%s inputs: max |dW - float64| / max |dW|: reduce-fused %.3g, separate %.3g; dgamma: %.3g, %.3g'
                  % (label, err[True], err[False], eg[True], eg[False]))
            assert err[True] <= max(2.0 * err[False], 2e-3), (label, err)
            assert eg[True] <= max(2.0 * eg[False], 2e-3), (label, eg)
    finally:
        nn_ops.USE_STEM_FUSED_BWD, nn_ops.USE_STEM_BN_REDUCE_FUSED = saved, saved_r


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_batchnorm_residual_relu_avgpool_matches_torch(dtype):
    """BatchNormAct2d.relu_pool with a residual (the tail of a residual block whose successor starts with the stride-2 pool):
    avg_pool2x2(relu(bn(x) + r)) against nn.BatchNorm2d + add + ReLU + F.avg_pool2d in float32 -- output, the gradients of x,
    r, gamma and beta, running statistics; even and odd extents (80 x 25 -> 40 x 12 drops a column)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from salsa_amd.crnn.nn_ops import BatchNormAct2d, _BnReluPool
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(17)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    for n, c, h, w in ((2, 256, 16, 25), (3, 64, 9, 7), (2, 128, 6, 10)):
        ref, fus = nn.BatchNorm2d(c).to(dev), BatchNormAct2d(c).to(dev)
        with torch.no_grad():
            ref.weight.copy_(torch.rand(c, device=dev, generator=g) + 0.5)
            ref.bias.copy_(torch.randn(c, device=dev, generator=g))
        fus.load_state_dict(ref.state_dict())
        x = (torch.randn((n, c, h, w), device=dev, generator=g) * 2 + 0.5).to(dtype).contiguous(memory_format=torch.channels_last)
        r = torch.randn((n, c, h, w), device=dev, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
        xa, ra = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
        xb, rb = x.float().clone().requires_grad_(True), r.float().clone().requires_grad_(True)
        ya = fus.relu_pool(xa, None, ra)
        yb = F.avg_pool2d(F.relu(ref(xb) + rb), 2)
        assert isinstance(ya.grad_fn, _BnReluPool._backward_cls) and ya.shape == yb.shape
        torch.testing.assert_close(ya.float(), yb, **(tol if dtype == torch.float32 else dict(rtol=2.0 ** -8, atol=2e-3)))
        gy = torch.randn(ya.shape, device=dev, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
        ya.backward(gy)
        yb.backward(gy.float())
        torch.testing.assert_close(xa.grad.float(), xb.grad, **tol)
        torch.testing.assert_close(ra.grad.float(), rb.grad, **tol)
        torch.testing.assert_close(fus.weight.grad, ref.weight.grad, rtol=tol['rtol'], atol=tol['atol'] * (n * h * w) ** 0.5)
        torch.testing.assert_close(fus.bias.grad, ref.bias.grad, rtol=tol['rtol'], atol=tol['atol'] * (n * h * w) ** 0.5)
        torch.testing.assert_close(fus.running_var, ref.running_var, rtol=1e-4, atol=1e-5)


def test_first_layer_statistics_epilogue_matches_the_stored_tensor():
    """salsa_nn_conv3x3_stem_stats (persistent first-layer forward that also leaves the BatchNorm's partial sums): same output
    as the plain launch, and the partial rows add up to the per-channel sum / sum of squares of the stored bf16 tensor --
    ragged sizes, a strided (time-cropped) input, more tiles than workgroups."""
    from salsa_amd import _lib
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    L = _lib.load()
    g = torch.Generator(device=dev).manual_seed(41)
    for n, cin, h, w in ((2, 7, 40, 70), (1, 7, 9, 33), (3, 4, 17, 5), (16, 7, 640, 200)):
        x = torch.randn((n, cin, h + 1, w), device=dev, generator=g)[:, :, :h]
        wq = nn_ops._stem_filter(torch.randn((64, cin, 3, 3), device=dev, generator=g) * 0.2)
        ref = nn_ops._conv_stem(x, wq)
        nb = L.salsa_nn_conv3x3_stem_stats_blocks(n, h, w)
        part = torch.full((nb * 128,), float('nan'), dtype=torch.float64, device=dev)
        y = torch.empty_like(ref)
        rc = L.salsa_nn_conv3x3_stem_stats(nn_ops._ptr(x), x.stride(0), x.stride(1), nn_ops._ptr(wq), nn_ops._ptr(y), nn_ops._ptr(part), n, cin,
                                           h, w, nn_ops._stream(x))
        assert rc == 0 and torch.equal(y, ref)
        tot, yf = part.view(nb, 2, 64).sum(0), y.double()
        torch.testing.assert_close(tot[0], yf.sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(tot[1], (yf * yf).sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3)


def test_fused_seld_loss_matches_the_eager_loss():
    """salsa_nn_seld_loss / _bwd (loss + gradients in one launch each) against the eager torch expression of crnn/loss.py
    (reference models/interfaces.py:304-355): the three values to 1e-6 relative, the gradients w.r.t. both predictions to 1e-6
    of their scale, also when the detached parts receive gradients of their own; a batch with no active class gives nan in
    both (0 / 0), as the reference does."""
    from salsa_amd.crnn import loss as L
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(5)
    B, T, nc = 8, 80, 12
    sed = (torch.rand(B, T, nc, generator=g) < 0.2).float().to(dev)
    doa_gt = (torch.rand(B, T, 3 * nc, generator=g) * 2 - 1).to(dev) * sed.repeat(1, 1, 3)
    logit0 = (torch.randn(B, T, nc, generator=g) * 3).to(dev)
    doa0 = torch.tanh(torch.randn(B, T, 3 * nc, generator=g)).to(dev)
    doa0[0, 0, :4] = doa_gt[0, 0, :4]                                   # exact hits: sign(0) = 0 in both
    res = {}
    old = L.FUSED_LOSS
    try:
        for fused in (False, True):
            L.FUSED_LOSS = fused
            logit, doa = logit0.clone().requires_grad_(True), doa0.clone().requires_grad_(True)
            out = L.seld_loss({'event_frame_logit': logit, 'doa_frame_output': doa}, sed, doa_gt)
            (out[0] * 1.7 + out[1] * 0.25 - out[2] * 0.5).backward()
            res[fused] = ([float(o.detach()) for o in out], logit.grad.clone(), doa.grad.clone())
            logit.grad = doa.grad = None
            out = L.seld_loss({'event_frame_logit': logit, 'doa_frame_output': doa}, sed, doa_gt)
            out[0].backward()                                          # the trainer's case: only the total is differentiated
            res[fused] += (logit.grad.clone(), doa.grad.clone())
        L.FUSED_LOSS = True
        nan = L.seld_loss({'event_frame_logit': logit0, 'doa_frame_output': doa0}, torch.zeros_like(sed), doa_gt)
        assert torch.isnan(nan[2]) and torch.isnan(nan[0]) and torch.isfinite(nan[1])
    finally:
        L.FUSED_LOSS = old
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 1e-6 * abs(b) + 1e-7, (res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        assert a.shape == b.shape and (a - b).abs().max() <= 1e-6 * b.abs().max(), (a - b).abs().max()


def test_hip_frequency_mean_matches_torch_forward_and_backward():
    """salsa_nn_freq_mean_fwd / _bwd (the decoder's mean over frequency + transpose, one pass, time-major float32 output) against
    ``feat.float().mean(3).transpose(1, 2)``: forward to float32 rounding of a 12-term sum, backward exactly g / F rounded to bf16."""
    from salsa_amd.crnn import nn_ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(11)
    for shape in ((3, 512, 40, 12), (2, 64, 7, 5)):
        x = torch.randn(shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = nn_ops.freq_mean_sequence(x)
        assert y.shape == (shape[0], shape[2], shape[1]) and y.dtype == torch.float32
        assert y.transpose(0, 1).is_contiguous()                                   # a view of the time-major buffer
        ref = x.detach().float().mean(dim=3).transpose(1, 2)
        assert torch.allclose(y, ref, rtol=1e-6, atol=1e-6)
        gy = torch.randn(y.shape, generator=g).to(dev)
        y.backward(gy)
        want = (gy.transpose(1, 2).unsqueeze(3) / shape[3]).expand(shape).to(torch.bfloat16)
        assert x.grad.shape == x.shape and x.grad.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(x.grad, want)
    xf = torch.randn(2, 64, 7, 5, device=dev)                                      # float32 input: the torch expression
    assert torch.equal(nn_ops.freq_mean_sequence(xf), xf.mean(dim=3).transpose(1, 2))


def test_eval_shortcut_folding_matches_the_unfolded_block():
    """Inference under bf16 autocast: the 1x1 shortcut's BatchNorm folded into its GEMM, its shift added by conv2's epilogue
    (nn_ops.folded_shortcut) against bn(conv1x1(x)) as a separate pass -- per block to bf16 rounding of the two differently
    rounded intermediates, and on the whole model's outputs."""
    from salsa_amd.crnn import SeldCRNN, nn_ops
    from salsa_amd.crnn.model import ResBlock
    from salsa_amd.crnn.testing import seeded_fill
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    old = nn_ops.FOLD_SHORTCUT
    try:
        blk = ResBlock(64, 128, 2).to(dev).eval()
        with torch.no_grad():
            for bn in (blk.bn1, blk.bn2, blk.short_bn):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.5, 0.5)
                bn.running_mean.uniform_(-0.3, 0.3)
                bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(2, 64, 32, 20, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        outs = {}
        for fold in (False, True):
            nn_ops.FOLD_SHORTCUT = fold
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
                outs[fold] = blk(x).float()
        scale = float(outs[False].abs().max())
        assert outs[True].shape == outs[False].shape and (outs[True] - outs[False]).abs().max() <= 0.02 * scale
        assert (outs[True] - outs[False]).abs().mean() <= 2e-3 * scale
        m = SeldCRNN()
        seeded_fill(m, 7)
        m = m.to(dev).eval()
        xin = torch.randn(2, 7, 160, 200, device=dev)
        res = {}
        for fold in (False, True):
            nn_ops.FOLD_SHORTCUT = fold
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
                res[fold] = {k: v.float() for k, v in m(xin).items()}
        for k in res[False]:
            s = float(res[False][k].abs().max()) + 1e-6
            assert (res[True][k] - res[False][k]).abs().max() <= 0.03 * s, k
    finally:
        nn_ops.FOLD_SHORTCUT = old


@pytest.mark.parametrize('shape', [(4, 128, 128, 40, 50), (3, 256, 512, 20, 12), (2, 64, 128, 33, 25), (5, 512, 512, 7, 12)])
def test_wide_conv_statistics_epilogue_matches_sums_of_its_output(shape):
    """salsa_nn_conv3x3_wide_stats: the convolution's output is bit-identical to the plain kernel's, and the float64 partial rows
    it leaves add up to the per-channel sum / sum of squares of that (bf16-rounded) output -- every tile variant (512 / 256 pixels
    x 128 / 64 channels), maps whose pixel count is not a multiple of the tile (masked tail)."""
    from salsa_amd import _lib
    from salsa_amd.crnn import nn_ops
    N, cin, cout, H, W = shape
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(sum(shape))
    x = torch.randn((N, cin, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert _lib.load().salsa_nn_conv3x3_wide_supported(N, H, W, cin, cout)
    nb = _lib.load().salsa_nn_conv3x3_wide_stats_blocks(N, H, W, cin, cout)
    assert nb > 0
    part = torch.full((nb, 2, cout), float('nan'), dtype=torch.float64, device=dev)
    y_plain = nn_ops._conv_wide(x, w)
    y = nn_ops._conv_wide(x, w, stats_part=part)
    assert torch.equal(y, y_plain)
    yf = y.double()
    s, q = part[:, 0].sum(0), part[:, 1].sum(0)
    ref_s, ref_q = yf.sum(dim=(0, 2, 3)), (yf * yf).sum(dim=(0, 2, 3))
    assert torch.isfinite(part).all()
    assert (s - ref_s).abs().max() <= 1e-5 * yf.abs().sum(dim=(0, 2, 3)).max()         # float32 partial sums inside a tile
    assert ((q - ref_q).abs() / ref_q).max() <= 1e-5


def test_gpu_training_step_matches_reference_modules():
    """Fixture g16 (the REFERENCE encoder + decoder in train() mode, the reference loss, backward; tools/make_golden_crnn.py) on
    the product path: Trainer's channels-last model on the GPU.
      * float32 (no autocast; the fused BatchNorm / pool / GRU kernels in float32): loss to 1e-4, the seven named gradients to
        cosine > 0.9999, the running statistics to 1e-4 -- the training semantics ARE the reference's;
      * bf16 autocast with every hand-written HIP layer on (the benchmarked path): loss to 2 %, statistics to 2 %, and every
        gradient at least as close to the reference as the SAME model on torch / MIOpen bf16 layers is (cosine >= torch's - 0.02:
        at this batch of 4 the early layers' gradients carry ~0.91 cosine of bf16 noise on either path; tools/probes/g16_probe.py).
    (float32 CPU: tests/test_crnn_cpu.py, the same fixture.)"""
    from salsa_amd.crnn import model as M, nn_ops
    from salsa_amd.crnn.loss import seld_loss
    from salsa_amd.crnn.testing import dropout_off, g16_batch, seeded_fill
    from salsa_amd.crnn.train import Trainer
    meta, a = load_golden('g16_crnn_train')

    def run(amp, hip):
        nn_ops.USE_HIP_POOL = nn_ops.USE_HIP_BN = nn_ops.USE_HIP_CONV = hip
        M.FUSED_GRU = hip
        try:
            tr = Trainer('cuda:0', total_steps=10, amp_dtype=amp)
            seeded_fill(tr.raw_model, meta['weight_seed'])
            nn_ops.invalidate_conv_caches(tr.raw_model)
            x, sed, doa = (t.cuda() for t in g16_batch(meta))
            tr.model.train()
            with dropout_off(tr.raw_model):
                with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp is not None):
                    pred = tr.model(tr._input_layout(x))
                loss, sed_l, doa_l = seld_loss(pred, sed, doa)
                loss.backward()
        finally:
            nn_ops.USE_HIP_POOL = nn_ops.USE_HIP_BN = nn_ops.USE_HIP_CONV = True
            M.FUSED_GRU = True
        params, cos = dict(tr.raw_model.named_parameters()), {}
        for k, st in meta['grad_strides'].items():
            got = params[k].grad.float().reshape(-1)[::st].cpu().numpy().astype(np.float64)
            ref = a['grad:' + k].astype(np.float64)
            cos[k] = (float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref))), float(np.linalg.norm(got) / np.linalg.norm(ref)))
        sd = tr.raw_model.state_dict()
        stats = {k: sd[k].float().cpu().numpy() for k in (x_[5:] for x_ in a.keys() if x_.startswith('stat:'))}
        return [float(loss.detach()), float(sed_l.detach()), float(doa_l.detach())], cos, stats

    loss, cos, stats = run(None, True)                                   # float32 on the GPU
    np.testing.assert_allclose(loss, a['loss'], rtol=1e-4)
    for k, (c, r) in cos.items():
        assert c > 0.9999 and abs(r - 1) < 1e-3, (k, c, r)
    for k, v in stats.items():
        np.testing.assert_allclose(v, a['stat:' + k], rtol=1e-4, atol=1e-5, err_msg=k)
    loss, cos, stats = run(torch.bfloat16, True)                         # the benchmarked path
    _, cos_torch, _ = run(torch.bfloat16, False)                         # the same model on torch / MIOpen bf16 layers
    np.testing.assert_allclose(loss, a['loss'], rtol=2e-2)
    for k, (c, r) in cos.items():
        assert c >= cos_torch[k][0] - 0.02 and c > 0.85 and 0.9 < r < 1.1, (k, c, r, cos_torch[k])
    for k, v in stats.items():
        np.testing.assert_allclose(v, a['stat:' + k], rtol=2e-2, atol=2e-3, err_msg=k)


def test_config5_sub_batch_is_batch_invariant():
    """BASELINE config 5's REAL sub-batch -- 32 x (7, 4800, 200), the largest tensors any kernel here sees (3.9-GB stem output;
    reference path models/seld_models.py:110-117) -- had a timing but no check (round-3 review): the eval forward of the first
    two clips inside the batch of 32 must equal the forward of those two clips alone (the persistent tile walks, the folded
    BatchNorm epilogues and the GRU scans all depend on the batch size; no output may)."""
    from salsa_amd.crnn.testing import seeded_fill
    from salsa_amd.crnn.train import Trainer
    tr = Trainer('cuda:0', total_steps=10)
    seeded_fill(tr.raw_model, 11)
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(32, 7, 4800, 200, device='cuda', generator=g)
    x[:, 4:] *= (torch.rand(32, 3, 4800, 200, device='cuda', generator=g) < 0.25)
    p32, d32 = tr.infer(x)
    p2, d2 = tr.infer(x[:2].contiguous())
    assert p32.shape == (32, 600, 12) and d32.shape == (32, 600, 36)
    assert bool(torch.isfinite(p32).all()) and bool(torch.isfinite(d32).all())
    np.testing.assert_allclose(p32[:2].cpu().numpy(), p2.cpu().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(d32[:2].cpu().numpy(), d2.cpu().numpy(), rtol=0, atol=1e-6)
    p_last, d_last = tr.infer(x[30:].contiguous())
    np.testing.assert_allclose(p32[30:].cpu().numpy(), p_last.cpu().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(d32[30:].cpu().numpy(), d_last.cpu().numpy(), rtol=0, atol=1e-6)


def test_deterministic_mode_gives_bit_equal_weight_gradients():
    """nn_ops.set_deterministic (include/salsa_nn.h: salsa_nn_set_deterministic): with it on, every weight-gradient kernel (64 -> 64,
    first layer, wide, 1x1) and the GRU bias column sums write per-workgroup partial slabs that one launch adds in slab order,
    so two identical training passes -- same parameters, same batch, same dropout seeds -- leave BIT-EQUAL gradients on every
    parameter (round-3 review: float atomics summed the partials in arrival order).  The deterministic gradients agree with the
    atomic ones to float32 rounding of differently ordered sums."""
    from salsa_amd.crnn import nn_ops
    from salsa_amd.crnn.loss import seld_loss
    from salsa_amd.crnn.train import Trainer, synthetic_batch
    tr = Trainer('cuda:0', total_steps=10)
    with torch.no_grad():
        for m in tr.raw_model.modules():                      # (zero-initialised bn2 weights would silence half the network)
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
    x, sed, doa = synthetic_batch(8, 'cuda:0', seed=3)
    params = [p for p in tr.raw_model.parameters()]

    def grads(seed):
        torch.manual_seed(seed)
        tr.raw_model.zero_grad(set_to_none=True)
        tr.model.train()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            pred = tr.model(tr._input_layout(x))
        seld_loss(pred, sed, doa)[0].backward()
        torch.cuda.synchronize()
        return [p.grad.detach().float().clone() for p in params]

    dev = torch.device('cuda', torch.cuda.current_device())
    try:
        nn_ops.set_deterministic(False)                                          # the atomics, for comparison
        assert not nn_ops.is_deterministic()
        base = grads(5)
        nn_ops.set_deterministic(True)
        assert nn_ops.is_deterministic()
        nn_ops._DET_WS[dev].view(torch.float32).fill_(float('nan'))               # (the slabs are never cleared: every element of a slab
        a = grads(5)                                                             #  is written by its workgroup before it is read)
        nn_ops._DET_WS[dev].view(torch.float32).fill_(float('nan'))
        b = grads(5)
    finally:
        nn_ops._DET_USER[0] = None                                               # back to the default (on, selected per forward)
    names = [n for n, _ in tr.raw_model.named_parameters()]
    for n, ga, gb, g0 in zip(names, a, b, base):
        assert torch.equal(ga, gb), n                                            # bit-equal, run to run
        scale = float(g0.abs().max()) + 1e-12
        assert float((ga - g0).abs().max()) <= 2e-3 * scale, (n, float((ga - g0).abs().max()), scale)   # same gradient as the atomic path
    # the library refuses a workspace that is too small for a shape instead of writing past it
    import ctypes as C
    from salsa_amd import _lib
    small = torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
    _lib.load().salsa_nn_set_deterministic(C.c_void_p(small.data_ptr()), small.numel())
    try:
        with pytest.raises(RuntimeError):
            grads(5)
    finally:
        nn_ops.set_deterministic(False)
        nn_ops._DET_USER[0] = None               # back to the default: the next differentiable forward selects its device's workspace


def test_one_launch_adam_matches_torch_adam():
    """optim.HipAdam (salsa_nn_adam_step: every tensor of a parameter group in one launch) against torch.optim.Adam on the same
    gradients: contiguous and channels-last tensors, sizes that are not multiples of 4, tensors that are 4-byte-aligned views, more
    than one 8192-element chunk, with and without weight decay, a changing learning rate; and the state dict loads into torch's."""
    from salsa_amd.crnn.optim import HipAdam
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(7)
    for wd in (0.0, 0.01):
        base = torch.randn(100003, generator=g)
        shapes = [(64, 64, 3, 3), (512,), (7,), (3, 5), (20011,), (12, 256)]
        ref_p, our_p = [], []
        for sh in shapes:
            t = torch.randn(sh, generator=g)
            a, b = t.clone().to(dev), t.clone().to(dev)
            if len(sh) == 4:
                a, b = a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)
            ref_p.append(torch.nn.Parameter(a)); our_p.append(torch.nn.Parameter(b))
        va, vb = base.clone().to(dev), base.clone().to(dev)                      # an odd-offset view: not 16-byte aligned
        ref_p.append(torch.nn.Parameter(va[1:50002])); our_p.append(torch.nn.Parameter(vb[1:50002]))
        ref = torch.optim.Adam(ref_p, lr=3e-4, weight_decay=wd)
        our = HipAdam(our_p, lr=3e-4, weight_decay=wd)
        for it in range(6):
            lr = 3e-4 * (1.0 - 0.1 * it)
            for opt in (ref, our):
                opt.param_groups[0]['lr'] = lr
            for a, b in zip(ref_p, our_p):
                gr = torch.randn(a.shape, generator=g).to(dev) * (10.0 ** (it - 3))
                if a.dim() == 4:
                    gr = gr.contiguous(memory_format=torch.channels_last)
                a.grad, b.grad = gr.clone(), gr.clone()
            ref.step(); our.step()
            for a, b in zip(ref_p, our_p):
                torch.testing.assert_close(b.detach(), a.detach(), rtol=2e-6, atol=1e-8)
        for a, b in zip(ref_p, our_p):
            for key in ('exp_avg', 'exp_avg_sq'):          # (the first moment cancels: tolerance relative to the tensor's scale)
                want = ref.state[a][key]
                torch.testing.assert_close(our.state[b][key], want, rtol=2e-6, atol=1e-6 * float(want.abs().max()))
            assert float(our.state[b]['step']) == float(ref.state[a]['step']) == 6.0
        again = torch.optim.Adam(our_p, lr=3e-4, weight_decay=wd)
        again.load_state_dict(our.state_dict())                                   # the same state layout and keys
        assert float(again.state[our_p[0]]['step']) == 6.0
        # round-4 advice: every parameter owns its step scalar (a shared tensor object, loaded into torch's single-tensor / foreach
        # paths, advanced once per PARAMETER); torch's state loads back with the steps as own host scalars
        steps = [our.state[b]['step'] for b in our_p]
        assert len({id(t) for t in steps}) == len(steps) and 'step_tensor' not in our.param_groups[0]
        assert len({id(again.state[b]['step']) for b in our_p}) == len(our_p)
        fused = torch.optim.Adam(ref_p, lr=3e-4, weight_decay=wd, fused=True)     # keeps `step` on the device
        import copy
        fused.load_state_dict(copy.deepcopy(ref.state_dict()))    # (deep copies: load_state_dict keeps same-device tensors as they are)
        back = HipAdam(our_p, lr=3e-4, weight_decay=wd)
        back.load_state_dict(copy.deepcopy(fused.state_dict()))
        assert all(not back.state[b]['step'].is_cuda and float(back.state[b]['step']) == 6.0 for b in our_p)
        assert len({id(back.state[b]['step']) for b in our_p}) == len(our_p)
        # a parameter whose first gradient comes late keeps ITS OWN count (one launch per distinct count)
        late = torch.nn.Parameter(torch.randn(33, generator=g).to(dev))
        late_ref = torch.nn.Parameter(late.detach().clone())
        back.add_param_group({'params': [late]})
        ref.add_param_group({'params': [late_ref]})
        for a, b in zip(ref_p + [late_ref], our_p + [late]):
            gr = torch.randn(a.shape, generator=g).to(dev)
            if a.dim() == 4:
                gr = gr.contiguous(memory_format=torch.channels_last)
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step(); back.step()
        assert float(back.state[late]['step']) == 1.0 and float(back.state[our_p[0]]['step']) == 7.0
        torch.testing.assert_close(late.detach(), late_ref.detach(), rtol=2e-6, atol=1e-8)
        torch.testing.assert_close(our_p[0].detach(), ref_p[0].detach(), rtol=2e-6, atol=1e-8)


def test_relu_bit_planes_leave_every_gradient_bit_equal():
    """The residual blocks' second BatchNorm: the backward's ReLU mask from the bit plane the forward leaves
    (salsa_nn_bn_train_fwd_bits / _pool_bits, salsa_nn_bn_bwd relu = 2 / salsa_nn_bn_bwd_pool_bits) instead of from the stored
    output / the residual -- the bits ARE the tests the backward made, so with the deterministic reductions every gradient of the
    network and the input gradient are bit-equal with the planes on and off; also at a map with an odd height and width (the pool
    drops the last row / column: their bytes are never written and must not matter) and in float32."""
    from salsa_amd.crnn import nn_ops
    from salsa_amd.crnn.loss import seld_loss
    from salsa_amd.crnn.train import Trainer, synthetic_batch
    tr = Trainer('cuda:0', total_steps=10)
    with torch.no_grad():
        for m in tr.raw_model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
    x, sed, doa = synthetic_batch(4, 'cuda:0', seed=11)
    params = [p for p in tr.raw_model.parameters()]

    def grads(on):
        nn_ops.USE_BN_RELU_BITS = on
        torch.manual_seed(9)
        tr.raw_model.zero_grad(set_to_none=True)
        tr.model.train()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            pred = tr.model(tr._input_layout(x))
        seld_loss(pred, sed, doa)[0].backward()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in params]

    try:
        a, b = grads(True), grads(False)
        names = [n for n, _ in tr.raw_model.named_parameters()]
        assert nn_ops.is_deterministic()
        for n, ga, gb in zip(names, a, b):
            assert torch.equal(ga, gb), n
        # the layers alone, ragged pooled map, both dtypes, gradient of the input and of the residual too
        g = torch.Generator(device='cuda').manual_seed(1)
        for dtype in (torch.bfloat16, torch.float32):
            for pool in (False, True):
                outs = []
                for on in (True, False):
                    nn_ops.USE_BN_RELU_BITS = on
                    bn = nn_ops.BatchNormAct2d(64).cuda().train()
                    with torch.no_grad():
                        bn.weight.copy_(torch.linspace(0.5, 1.5, 64)); bn.bias.copy_(torch.linspace(-0.3, 0.3, 64))
                    gg = torch.Generator(device='cuda').manual_seed(5)
                    xx = torch.randn((3, 64, 9, 7), device='cuda', generator=gg).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                    rr = torch.randn((3, 64, 9, 7), device='cuda', generator=gg).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                    y = bn.relu_pool(xx, residual=rr) if pool else bn(xx, residual=rr, relu=True)
                    gy = torch.randn(y.shape, device='cuda', generator=gg).to(dtype)
                    y.backward(gy)
                    outs.append((y.detach().clone(), xx.grad.clone(), rr.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
                for u, v in zip(*outs):
                    assert torch.equal(u, v), (dtype, pool)
    finally:
        nn_ops.USE_BN_RELU_BITS = True
