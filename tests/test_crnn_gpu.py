"""GPU tests of the CRNN consumer (MIOpen / rocBLAS through torch): forward against the reference-model golden, and a
few bf16 training steps that must reduce the loss on a fixed batch."""
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
