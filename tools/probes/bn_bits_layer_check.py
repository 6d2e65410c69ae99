import sys, torch
sys.path.insert(0, '/root/repo')
from salsa_amd.crnn import nn_ops
nn_ops.set_deterministic(True, 'cuda:0')
for dtype in (torch.bfloat16, torch.float32):
    for pool in (False, True):
        for shape in ((3, 64, 9, 7), (8, 64, 80, 50), (8, 256, 20, 12)):
            outs = []
            for on in (True, False):
                nn_ops.USE_BN_RELU_BITS = on
                bn = nn_ops.BatchNormAct2d(shape[1]).cuda().train()
                with torch.no_grad():
                    bn.weight.copy_(torch.linspace(0.5, 1.5, shape[1])); bn.bias.copy_(torch.linspace(-0.3, 0.3, shape[1]))
                gg = torch.Generator(device='cuda').manual_seed(5)
                xx = torch.randn(shape, device='cuda', generator=gg).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                rr = torch.randn(shape, device='cuda', generator=gg).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                y = bn.relu_pool(xx, residual=rr) if pool else bn(xx, residual=rr, relu=True)
                gy = torch.randn(y.shape, device='cuda', generator=gg).to(dtype)
                y.backward(gy)
                outs.append((y.detach().clone(), xx.grad.clone(), rr.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
            print(dtype, pool, shape, [bool(torch.equal(u, v)) for u, v in zip(*outs)], [float((u.float()-v.float()).abs().max()) for u, v in zip(*outs)])
