import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from salsa_amd.extractor import SalsaExtractor
from bench import make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
host = make_batch(2021, B, 1440000)
dev = torch.device('cuda:0')
audio = torch.from_numpy(host).to(dev)
K = 20
for nstreams in (1, 2, 3):
    exs = [SalsaExtractor(device=dev) for _ in range(nstreams)]
    outs = [torch.empty((B, 7, 4801, 200), dtype=torch.float32, device=dev) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    for i in range(4):
        with torch.cuda.stream(streams[i % nstreams]):
            exs[i % nstreams].extract(audio, out=outs[i % nstreams])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        j = i % nstreams
        with torch.cuda.stream(streams[j]):
            exs[j].extract(audio, out=outs[j])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('streams', nstreams, 'ms/step %.4f' % (1e3 * dt / K), 'audio-s/s %.0f' % (B * 60 * K / dt))
    assert torch.equal(outs[0], outs[-1])
