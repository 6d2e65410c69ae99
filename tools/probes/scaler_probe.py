import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from salsa_amd.extractor import SalsaExtractor
from salsa_amd.synth import synth_clip
dev = torch.device('cuda:0')
for fmt, fmax, secs, B, ft in (('foa', 9000, 60, 32, 'salsa'), ('mic', 4000, 8, 32, 'salsa'), ('mic', 2000, 60, 32, 'salsa_lite')):
    ys = np.stack([synth_clip(2021 + i % 4, secs * 24000) for i in range(B)])
    a = torch.from_numpy(ys).to(dev)
    for scaler in (False, True):
        ex = SalsaExtractor(audio_format=fmt, fmax_doa=fmax, feature_type=ft, device=dev)
        F = ex.output_shape(secs * 24000)[2]
        if scaler:
            ex.set_scaler(torch.full((4, 1, F), -60.0, device=dev), torch.full((4, 1, F), 12.0, device=dev))
        out = ex.extract(a)
        for _ in range(3): ex.extract(a, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ex.extract(a, out=out)
        torch.cuda.synchronize(); print(ft, fmt, secs, 's scaler' if scaler else 's plain ', '%.4f ms' % ((time.perf_counter() - t0) / 20 * 1e3), float(out.float().abs().sum()))
