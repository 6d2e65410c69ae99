"""tools/probes/lite_std_check.py [save.npz | compare a.npz b.npz]: SALSA-Lite / IPD features of a seeded batch from the library under SALSA_HIP_LIB
(SHA-256 printed, arrays optionally saved) -- run once per build; `compare` says where two builds differ.  The Lite STD instantiation must be
bit-identical to the general kernel."""
import hashlib, sys
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == 'compare':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        d = x != y
        print(k, 'differing elements', int(d.sum()), 'of', d.size)
        if d.any():
            for c in range(7):
                dc = d[:, c]
                if dc.any():
                    idx = np.argwhere(dc)
                    print('  channel', c, int(dc.sum()), 'max |diff|', float(np.abs(x[:, c] - y[:, c]).max()), 'bins', int(idx[:, 2].min()), '..', int(idx[:, 2].max()),
                          'first', idx[0].tolist(), float(x[:, c][tuple(idx[0])]), float(y[:, c][tuple(idx[0])]))
    sys.exit(0)
import torch
sys.path.insert(0, '.')
from salsa_amd.extractor import SalsaExtractor
from salsa_amd.synth import synth_clip
ys = np.stack([synth_clip(900 + i, 24000 * 5 + 137) for i in range(6)])
ys[5, :, :3000] = 0.0                                   # a silent stretch (tiny-product rescue, zero phases)
ys[4] *= 1e-18                                          # tiny spectra
a = torch.from_numpy(ys).cuda()
keep = {}
for ft, fmax in (('salsa_lite', 2000), ('salsa_ipd', 2000), ('salsa_lite', 4000), ('salsa_lite', 9000)):
    out = SalsaExtractor(audio_format='mic', feature_type=ft, fmax_doa=fmax).extract(a).cpu().numpy()
    keep['%s_%d' % (ft, fmax)] = out
    print(ft, fmax, tuple(out.shape), hashlib.sha256(out.tobytes()).hexdigest()[:16])
if len(sys.argv) > 1:
    np.savez(sys.argv[1], **keep)
