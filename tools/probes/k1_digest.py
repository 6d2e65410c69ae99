"""SHA-256 of the feature arrays of a fixed set of inputs through whatever library SALSA_HIP_LIB names: two builds that print the same
digests are bit-identical on config 2 (32 x 60 s FOA), MIC chunks, a ragged FOA batch, FOA with a fused scaler and interleaved audio."""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_batch
from salsa_amd import _lib
from salsa_amd.extractor import SalsaExtractor
from salsa_amd.synth import synth_clip
dev = torch.device('cuda:0')
def dig(t): return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
print('library flags: %r' % _lib.build_flags())
a = torch.from_numpy(make_batch(2021, 32, 60 * 24000)).to(dev)
print('config2 foa   ', dig(SalsaExtractor(device=dev).extract(a)))
ex = SalsaExtractor(device=dev); ex.set_scaler(np.full((4, 1, 200), -60.0, np.float32), np.full((4, 1, 200), 12.0, np.float32))
print('config2 scaler', dig(ex.extract(a[:4])))
m = torch.from_numpy(np.stack([synth_clip(4021 + i, 8 * 24000) for i in range(8)])).to(dev)
print('mic chunks    ', dig(SalsaExtractor(audio_format='mic', fmax_doa=4000, device=dev).extract(m)))
r = torch.from_numpy(np.stack([synth_clip(17 + i, 14700) for i in range(3)])).to(dev)
print('ragged foa    ', dig(SalsaExtractor(device=dev).extract(r)))
it = torch.from_numpy(np.ascontiguousarray(np.stack([synth_clip(27 + i, 72000) for i in range(2)]).transpose(0, 2, 1))).to(dev)
print('interleaved   ', dig(SalsaExtractor(audio_layout='interleaved', device=dev).extract(it)))
print('nocompress    ', dig(SalsaExtractor(is_compress_high_freq=False, device=dev).extract(r)))
print('lite mic      ', dig(SalsaExtractor(audio_format='mic', feature_type='salsa_lite', fmax_doa=2000, device=dev).extract(m)))
print('ipd mic       ', dig(SalsaExtractor(audio_format='mic', feature_type='salsa_ipd', fmax_doa=2000, device=dev).extract(r)))
print('nfft256       ', dig(SalsaExtractor(n_fft=256, hop_len=150, win_len=256, device=dev).extract(r)))
