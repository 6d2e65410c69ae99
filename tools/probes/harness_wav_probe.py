"""tools/probes/harness_wav_probe.py [clips]: the file-to-file harness on a tree of 16-bit PCM WAV clips (what TNSSE2021 ships) against the
float32 .npy tree bench.py's harness leg uses: audio-s/s of extract_features(task='feature'), warm."""
import os, shutil, sys, tempfile, time
import numpy as np, yaml
sys.path.insert(0, '.')
from scipy.io import wavfile
from bench import make_batch
from salsa_amd import features

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 60 * 24000
host = make_batch(2021, 8, n)
base = '/dev/shm' if os.path.isdir('/dev/shm') and shutil.disk_usage('/dev/shm').free > 12e9 else None
for kind in ('npy', 'wav16', 'wav32f'):
    tmp = tempfile.mkdtemp(dir=base)
    try:
        d = os.path.join(tmp, 'data', 'foa_dev'); os.makedirs(d); os.makedirs(os.path.join(tmp, 'data', 'foa_eval'))
        for i in range(n_clips):
            y = host[i % 8]
            if kind == 'npy':
                np.save(os.path.join(d, 'c%03d.npy' % i), y)
            elif kind == 'wav16':
                wavfile.write(os.path.join(d, 'c%03d.wav' % i), 24000, np.clip(y.T / np.abs(y).max() * 32767, -32768, 32767).astype(np.int16))
            else:
                wavfile.write(os.path.join(d, 'c%03d.wav' % i), 24000, y.T)
        cfg = {'data_dir': os.path.join(tmp, 'data'), 'feature_dir': os.path.join(tmp, 'feat'),
               'data': {'format': 'foa', 'fs': 24000, 'n_fft': 512, 'win_len': 512, 'hop_len': 300, 'fmin_doa': 50, 'fmax_doa': 9000}}
        yaml.safe_dump(cfg, open(os.path.join(tmp, 'cfg.yml'), 'w'))
        for rep in range(3):
            t0 = time.perf_counter()
            features.extract_features(data_config=os.path.join(tmp, 'cfg.yml'), task='feature', batch_size=32)
            dt = time.perf_counter() - t0
            print(kind, 'run', rep, '%.3f s' % dt, '%.1f k audio-s/s' % (n_clips * 60 / dt / 1e3), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
