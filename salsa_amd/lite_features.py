"""Drop-in surface of dataset/salsa_lite_feature_extraction.py (SALSA-Lite / SALSA-IPD, MIC only) on the MI355X path.
extract_features mirrors :18-137: same config keys, directory naming (no ``cond`` field, :69), split iteration and
(7, T, cutoff_bin-lower_bin) float32 'feature' per clip, then the shared compute_scaler."""
import os
import sys
from timeit import default_timer as timer

import numpy as np

from .features import _cli, _extract_split, _parse, compute_scaler

import logging  # noqa: E402
_log = logging.getLogger('salsa_amd.features')


def extract_features(data_config: str = 'configs/tnsse2021_salsa_lite_feature_config.yml',
                     feature_type: str = 'salsa_lite',
                     task: str = 'feature_scaler',
                     batch_size: int = 32) -> None:
    assert feature_type in ['salsa_lite', 'salsa_ipd'], 'Invalid feature type {}'.format(feature_type)
    cfg, audio_format, fs, n_fft, hop_length, win_length, fmin_doa, fmax_doa = _parse(data_config)
    fmax_doa = int(np.min((fmax_doa, fs // 2)))
    feature_description = '{}fs_{}nfft_{}nhop_{}fmaxdoa'.format(fs, n_fft, hop_length, int(fmax_doa))
    _log.info('feature directory name: %s', feature_description)
    assert audio_format == 'mic', 'SALSA-Lite and SALSA-IPD are only for MIC format!'
    splits = ['mic_dev', 'mic_eval']
    if task in ['feature_scaler', 'feature']:
        from .extractor import SalsaExtractor
        ex = SalsaExtractor(fs=fs, n_fft=n_fft, hop_len=hop_length, win_len=win_length, fmin_doa=fmin_doa,
                            fmax_doa=fmax_doa, audio_format='mic', feature_type=feature_type)
        from . import features as _f
        scaler = {} if (task == 'feature_scaler' and _f.FUSED_SCALER) else None    # (as features.extract_features: statistics taken on the device)
        for split in splits:
            _log.info('split %s: extracting on %s', split, ex.device)
            start_time = timer()
            audio_dir = os.path.join(cfg['data_dir'], split)
            feature_dir = os.path.join(cfg['feature_dir'], feature_type, audio_format, feature_description, split)
            _extract_split(ex, audio_dir, feature_dir, fs, batch_size, scaler=scaler if split.endswith('_dev') else None)
            _log.info('split %s: done in %.3f s', split, timer() - start_time)
        if scaler is not None and scaler.get('sums') is not None and not scaler.get('unavailable'):
            feature_dir = os.path.join(cfg['feature_dir'], feature_type, audio_format, feature_description)
            _f.write_scaler_from_sums(feature_dir, audio_format, scaler['sums'], scaler['n'])
            return
    if task in ['feature_scaler', 'scaler']:
        feature_dir = os.path.join(cfg['feature_dir'], feature_type, audio_format, feature_description)
        compute_scaler(feature_dir=feature_dir, audio_format=audio_format)


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO, format='%(message)s')      # progress lines on the console, as a CLI should
    _cli(extract_features, sys.argv[1:])
