"""ctypes binding of libsalsa_hip.so (C ABI in include/salsa_hip.h).  There is no CPU fallback: if the HIP library
is missing this module raises, loudly, with the build command."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SALSA_HIP_LIB') or os.path.join(_HERE, 'lib', 'libsalsa_hip.so')   # (env override: A/B probes of kernel variants)
SRC_PATH = os.path.join(_HERE, 'csrc', 'salsa_kernels.hip')
GRU_SRC_PATH = os.path.join(_HERE, 'csrc', 'gru_scan.hip')
NN_SRC_PATH = os.path.join(_HERE, 'csrc', 'nn_ops.hip')
CONV_SRC_PATH = os.path.join(_HERE, 'csrc', 'conv_mfma.hip')
CONV_WIDE_SRC_PATH = os.path.join(_HERE, 'csrc', 'conv_wide.hip')
CONV_1X1_SRC_PATH = os.path.join(_HERE, 'csrc', 'conv_1x1.hip')

FORMAT = {'foa': 0, 'mic': 1}
FEATURE = {'salsa': 0, 'salsa_lite': 1, 'salsa_ipd': 2}
LAYOUT = {'planar': 0, 'interleaved': 1}
FLAG_FLEX, FLAG_NO_CLIP_FREQS, FLAG_CLIP_SPATIAL_ALIAS = 1, 2, 4
FLAG_FORCE_F64 = 8  # verification: the all-float64 instantiation of the covariance / eigen kernel (include/salsa_hip.h)
PIPE_SPLIT_PAIRS, PIPE_GRAPH = 1, 2
MAX_KERNELS = 32

PARTIAL = 1     # salsa_extract_batch in the prefix-issue measurement mode (include/salsa_hip.h: SALSA_PARTIAL)
E_INVAL, E_NFFT, E_FORMAT, E_BINS, E_WORKSPACE, E_HIP = -1, -2, -3, -4, -5, -6


class SalsaParams(C.Structure):
    _fields_ = [('fs', C.c_int), ('n_fft', C.c_int), ('hop_len', C.c_int), ('win_len', C.c_int),
                ('fmin_doa', C.c_int), ('fmax_doa', C.c_int), ('cond_num', C.c_double), ('n_hopframes', C.c_int),
                ('is_tracking', C.c_int), ('is_compress_high_freq', C.c_int), ('audio_format', C.c_int),
                ('feature_type', C.c_int), ('audio_layout', C.c_int), ('flags', C.c_int),
                ('floor_mask_ratio', C.c_double), ('fmax_spec', C.c_int), ('reserved', C.c_int)]


_lib = None


def build_command():
    return ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', LIB_PATH, SRC_PATH, GRU_SRC_PATH, NN_SRC_PATH, CONV_SRC_PATH, CONV_WIDE_SRC_PATH, CONV_1X1_SRC_PATH]


def load():
    """Load libsalsa_hip.so; raise if it has not been built (the product never falls back to CPU code)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libsalsa_hip.so is missing (%s). Build it with `python -c "import __graft_entry__ as g; '
                           'g.build()"` or: %s' % (LIB_PATH, ' '.join(build_command())))
    L = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.salsa_abi_version.restype = C.c_int
    L.salsa_build_flags.restype = C.c_char_p
    L.salsa_last_error.restype = C.c_char_p
    L.salsa_bin_limits.argtypes = [C.c_int] * 4 + [ip, ip, ip]
    L.salsa_compress_matrix.argtypes = [C.c_int, C.c_int, fp]
    L.salsa_plan_create.argtypes = [C.POINTER(SalsaParams), C.POINTER(vp)]
    L.salsa_plan_destroy.argtypes = [vp]
    L.salsa_output_shape.argtypes = [vp, C.c_int64, ip, C.POINTER(C.c_int64), ip]
    L.salsa_workspace_bytes.restype = C.c_size_t
    L.salsa_workspace_bytes.argtypes = [vp, C.c_int, C.c_int64]
    L.salsa_extract_batch.argtypes = [vp, vp, C.c_int, C.c_int64, vp, vp, C.c_size_t, vp]
    L.salsa_logspec_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, vp, vp]
    L.salsa_eigvec_workspace_bytes.restype = C.c_size_t
    L.salsa_eigvec_workspace_bytes.argtypes = [vp, C.c_int, C.c_int, C.c_int64]
    L.salsa_eigvec_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int, vp, vp, vp, C.c_size_t, vp]
    L.salsa_eigvec_feature_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int, vp, vp, C.c_size_t, vp]
    L.salsa_plan_set_stats.argtypes = [vp, vp]
    L.salsa_plan_set_fused.argtypes = [vp, C.c_int]
    L.salsa_plan_set_timing.argtypes = [vp, C.c_int]
    L.salsa_plan_read_timing.argtypes = [vp, fp, C.POINTER(C.c_char_p), ip]
    L.salsa_plan_set_groups.argtypes = [vp, C.c_int]
    L.salsa_plan_set_pipeline.argtypes = [vp, C.c_int, C.c_int]
    L.salsa_plan_set_scaler.argtypes = [vp, vp, vp]
    L.salsa_gru_scan_fwd.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_gru_scan_fwd_regw.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_gru_scan_bwd_regw.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_gru_scan_bwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_avgpool2x2_fwd.argtypes = [vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_avgpool2x2_bwd.argtypes = [vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_c64.argtypes = [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_c64_bias_act.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_c64_bias_act_pool.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_c64_wrw.argtypes = [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_stem.argtypes = [vp, C.c_int64, C.c_int64, vp, vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_wide_supported.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.salsa_nn_conv3x3_wide.argtypes = [vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_wide_stats.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_wide_stats_blocks.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.salsa_nn_conv3x3_wide_wrw_supported.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.salsa_nn_conv3x3_wide_table_len.restype = C.c_int64
    L.salsa_nn_conv3x3_wide_table_len.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.salsa_nn_conv3x3_wide_tile_count.restype = C.c_int64
    L.salsa_nn_conv3x3_wide_tile_count.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.salsa_nn_conv3x3_wide_tables.argtypes = [C.c_int64, C.c_int, C.c_int, vp, vp, vp]
    L.salsa_nn_conv3x3_wide_wrw.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_wide_bias_act.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_bn_supported.argtypes = [C.c_int, C.c_int64, C.c_int]
    L.salsa_nn_bn_workspace_bytes.restype = C.c_size_t
    L.salsa_nn_bn_workspace_bytes.argtypes = [C.c_int, C.c_int64, C.c_int]
    L.salsa_nn_bn_train_fwd.argtypes = [vp, vp, vp, C.c_int, C.c_int64, C.c_int, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp,
                                        vp, C.c_int, C.c_float, C.c_uint32, vp, vp, C.c_int, vp]
    L.salsa_nn_bn_train_fwd_pool.argtypes = [vp, vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, C.c_float, vp, vp,
                                             vp, vp, vp, vp, vp, C.c_int, vp]
    L.salsa_nn_bn_bwd_pool.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.salsa_nn_bn_train_fwd_bits.argtypes = [vp, vp, vp, C.c_int, C.c_int64, C.c_int, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp,
                                             vp, C.c_int, C.c_float, C.c_uint32, vp, vp, C.c_int, vp, vp]
    L.salsa_nn_bn_train_fwd_pool_bits.argtypes = [vp, vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, C.c_float, vp, vp,
                                                  vp, vp, vp, vp, vp, C.c_int, vp, vp]
    L.salsa_nn_bn_bwd_pool_bits.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.salsa_nn_conv_filter_bank.argtypes = [vp, C.c_int, C.c_int, vp]
    L.salsa_nn_seld_loss.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int, C.c_float, C.c_float, vp, vp, vp, vp, vp]
    L.salsa_nn_seld_loss_bwd.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]
    L.salsa_nn_freq_mean_fwd.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_freq_mean_bwd.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_colsum2.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int, vp]
    L.salsa_nn_adam_step.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, vp]
    L.salsa_nn_set_deterministic.argtypes = [vp, C.c_size_t]
    L.salsa_nn_bn_train_finalize.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp]
    L.salsa_nn_conv3x3_c64_xform_stats.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_uint32, C.c_int64, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_c64_wrw_xform.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_uint32, C.c_int64, C.c_int, C.c_int, vp]
    L.salsa_nn_get_deterministic.argtypes = []
    L.salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes.restype = C.c_size_t
    L.salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.salsa_nn_conv3x3_stem_wrw_bnf.argtypes = [vp, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_size_t, C.c_int64,
                                                C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_c64_stats_blocks.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.salsa_nn_conv3x3_c64_stats.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_stem_stats_blocks.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.salsa_nn_conv3x3_stem_stats.argtypes = [vp, C.c_int64, C.c_int64, vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_stem_wrw.argtypes = [vp, C.c_int64, C.c_int64, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_conv3x3_stem_wrw_bn.argtypes = [vp, C.c_int64, C.c_int64, vp, vp, vp, C.c_int, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp]
    L.salsa_nn_conv1x1_supported.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.salsa_nn_conv1x1.argtypes = [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]
    L.salsa_nn_conv1x1_wrw_supported.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.salsa_nn_conv1x1_wrw.argtypes = [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]
    L.salsa_nn_bn_eval_fwd.argtypes = [vp, vp, vp, C.c_int, C.c_int64, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    L.salsa_nn_bn_bwd.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp,
                                  C.c_float, C.c_uint32, vp]
    L.salsa_scaler_accumulate.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, vp, vp]
    L.salsa_normalize_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, vp, vp, vp]
    L.salsa_augment_batch.argtypes = [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.salsa_to_freq_major.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, vp, vp]
    L.salsa_selftest_decibel.argtypes = [vp, vp, C.c_int64, vp]
    L.salsa_pcm_to_planar.argtypes = [vp, C.c_int, C.c_int, C.c_int64, vp, vp]
    L.salsa_resample_batch.argtypes = [vp, C.c_int, C.c_int64, vp, C.c_int64, C.c_int64, C.c_double, vp, vp, C.c_int, C.c_int, vp, vp]
    L.salsa_multichannel_workspace_bytes.restype = C.c_size_t
    L.salsa_multichannel_workspace_bytes.argtypes = [vp, C.c_int, C.c_int, C.c_int64]
    L.salsa_extract_multichannel.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, vp, vp, C.c_size_t, vp]
    flags = L.salsa_build_flags().decode()
    if flags:                                        # an A/B or probe library (SALSA_HIP_LIB / tools/dev_build.sh with -D...): say so, loudly
        import sys
        sys.stderr.write('salsa_amd: %s was NOT built with the product\'s flags:%s%s\n'
                         % (LIB_PATH, flags, ' -- a PROBE build computes wrong results on purpose' if ' PROBE' in flags else ''))
    _lib = L
    return L


def build_flags() -> str:
    """'' for the product's build; otherwise the probe marker / overridden tunables the loaded library reports (salsa_build_flags)."""
    return load().salsa_build_flags().decode()


def last_error() -> str:
    return load().salsa_last_error().decode()


EXPORTS = ['salsa_abi_version', 'salsa_build_flags', 'salsa_last_error', 'salsa_bin_limits', 'salsa_compress_matrix', 'salsa_plan_create',
           'salsa_plan_destroy', 'salsa_output_shape', 'salsa_workspace_bytes', 'salsa_extract_batch',
           'salsa_logspec_batch', 'salsa_eigvec_workspace_bytes', 'salsa_eigvec_batch', 'salsa_eigvec_feature_batch', 'salsa_plan_set_stats', 'salsa_plan_set_fused', 'salsa_plan_set_timing',
           'salsa_plan_read_timing', 'salsa_plan_set_groups', 'salsa_plan_set_pipeline', 'salsa_scaler_accumulate', 'salsa_normalize_batch', 'salsa_plan_set_scaler',
           'salsa_to_freq_major', 'salsa_augment_batch', 'salsa_selftest_decibel', 'salsa_multichannel_workspace_bytes', 'salsa_extract_multichannel',
           'salsa_resample_batch', 'salsa_pcm_to_planar']
GRU_EXPORTS = ['salsa_gru_scan_fwd', 'salsa_gru_scan_fwd_regw', 'salsa_gru_scan_bwd', 'salsa_gru_scan_bwd_regw']
NN_EXPORTS = ['salsa_nn_avgpool2x2_fwd', 'salsa_nn_avgpool2x2_bwd', 'salsa_nn_conv3x3_c64', 'salsa_nn_conv3x3_c64_bias_act', 'salsa_nn_conv3x3_c64_wrw', 'salsa_nn_conv3x3_stem', 'salsa_nn_conv3x3_c64_bias_act_pool', 'salsa_nn_conv3x3_wide_supported', 'salsa_nn_conv3x3_wide', 'salsa_nn_conv3x3_wide_bias_act', 'salsa_nn_conv3x3_wide_wrw_supported', 'salsa_nn_conv3x3_wide_table_len', 'salsa_nn_conv3x3_wide_tile_count', 'salsa_nn_conv3x3_wide_tables', 'salsa_nn_conv3x3_wide_wrw', 'salsa_nn_bn_supported', 'salsa_nn_bn_workspace_bytes', 'salsa_nn_bn_train_fwd',
              'salsa_nn_bn_eval_fwd', 'salsa_nn_bn_bwd', 'salsa_nn_bn_train_fwd_pool', 'salsa_nn_bn_bwd_pool', 'salsa_nn_conv_filter_bank', 'salsa_nn_conv3x3_c64_stats_blocks', 'salsa_nn_conv3x3_c64_stats', 'salsa_nn_conv3x3_stem_wrw', 'salsa_nn_conv3x3_stem_stats_blocks', 'salsa_nn_conv3x3_stem_stats', 'salsa_nn_conv3x3_stem_wrw_bn', 'salsa_nn_conv1x1_supported', 'salsa_nn_conv1x1', 'salsa_nn_conv1x1_wrw_supported', 'salsa_nn_conv1x1_wrw', 'salsa_nn_seld_loss', 'salsa_nn_seld_loss_bwd', 'salsa_nn_freq_mean_fwd',
              'salsa_nn_freq_mean_bwd', 'salsa_nn_colsum2', 'salsa_nn_conv3x3_wide_stats', 'salsa_nn_conv3x3_wide_stats_blocks', 'salsa_nn_set_deterministic', 'salsa_nn_get_deterministic', 'salsa_nn_conv3x3_stem_wrw_bnf', 'salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes', 'salsa_nn_bn_train_finalize', 'salsa_nn_conv3x3_c64_xform_stats', 'salsa_nn_conv3x3_c64_wrw_xform', 'salsa_nn_adam_step', 'salsa_nn_bn_train_fwd_bits',
              'salsa_nn_bn_train_fwd_pool_bits', 'salsa_nn_bn_bwd_pool_bits']
