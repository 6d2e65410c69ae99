/*
 * salsa_hip.h -- C ABI of libsalsa_hip.so, the MI355X (gfx950) SALSA / SALSA-Lite feature extractor.
 *
 * The upstream reference (thomeou/SALSA) is pure Python and has no FFI layer: its boundary is a set of Python
 * function signatures plus the (7, T, F) float32 feature array they produce.  Each entry point below replaces one of
 * those, batched over clips and operating on caller-owned DEVICE memory (plain pointers and sizes, no torch types);
 * salsa_amd/features.py re-exposes the reference's own signatures on top of them.
 *
 *   reference interface (file:line, relative to the upstream repo)               replaced by
 *   ---------------------------------------------------------------------------  ---------------------------------
 *   extract_features() per-file body, dataset/salsa_feature_extraction.py:353-377 salsa_extract_batch (SALSA)
 *   extract_features() per-file body, dataset/salsa_lite_feature_extraction.py:94-123  salsa_extract_batch (LITE/IPD)
 *   MagStftExtractor.extract, dataset/salsa_feature_extraction.py:177-201         salsa_logspec_batch
 *   extract_normalized_eigenvector, dataset/salsa_feature_extraction.py:17-129    salsa_eigvec_batch
 *   bin limits / freq_dim, dataset/salsa_feature_extraction.py:298-313 (+ lite :50-59)  salsa_bin_limits, salsa_output_shape
 *   MagStftExtractor.W, dataset/salsa_feature_extraction.py:152-175               salsa_compress_matrix (host)
 *   compute_scaler, dataset/salsa_feature_extraction.py:204-262                   salsa_scaler_accumulate
 *   Database.load_chunk_data normalisation, dataset/database.py:197-202           salsa_normalize_batch
 *   SeldDataset train transforms, utilities/transforms.py (datamodule.py:45-82)   salsa_augment_batch
 *   SalsaFeatures / SalsaLiteFeatures.__call__, contrib/salsa_flexible.py:237-265 (+ :286-400)   salsa_extract_batch with SALSA_FLAG_FLEX, salsa_to_freq_major
 *   librosa.load(sr=fs)'s resampling of a file of another rate, dataset/salsa_feature_extraction.py:353   salsa_resample_batch
 *   librosa.load's PCM -> float32 (channels, samples) conversion, dataset/salsa_feature_extraction.py:353    salsa_pcm_to_planar
 *
 * Conventions: every function returns 0 on success or a negative SALSA_E* code; salsa_last_error() gives the
 * message of the calling thread's last failure.  Device pointers are caller-owned; work is enqueued asynchronously
 * on the HIP stream passed in (NULL = the default stream) and nothing is allocated or synchronised inside the
 * extract calls, so they are hipGraph-capturable.  A plan is bound to the device current at salsa_plan_create.
 */
#ifndef SALSA_HIP_H
#define SALSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SALSA_ABI_VERSION 2

enum { SALSA_FORMAT_FOA = 0, SALSA_FORMAT_MIC = 1 };                      /* cfg['data']['format'] */
enum { SALSA_FEATURE_SALSA = 0, SALSA_FEATURE_LITE = 1, SALSA_FEATURE_IPD = 2 }; /* 'salsa' | 'salsa_lite' | 'salsa_ipd' */
enum { SALSA_LAYOUT_PLANAR = 0, SALSA_LAYOUT_INTERLEAVED = 1 };           /* [B][4][N] (librosa.load) | [B][N][4] (WAV) */

/* salsa_params.flags.  SALSA_FLAG_FLEX selects the semantics of the on-the-fly re-implementation contrib/salsa_flexible.py
 * (MIC-style features only), which differ from the dataset scripts: the noise-floor tracker follows the raw |X0| (not the
 * 3-frame RMS) with its initial floor clamped to 1e-6 (:118-120, :326-333); the coherence test gates even without
 * tracking -- and then switches a bin off for the rest of the clip once it fails (:336-337, :352-354); the frequency
 * normalisation vector is float32 (:188-190); the log-spectrogram is uncompressed over the SAME band as the spatial
 * features: bins [lower_bin, cutoff_bin(fmax_spec)) or, with SALSA_FLAG_NO_CLIP_FREQS, all n_fft/2+1 bins
 * (clip_freqs, :252-257); SALSA_FLAG_CLIP_SPATIAL_ALIAS zeroes spatial rows >= upper_bin of that band
 * (clip_spatial_alias, :262-263); fmax_doa is not clamped to fs/2.  cond_num = ew_thresh, n_hopframes =
 * covmat_avg_neighbours.  Fewer than 4 microphones: pad the audio with silent channels (the covariance gains zero
 * eigenvalues, the gate and the principal eigenvector are unchanged) and drop the padded output channels. */
enum { SALSA_FLAG_FLEX = 1, SALSA_FLAG_NO_CLIP_FREQS = 2, SALSA_FLAG_CLIP_SPATIAL_ALIAS = 4,
       /* verification switch: salsa_extract_batch / salsa_eigvec_feature_batch launch the all-float64 instantiation of the
        * covariance + eigen kernel instead of the production packed-float32 pair solve with its float64 cold list (same
        * spill, same masks), so a test can hold the two against each other on the GPU. */
       SALSA_FLAG_FORCE_F64 = 8 };

enum {
    SALSA_OK = 0,
    SALSA_PARTIAL = 1,     /* measurement mode only (salsa_plan_set_timing(plan, -1 | -2)): a PREFIX of the path was issued, the
                            * output planes hold whatever an earlier full call left there -- never returned by a plain call */
    SALSA_EINVAL = -1,     /* bad argument (NULL pointer, negative size, ...) */
    SALSA_ENFFT = -2,      /* n_fft not in {256, 512}: the reference's assert, salsa_feature_extraction.py:152,306 */
    SALSA_EFORMAT = -3,    /* unknown audio format: the reference's ValueError, :125,:332 ; lite requires MIC, lite :72 */
    SALSA_EBINS = -4,      /* upper_bin > cutoff_bin (lite :59) or an empty / oversized DOA band */
    SALSA_EWORKSPACE = -5, /* workspace smaller than salsa_workspace_bytes() */
    SALSA_EHIP = -6        /* a HIP runtime call failed (message in salsa_last_error) */
};

/* Mirrors the keyword arguments of the reference's extract_features() + the cfg['data'] block of its YAML. */
typedef struct salsa_params {
    int fs;                    /* 24000 */
    int n_fft;                 /* 512 | 256 */
    int hop_len;               /* 300 */
    int win_len;               /* <= n_fft */
    int fmin_doa;              /* Hz */
    int fmax_doa;              /* Hz (clamped to fs/2 like the reference) */
    double cond_num;           /* coherence threshold, default 5 */
    int n_hopframes;           /* default 3 ("do not change") */
    int is_tracking;           /* noise-floor tracking */
    int is_compress_high_freq; /* 200/100-bin compressed log-spectrogram */
    int audio_format;          /* SALSA_FORMAT_* */
    int feature_type;          /* SALSA_FEATURE_* */
    int audio_layout;          /* SALSA_LAYOUT_* */
    int flags;                 /* SALSA_FLAG_* (0 = the dataset scripts' semantics) */
    double floor_mask_ratio;   /* indicator_sig = mag > ratio * floor; 0 = the reference's 1.5 (:36; contrib kwarg) */
    int fmax_spec;             /* SALSA_FLAG_FLEX only: spectrogram cutoff in Hz; 0 = 9000 */
    int reserved;
} salsa_params;

typedef struct salsa_plan salsa_plan;

int salsa_abi_version(void);
/* "" for the product's build (salsa_amd/_lib.py build_command(): no -D at all).  Otherwise what this library was built with: " PROBE"
 * when -DSALSA_PROBE_BUILD admitted the timing-probe switches of salsa_amd/csrc (most of which compute wrong results on purpose), then
 * " NAME=value" for every overridden tunable -- so that an A/B or probe library loaded in place of the product identifies itself. */
const char *salsa_build_flags(void);
const char *salsa_last_error(void);

/* lower_bin, upper_bin (exclusive) and the lite spectrogram cutoff bin, bit-exact integer arithmetic of the reference */
int salsa_bin_limits(int fs, int n_fft, int fmin_doa, int fmax_doa, int *lower_bin, int *upper_bin, int *cutoff_bin);
/* host: the (freq_dim x n_fft/2+1) float32 row-major compression matrix W (what the kernels apply implicitly) */
int salsa_compress_matrix(int n_fft, int is_compress_high_freq, float *W_host);

int salsa_plan_create(const salsa_params *params, salsa_plan **out_plan);
int salsa_plan_destroy(salsa_plan *plan);
/* feature array shape for clips of n_samples: C = 7, T = 1 + n_samples / hop_len, F = 200 | 100 | n_fft/2 | cutoff-lower */
int salsa_output_shape(const salsa_plan *plan, int64_t n_samples, int *C, int64_t *T, int *F);
/* scratch bytes salsa_extract_batch needs for (batch, n_samples) */
size_t salsa_workspace_bytes(const salsa_plan *plan, int batch, int64_t n_samples);

/* d_audio: float32 [B][4][N] (planar) or [B][N][4] (interleaved) ; d_out: float32 [B][7][T][F], fully written. */
int salsa_extract_batch(salsa_plan *plan, const float *d_audio, int batch, int64_t n_samples, float *d_out,
                        void *d_workspace, size_t workspace_bytes, void *hip_stream);

/* MagStftExtractor.extract: d_audio float32 [B][C][N] planar (any C that is a multiple of 2) -> d_out [B][C][T][F] */
int salsa_logspec_batch(salsa_plan *plan, const float *d_audio, int batch, int n_channels, int64_t n_samples,
                        float *d_out, void *hip_stream);

/* extract_normalized_eigenvector: d_X complex64 [B][n_bins][n_frames][4] (the reference's (n_bins,n_frames,n_chans)),
 * d_out float64 [B][3][n_bins][n_frames].  Uses plan's cond_num / n_hopframes / is_tracking / audio_format / fs /
 * n_fft; lower_bin as given (it only enters the MIC normalisation).  d_gate (optional, may be NULL): uint8
 * [B][n_bins][n_frames], 0 = rejected by the noise gate, 1 = failed the coherence test, 2 = emitted. */
size_t salsa_eigvec_workspace_bytes(const salsa_plan *plan, int batch, int n_bins, int64_t n_frames);
int salsa_eigvec_batch(salsa_plan *plan, const float *d_X, int batch, int n_bins, int64_t n_frames, int lower_bin,
                       double *d_out, unsigned char *d_gate, void *d_workspace, size_t workspace_bytes,
                       void *hip_stream);

/* The same stage through the PRODUCTION feature kernel (the instantiation salsa_extract_batch launches: packed-float32 pair solve
 * + float64 cold list, float32 output, or the float64 one under SALSA_FLAG_FORCE_F64): d_X as above, d_feat float32
 * [B][7][n_frames][n_bins] of which planes 4-6 are written (time-major, like salsa_extract_batch's output with F = n_bins; planes
 * 0-3 are not touched).  Exists so that the reference's extract_normalized_eigenvector goldens -- rank-1 windows, eigenvalue
 * ratios straddling cond_num, u[0] ~ 0, silence -- reach the solver that ships, not only salsa_eigvec_batch's float64 one. */
int salsa_eigvec_feature_batch(salsa_plan *plan, const float *d_X, int batch, int n_bins, int64_t n_frames, int lower_bin,
                               float *d_feat, void *d_workspace, size_t workspace_bytes, void *hip_stream);

/* Schedule of salsa_extract_batch for the dataset scripts' main configuration (full SALSA, n_fft 512, n_hopframes 3, tracking on,
 * cond_num > 1; every other plan ignores the switch).  0: STFT -> tracker -> covariance / eigen, connected by the spectra spilled
 * to the workspace.  1 (round-5 stage a): STFT -> tracker as before, then ONE fused kernel that recomputes the STFT of a segment
 * of frames into a ring in LDS, rewrites the log-spectrogram and solves straight from the ring -- the spill is still written (the
 * tracker reads it) but no longer read back; it exists to measure the fused kernel against the two it replaces (measured:
 * slower, DESIGN.md section 6 -- the default stays 0).  2 (verification): as 1 with the deferred float64 records switched off,
 * i.e. every frame the packed solve hands back takes the fused kernel's in-step fallback.  Results are bit-identical in all three. */
int salsa_plan_set_fused(salsa_plan *plan, int mode);

/* Optional solver statistics (verification / study): d_counters = 4 device uint64, ADDED to by every covariance / eigen launch of
 * the plan: [0] work-list items (frame pairs), [1] gated frames in them, [2] frames handed to the float64 cold list (the packed
 * solve's `unsure` + small-pivot fallbacks), [3] tiles.  NULL (the default) detaches; the kernels then touch nothing. */
int salsa_plan_set_stats(salsa_plan *plan, unsigned long long *d_counters);

/* compute_scaler (salsa_feature_extraction.py:204-262) on device: accumulate float64 sum / sum-of-squares over time of
 * the first n_scaler_channels channels per frequency into d_sums [2][n_scaler_channels][n_freq] (zeroed by the caller
 * once; mean = sum/n, std = sqrt(sumsq/n - mean^2), population variance like sklearn's StandardScaler).  n_freq <= 256. */
int salsa_scaler_accumulate(const float *d_feat, int batch, int n_channels, int64_t n_frames, int n_freq,
                            int n_scaler_channels, double *d_sums, void *hip_stream);
/* normalise-on-load (dataset/database.py:197-202): d_feat[:, :n_scaler_channels] = (x - mean) / std in place;
 * d_mean / d_std float32 [n_scaler_channels][n_freq] (the scaler file's (4,1,F) arrays). */
int salsa_normalize_batch(float *d_feat, int batch, int n_channels, int64_t n_frames, int n_freq, int n_scaler_channels,
                          const float *d_mean, const float *d_std, void *hip_stream);

/* Fuse normalise-on-load into the extraction: with a scaler attached, salsa_extract_batch writes (x - mean) / std for the
 * 4 spectrogram channels (same float32 arithmetic as salsa_normalize_batch), saving the separate pass over the features.
 * d_mean / d_std: device float32 [4][F], caller-owned, must outlive the calls; (NULL, NULL) detaches. */
int salsa_plan_set_scaler(salsa_plan *plan, const float *d_mean, const float *d_std);

/* Feature rows [n_rows][n_frames][n_freq] float32 (time-major, what salsa_extract_batch writes; n_rows = batch * 7) ->
 * [n_rows][n_freq][n_frames] float64, the freq-major float64 array contrib/salsa_flexible.py returns (:264). */
int salsa_to_freq_major(const float *d_feat, int64_t n_rows, int64_t n_frames, int n_freq, double *d_out, void *hip_stream);

/* Training augmentation of a feature batch in one pass (utilities/transforms.py via dataset/datamodule.py:45-52, :73-82):
 * channel swap (TfmapRandomSwapChannelFoa :365-437 / ...Mic :440-523), RandomShiftUpDownNp (:286-320), then the
 * CompositeCutout rectangles (:58-283).  d_out: float32 [B][7][T][F]; d_in: the same block, possibly a view with its own
 * batch / channel strides in elements (e.g. extractor output cropped in time; rows stay contiguous).  The random draws are the
 * caller's: d_params int32 [B][40] = m0..m3 (swap bits; MIC uses three), shift (0 = none), up, 0, 0, top[8], h[8], left[8],
 * w[8] (h or w = 0: no rectangle; later rectangles win); d_uval float32 [B][8] in [0,1) and d_minmax float32 [B][2] give the
 * fill value min + (max - min) * u of the first 7 - n_zero_channels rows (the last n_zero_channels rows get 0). */
#define SALSA_AUGMENT_NPAR 40
int salsa_augment_batch(const float *d_in, int64_t in_batch_stride, int64_t in_channel_stride, float *d_out, int batch,
                        int64_t n_frames, int n_freq, int audio_format, int n_zero_channels, const int *d_params,
                        const float *d_uval, const float *d_minmax, void *hip_stream);

/* Per-kernel timing of salsa_extract_batch with HIP events recorded on the call's stream (for roofline reporting).
 * enable == 1 brackets each launch with an event pair.  enable = K > 1 launches every kernel of the call K times back to
 * back between ONE event pair (each kernel is idempotent on the audio / spill / mask buffers, so the results are those of
 * a plain call) and reports elapsed / K: the per-launch average without an event between launches, which is what a
 * rocprofv3 kernel trace of the plain call shows (an event pair around a single launch inflates it by ~12 %).
 * enable = -1 / -2 (measurement only, no events): salsa_extract_batch issues only a PREFIX of the path -- the STFT launch
 * alone / STFT + tracker -- on the buffers a full call left behind, so a caller can time prefixes of the real launch
 * sequence with its own clock and attribute the step to its kernels by differences that add up to the step exactly.
 * Such a call returns SALSA_PARTIAL (1), not SALSA_OK: its outputs are incomplete and no caller can mistake it for a plain
 * call.  The mode persists on the plan until salsa_plan_set_timing(plan, >= 0).
 * salsa_plan_read_timing synchronises on the events and returns the milliseconds per launch of the last call's kernels
 * in issue order (n_out <= SALSA_MAX_KERNELS) and their names. */
#define SALSA_MAX_KERNELS 32
int salsa_plan_set_timing(salsa_plan *plan, int enable);
int salsa_plan_read_timing(salsa_plan *plan, float *ms, const char **names, int *n_out);

/* Pipelined schedule of salsa_extract_batch (default: n_groups 1, flags 0 = three kernels in order on the caller's stream).
 * n_groups > 1 (at most 16, capped at the batch size): the batch is cut into clip ranges whose kernels run on plan-owned
 * streams forked from / joined to the caller's stream, so the latency-bound noise-floor tracker of one group overlaps the
 * STFT / eigen kernels of the others.  SALSA_PIPE_SPLIT_PAIRS: the STFT of a group is two launches (channels 0/1, then 2/3)
 * and the tracker, which only needs channel 0, starts after the first.  SALSA_PIPE_GRAPH: the fork/join is captured once per
 * (buffers, sizes) into a plan-owned hipGraph and every later call with the same arguments is ONE hipGraphLaunch on the
 * caller's stream (a call made while the caller's stream is itself being captured issues the fork/join eagerly, so it
 * becomes part of the caller's graph).  Results are bit-identical under every schedule.  salsa_plan_set_groups(n) keeps
 * the current flags. */
enum { SALSA_PIPE_SPLIT_PAIRS = 1, SALSA_PIPE_GRAPH = 2 };
int salsa_plan_set_pipeline(salsa_plan *plan, int n_groups, int flags);
int salsa_plan_set_groups(salsa_plan *plan, int n_groups);

/* contrib/salsa_flexible.py accepts any number of microphones (stacked_covmat_eigh :52-77).  Up to 4 go through
 * salsa_extract_batch (fewer than 4: pad with silent channels).  5 - SALSA_MAX_MICS go through this entry point: plan created
 * with SALSA_FLAG_FLEX (SALSA or SALSA-Lite, MIC), d_audio float32 planar [B][n_channels][N] with n_channels EVEN, 6 .. 16 (an
 * odd count: append one silent channel -- the covariance only gains a zero eigenvalue -- and drop its output planes),
 * d_out float32 [B][2*n_channels - 1][T][F]: n_channels log-spectrograms, then n_channels - 1 spatial planes.  The
 * N x N eigenproblem is solved by cyclic complex Jacobi in float64, one lane per gated TF bin (6 and 8 channels: fully
 * unrolled instantiations; 10 - 16: one instantiation with the channel count read at run time). */
#define SALSA_MAX_MICS 16
size_t salsa_multichannel_workspace_bytes(const salsa_plan *plan, int n_channels, int batch, int64_t n_samples);
int salsa_extract_multichannel(salsa_plan *plan, const float *d_audio, int n_channels, int batch, int64_t n_samples,
                               float *d_out, void *d_workspace, size_t workspace_bytes, void *hip_stream);

/* Diagnostic: the kernels' float32 dB conversion 10*log10(max(1e-10, p)) (librosa.power_to_db(ref=1, amin=1e-10, top_db=None),
 * salsa_feature_extraction.py:194-195) applied elementwise to d_power[n] -> d_db[n].  It is the SAME device function the STFT
 * kernel inlines (hardware v_log_f32 times a constant); exported so a test can bound its error over the whole float32
 * exponent range instead of over whatever dynamic range a test clip happens to have. */
int salsa_selftest_decibel(const float *d_power, float *d_db, int64_t n, void *hip_stream);

/* The resampling step of the reference's loader: librosa.load(path, sr=fs, mono=False, dtype=float32)
 * (dataset/salsa_feature_extraction.py:353, salsa_lite_feature_extraction.py:93) on a file whose native rate is not fs ->
 * librosa 0.8.0 core/audio.py::resample(res_type='kaiser_best', fix=True) -> resampy 0.2.2 (requirements.yml:181)
 * resample / interpn.py::resample_f.  d_x: float32 [n_rows][n_in] (one row per channel of a clip); d_y: float32
 * [n_rows][n_out_fixed]: n_out = int(n_in * sample_ratio) samples computed (each tap float32(float64(y) + weight * x), left
 * wing then right wing: the sequential reference loop bit for bit), then zeros up to n_out_fixed = ceil(n_in * sample_ratio)
 * (librosa's fix_length).  d_interp_win / d_interp_delta: float64 [n_win], the filter half-window (scaled by sample_ratio when
 * < 1) and its first differences; num_table: table entries per zero crossing (512 for kaiser_best); d_time_register: float64
 * [n_out], the reference's sequentially accumulated read positions (0, 1/ratio, 1/ratio + 1/ratio, ...).  salsa_amd/resample.py
 * builds all three. */
int salsa_resample_batch(const float *d_x, int n_rows, int64_t n_in, float *d_y, int64_t n_out, int64_t n_out_fixed, double sample_ratio,
                         const double *d_interp_win, const double *d_interp_delta, int n_win, int num_table,
                         const double *d_time_register, void *hip_stream);

/* The sample conversion inside librosa.load(path, sr=fs, mono=False, dtype=float32) (dataset/salsa_feature_extraction.py:353, lite :93:
 * soundfile reads the WAV's interleaved PCM frames as float32, librosa transposes to (channels, samples)).  d_pcm: the file's data chunk as
 * it is on disk, [n_frames][n_channels] samples of `sample_format`, aligned to one frame; d_out: float32 [n_channels][n_frames] (what
 * salsa_extract_batch takes as one planar clip).  int16 / 2^15, int32 / 2^31, (uint8 - 128) / 2^7, float32 as is: libsndfile's normalisation,
 * exact in float32.  Lets a loader upload raw file bytes (half the PCIe traffic for 16-bit clips) and do no arithmetic on the host. */
#define SALSA_PCM_S16 1
#define SALSA_PCM_S32 2
#define SALSA_PCM_U8 3
#define SALSA_PCM_F32 4
int salsa_pcm_to_planar(const void *d_pcm, int sample_format, int n_channels, int64_t n_frames, float *d_out, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SALSA_HIP_H */
