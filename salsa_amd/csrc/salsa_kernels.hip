// salsa_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the SALSA / SALSA-Lite feature extractor + the C ABI
// declared in include/salsa_hip.h.  Written for wave64 / 160 KiB LDS / HBM3E; no CUDA or multi-backend paths.
//
// Pipeline of salsa_extract_batch (full SALSA), all on the caller's stream:
//   K1 stft_kernel      one wave per packed 512-point complex FFT (two real channels), Stockham radix-8 through LDS;
//                       unpacks to the 4 channel spectra, writes the log-spectrogram channels 0-3 straight to the
//                       output and spills the DOA band of the spectra (float32-rounded, like the reference's
//                       complex64 STFT) to the workspace as Xs[b][t][c][bin].
//   K2 tracker_kernel   one lane per (clip, bin): 3-frame RMS of channel 0 and the sequential noise-floor tracker
//                       in float64 -> valid[b][t][bin].
//   K3 cov_eig_kernel   one lane per TF bin (64 consecutive bins of one frame per wave, coalesced): 7-frame Hermitian
//                       covariance accumulated in registers, eigen-gate + principal eigenvector (salsa_math.h),
//                       FOA / MIC normalisation, writes channels 4-6 (zeros where gated).
// SALSA-Lite / IPD is K1 alone (log-spectrogram + inter-channel phase fused into the unpack).
//
// Arithmetic types follow the reference (see DESIGN.md "Precision"): STFT evaluated in float64 and rounded to
// float32, log-spectrogram in float32, tracker / covariance / eigen-solve in float64.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/salsa_hip.h"
#include "salsa_math.h"

using salsa::cplx;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *a = "", long b = 0)
{
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            snprintf(g_err, sizeof(g_err), "%s failed: %s", #expr, hipGetErrorString(e_));        \
            return SALSA_EHIP;                                                                    \
        }                                                                                         \
    } while (0)

struct KParams {
    int B;
    long N;   // samples per channel
    long T;   // frames
    int hop;
    int lower, upper, nd; // DOA band [lower, upper), nd = upper - lower
    int cutoff;           // lite: spectrogram band [lower, cutoff)
    int F;                // feature bins per frame
    int OC;               // output channels (7, or 4 for logspec-only)
    int ident;            // identity rows of W (192 | 96 | n_fft/2)
    int compress;
    int layout;
    int feature;          // SALSA_FEATURE_* ; 3 = logspec only
    int format;
    int tracking;
    int n_hop;
    double cond;
    double delta;         // 2 pi fs / (n_fft * 343)
};

constexpr int FEATURE_LOGSPEC_ONLY = 3;

// ------------------------------------------------------------------------------------------------------------ K1
template <int N> struct fft_cfg {
    static constexpr int R = (N == 512) ? 8 : 4; // points per lane; 64 lanes per transform either way
    static constexpr int PADN = N + N / 8;       // one pad element every 8: stride-R scatter of 16-B elements is conflict-free
};

__device__ __forceinline__ int padi(int i) { return i + (i >> 3); }

constexpr int K1_FRAMES_PER_ITER = 2;
constexpr int K1_ITERS = 4;
constexpr int K1_FRAMES_PER_BLOCK = K1_FRAMES_PER_ITER * K1_ITERS;

template <int N, typename T>
__global__ __launch_bounds__(256) void stft_kernel(const KParams kp, const float *__restrict__ audio,
                                                   const double *__restrict__ window,
                                                   const cplx<double> *__restrict__ tw, float *__restrict__ out,
                                                   float2 *__restrict__ Xs)
{
    constexpr int R = fft_cfg<N>::R;
    constexpr int PADN = fft_cfg<N>::PADN;
    constexpr int NB = N / 2 + 1;
    __shared__ cplx<T> buf[4][PADN];
    __shared__ float pw[K1_FRAMES_PER_ITER][4][NB];

    const int tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63;
    const int fr = w >> 1, pair = w & 1;
    const int b = blockIdx.y;
    const long t0 = (long)blockIdx.x * K1_FRAMES_PER_BLOCK;
    const long Ns = kp.N, Tn = kp.T;
    cplx<T> *mybuf = buf[w];

    // window and twiddles of this lane's elements (same for every frame)
    T win[R];
#pragma unroll
    for (int r = 0; r < R; r++) win[r] = (T)window[salsa::stockham_in(lane, r, N, R)];

    for (int it = 0; it < K1_ITERS; it++) {
        const long t = t0 + it * K1_FRAMES_PER_ITER + fr;
        const bool live = t < Tn;
        cplx<T> v[R];
        // ---- load + window: z[n] = w[n] * (y_c0[n] + i y_c1[n]) for this wave's channel pair
        {
            const long base = t * kp.hop - N / 2;
            const bool interior = live && base >= 0 && base + N <= Ns;
            const int c0 = 2 * pair;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int n = salsa::stockham_in(lane, r, N, R);
                float y0 = 0.f, y1 = 0.f;
                if (live) {
                    const long s = interior ? base + n : salsa::reflect_index(base + n, Ns);
                    if (kp.layout == SALSA_LAYOUT_PLANAR) {
                        y0 = audio[((long)b * 4 + c0) * Ns + s];
                        y1 = audio[((long)b * 4 + c0 + 1) * Ns + s];
                    } else {
                        const float2 yy = *reinterpret_cast<const float2 *>(audio + ((long)b * Ns + s) * 4 + c0);
                        y0 = yy.x;
                        y1 = yy.y;
                    }
                }
                v[r] = {win[r] * (T)y0, win[r] * (T)y1};
            }
        }
        // ---- Stockham passes, in place in this wave's LDS buffer
        salsa::dftR<R>(v); // pass p = 1 (no twiddles)
#pragma unroll
        for (int r = 0; r < R; r++) mybuf[padi(salsa::stockham_out(lane, r, 1, R))] = v[r];
#pragma unroll
        for (int p = R; p < N; p *= R) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; r++) v[r] = mybuf[padi(salsa::stockham_in(lane, r, N, R))];
            __syncthreads();
#pragma unroll
            for (int r = 1; r < R; r++) {
                const cplx<double> wd = tw[salsa::stockham_tw(lane, r, p, N, R)];
                v[r] = salsa::cmul(v[r], cplx<T>{(T)wd.re, (T)wd.im});
            }
            salsa::dftR<R>(v);
#pragma unroll
            for (int r = 0; r < R; r++) mybuf[padi(salsa::stockham_out(lane, r, p, R))] = v[r];
        }
        __syncthreads();

        // ---- unpack the two packed transforms of this frame into 4 channel spectra; 128 lanes per frame
        const cplx<T> *za = buf[fr * 2 + 0], *zb = buf[fr * 2 + 1];
        const int j = pair * 64 + lane;
        float *o = out + (long)b * kp.OC * Tn * kp.F; // [OC][T][F] of this clip
        for (int k = j; k <= N / 2; k += 128) {
            const int km = (N - k) & (N - 1);
            cplx<T> X[4];
            salsa::unpack_pair(za[padi(k)], za[padi(km)], X[0], X[1]);
            salsa::unpack_pair(zb[padi(k)], zb[padi(km)], X[2], X[3]);
            float2 xf[4];
            float pc[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                xf[c] = make_float2((float)X[c].re, (float)X[c].im); // the reference stores its STFT as complex64
                pc[c] = xf[c].x * xf[c].x + xf[c].y * xf[c].y;
            }
            if (!live) continue;
            if (kp.feature == SALSA_FEATURE_SALSA || kp.feature == FEATURE_LOGSPEC_ONLY) {
                if (kp.feature == SALSA_FEATURE_SALSA && k >= kp.lower && k < kp.upper) {
#pragma unroll
                    for (int c = 0; c < 4; c++) Xs[(((long)b * Tn + t) * 4 + c) * kp.nd + (k - kp.lower)] = xf[c];
                }
                if (k >= 1 && k <= kp.ident) {
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        o[((long)c * Tn + t) * kp.F + (k - 1)] = 10.0f * log10f(fmaxf(1e-10f, pc[c]));
                } else if (k > kp.ident) {
#pragma unroll
                    for (int c = 0; c < 4; c++) pw[fr][c][k] = pc[c];
                }
            } else { // SALSA-Lite / SALSA-IPD (salsa_lite_feature_extraction.py:103-120)
                if (k >= kp.lower && k < kp.cutoff) {
                    const int f = k - kp.lower;
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        o[((long)c * Tn + t) * kp.F + f] = 10.0f * log10f(fmaxf(1e-10f, pc[c]));
                    const double x0r = xf[0].x, x0i = xf[0].y;
                    const double scale = kp.feature == SALSA_FEATURE_IPD ? 3.14159265358979323846
                                                                         : kp.delta * (double)(k == 0 ? 1 : k);
#pragma unroll
                    for (int c = 1; c < 4; c++) {
                        float ph = 0.f;
                        if (f < kp.upper) { // ":120 phase_vector[:, :, upper_bin:] = 0" indexes the CROPPED axis
                            double wr = (double)xf[c].x * x0r + (double)xf[c].y * x0i;
                            double wi = (double)xf[c].y * x0r - (double)xf[c].x * x0i;
                            const double m = fmax(fabs(wr), fabs(wi));
                            if (m < 1e-30 && m > 0.0) { wr *= 0x1p200; wi *= 0x1p200; } // keep the float cast normal
                            ph = (float)((double)atan2f((float)wi, (float)wr) / scale);
                        }
                        o[((long)(3 + c) * Tn + t) * kp.F + f] = ph;
                    }
                }
            }
        }
        __syncthreads();
        // ---- compressed high-frequency rows of W: mean-like sum of 8 (last row 7) bins times 1/8
        if (kp.feature != SALSA_FEATURE_LITE && kp.feature != SALSA_FEATURE_IPD && kp.compress && live) {
            const int c = j & 3, gi = j >> 2;
            const int ng = kp.F - kp.ident;
            if (gi < ng) {
                const int start = kp.ident + 1 + 8 * gi;
                const int cnt = gi < ng - 1 ? 8 : 7;
                float acc = 0.f;
                for (int q = 0; q < cnt; q++) acc += 0.125f * pw[fr][c][start + q];
                o[((long)c * Tn + t) * kp.F + kp.ident + gi] = 10.0f * log10f(fmaxf(1e-10f, acc));
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------ K2
// One lane per (clip, bin); sequential over time (the tracker state depends on every earlier frame).
constexpr int TR_CHUNK = 32;

__global__ __launch_bounds__(64) void tracker_kernel(const KParams kp, const float2 *__restrict__ Xs,
                                                     unsigned char *__restrict__ valid)
{
    const long idx = (long)blockIdx.x * 64 + threadIdx.x;
    if (idx >= (long)kp.B * kp.nd) return;
    const int b = (int)(idx / kp.nd), bin = (int)(idx % kp.nd);
    const long Tn = kp.T;
    const float2 *x0 = Xs + ((long)b * Tn * 4) * kp.nd + bin; // channel 0 of frame t at x0[t*4*nd]
    const long stride = 4L * kp.nd;
    auto P = [&](long t) {
        t %= Tn;
        if (t < 0) t += Tn;
        const float2 x = x0[t * stride];
        const double re = x.x, im = x.y;
        return re * re + im * im;
    };
    // noise_floor = 0.5 * mean(mag[0:5])  (:58)
    const long n0 = Tn < 5 ? Tn : 5;
    double acc = 0.0;
    for (long t = 0; t < n0; t++) acc += sqrt((((0.0 + P(t)) + P(t - 1)) + P(t - 2)) / 3);
    salsa::tracker_state st = {0.5 * (acc / (double)n0), 3};
    double p1 = P(-1), p2 = P(-2);
    unsigned char *vout = valid + (long)b * Tn * kp.nd + bin;
    // The recurrence is sequential but its inputs are not: keep TR_CHUNK frames of channel 0 in registers and fetch
    // the next chunk while stepping through the current one, so the ~2 us strided-load latency is paid once per
    // chunk instead of once per frame.
    float2 cur[TR_CHUNK], nxt[TR_CHUNK];
#pragma unroll
    for (int i = 0; i < TR_CHUNK; i++) cur[i] = x0[(i < Tn ? i : Tn - 1) * stride];
    for (long c0 = 0; c0 < Tn; c0 += TR_CHUNK) {
#pragma unroll
        for (int i = 0; i < TR_CHUNK; i++) {
            const long tn = c0 + TR_CHUNK + i;
            nxt[i] = x0[(tn < Tn ? tn : Tn - 1) * stride];
        }
#pragma unroll
        for (int i = 0; i < TR_CHUNK; i++) {
            const long t = c0 + i;
            if (t < Tn) {
                const double re = cur[i].x, im = cur[i].y;
                const double p0 = re * re + im * im;
                const double mag = sqrt((((0.0 + p0) + p1) + p2) / 3);
                vout[t * kp.nd] = salsa::tracker_step(st, mag) ? 1 : 0;
                p2 = p1;
                p1 = p0;
            }
        }
#pragma unroll
        for (int i = 0; i < TR_CHUNK; i++) cur[i] = nxt[i];
    }
}

// ------------------------------------------------------------------------------------------------------------ K3
// One block per (clip, frame); lane = bin.  FEAT: write float32 channels 4-6 of the feature array (zeros above the
// DOA band up to F); otherwise write the float64 (3, n_bins, n_frames) array of extract_normalized_eigenvector.
template <bool FEAT>
__global__ __launch_bounds__(256) void cov_eig_kernel(const KParams kp, const float2 *__restrict__ Xs,
                                                      const unsigned char *__restrict__ valid,
                                                      float *__restrict__ out_feat, double *__restrict__ out_eig,
                                                      unsigned char *__restrict__ gate)
{
    const long bt = blockIdx.x;
    const int bin = blockIdx.y * 256 + threadIdx.x;
    const long Tn = kp.T;
    const int b = (int)(bt / Tn);
    const long t = bt % Tn;
    const int limit = FEAT ? kp.F : kp.nd;
    if (bin >= limit) return;
    double e[3] = {0.0, 0.0, 0.0};
    unsigned char g = 0;
    if (bin < kp.nd) {
        const bool sig = kp.tracking ? (valid[bt * kp.nd + bin] != 0) : true;
        if (sig) {
            salsa::herm4<double> R = {};
            for (int k = -kp.n_hop; k <= kp.n_hop; k++) {
                long tt = (t + k) % Tn; // np.pad(..., 'wrap') on the time axis (:43)
                if (tt < 0) tt += Tn;
                const float2 *xp = Xs + (((long)b * Tn + tt) * 4) * kp.nd + bin;
                cplx<double> x[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float2 xv = xp[(long)c * kp.nd];
                    x[c] = {(double)xv.x, (double)xv.y};
                }
                salsa::herm4_rank1_add(R, x);
            }
            const salsa::eig_result<double> er = salsa::herm4_gate_eigvec(R, kp.cond, !kp.tracking);
            g = er.rank1 ? 2 : 1;
            if (er.rank1 || !kp.tracking) { // :111-112 the coherence test only gates when tracking
                if (kp.format == SALSA_FORMAT_FOA) salsa::normalise_foa(er.u, e);
                else salsa::normalise_mic(er.u, kp.delta * (double)(bin + kp.lower), e);
                if (!kp.tracking) g = 2;
            }
        }
    }
    if (FEAT) {
#pragma unroll
        for (int i = 0; i < 3; i++) out_feat[(((long)b * kp.OC + 4 + i) * Tn + t) * kp.F + bin] = (float)e[i];
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) out_eig[(((long)b * 3 + i) * kp.nd + bin) * Tn + t] = e[i];
        if (gate) gate[((long)b * kp.nd + bin) * Tn + t] = g;
    }
}

// reference layout (n_bins, n_frames, 4) complex64 -> internal Xs[b][t][c][bin]
__global__ void relayout_kernel(const float2 *__restrict__ X, float2 *__restrict__ Xs, int B, int nb, long Tn)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * nb * Tn * 4;
    if (idx >= total) return;
    const int bin = (int)(idx % nb);
    long r = idx / nb;
    const int c = (int)(r % 4);
    r /= 4;
    const long t = r % Tn;
    const int b = (int)(r / Tn);
    Xs[idx] = X[(((long)b * nb + bin) * Tn + t) * 4 + c];
}

} // namespace

// ================================================================================================== plan + C ABI
struct salsa_plan {
    salsa_params p;
    int device;
    int lower, upper, cutoff, nd, F, ident;
    double delta;
    double *d_window;
    cplx<double> *d_tw;
    int timing;
    int n_kernels;
    hipEvent_t ev[SALSA_MAX_KERNELS + 1];
    const char *names[SALSA_MAX_KERNELS];
};

extern "C" {

int salsa_abi_version(void) { return SALSA_ABI_VERSION; }
const char *salsa_last_error(void) { return g_err; }

int salsa_bin_limits(int fs, int n_fft, int fmin_doa, int fmax_doa, int *lower_bin, int *upper_bin, int *cutoff_bin)
{
    if (fs <= 0 || n_fft <= 0 || !lower_bin || !upper_bin) return fail(SALSA_EINVAL, "salsa_bin_limits: bad argument%s");
    // salsa_feature_extraction.py:298-304: fmax = min(fmax, fs//2); int(floor(f * n_fft / float(fs))); lower = max(1, lower)
    const int fmax = fmax_doa < fs / 2 ? fmax_doa : fs / 2;
    int lo = (int)floor((double)((int64_t)fmin_doa * n_fft) / (double)fs);
    const int up = (int)floor((double)((int64_t)fmax * n_fft) / (double)fs);
    if (lo < 1) lo = 1;
    *lower_bin = lo;
    *upper_bin = up;
    if (cutoff_bin) *cutoff_bin = (int)floor((double)((int64_t)9000 * n_fft) / (double)fs); // lite :57-58
    return SALSA_OK;
}

static int freq_dim(int n_fft, int compress)
{
    if (n_fft != 512 && n_fft != 256) return -1;
    if (compress) return n_fft == 512 ? 200 : 100;
    return n_fft / 2;
}

int salsa_compress_matrix(int n_fft, int compress, float *W)
{
    const int F = freq_dim(n_fft, compress);
    if (F < 0) return fail(SALSA_ENFFT, "nfft is not 512 or 256%s");
    if (!W) return fail(SALSA_EINVAL, "salsa_compress_matrix: NULL output%s");
    const int nb = n_fft / 2 + 1;
    memset(W, 0, sizeof(float) * (size_t)F * nb);
    const int ident = compress ? (n_fft == 512 ? 192 : 96) : n_fft / 2;
    for (int i = 0; i < ident; i++) W[(size_t)i * nb + i + 1] = 1.0f;
    for (int i = ident; i < F; i++) {
        const int cnt = i < F - 1 ? 8 : 7;
        for (int k = 0; k < cnt; k++) W[(size_t)i * nb + ident + 1 + (i - ident) * 8 + k] = 0.125f;
    }
    return SALSA_OK;
}

int salsa_plan_create(const salsa_params *params, salsa_plan **out_plan)
{
    if (!params || !out_plan) return fail(SALSA_EINVAL, "salsa_plan_create: NULL argument%s");
    const salsa_params &p = *params;
    if (p.n_fft != 512 && p.n_fft != 256) return fail(SALSA_ENFFT, "only 256 or 512 fft is supported%s");
    if (p.fs <= 0 || p.hop_len <= 0 || p.win_len <= 0 || p.win_len > p.n_fft)
        return fail(SALSA_EINVAL, "bad fs / hop_len / win_len (window length must be <= nfft)%s");
    if (p.audio_format != SALSA_FORMAT_FOA && p.audio_format != SALSA_FORMAT_MIC)
        return fail(SALSA_EFORMAT, "Unknown audio format%s");
    if (p.feature_type < SALSA_FEATURE_SALSA || p.feature_type > SALSA_FEATURE_IPD)
        return fail(SALSA_EINVAL, "Invalid feature type%s");
    if (p.feature_type != SALSA_FEATURE_SALSA && p.audio_format != SALSA_FORMAT_MIC)
        return fail(SALSA_EFORMAT, "SALSA-Lite and SALSA-IPD are only for MIC format!%s");
    if (p.n_hopframes < 0 || p.n_hopframes > 16) return fail(SALSA_EINVAL, "n_hopframes out of range%s");
    salsa_plan *pl = new salsa_plan();
    memset(pl, 0, sizeof(*pl));
    pl->p = p;
    salsa_bin_limits(p.fs, p.n_fft, p.fmin_doa, p.fmax_doa, &pl->lower, &pl->upper, &pl->cutoff);
    const int nbins = p.n_fft / 2 + 1;
    if (p.feature_type == SALSA_FEATURE_SALSA) {
        pl->F = freq_dim(p.n_fft, p.is_compress_high_freq);
        pl->ident = p.is_compress_high_freq ? (p.n_fft == 512 ? 192 : 96) : p.n_fft / 2;
        pl->nd = pl->upper - pl->lower;
        if (pl->nd < 0 || pl->nd > pl->F || pl->upper > nbins) {
            delete pl;
            return fail(SALSA_EBINS, "DOA band [lower_bin, upper_bin) does not fit the feature axis%s");
        }
    } else {
        if (pl->upper > pl->cutoff) {
            delete pl;
            return fail(SALSA_EBINS, "Upper bin for spatial feature is higher than cutoff bin for spectrogram!%s");
        }
        if (pl->cutoff > nbins) pl->cutoff = nbins; // numpy slicing clips [lower:cutoff] at n_bins
        pl->F = pl->cutoff - pl->lower;
        pl->ident = 0;
        pl->nd = 0;
        if (pl->F <= 0) {
            delete pl;
            return fail(SALSA_EBINS, "empty spectrogram band%s");
        }
    }
    pl->delta = 2.0 * 3.14159265358979323846 * p.fs / (p.n_fft * 343.0);
    if (hipGetDevice(&pl->device) != hipSuccess) {
        delete pl;
        return fail(SALSA_EHIP, "hipGetDevice failed (no HIP device?)%s");
    }
    // window: scipy.signal.get_window('hann', win, fftbins=True), centre-padded to n_fft ; twiddles W_N^m
    double *hw = new double[p.n_fft];
    cplx<double> *htw = new cplx<double>[p.n_fft];
    const int lpad = (p.n_fft - p.win_len) / 2;
    for (int i = 0; i < p.n_fft; i++) hw[i] = 0.0;
    for (int n = 0; n < p.win_len; n++) hw[lpad + n] = 0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * n / p.win_len);
    for (int m = 0; m < p.n_fft; m++)
        htw[m] = {cos(-2.0 * 3.14159265358979323846 * m / p.n_fft), sin(-2.0 * 3.14159265358979323846 * m / p.n_fft)};
    hipError_t e1 = hipMalloc((void **)&pl->d_window, sizeof(double) * p.n_fft);
    hipError_t e2 = hipMalloc((void **)&pl->d_tw, sizeof(cplx<double>) * p.n_fft);
    if (e1 == hipSuccess && e2 == hipSuccess) {
        e1 = hipMemcpy(pl->d_window, hw, sizeof(double) * p.n_fft, hipMemcpyHostToDevice);
        e2 = hipMemcpy(pl->d_tw, htw, sizeof(cplx<double>) * p.n_fft, hipMemcpyHostToDevice);
    }
    delete[] hw;
    delete[] htw;
    if (e1 != hipSuccess || e2 != hipSuccess) {
        salsa_plan_destroy(pl);
        return fail(SALSA_EHIP, "plan table upload failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    for (int i = 0; i <= SALSA_MAX_KERNELS; i++) pl->ev[i] = nullptr;
    *out_plan = pl;
    return SALSA_OK;
}

int salsa_plan_destroy(salsa_plan *pl)
{
    if (!pl) return SALSA_OK;
    if (pl->d_window) (void)hipFree(pl->d_window);
    if (pl->d_tw) (void)hipFree(pl->d_tw);
    for (int i = 0; i <= SALSA_MAX_KERNELS; i++)
        if (pl->ev[i]) (void)hipEventDestroy(pl->ev[i]);
    delete pl;
    return SALSA_OK;
}

int salsa_output_shape(const salsa_plan *pl, int64_t n_samples, int *C, int64_t *T, int *F)
{
    if (!pl || n_samples < 0) return fail(SALSA_EINVAL, "salsa_output_shape: bad argument%s");
    if (C) *C = 7;
    if (T) *T = 1 + n_samples / pl->p.hop_len;
    if (F) *F = pl->F;
    return SALSA_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t salsa_workspace_bytes(const salsa_plan *pl, int batch, int64_t n_samples)
{
    if (!pl || batch <= 0 || n_samples <= 0 || pl->p.feature_type != SALSA_FEATURE_SALSA) return 0;
    const size_t T = 1 + n_samples / pl->p.hop_len;
    return align256((size_t)batch * T * 4 * pl->nd * sizeof(float2)) + align256((size_t)batch * T * pl->nd) + 256;
}

size_t salsa_eigvec_workspace_bytes(const salsa_plan *pl, int batch, int n_bins, int64_t n_frames)
{
    if (!pl || batch <= 0 || n_bins <= 0 || n_frames <= 0) return 0;
    return align256((size_t)batch * n_frames * 4 * n_bins * sizeof(float2)) + align256((size_t)batch * n_frames * n_bins) + 256;
}

static KParams make_kparams(const salsa_plan *pl, int batch, int64_t n_samples)
{
    KParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.B = batch;
    kp.N = n_samples;
    kp.T = 1 + n_samples / pl->p.hop_len;
    kp.hop = pl->p.hop_len;
    kp.lower = pl->lower;
    kp.upper = pl->upper;
    kp.nd = pl->nd;
    kp.cutoff = pl->cutoff;
    kp.F = pl->F;
    kp.OC = 7;
    kp.ident = pl->ident;
    kp.compress = pl->p.is_compress_high_freq;
    kp.layout = pl->p.audio_layout;
    kp.feature = pl->p.feature_type;
    kp.format = pl->p.audio_format;
    kp.tracking = pl->p.is_tracking;
    kp.n_hop = pl->p.n_hopframes;
    kp.cond = pl->p.cond_num;
    kp.delta = pl->delta;
    return kp;
}

static void mark(salsa_plan *pl, hipStream_t s, const char *name)
{
    if (!pl->timing) return;
    if (name && pl->n_kernels < SALSA_MAX_KERNELS) pl->names[pl->n_kernels] = name;
    const int slot = name ? pl->n_kernels + 1 : 0;
    if (slot > SALSA_MAX_KERNELS) return;
    if (!pl->ev[slot]) (void)hipEventCreate(&pl->ev[slot]);
    (void)hipEventRecord(pl->ev[slot], s);
    if (name) pl->n_kernels++;
    else pl->n_kernels = 0;
}

static int launch_stft(salsa_plan *pl, const KParams &kp, const float *d_audio, float *d_out, float2 *Xs, hipStream_t s)
{
    dim3 grid((unsigned)((kp.T + K1_FRAMES_PER_BLOCK - 1) / K1_FRAMES_PER_BLOCK), (unsigned)kp.B);
    if (pl->p.n_fft == 512)
        hipLaunchKernelGGL((stft_kernel<512, double>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
    else
        hipLaunchKernelGGL((stft_kernel<256, double>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

int salsa_extract_batch(salsa_plan *pl, const float *d_audio, int batch, int64_t n_samples, float *d_out,
                        void *d_workspace, size_t workspace_bytes, void *hip_stream)
{
    if (!pl || !d_audio || !d_out || batch <= 0 || n_samples <= 0)
        return fail(SALSA_EINVAL, "salsa_extract_batch: bad argument%s");
    if (n_samples <= pl->p.n_fft / 2)
        return fail(SALSA_EINVAL, "clip shorter than n_fft/2 samples cannot be reflect-padded%s");
    hipStream_t s = (hipStream_t)hip_stream;
    KParams kp = make_kparams(pl, batch, n_samples);
    const bool full = pl->p.feature_type == SALSA_FEATURE_SALSA;
    float2 *Xs = nullptr;
    unsigned char *valid = nullptr;
    if (full) {
        const size_t need = salsa_workspace_bytes(pl, batch, n_samples);
        if (!d_workspace || workspace_bytes < need) return fail(SALSA_EWORKSPACE, "workspace too small%s (need %ld bytes)", "", (long)need);
        Xs = (float2 *)d_workspace;
        valid = (unsigned char *)d_workspace + align256((size_t)batch * kp.T * 4 * kp.nd * sizeof(float2));
    }
    mark(pl, s, nullptr);
    int rc = launch_stft(pl, kp, d_audio, d_out, Xs, s);
    if (rc) return rc;
    mark(pl, s, "stft_logspec");
    if (!full) return SALSA_OK;
    if (kp.nd > 0 && kp.tracking) {
        const long n = (long)kp.B * kp.nd;
        hipLaunchKernelGGL(tracker_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, kp, Xs, valid);
        HIP_TRY(hipGetLastError());
    }
    mark(pl, s, "noise_floor_tracker");
    {
        dim3 grid((unsigned)((long)kp.B * kp.T), (unsigned)((kp.F + 255) / 256));
        hipLaunchKernelGGL((cov_eig_kernel<true>), grid, dim3(256), 0, s, kp, Xs, valid, d_out, (double *)nullptr,
                           (unsigned char *)nullptr);
        HIP_TRY(hipGetLastError());
    }
    mark(pl, s, "cov_eig");
    return SALSA_OK;
}

int salsa_logspec_batch(salsa_plan *pl, const float *d_audio, int batch, int n_channels, int64_t n_samples,
                        float *d_out, void *hip_stream)
{
    if (!pl || !d_audio || !d_out || batch <= 0 || n_samples <= 0) return fail(SALSA_EINVAL, "salsa_logspec_batch: bad argument%s");
    if (n_channels != 4) return fail(SALSA_EINVAL, "salsa_logspec_batch: n_channels must be 4 (pad with silent channels)%s");
    if (n_samples <= pl->p.n_fft / 2) return fail(SALSA_EINVAL, "clip shorter than n_fft/2 samples cannot be reflect-padded%s");
    KParams kp = make_kparams(pl, batch, n_samples);
    kp.feature = FEATURE_LOGSPEC_ONLY;
    kp.OC = 4;
    kp.layout = SALSA_LAYOUT_PLANAR;
    kp.F = freq_dim(pl->p.n_fft, pl->p.is_compress_high_freq);
    kp.ident = pl->p.is_compress_high_freq ? (pl->p.n_fft == 512 ? 192 : 96) : pl->p.n_fft / 2;
    return launch_stft(pl, kp, d_audio, d_out, nullptr, (hipStream_t)hip_stream);
}

int salsa_eigvec_batch(salsa_plan *pl, const float *d_X, int batch, int n_bins, int64_t n_frames, int lower_bin,
                       double *d_out, unsigned char *d_gate, void *d_workspace, size_t workspace_bytes,
                       void *hip_stream)
{
    if (!pl || !d_X || !d_out || batch <= 0 || n_bins <= 0 || n_frames <= 0)
        return fail(SALSA_EINVAL, "salsa_eigvec_batch: bad argument%s");
    const size_t need = salsa_eigvec_workspace_bytes(pl, batch, n_bins, n_frames);
    if (!d_workspace || workspace_bytes < need) return fail(SALSA_EWORKSPACE, "workspace too small%s (need %ld bytes)", "", (long)need);
    hipStream_t s = (hipStream_t)hip_stream;
    KParams kp = make_kparams(pl, batch, 0);
    kp.T = n_frames;
    kp.nd = n_bins;
    kp.lower = lower_bin;
    kp.upper = lower_bin + n_bins;
    kp.F = n_bins;
    kp.feature = SALSA_FEATURE_SALSA;
    float2 *Xs = (float2 *)d_workspace;
    unsigned char *valid = (unsigned char *)d_workspace + align256((size_t)batch * n_frames * 4 * n_bins * sizeof(float2));
    const long total = (long)batch * n_bins * n_frames * 4;
    hipLaunchKernelGGL(relayout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float2 *)d_X, Xs, batch, n_bins, (long)n_frames);
    HIP_TRY(hipGetLastError());
    if (kp.tracking) {
        const long n = (long)kp.B * kp.nd;
        hipLaunchKernelGGL(tracker_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, kp, Xs, valid);
        HIP_TRY(hipGetLastError());
    }
    dim3 grid((unsigned)((long)kp.B * kp.T), (unsigned)((n_bins + 255) / 256));
    hipLaunchKernelGGL((cov_eig_kernel<false>), grid, dim3(256), 0, s, kp, Xs, valid, (float *)nullptr, d_out, d_gate);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

int salsa_plan_set_timing(salsa_plan *pl, int enable)
{
    if (!pl) return fail(SALSA_EINVAL, "salsa_plan_set_timing: NULL plan%s");
    pl->timing = enable != 0;
    pl->n_kernels = 0;
    return SALSA_OK;
}

int salsa_plan_read_timing(salsa_plan *pl, float *ms, const char **names, int *n_out)
{
    if (!pl || !ms || !n_out) return fail(SALSA_EINVAL, "salsa_plan_read_timing: NULL argument%s");
    *n_out = 0;
    if (!pl->timing || pl->n_kernels == 0) return SALSA_OK;
    HIP_TRY(hipEventSynchronize(pl->ev[pl->n_kernels]));
    for (int i = 0; i < pl->n_kernels; i++) {
        HIP_TRY(hipEventElapsedTime(&ms[i], pl->ev[i], pl->ev[i + 1]));
        if (names) names[i] = pl->names[i];
    }
    *n_out = pl->n_kernels;
    return SALSA_OK;
}

} // extern "C"
