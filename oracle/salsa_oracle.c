/*
 * oracle/salsa_oracle.c -- CPU restatement of the reference SALSA / SALSA-Lite feature path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the timed CPU baseline ("port").  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product (salsa_amd/, libsalsa_hip.so)
 * never links, imports or falls back to it.
 *
 * Parity status: PINNED against outputs of the reference itself.  The reference (pure Python, no tests of its own)
 * is imported unmodified by tools/make_golden.py in the build container and its outputs are committed under
 * tests/golden/; tests/test_oracle_golden.py checks every function below against them (float64 stages <= 1e-9,
 * float32 stages to float32 round-off, gates / indices exact).  The STFT itself is third-party arithmetic
 * (librosa==0.8.0, requirements.yml:101, absent from /root/reference): restated from its published algorithm
 * (tools/ref_shims.py docstring) and pinned by fixture g8 + torch.stft cross-check.
 *
 * Each function cites the reference lines it follows (paths relative to /root/reference/).
 * Arithmetic types follow the reference exactly: STFT evaluated in float64 and rounded to complex64; log-spectrogram
 * in float32; noise-floor tracker, covariance and eigen-decomposition in float64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static int g_threads = 1;

void salsa_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int salsa_oracle_get_threads(void) { return g_threads; }
int salsa_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------ bin limits
 * dataset/salsa_feature_extraction.py:298-304 ; salsa_lite_feature_extraction.py:50-58.
 *   fmax = min(fmax, fs // 2); lower = int(floor(fmin*n_fft/float(fs))); upper = int(floor(fmax*n_fft/float(fs)));
 *   lower = max(1, lower); cutoff (lite) = int(floor(9000*n_fft/float(fs))).
 * The products are formed in double exactly as Python does (int*int exact, then / float). */
int salsa_oracle_bin_limits(int fs, int n_fft, int fmin_doa, int fmax_doa, int *lower, int *upper, int *cutoff)
{
    int fmax = fmax_doa < fs / 2 ? fmax_doa : fs / 2;
    int lo = (int)floor((double)((int64_t)fmin_doa * n_fft) / (double)fs);
    int up = (int)floor((double)((int64_t)fmax * n_fft) / (double)fs);
    if (lo < 1) lo = 1;
    *lower = lo;
    *upper = up;
    if (cutoff) *cutoff = (int)floor((double)((int64_t)9000 * n_fft) / (double)fs);
    return 0;
}

/* freq_dim, salsa_feature_extraction.py:306-313 */
int salsa_oracle_freq_dim(int n_fft, int compress)
{
    if (n_fft != 512 && n_fft != 256) return -1;
    if (compress) return n_fft == 512 ? 200 : 100;
    return n_fft / 2;
}

/* ------------------------------------------------------------------------------------------------ W matrix
 * MagStftExtractor.__init__, salsa_feature_extraction.py:152-175.  W is (freq_dim, n_fft/2+1) float32 row-major. */
int salsa_oracle_W(int n_fft, int compress, float *W)
{
    int F = salsa_oracle_freq_dim(n_fft, compress);
    if (F < 0) return -1;
    int nb = n_fft / 2 + 1;
    memset(W, 0, sizeof(float) * (size_t)F * nb);
    if (!compress) {
        for (int i = 0; i < n_fft / 2; i++) W[(size_t)i * nb + i + 1] = 1.0f;
        return 0;
    }
    int ident = n_fft == 512 ? 192 : 96; /* rows 0..ident-1 -> bin i+1 */
    for (int i = 0; i < ident; i++) W[(size_t)i * nb + i + 1] = 1.0f;
    for (int i = ident; i < F; i++) {
        int start = ident + 1 + (i - ident) * 8;
        int cnt = (i < F - 1) ? 8 : 7;
        for (int k = 0; k < cnt; k++) W[(size_t)i * nb + start + k] = 1.0f / 8;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ STFT
 * librosa.stft(y, n_fft, hop, win_length, window='hann', center=True, pad_mode='reflect') as called at
 * salsa_feature_extraction.py:186-192, :360-361 and salsa_lite_feature_extraction.py:97-98.
 * T = 1 + N // hop.  Output complex64 (float64 evaluation, rounded on store). */
long salsa_oracle_n_frames(long N, int hop) { return 1 + N / hop; }

static void make_window(int n_fft, int win, double *w)
{
    /* scipy.signal.get_window('hann', win, fftbins=True) == 0.5 - 0.5 cos(2 pi n / win); pad_center to n_fft */
    int lpad = (n_fft - win) / 2;
    for (int i = 0; i < n_fft; i++) w[i] = 0.0;
    for (int n = 0; n < win; n++) w[lpad + n] = 0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)win);
}

static inline long reflect_idx(long i, long N)
{
    /* np.pad(mode='reflect'): ... y[2] y[1] | y[0] y[1] ... y[N-1] | y[N-2] ...   (period 2N-2) */
    if (N == 1) return 0;
    long p = 2 * (N - 1);
    i %= p;
    if (i < 0) i += p;
    return i < N ? i : p - i;
}

/* in-place iterative radix-2 complex FFT, float64; tw = exp(-2 pi i k / n), k < n/2 */
static void fft_c2c(double *re, double *im, int n, const double *twr, const double *twi)
{
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int s = 0; s < n; s += len)
            for (int k = 0; k < half; k++) {
                double wr = twr[k * step], wi = twi[k * step];
                double xr = re[s + k + half], xi = im[s + k + half];
                double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
                re[s + k + half] = re[s + k] - tr;
                im[s + k + half] = im[s + k] - ti;
                re[s + k] += tr;
                im[s + k] += ti;
            }
    }
}

/* one channel: y[N] float32 -> out[(n_fft/2+1)][T] complex64 (interleaved re,im), bin-major like librosa */
int salsa_oracle_stft(const float *y, long N, int n_fft, int hop, int win, float *out)
{
    if (n_fft & (n_fft - 1)) return -1;
    long T = salsa_oracle_n_frames(N, hop);
    int nb = n_fft / 2 + 1;
    double *w = (double *)malloc(sizeof(double) * n_fft);
    double *twr = (double *)malloc(sizeof(double) * n_fft / 2), *twi = (double *)malloc(sizeof(double) * n_fft / 2);
    make_window(n_fft, win, w);
    for (int k = 0; k < n_fft / 2; k++) {
        twr[k] = cos(-2.0 * M_PI * k / n_fft);
        twi[k] = sin(-2.0 * M_PI * k / n_fft);
    }
#pragma omp parallel num_threads(g_threads)
    {
        double *re = (double *)malloc(sizeof(double) * n_fft), *im = (double *)malloc(sizeof(double) * n_fft);
#pragma omp for schedule(static)
        for (long t = 0; t < T; t++) {
            long base = t * hop - n_fft / 2;
            for (int n = 0; n < n_fft; n++) {
                re[n] = w[n] * (double)y[reflect_idx(base + n, N)];
                im[n] = 0.0;
            }
            fft_c2c(re, im, n_fft, twr, twi);
            for (int k = 0; k < nb; k++) {
                out[((size_t)k * T + t) * 2 + 0] = (float)re[k];
                out[((size_t)k * T + t) * 2 + 1] = (float)im[k];
            }
        }
        free(re);
        free(im);
    }
    free(w); free(twr); free(twi);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ log-spectrogram
 * MagStftExtractor.extract, salsa_feature_extraction.py:177-201:
 *   spec = np.abs(stft) (float32 hypot) ; spec = dot(W, spec**2).T (float32) ; 10*log10(max(1e-10, spec)) (float32)
 * stft: [nb][T] c64 of ONE channel ; out: [T][F] float32. */
static void logspec_from_stft(const float *stft, long T, int n_fft, int compress, float *out)
{
    int nb = n_fft / 2 + 1;
    int F = salsa_oracle_freq_dim(n_fft, compress);
    int ident = compress ? (n_fft == 512 ? 192 : 96) : n_fft / 2;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long t = 0; t < T; t++) {
        for (int f = 0; f < F; f++) {
            float acc;
            if (f < ident) {
                int k = f + 1;
                float a = hypotf(stft[((size_t)k * T + t) * 2], stft[((size_t)k * T + t) * 2 + 1]);
                acc = a * a;
            } else {
                int start = ident + 1 + (f - ident) * 8;
                int cnt = (f < F - 1) ? 8 : 7;
                acc = 0.0f;
                for (int j = 0; j < cnt; j++) {
                    int k = start + j;
                    float a = hypotf(stft[((size_t)k * T + t) * 2], stft[((size_t)k * T + t) * 2 + 1]);
                    acc += 0.125f * (a * a);
                }
            }
            (void)nb;
            out[(size_t)t * F + f] = 10.0f * log10f(fmaxf(1e-10f, acc));
        }
    }
}

/* audio [C][N] -> out [C][T][F] */
int salsa_oracle_logspec(const float *audio, int C, long N, int n_fft, int hop, int win, int compress, float *out)
{
    long T = salsa_oracle_n_frames(N, hop);
    int nb = n_fft / 2 + 1;
    int F = salsa_oracle_freq_dim(n_fft, compress);
    if (F < 0) return -1;
    float *st = (float *)malloc(sizeof(float) * 2 * (size_t)nb * T);
    for (int c = 0; c < C; c++) {
        salsa_oracle_stft(audio + (size_t)c * N, N, n_fft, hop, win, st);
        logspec_from_stft(st, T, n_fft, compress, out + (size_t)c * T * F);
    }
    free(st);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ 4x4 Hermitian eigen
 * Stands in for np.linalg.svd(Rxx1) at salsa_feature_extraction.py:103.  Rxx1 is Hermitian PSD, so its singular
 * values are its eigenvalues (|.| of them) and u[:,0] is the principal eigenvector up to a unit phase; every use the
 * reference makes of u (:118, :121) is invariant to that phase.  Cyclic complex Jacobi, float64.
 * A (in, destroyed): ar/ai 4x4 row-major ; on return lam[4] sorted descending by |.|, v = eigenvector of lam[0]. */
static void herm4_eig(double ar[4][4], double ai[4][4], double lam[4], double vr[4], double vi[4])
{
    double Vr[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}, Vi[4][4] = {{0}};
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < 4; p++) {
            diag += ar[p][p] * ar[p][p];
            for (int q = p + 1; q < 4; q++) off += ar[p][q] * ar[p][q] + ai[p][q] * ai[p][q];
        }
        if (off <= 1e-34 * diag || off == 0.0) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double br = ar[p][q], bi = ai[p][q];
                double b = hypot(br, bi);
                if (b == 0.0) continue;
                /* unit phase e = a_pq/|a_pq| ; 2x2 real problem [[app, b],[b, aqq]] */
                double er = br / b, ei = bi / b;
                double theta = (ar[q][q] - ar[p][p]) / (2.0 * b);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                /* J = [[c, s],[-s conj(e), c conj(e)]] on columns (p,q):  col_p' = c col_p - s conj(e) col_q ;
                 * col_q' = s col_p + c conj(e) col_q.   A <- J^H A J ,  V <- V J. */
                for (int k = 0; k < 4; k++) { /* columns of A */
                    double pr = ar[k][p], pi = ai[k][p], qr = ar[k][q], qi = ai[k][q];
                    double cqr = qr * er + qi * ei, cqi = qi * er - qr * ei; /* conj(e) * a_kq */
                    ar[k][p] = c * pr - s * cqr; ai[k][p] = c * pi - s * cqi;
                    ar[k][q] = s * pr + c * cqr; ai[k][q] = s * pi + c * cqi;
                }
                for (int k = 0; k < 4; k++) { /* rows of A: row_p' = c row_p - s e row_q ; row_q' = s row_p + c e row_q */
                    double pr = ar[p][k], pi = ai[p][k], qr = ar[q][k], qi = ai[q][k];
                    double eqr = qr * er - qi * ei, eqi = qi * er + qr * ei; /* e * a_qk */
                    ar[p][k] = c * pr - s * eqr; ai[p][k] = c * pi - s * eqi;
                    ar[q][k] = s * pr + c * eqr; ai[q][k] = s * pi + c * eqi;
                }
                ar[p][q] = ai[p][q] = ar[q][p] = ai[q][p] = 0.0;
                ai[p][p] = ai[q][q] = 0.0;
                for (int k = 0; k < 4; k++) {
                    double pr = Vr[k][p], pi = Vi[k][p], qr = Vr[k][q], qi = Vi[k][q];
                    double cqr = qr * er + qi * ei, cqi = qi * er - qr * ei;
                    Vr[k][p] = c * pr - s * cqr; Vi[k][p] = c * pi - s * cqi;
                    Vr[k][q] = s * pr + c * cqr; Vi[k][q] = s * pi + c * cqi;
                }
            }
    }
    int idx[4] = {0, 1, 2, 3};
    for (int i = 0; i < 4; i++) lam[i] = fabs(ar[i][i]);
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 4; j++)
            if (lam[idx[j]] > lam[idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    double l2[4];
    for (int i = 0; i < 4; i++) l2[i] = lam[idx[i]];
    for (int i = 0; i < 4; i++) lam[i] = l2[i];
    for (int k = 0; k < 4; k++) { vr[k] = Vr[k][idx[0]]; vi[k] = Vi[k][idx[0]]; }
}

/* ------------------------------------------------------------------------------------------------ eigenvector stage
 * extract_normalized_eigenvector, salsa_feature_extraction.py:17-129.
 * X: [n_bins][n_frames][4] complex64 (interleaved).  out: [3][n_bins][n_frames] float64 (zeros where gated).
 * sig_mask (optional, may be NULL): [n_bins][n_frames] u8 = indicator_sig (:87).
 * rank_mask (optional): [n_bins][n_frames] u8: 0 = not evaluated, 1 = coherence test failed, 2 = passed (:106).
 * margin (optional): [n_bins][n_frames] float64 = (s0 - s1*cond)/s0 for evaluated bins (gate margin), else NaN.
 * format: 0 'foa' (:117-119), 1 'mic' (:120-123).  Per-bin state is independent, so the loop nest is bin-major. */
int salsa_oracle_eigvec(const float *X, int n_bins, long n_frames, double cond, int n_hop, int tracking, int format,
                        int fs, int n_fft, int lower_bin, double *out, unsigned char *sig_mask,
                        unsigned char *rank_mask, double *margin)
{
    const int n_ch = 4;
    if (format != 0 && format != 1) return -2; /* ValueError('audio format ... is not valid') :125 */
    const double floor_up = 1 + 0.02, floor_up_slow = 1 + 0.1 * 0.02, floor_down = 1 - 0.02, snr_ratio = 1.5; /* :31-36 */
    const double delta = (fs > 0 && n_fft > 0) ? 2.0 * M_PI * fs / (n_fft * 343.0) : 0.0;                     /* :39-40 */
    const long T = n_frames;
    const int L = 2 * n_hop + 1;
    memset(out, 0, sizeof(double) * 3 * (size_t)n_bins * T);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int b = 0; b < n_bins; b++) {
        const float *Xb = X + (size_t)b * T * n_ch * 2;
        double *mag = (double *)malloc(sizeof(double) * T);
        /* :47-55  3-frame running RMS of channel 0 over the wrap-padded time axis */
        for (long t = 0; t < T; t++) {
            double acc = 0.0;
            for (int k = 0; k < 3; k++) {
                long tt = ((t - k) % T + T) % T;
                double a = hypot((double)Xb[(tt * n_ch) * 2], (double)Xb[(tt * n_ch) * 2 + 1]);
                acc = acc + a * a;
            }
            mag[t] = sqrt(acc / 3);
        }
        /* :58  noise_floor = 0.5 * mean(first 5 frames) */
        long n0 = T < 5 ? T : 5;
        double nf = 0.0;
        for (long t = 0; t < n0; t++) nf += mag[t];
        nf = 0.5 * (nf / (double)n0);
        long countdown = 3; /* :30 */
        for (long t = 0; t < T; t++) {
            double xf = mag[t];
            /* :65-85 */
            int above = xf > nf;
            if (above) {
                countdown -= 1;
                if (countdown < 0) nf = floor_up_slow * nf;
                else nf = floor_up * nf;
            } else {
                countdown = 3;
                nf = floor_down * nf;
            }
            if (nf < 1e-6) nf = 1e-6;
            int sig = xf > snr_ratio * nf; /* :87 */
            if (sig_mask) sig_mask[(size_t)b * T + t] = (unsigned char)sig;
            if (rank_mask) rank_mask[(size_t)b * T + t] = 0;
            if (margin) margin[(size_t)b * T + t] = NAN;
            int valid = tracking ? sig : 1; /* :90-93 */
            if (!valid) continue;
            /* :99-100  Rxx = X1^T conj(X1) / (2*n_hop+1) over frames t-n_hop..t+n_hop (wrap) */
            double ar[4][4] = {{0}}, ai[4][4] = {{0}};
            for (int k = -n_hop; k <= n_hop; k++) {
                long tt = ((t + k) % T + T) % T;
                const float *x = Xb + tt * n_ch * 2;
                for (int i = 0; i < 4; i++)
                    for (int j = 0; j < 4; j++) {
                        double xr = x[2 * i], xi = x[2 * i + 1], yr = x[2 * j], yi = x[2 * j + 1];
                        ar[i][j] += xr * yr + xi * yi; /* x_i * conj(x_j) */
                        ai[i][j] += xi * yr - xr * yi;
                    }
            }
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) { ar[i][j] /= (double)L; ai[i][j] /= (double)L; }
            double lam[4], ur[4], ui[4];
            herm4_eig(ar, ai, lam, ur, ui);
            int rank1 = lam[0] > lam[1] * cond; /* :106 */
            if (rank_mask) rank_mask[(size_t)b * T + t] = rank1 ? 2 : 1;
            if (margin) margin[(size_t)b * T + t] = (lam[0] - lam[1] * cond) / lam[0];
            if (tracking) valid = valid && rank1; /* :111-112 (ignored when not tracking) */
            if (!valid) continue;
            double e[3];
            if (format == 0) {
                /* :118-119  real(u[1:,0]/u[0,0]) then L2-normalise (no guard: IEEE inf/nan propagate) */
                double den = ur[0] * ur[0] + ui[0] * ui[0], ss = 0.0;
                for (int i = 0; i < 3; i++) {
                    e[i] = (ur[i + 1] * ur[0] + ui[i + 1] * ui[0]) / den;
                    ss += e[i] * e[i];
                }
                ss = sqrt(ss);
                for (int i = 0; i < 3; i++) e[i] = e[i] / ss;
            } else {
                /* :121-123  angle(u[1:,0]*conj(u[0,0])) / (delta*(ibin+lower_bin)) */
                for (int i = 0; i < 3; i++) {
                    double pr = ur[i + 1] * ur[0] + ui[i + 1] * ui[0];
                    double pi = ui[i + 1] * ur[0] - ur[i + 1] * ui[0];
                    e[i] = atan2(pi, pr) / (delta * (double)(b + lower_bin));
                }
            }
            for (int i = 0; i < 3; i++) out[((size_t)i * n_bins + b) * T + t] = e[i]; /* :127 */
        }
        free(mag);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ full SALSA, one clip
 * extract_features per-file body, salsa_feature_extraction.py:353-377.
 * audio [4][N] float32 -> out [7][T][F] float32 (the array written as 'feature' at :380-382).
 * aux_rank / aux_margin (optional): [upper-lower][T] gate diagnostics from the eigenvector stage. */
int salsa_oracle_extract_salsa(const float *audio, long N, int fs, int n_fft, int hop, int win, int fmin_doa,
                               int fmax_doa, double cond, int n_hop, int tracking, int compress, int format,
                               float *out, unsigned char *aux_rank, double *aux_margin)
{
    const int C = 4;
    int lower, upper;
    int F = salsa_oracle_freq_dim(n_fft, compress);
    if (F < 0) return -1; /* assert n_fft in (256,512) :306 */
    if (format != 0 && format != 1) return -2;
    salsa_oracle_bin_limits(fs, n_fft, fmin_doa, fmax_doa, &lower, &upper, NULL);
    long T = salsa_oracle_n_frames(N, hop);
    int nb = n_fft / 2 + 1, nd = upper - lower;
    if (nd < 0 || nd > F) return -3;
    float *st = (float *)malloc(sizeof(float) * 2 * (size_t)nb * T);
    float *X = (float *)malloc(sizeof(float) * 2 * (size_t)(nd > 0 ? nd : 1) * T * C);
    for (int c = 0; c < C; c++) {
        salsa_oracle_stft(audio + (size_t)c * N, N, n_fft, hop, win, st);          /* :186 and :360 (same STFT) */
        logspec_from_stft(st, T, n_fft, compress, out + (size_t)c * T * F);        /* :194-195 */
        for (int b = 0; b < nd; b++)                                                /* :365-366 */
            for (long t = 0; t < T; t++) {
                X[(((size_t)b * T + t) * C + c) * 2] = st[((size_t)(b + lower) * T + t) * 2];
                X[(((size_t)b * T + t) * C + c) * 2 + 1] = st[((size_t)(b + lower) * T + t) * 2 + 1];
            }
    }
    double *ev = (double *)malloc(sizeof(double) * 3 * (size_t)(nd > 0 ? nd : 1) * T);
    salsa_oracle_eigvec(X, nd, T, cond, n_hop, tracking, format, fs, n_fft, lower, ev, NULL, aux_rank, aux_margin);
    /* :372-377  transpose (0,2,1) into zeros (3,T,F) and stack under the 4 log-spec channels */
    for (int i = 0; i < 3; i++)
        for (long t = 0; t < T; t++) {
            float *o = out + ((size_t)(4 + i) * T + t) * F;
            for (int f = 0; f < F; f++) o[f] = 0.0f;
            for (int b = 0; b < nd; b++) o[b] = (float)ev[((size_t)i * nd + b) * T + t];
        }
    free(ev); free(X); free(st);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ SALSA-Lite / IPD
 * salsa_lite_feature_extraction.py:94-123.  audio [4][N] -> out [7][T][cutoff-lower] float32.
 * ipd = 0: 'salsa_lite' (divide by delta*k, :115) ; ipd = 1: 'salsa_ipd' (divide by pi, :113).
 * Note :120 zeroes CROPPED indices >= upper_bin (i.e. absolute bins >= upper_bin+lower_bin) -- reproduced as is. */
int salsa_oracle_extract_lite(const float *audio, long N, int fs, int n_fft, int hop, int win, int fmin_doa,
                              int fmax_doa, int ipd, float *out)
{
    const int C = 4;
    int lower, upper, cutoff;
    salsa_oracle_bin_limits(fs, n_fft, fmin_doa, fmax_doa, &lower, &upper, &cutoff);
    if (upper > cutoff) return -4; /* assert :59 */
    long T = salsa_oracle_n_frames(N, hop);
    int nb = n_fft / 2 + 1, F = cutoff - lower;
    if (F <= 0) return -3;
    const double delta = 2.0 * M_PI * fs / (n_fft * 343.0); /* :62-63 */
    float *st = (float *)malloc(sizeof(float) * 2 * (size_t)nb * T * C);
    for (int c = 0; c < C; c++) salsa_oracle_stft(audio + (size_t)c * N, N, n_fft, hop, win, st + (size_t)c * 2 * nb * T);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long t = 0; t < T; t++) {
        for (int f = 0; f < F; f++) {
            int k = f + lower;
            const float *x0 = st + ((size_t)k * T + t) * 2;
            for (int c = 0; c < C; c++) {
                const float *xc = st + (size_t)c * 2 * nb * T + ((size_t)k * T + t) * 2;
                float a = hypotf(xc[0], xc[1]); /* :103-105 float32 */
                out[((size_t)c * T + t) * F + f] = 10.0f * log10f(fmaxf(1e-10f, a * a));
                if (c > 0) {
                    /* :111  angle(X_c * conj(X_0)) in float64 */
                    double pr = (double)xc[0] * x0[0] + (double)xc[1] * x0[1];
                    double pi = (double)xc[1] * x0[0] - (double)xc[0] * x0[1];
                    double ph = atan2(pi, pr);
                    if (ipd) ph = ph / M_PI;
                    else ph = ph / (delta * (double)(k == 0 ? 1 : k)); /* freq_vector[0] = 1 :65 */
                    if (f >= upper) ph = 0.0; /* :120 (index into the cropped axis) */
                    out[((size_t)(3 + c) * T + t) * F + f] = (float)ph;
                }
            }
        }
    }
    free(st);
    return 0;
}

/* ================================================================================================ contrib on-the-fly
 * contrib/salsa_flexible.py: SpatialFeaturesAbstract.__init__ (:169-193), .spectrograms (:195-221),
 * .__call__ (:237-265), SalsaNoiseFloorTracker (:80-146), stacked_covmat_eigh (:52-77), SalsaFeatures.features
 * (:286-367), SalsaLiteFeatures.features (:386-400).  Arbitrary channel count (here 2..SALSA_FLEX_MAXCH).
 *
 * n x n Hermitian eigen-decomposition standing in for np.linalg.eigh(covmats, UPLO='U') (:75): cyclic complex Jacobi in
 * float64, eigenvalues returned ASCENDING and signed like LAPACK's, v = eigenvector of the largest one (any unit phase:
 * :362 only uses conj(v0)*v_i). */
#define SALSA_FLEX_MAXCH 16
static void hermn_eigh(int n, double ar[SALSA_FLEX_MAXCH][SALSA_FLEX_MAXCH], double ai[SALSA_FLEX_MAXCH][SALSA_FLEX_MAXCH],
                       double *lam, double *vr, double *vi)
{
    double Vr[SALSA_FLEX_MAXCH][SALSA_FLEX_MAXCH] = {{0}}, Vi[SALSA_FLEX_MAXCH][SALSA_FLEX_MAXCH] = {{0}};
    for (int i = 0; i < n; i++) Vr[i][i] = 1.0;
    for (int sweep = 0; sweep < 80; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < n; p++) {
            diag += ar[p][p] * ar[p][p];
            for (int q = p + 1; q < n; q++) off += ar[p][q] * ar[p][q] + ai[p][q] * ai[p][q];
        }
        if (off <= 1e-34 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double br = ar[p][q], bi = ai[p][q], b = hypot(br, bi);
                if (b == 0.0) continue;
                const double er = br / b, ei = bi / b;
                const double theta = (ar[q][q] - ar[p][p]) / (2.0 * b);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) { /* A <- A J */
                    const double pr = ar[k][p], pi = ai[k][p], qr = ar[k][q], qi = ai[k][q];
                    const double cqr = qr * er + qi * ei, cqi = qi * er - qr * ei;
                    ar[k][p] = c * pr - s * cqr; ai[k][p] = c * pi - s * cqi;
                    ar[k][q] = s * pr + c * cqr; ai[k][q] = s * pi + c * cqi;
                }
                for (int k = 0; k < n; k++) { /* A <- J^H A */
                    const double pr = ar[p][k], pi = ai[p][k], qr = ar[q][k], qi = ai[q][k];
                    const double eqr = qr * er - qi * ei, eqi = qi * er + qr * ei;
                    ar[p][k] = c * pr - s * eqr; ai[p][k] = c * pi - s * eqi;
                    ar[q][k] = s * pr + c * eqr; ai[q][k] = s * pi + c * eqi;
                }
                ar[p][q] = ai[p][q] = ar[q][p] = ai[q][p] = 0.0;
                ai[p][p] = ai[q][q] = 0.0;
                for (int k = 0; k < n; k++) { /* V <- V J */
                    const double pr = Vr[k][p], pi = Vi[k][p], qr = Vr[k][q], qi = Vi[k][q];
                    const double cqr = qr * er + qi * ei, cqi = qi * er - qr * ei;
                    Vr[k][p] = c * pr - s * cqr; Vi[k][p] = c * pi - s * cqi;
                    Vr[k][q] = s * pr + c * cqr; Vi[k][q] = s * pi + c * cqi;
                }
            }
    }
    int idx[SALSA_FLEX_MAXCH];
    for (int i = 0; i < n; i++) idx[i] = i;
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++)
            if (ar[idx[j]][idx[j]] < ar[idx[i]][idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    for (int i = 0; i < n; i++) lam[i] = ar[idx[i]][idx[i]];
    for (int k = 0; k < n; k++) { vr[k] = Vr[k][idx[n - 1]]; vi[k] = Vi[k][idx[n - 1]]; }
}

/* Bin limits of SpatialFeaturesAbstract.__init__ (:177-184): like the dataset scripts' but WITHOUT the fs/2 clamp on
 * fmax_doa and with the spectrogram cutoff taken from fmax_spec.  Returns -4 for the assert at :183. */
int salsa_oracle_flex_bins(int fs, int n_fft, int fmin_doa, int fmax_doa, int fmax_spec, int *lower, int *upper, int *cutoff)
{
    int lo = (int)floor((double)((long)fmin_doa * n_fft) / (double)fs);
    if (lo < 1) lo = 1;
    *lower = lo;
    *upper = (int)floor((double)((long)fmax_doa * n_fft) / (double)fs);
    *cutoff = (int)floor((double)((long)fmax_spec * n_fft) / (double)fs);
    return *upper <= *cutoff ? 0 : -4;
}

/* obj(wavchans, clip_freqs, clip_spatial_alias, **kw) of SalsaFeatures (lite = 0) / SalsaLiteFeatures (lite = 1).
 * audio [C][N] float32 -> out_spec [C][F][T] float32 (the float32 values the reference upcasts at :264) and
 * out_spatial [C-1][F][T] float64; F = cutoff-lower (clip_freqs) or n_fft/2+1.  Returns F, or a negative error. */
int salsa_oracle_flex(const float *audio, int C, long N, int fs, int n_fft, int hop, int fmin_doa, int fmax_doa,
                      int fmax_spec, int lite, int clip_freqs, int clip_alias, double ew_thresh, int n_neigh,
                      int tracking, double floor_mask_ratio, float *out_spec, double *out_spatial)
{
    if (C < 2 || C > SALSA_FLEX_MAXCH) return -2;
    int lower, upper, cutoff;
    if (salsa_oracle_flex_bins(fs, n_fft, fmin_doa, fmax_doa, fmax_spec, &lower, &upper, &cutoff)) return -4;
    const int nb = n_fft / 2 + 1;
    const int lo = clip_freqs ? lower : 0, hi = clip_freqs ? (cutoff < nb ? cutoff : nb) : nb; /* numpy slicing clips */
    const int F = hi - lo;
    if (F <= 0) return -3;
    const long T = salsa_oracle_n_frames(N, hop);
    const double delta = 2.0 * M_PI * fs / (n_fft * 343.0); /* :186 */
    float *st = (float *)malloc(sizeof(float) * 2 * (size_t)nb * T * C);
    for (int c = 0; c < C; c++) salsa_oracle_stft(audio + (size_t)c * N, N, n_fft, hop, n_fft, st + (size_t)c * 2 * nb * T);
#define STF(c, k, t) (st + (size_t)(c) * 2 * nb * T + ((size_t)(k) * T + (t)) * 2)
    memset(out_spatial, 0, sizeof(double) * (size_t)(C - 1) * F * T);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int f = 0; f < F; f++) {
        const int k = f + lo;
        /* norm_freq (:188-190): float32 arange, [0] = 1, multiplied IN float32 by delta */
        const double nf = (double)((float)(k == 0 ? 1 : k) * (float)delta);
        for (int c = 0; c < C; c++) /* :211-215 float32 power and dB */
            for (long t = 0; t < T; t++) {
                const float *x = STF(c, k, t);
                const float a = hypotf(x[0], x[1]);
                out_spec[((size_t)c * F + f) * T + t] = 10.0f * log10f(fmaxf(1e-10f, a * a));
            }
        if (lite) { /* :397-399 */
            for (int c = 1; c < C; c++)
                for (long t = 0; t < T; t++) {
                    const float *x0 = STF(0, k, t), *xc = STF(c, k, t);
                    const double pr = (double)x0[0] * xc[0] + (double)x0[1] * xc[1];
                    const double pi = (double)x0[0] * xc[1] - (double)x0[1] * xc[0]; /* conj(X0) * Xc */
                    out_spatial[((size_t)(c - 1) * F + f) * T + t] = atan2(pi, pr) / nf;
                }
            continue;
        }
        /* :326-333 raw |X0| and the initial floor (clamped to epsilon by the tracker's constructor :118-120) */
        double floor_ = 0.0;
        const long n0 = T < 5 ? T : 5;
        for (long t = 0; t < n0; t++) floor_ += hypot((double)STF(0, k, t)[0], (double)STF(0, k, t)[1]);
        floor_ = 0.5 * (floor_ / (double)n0);
        if (floor_ < 1e-6) floor_ = 1e-6;
        long count = 0; /* :123 */
        int alive = 1;  /* is_tracking=False: ONE allpass mask array is created (:336-337) and then narrowed in place by
                           'mask[mask] = good_coherence_mask' (:354) -- a bin that fails once stays off for the clip */
        for (long t = 0; t < T; t++) {
            const double m = hypot((double)STF(0, k, t)[0], (double)STF(0, k, t)[1]);
            int mask;
            if (tracking) { /* SalsaNoiseFloorTracker.__call__ :125-146 */
                const int above = m > floor_;
                count += above;
                if (above && count <= 3) floor_ *= 1.02;
                if (above && count > 3) floor_ *= 1.002;
                if (!above) floor_ *= 0.98;
                if (floor_ < 1e-6) floor_ = 1e-6;
                if (!above) count = 0;
                mask = m > floor_mask_ratio * floor_;
            } else {
                mask = alive;
            }
            if (!mask) continue;
            double ar[SALSA_FLEX_MAXCH][SALSA_FLEX_MAXCH] = {{0}}, ai[SALSA_FLEX_MAXCH][SALSA_FLEX_MAXCH] = {{0}};
            for (int d = -n_neigh; d <= n_neigh; d++) { /* :347-349, :70-74: SUM (not mean) of x_i conj(x_j), wrap in time */
                const long tt = ((t + d) % T + T) % T;
                for (int i = 0; i < C; i++)
                    for (int j = 0; j < C; j++) {
                        const float *xi = STF(i, k, tt), *xj = STF(j, k, tt);
                        ar[i][j] += (double)xi[0] * xj[0] + (double)xi[1] * xj[1];
                        ai[i][j] += (double)xi[1] * xj[0] - (double)xi[0] * xj[1];
                    }
            }
            double lam[SALSA_FLEX_MAXCH], ur[SALSA_FLEX_MAXCH], ui[SALSA_FLEX_MAXCH];
            hermn_eigh(C, ar, ai, lam, ur, ui);
            const int good = lam[C - 1] > lam[C - 2] * ew_thresh; /* :353 */
            if (!tracking) alive = good;
            if (!good) continue;
            for (int i = 1; i < C; i++) { /* :362-363 angle(conj(u0) u_i) / norm_freq */
                const double pr = ur[0] * ur[i] + ui[0] * ui[i], pi = ur[0] * ui[i] - ui[0] * ur[i];
                out_spatial[((size_t)(i - 1) * F + f) * T + t] = atan2(pi, pr) / nf;
            }
        }
    }
    if (clip_alias) /* :262-263 index into the (possibly cropped) axis */
        for (int i = 0; i < C - 1; i++)
            for (int f = upper; f < F; f++) memset(out_spatial + ((size_t)i * F + f) * T, 0, sizeof(double) * T);
#undef STF
    free(st);
    return F;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Resampling on load: librosa.load(path, sr=fs, mono=False, dtype=np.float32) (dataset/salsa_feature_extraction.py:353,
 * salsa_lite_feature_extraction.py:93) of a file whose native rate is not fs -> librosa 0.8.0 core/audio.py::resample
 * (res_type='kaiser_best', fix=True) -> resampy 0.2.2 (requirements.yml:181).  THIRD-PARTY ARITHMETIC, absent from
 * /root/reference and from the image: PARITY UNPINNED for this function -- restated from resampy's published
 * interpn.py::resample_f (a numba loop), whose order is kept: for each output sample, the left wing of the filter from
 * sample n = int(time_register) downwards, then the right wing from n + 1 upwards; the filter value is linearly
 * interpolated between table entries; y (float32, the dtype of x) is updated in place, i.e. rounded to float32 after
 * every tap; time_register accumulates 1 / sample_ratio in float64.  x: [n_rows][n_in]; y: [n_rows][n_fix] with
 * n_out = int(n_in * sample_ratio) computed samples followed by zeros (librosa util.fix_length to
 * ceil(n_in * sample_ratio)).  interp_win / interp_delta / num_table: resampy core.py::resample's locals (the caller
 * scales the window by sample_ratio when < 1 and takes np.diff, as core.py does). */
int salsa_oracle_resample(const float *x, int n_rows, long n_in, float *y, long n_out, long n_fix, double sample_ratio,
                          const double *interp_win, const double *interp_delta, int nwin, int num_table)
{
    if (!x || !y || n_rows <= 0 || n_in <= 0 || n_out < 0 || n_fix < n_out || !(sample_ratio > 0.0)) return -1;
    const double scale = sample_ratio < 1.0 ? sample_ratio : 1.0;
    const double time_increment = 1.0 / sample_ratio;
    const int index_step = (int)(scale * num_table);
    if (index_step < 1) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int r = 0; r < n_rows; r++) {
        const float *xr = x + (long)r * n_in;
        float *yr = y + (long)r * n_fix;
        double time_register = 0.0;
        for (long t = 0; t < n_fix; t++) yr[t] = 0.f;
        for (long t = 0; t < n_out; t++) {
            const long n = (long)time_register;                       /* top bits: index into the input */
            double frac = scale * (time_register - (double)n);
            double index_frac = frac * num_table;
            int offset = (int)index_frac;
            double eta = index_frac - offset;
            long m = (nwin - offset) / index_step;
            long i_max = n + 1 < m ? n + 1 : m;
            for (long i = 0; i < i_max; i++) {                        /* left wing */
                const double weight = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step];
                yr[t] = (float)((double)yr[t] + weight * (double)xr[n - i]);
            }
            frac = scale - frac;                                      /* invert P */
            index_frac = frac * num_table;
            offset = (int)index_frac;
            eta = index_frac - offset;
            m = (nwin - offset) / index_step;
            long k_max = n_in - n - 1 < m ? n_in - n - 1 : m;
            for (long k = 0; k < k_max; k++) {                        /* right wing */
                const double weight = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step];
                yr[t] = (float)((double)yr[t] + weight * (double)xr[n + k + 1]);
            }
            time_register += time_increment;
        }
    }
    return 0;
}
