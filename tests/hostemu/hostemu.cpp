// tests/hostemu/hostemu.cpp -- CPU unit-test harness for salsa_amd/csrc/salsa_math.h (the per-thread arithmetic of the
// HIP kernels).  TEST INFRASTRUCTURE ONLY: it lets `pytest -m "not gpu"` exercise the kernels' butterflies, Stockham
// addressing and eigen-gate on CPU (no GPU in the build container).  The product never loads it and has no CPU path.
#include <string.h>
#include "../../salsa_amd/csrc/salsa_math.h"
#include <vector>
using namespace salsa;

extern "C" {

// One N-point complex FFT through the kernel's Stockham addressing + dftR butterflies (float64 or float32).
int hostemu_fft(const double *in_re, const double *in_im, int N, int use_f32, double *out_re, double *out_im)
{
    const int R = (N == 512) ? 8 : (N == 256 ? 4 : 0);
    if (!R) return -1;
    std::vector<cplx<double>> tw(N);
    for (int m = 0; m < N; m++) tw[m] = {cos(-2.0 * M_PI * m / N), sin(-2.0 * M_PI * m / N)};
    if (use_f32) {
        std::vector<cplx<float>> x(N), y(N);
        for (int n = 0; n < N; n++) x[n] = {(float)in_re[n], (float)in_im[n]};
        for (int p = 1; p < N; p *= R) {
            for (int i = 0; i < N / R; i++) {
                cplx<float> v[8];
                for (int r = 0; r < R; r++) {
                    v[r] = x[stockham_in(i, r, N, R)];
                    int e = stockham_tw(i, r, p, N, R);
                    v[r] = cmul(v[r], cplx<float>{(float)tw[e].re, (float)tw[e].im});
                }
                if (R == 8) dftR<8>(v); else dftR<4>(v);
                for (int r = 0; r < R; r++) y[stockham_out(i, r, p, R)] = v[r];
            }
            x.swap(y);
        }
        for (int n = 0; n < N; n++) { out_re[n] = x[n].re; out_im[n] = x[n].im; }
        return 0;
    }
    std::vector<cplx<double>> x(N), y(N);
    for (int n = 0; n < N; n++) x[n] = {in_re[n], in_im[n]};
    for (int p = 1; p < N; p *= R) {
        for (int i = 0; i < N / R; i++) {
            cplx<double> v[8];
            for (int r = 0; r < R; r++) {
                v[r] = x[stockham_in(i, r, N, R)];
                v[r] = cmul(v[r], tw[stockham_tw(i, r, p, N, R)]);
            }
            if (R == 8) dftR<8>(v); else dftR<4>(v);
            for (int r = 0; r < R; r++) y[stockham_out(i, r, p, R)] = v[r];
        }
        x.swap(y);
    }
    for (int n = 0; n < N; n++) { out_re[n] = x[n].re; out_im[n] = x[n].im; }
    return 0;
}

// two real sequences through one packed complex FFT + unpack_pair
int hostemu_rfft_pair(const double *x0, const double *x1, int N, double *X0 /*[N/2+1][2]*/, double *X1)
{
    std::vector<double> re(N), im(N);
    if (hostemu_fft(x0, x1, N, 0, re.data(), im.data())) return -1;
    for (int k = 0; k <= N / 2; k++) {
        cplx<double> a = {re[k], im[k]}, b = {re[(N - k) % N], im[(N - k) % N]}, u, v;
        unpack_pair(a, b, u, v);
        X0[2 * k] = u.re; X0[2 * k + 1] = u.im;
        X1[2 * k] = v.re; X1[2 * k + 1] = v.im;
    }
    return 0;
}

// The eigenvector stage exactly as the cov/eig kernel evaluates it per TF bin.
// X [nb][nt][4] c64 ; out [3][nb][nt] f64 ; rank [nb][nt] u8 (0 not evaluated, 1 fail, 2 pass)
int hostemu_eigvec(const float *X, int nb, long nt, double cond, int n_hop, int tracking, int format, double delta,
                   int lower_bin, double *out, unsigned char *rank)
{
    for (size_t i = 0; i < (size_t)3 * nb * nt; i++) out[i] = 0.0;
    for (int b = 0; b < nb; b++) {
        const float *Xb = X + (size_t)b * nt * 8;
        auto pw = [&](long t) {
            t = ((t % nt) + nt) % nt;
            double re = Xb[t * 8], im = Xb[t * 8 + 1];
            return re * re + im * im;
        };
        auto mag = [&](long t) { return sqrt(((0.0 + pw(t)) + pw(t - 1) + pw(t - 2)) / 3); };
        long n0 = nt < 5 ? nt : 5;
        double acc = 0;
        for (long t = 0; t < n0; t++) acc += mag(t);
        double fl = 0.5 * (acc / (double)n0);
        int cd = 3;
        for (long t = 0; t < nt; t++) {
            bool sig = tracker_step(fl, cd, mag(t));
            rank[(size_t)b * nt + t] = 0;
            if (tracking && !sig) continue;
            herm4<double> R = {};
            for (int k = -n_hop; k <= n_hop; k++) {
                long tt = (((t + k) % nt) + nt) % nt;
                cplx<double> x[4];
                for (int c = 0; c < 4; c++) x[c] = {(double)Xb[tt * 8 + 2 * c], (double)Xb[tt * 8 + 2 * c + 1]};
                herm4_rank1_add(R, x);
            }
            eig_result<double> er = herm4_gate_eigvec(R, cond, 1.0 / cond, !tracking, format != 0);
            rank[(size_t)b * nt + t] = er.rank1 ? 2 : 1;
            if (tracking && !er.rank1) continue;
            double e[3];
            if (er.col0) { // the kernel's gated fast path
                if (format == 0) normalise_foa_col0(er.u, e);
                else normalise_mic_col0(er.u, delta * (double)(b + lower_bin), e);
            } else if (format == 0) normalise_foa(er.u, e, !tracking);
            else normalise_mic(er.u, delta * (double)(b + lower_bin), e);
            for (int i = 0; i < 3; i++) out[((size_t)i * nb + b) * nt + t] = e[i];
        }
    }
    return 0;
}

long hostemu_reflect(long i, long N) { return reflect_index(i, N); }

} // extern "C"

extern "C" {
// Direct access to the per-bin solver: R packed as d[4], o[6] (re,im) -> rank1 flag, u[4] (re,im)
int hostemu_solve(const double *d, const double *o, double cond, int need_vec, int *rank1, double *u)
{
    herm4<double> R;
    for (int i = 0; i < 4; i++) R.d[i] = d[i];
    for (int k = 0; k < 6; k++) R.o[k] = {o[2 * k], o[2 * k + 1]};
    eig_result<double> er = herm4_gate_eigvec(R, cond, cond > 0 ? 1.0 / cond : 0.0, need_vec != 0);
    *rank1 = er.rank1 ? 1 : 0;
    for (int i = 0; i < 4; i++) { u[2 * i] = er.u[i].re; u[2 * i + 1] = er.u[i].im; }
    return 0;
}
}

extern "C" {
// The per-bin solve + normalisation as the kernel's emit path runs it, in float64 or (study only: DESIGN "precision policy")
// with the whole solve instantiated in float32 on the float64-accumulated covariance.  gated != 0: the tracking-on mode (gate
// decides, column-0 fast path allowed); returns rank1, whether the fast path was taken, and the 3 feature values.
int hostemu_feature(const double *d, const double *o, double cond, int gated, int format, double dk, int use_f32, int *rank1,
                    int *col0, double *e3)
{
    herm4<double> R;
    for (int i = 0; i < 4; i++) R.d[i] = d[i];
    for (int k = 0; k < 6; k++) R.o[k] = {o[2 * k], o[2 * k + 1]};
    e3[0] = e3[1] = e3[2] = 0.0;
    if (use_f32) {
        herm4<float> Rf;
        const double sc = pow2_unscale(R.d[0] + R.d[1] + R.d[2] + R.d[3]); // exact power of two: keep float32 in range
        for (int i = 0; i < 4; i++) Rf.d[i] = (float)(R.d[i] * sc);
        for (int k = 0; k < 6; k++) Rf.o[k] = {(float)(R.o[k].re * sc), (float)(R.o[k].im * sc)};
        eig_result<float> er = herm4_gate_eigvec(Rf, (float)cond, (float)(1.0 / cond), !gated, format != 0);
        *rank1 = er.rank1;
        *col0 = er.col0;
        if (gated && !er.rank1) return 0;
        float e[3];
        if (er.col0) { if (format == 0) normalise_foa_col0(er.u, e); else normalise_mic_col0(er.u, (float)dk, e); }
        else if (format == 0) normalise_foa(er.u, e, !gated);
        else normalise_mic(er.u, (float)dk, e);
        for (int i = 0; i < 3; i++) e3[i] = e[i];
        return 0;
    }
    eig_result<double> er = herm4_gate_eigvec(R, cond, 1.0 / cond, !gated, format != 0);
    *rank1 = er.rank1;
    *col0 = er.col0;
    if (gated && !er.rank1) return 0;
    if (er.col0) { if (format == 0) normalise_foa_col0(er.u, e3); else normalise_mic_col0(er.u, dk, e3); }
    else if (format == 0) normalise_foa(er.u, e3, !gated);
    else normalise_mic(er.u, dk, e3);
    return 0;
}
}

extern "C" {
// Round 4: the packed-float32 pair solve (herm4_gate_eigvec_pk) beside the float64 solve on the SAME float64 covariances, as
// the kernel forms them: item i = 8 consecutive frames x 4 channels of complex64 spectra of one bin; frame windows 0..6 and
// 1..7 are the pair.  format 0 = FOA, 1 = MIC (dk = delta * k).  Outputs per (item, frame j): rank64 (float64 gate),
// margin64 (q(mu1 / cond) on the trace-1..2 scale), e64[3]; pass32 / unsure32 flags and e32[3] of the packed solve.
int hostemu_pk_pairs(const float *X, long n, double cond, int format, double dk, int cov32, unsigned char *rank64,
                     double *margin64, double *e64, unsigned char *pass32, unsigned char *unsure32, float *e32)
{
    for (long i = 0; i < n; i++) {
        const float *x = X + i * 8 * 4 * 2;
        auto frame = [&](int k, herm4<double> &A) {
            cplx<double> v[4];
            for (int c = 0; c < 4; c++) v[c] = {(double)x[(k * 4 + c) * 2], (double)x[(k * 4 + c) * 2 + 1]};
            herm4_rank1_add(A, v);
        };
        herm4<double> Rc = {};
        for (int k = 1; k <= 6; k++) frame(k, Rc);
        herm4<double> R[2] = {Rc, Rc};
        frame(0, R[0]);
        frame(7, R[1]);
        herm4<pk2f> A;
        for (int j = 0; j < 2; j++) {
            eig_result<double> er = herm4_gate_eigvec<0>(R[j], cond, 1.0 / cond, false, format != 0);
            rank64[2 * i + j] = er.rank1;
            margin64[2 * i + j] = er.margin;
            double e[3] = {0, 0, 0};
            if (er.rank1) {
                if (er.col0) { if (format == 0) normalise_foa_col0(er.u, e); else normalise_mic_col0(er.u, dk, e); }
                else if (format == 0) normalise_foa(er.u, e, false);
                else normalise_mic(er.u, dk, e);
            }
            for (int c = 0; c < 3; c++) e64[(2 * i + j) * 3 + c] = e[c];
            const double tr = R[j].d[0] + R[j].d[1] + R[j].d[2] + R[j].d[3];
            const float sc = (float)pow2_unscale(tr); // (the kernel: convert, then scale by the power of two in float32)
            for (int c = 0; c < 4; c++) A.d[c][j] = (float)R[j].d[c] * sc;
            for (int k = 0; k < 6; k++) { A.o[k].re[j] = (float)R[j].o[k].re * sc; A.o[k].im[j] = (float)R[j].o[k].im * sc; }
        }
        int odd = 0;
        if (cov32) { // the kernel's float32 covariance: (re, im)-packed accumulation, shared frames first, then one more each
            auto fr = [&](int k, pk2f *v) { for (int c = 0; c < 4; c++) v[c] = pk2f{x[(k * 4 + c) * 2], x[(k * 4 + c) * 2 + 1]}; };
            cov4pk Cc = {}, C0, C1;
            pk2f v[4];
            for (int k = 1; k <= 6; k++) { fr(k, v); cov4pk_rank1(Cc, Cc, v); }
            fr(0, v); cov4pk_rank1(C0, Cc, v);
            fr(7, v); cov4pk_rank1(C1, Cc, v);
            A = herm4_pk_from_windows(C0, C1, odd);
        }
        pk_eig r = format ? herm4_gate_eigvec_pk<true>(A, (float)cond, (float)(1.0 / cond), 3 & ~odd)
                          : herm4_gate_eigvec_pk<false>(A, (float)cond, (float)(1.0 / cond), 3 & ~odd);
        r.unsure |= odd;
        pk2f e[3] = {pk_splat(0.f), pk_splat(0.f), pk_splat(0.f)};
        if (r.pass) {
            if (format == 0) normalise_foa_pk(r, e);
            else normalise_mic_pk(r, (float)dk, e);
        }
        for (int j = 0; j < 2; j++) {
            pass32[2 * i + j] = (r.pass >> j) & 1;
            unsure32[2 * i + j] = (r.unsure >> j) & 1;
            for (int c = 0; c < 3; c++) e32[(2 * i + j) * 3 + c] = ((r.pass >> j) & 1) ? e[c][j] : 0.f;
        }
    }
    return 0;
}
}

extern "C" {
// Round 6, the error budget of the packed solve's gate certificate: for every frame of every pair item
// (layout as hostemu_pk_pairs): t32 [2n][7] = the Taylor coefficients t0..t3 of the characteristic quartic at c = mu1 / cond, mu1, the
// last Newton step and c as the float32 pair code computes them from its float32 covariance; t64 [2n][7] = t0..t3 of the EXACT
// covariance's quartic (long double) evaluated at that same float32 c, the exact mu1 (Newton in long double), 0, c.  So
// |t32[k] - t64[k]|, k < 4, is the coefficient + evaluation error alone (what SALSA_PK_GATE_TOL must cover, whether or not Newton had
// converged), and |t32[4] - t64[4]| the error of mu1 (meaningful where the kernel's convergence test |last| <= 4e-6 mu1 holds).
int hostemu_pk_coeffs(const float *X, long n, double cond, double *t32, double *t64)
{
    typedef long double L;
    for (long i = 0; i < n; i++) {
        const float *x = X + i * 8 * 4 * 2;
        auto fr = [&](int k, pk2f *v) { for (int c = 0; c < 4; c++) v[c] = pk2f{x[(k * 4 + c) * 2], x[(k * 4 + c) * 2 + 1]}; };
        cov4pk Cc = {}, C0, C1;
        pk2f v[4];
        for (int k = 1; k <= 6; k++) { fr(k, v); cov4pk_rank1(Cc, Cc, v); }
        fr(0, v); cov4pk_rank1(C0, Cc, v);
        fr(7, v); cov4pk_rank1(C1, Cc, v);
        int odd = 0;
        const herm4<pk2f> A = herm4_pk_from_windows(C0, C1, odd);
        pk2f st[7];
        for (int q = 0; q < 7; q++) st[q] = pk_splat(0.f);
        (void)herm4_gate_eigvec_pk<false>(A, (float)cond, (float)(1.0 / cond), 3 & ~odd, st);
        for (int j = 0; j < 2; j++) {
            herm4<L> R = {};
            for (int k = j; k < j + 7; k++) {
                cplx<L> w[4];
                for (int c = 0; c < 4; c++) w[c] = {(L)x[(k * 4 + c) * 2], (L)x[(k * 4 + c) * 2 + 1]};
                herm4_rank1_add(R, w);
            }
            const double tr = (double)(R.d[0] + R.d[1] + R.d[2] + R.d[3]);
            double *o32 = t32 + (2 * i + j) * 7, *o64 = t64 + (2 * i + j) * 7;
            for (int q = 0; q < 7; q++) o32[q] = (double)st[q][j], o64[q] = 0.0;
            if (!(tr > 0.0) || ((odd >> j) & 1)) { for (int q = 0; q < 7; q++) o32[q] = o64[q] = NAN; continue; }
            // the power of two the float32 code scaled ITS trace by (a trace within rounding of a power of two may land on either side)
            const double tr32 = (double)A.d[0][j] + (double)A.d[1][j] + (double)A.d[2][j] + (double)A.d[3][j];
            const L sc = (L)exp2(rint(log2(tr32 / tr)));
            for (int c = 0; c < 4; c++) R.d[c] *= sc;
            for (int k = 0; k < 6; k++) { R.o[k].re *= sc; R.o[k].im *= sc; }
            L nrm = 0;
            for (int k = 0; k < 6; k++) nrm += R.o[k].re * R.o[k].re + R.o[k].im * R.o[k].im;
            const L e1 = R.d[0] + R.d[1] + R.d[2] + R.d[3];
            const L e2 = R.d[0] * R.d[1] + R.d[0] * R.d[2] + R.d[0] * R.d[3] + R.d[1] * R.d[2] + R.d[1] * R.d[3] + R.d[2] * R.d[3] - nrm;
            const minors4<L> m = herm4_minors(R);
            L dg[4];
            herm4_adj_diag(R, m, dg);
            const L e3 = dg[0] + dg[1] + dg[2] + dg[3];
            const L e4 = m.s0 * m.c5 - re_mul(m.s1, m.c4) + re_mul(m.s2, m.c3) + re_mul(m.s3, m.c2) - re_mul(m.s4, m.c1) + re_mul(m.s5, m.c0);
            const L a3 = -e1, a2 = e2, a1 = -e3, a0 = e4;
            L xr = e1; // Newton from the trace (>= mu1), monotone descent
            for (int it = 0; it < 400; it++) {
                const L q = (((xr + a3) * xr + a2) * xr + a1) * xr + a0;
                const L dq = ((4 * xr + 3 * a3) * xr + 2 * a2) * xr + a1;
                if (!(dq > 0)) break;
                const L stp = q / dq;
                if (!(stp > 0)) break;
                xr -= stp;
            }
            const L c = (L)st[6][j];
            o64[0] = (double)((((c + a3) * c + a2) * c + a1) * c + a0);
            o64[1] = (double)(((4 * c + 3 * a3) * c + 2 * a2) * c + a1);
            o64[2] = (double)((6 * c + 3 * a3) * c + a2);
            o64[3] = (double)(4 * c + a3);
            o64[4] = (double)xr;
            o64[6] = (double)c;
        }
    }
    return 0;
}
}
