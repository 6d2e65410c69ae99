// salsa_math.h -- per-thread arithmetic of the MI355X SALSA kernels (host+device inline templates).
//
// Everything here is register-level math with no memory traffic: the radix-8/4 butterflies of the LDS-staged FFT, the
// packed-real-FFT unpack, the 4x4 Hermitian covariance / eigen-gate / principal-eigenvector solver and the FOA / MIC
// normalisations.  The kernels in salsa_kernels.hip own all addressing and staging.  The same header compiles with
// g++ (tests/hostemu) so this arithmetic is unit-tested on CPU against the oracle before it ever runs on a GPU; that
// host build is a test harness, never a fallback of the product.
//
// Reference semantics implemented (paths relative to the upstream repo):
//   dataset/salsa_feature_extraction.py:99-127  covariance, coherence gate s0 > s1*cond, FOA / MIC normalisation
//   dataset/salsa_feature_extraction.py:53-93   3-frame RMS + noise-floor tracker step
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SALSA_HD __host__ __device__ __forceinline__
#else
#define SALSA_HD inline
#endif

namespace salsa {

template <typename T> struct cplx { T re, im; };

template <typename T> SALSA_HD cplx<T> cmul(cplx<T> a, cplx<T> b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <typename T> SALSA_HD cplx<T> cmulc(cplx<T> a, cplx<T> b) { return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; } // a*conj(b)
template <typename T> SALSA_HD cplx<T> cadd(cplx<T> a, cplx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T> SALSA_HD cplx<T> csub(cplx<T> a, cplx<T> b) { return {a.re - b.re, a.im - b.im}; }
template <typename T> SALSA_HD cplx<T> cconj(cplx<T> a) { return {a.re, -a.im}; }
template <typename T> SALSA_HD cplx<T> cmul_mi(cplx<T> a) { return {a.im, -a.re}; } // a * (-i)

// ---------------------------------------------------------------------------------------------------------------
// np.pad(mode='reflect') index (librosa.stft center=True): period 2N-2, no edge duplication.
SALSA_HD long reflect_index(long i, long N)
{
    if (i >= 0 && i < N) return i;
    if (N == 1) return 0;
    long p = 2 * (N - 1);
    i %= p;
    if (i < 0) i += p;
    return i < N ? i : p - i;
}

// ---------------------------------------------------------------------------------------------------------------
// In-register forward DFTs (decimation in frequency).  Outputs in natural order.
template <typename T> SALSA_HD void dft2(cplx<T> &a, cplx<T> &b)
{
    cplx<T> t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

template <typename T> SALSA_HD void dft4(cplx<T> *x) // x[0..3] -> X[0..3]
{
    cplx<T> e0 = cadd(x[0], x[2]), o0 = csub(x[0], x[2]);
    cplx<T> e1 = cadd(x[1], x[3]), o1 = cmul_mi(csub(x[1], x[3]));
    x[0] = cadd(e0, e1);
    x[2] = csub(e0, e1);
    x[1] = cadd(o0, o1);
    x[3] = csub(o0, o1);
}

template <typename T> SALSA_HD void dft8(cplx<T> *x) // x[0..7] -> X[0..7]
{
    const T h = (T)0.70710678118654752440;
    cplx<T> s[4], d[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        s[j] = cadd(x[j], x[j + 4]);
        d[j] = csub(x[j], x[j + 4]);
    }
    // d[j] *= W8^j : W8^1 = (1-i)/sqrt2, W8^2 = -i, W8^3 = (-1-i)/sqrt2
    d[1] = {(d[1].re + d[1].im) * h, (d[1].im - d[1].re) * h};
    d[2] = cmul_mi(d[2]);
    d[3] = {(d[3].im - d[3].re) * h, -(d[3].re + d[3].im) * h};
    dft4(s);
    dft4(d);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        x[2 * m] = s[m];
        x[2 * m + 1] = d[m];
    }
}

template <int R, typename T> SALSA_HD void dftR(cplx<T> *x)
{
    if (R == 8) dft8(x);
    else if (R == 4) dft4(x);
    else dft2(x[0], x[1]);
}

// Two real channels packed as z = x0 + i*x1, Z = FFT(z).  a = Z[k], b = Z[(N-k) mod N]:
//   X0[k] = (a + conj(b))/2 ,  X1[k] = (a - conj(b))/(2i)
template <typename T> SALSA_HD void unpack_pair(cplx<T> a, cplx<T> b, cplx<T> &x0, cplx<T> &x1)
{
    x0 = {(a.re + b.re) * (T)0.5, (a.im - b.im) * (T)0.5};
    x1 = {(a.im + b.im) * (T)0.5, (b.re - a.re) * (T)0.5};
}

// ---------------------------------------------------------------------------------------------------------------
// 4x4 Hermitian matrix: 4 real diagonals + 6 complex upper off-diagonals in the order (01,02,03,12,13,23).
template <typename T> struct herm4 {
    T d[4];
    cplx<T> o[6];
};

// acc += x x^H  (x = 4 channels of one frame).  salsa_feature_extraction.py:99-100 (the /7 is a common scale that
// neither the gate s0 > s1*cond nor the eigenvector depends on; it is dropped).
template <typename T> SALSA_HD void herm4_rank1_add(herm4<T> &A, const cplx<T> *x)
{
#pragma unroll
    for (int i = 0; i < 4; i++) A.d[i] += x[i].re * x[i].re + x[i].im * x[i].im;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i + 1; j < 4; j++, k++) {
            A.o[k].re += x[i].re * x[j].re + x[i].im * x[j].im;
            A.o[k].im += x[i].im * x[j].re - x[i].re * x[j].im;
        }
}

// C = A*A (Hermitian)
template <typename T> SALSA_HD herm4<T> herm4_square(const herm4<T> &A)
{
    herm4<T> C;
    const cplx<T> a01 = A.o[0], a02 = A.o[1], a03 = A.o[2], a12 = A.o[3], a13 = A.o[4], a23 = A.o[5];
    T n01 = a01.re * a01.re + a01.im * a01.im, n02 = a02.re * a02.re + a02.im * a02.im;
    T n03 = a03.re * a03.re + a03.im * a03.im, n12 = a12.re * a12.re + a12.im * a12.im;
    T n13 = a13.re * a13.re + a13.im * a13.im, n23 = a23.re * a23.re + a23.im * a23.im;
    C.d[0] = A.d[0] * A.d[0] + n01 + n02 + n03;
    C.d[1] = A.d[1] * A.d[1] + n01 + n12 + n13;
    C.d[2] = A.d[2] * A.d[2] + n02 + n12 + n23;
    C.d[3] = A.d[3] * A.d[3] + n03 + n13 + n23;
    cplx<T> t;
    T s;
    s = A.d[0] + A.d[1]; t = cadd(cmulc(a02, a12), cmulc(a03, a13));            // a02 conj(a12) + a03 conj(a13)
    C.o[0] = {a01.re * s + t.re, a01.im * s + t.im};
    s = A.d[0] + A.d[2]; t = cadd(cmul(a01, a12), cmulc(a03, a23));             // a01 a12 + a03 conj(a23)
    C.o[1] = {a02.re * s + t.re, a02.im * s + t.im};
    s = A.d[0] + A.d[3]; t = cadd(cmul(a01, a13), cmul(a02, a23));              // a01 a13 + a02 a23
    C.o[2] = {a03.re * s + t.re, a03.im * s + t.im};
    s = A.d[1] + A.d[2]; t = cadd(cmul(cconj(a01), a02), cmulc(a13, a23));      // conj(a01) a02 + a13 conj(a23)
    C.o[3] = {a12.re * s + t.re, a12.im * s + t.im};
    s = A.d[1] + A.d[3]; t = cadd(cmul(cconj(a01), a03), cmul(a12, a23));       // conj(a01) a03 + a12 a23
    C.o[4] = {a13.re * s + t.re, a13.im * s + t.im};
    s = A.d[2] + A.d[3]; t = cadd(cmul(cconj(a02), a03), cmul(cconj(a12), a13)); // conj(a02) a03 + conj(a12) a13
    C.o[5] = {a23.re * s + t.re, a23.im * s + t.im};
    return C;
}

template <typename T> SALSA_HD T herm4_frob2(const herm4<T> &A) // ||A||_F^2 = tr(A^2)
{
    T s = A.d[0] * A.d[0] + A.d[1] * A.d[1] + A.d[2] * A.d[2] + A.d[3] * A.d[3];
    T o = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) o += A.o[k].re * A.o[k].re + A.o[k].im * A.o[k].im;
    return s + 2 * o;
}

template <typename T> SALSA_HD T herm4_trace_prod(const herm4<T> &A, const herm4<T> &B) // tr(A B)
{
    T s = A.d[0] * B.d[0] + A.d[1] * B.d[1] + A.d[2] * B.d[2] + A.d[3] * B.d[3];
    T o = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) o += A.o[k].re * B.o[k].re + A.o[k].im * B.o[k].im;
    return s + 2 * o;
}

// element (i,j) of a Hermitian matrix from its packed upper triangle
template <typename T> SALSA_HD cplx<T> herm4_at(const herm4<T> &A, int i, int j)
{
    if (i == j) return {A.d[i], (T)0};
    const int idx[4][4] = {{-1, 0, 1, 2}, {0, -1, 3, 4}, {1, 3, -1, 5}, {2, 4, 5, -1}};
    cplx<T> v = A.o[idx[i][j]];
    return i < j ? v : cconj(v);
}

template <typename T> SALSA_HD cplx<T> det3(cplx<T> a, cplx<T> b, cplx<T> c, cplx<T> d, cplx<T> e, cplx<T> f, cplx<T> g,
                                            cplx<T> h, cplx<T> i)
{
    // | a b c ; d e f ; g h i |
    cplx<T> m0 = csub(cmul(e, i), cmul(f, h));
    cplx<T> m1 = csub(cmul(d, i), cmul(f, g));
    cplx<T> m2 = csub(cmul(d, h), cmul(e, g));
    return cadd(csub(cmul(a, m0), cmul(b, m1)), cmul(c, m2));
}

// Result of the per-TF-bin solve.
template <typename T> struct eig_result {
    bool rank1;      // coherence test s0 > s1*cond (salsa_feature_extraction.py:106)
    T margin;        // diagnostic: q(c)-based signed distance proxy (negative = one root above c)
    cplx<T> u[4];    // principal eigenvector (arbitrary scale and phase); valid when computed
};

// Gate + principal eigenvector of a Hermitian PSD 4x4 R (any positive scale).
//
// Eigenvalues: R is normalised to trace 1 (A); the characteristic quartic q comes from the power sums tr(A^k), k<=4,
// through Newton's identities; mu1 = largest root by Newton's iteration started at sqrt(tr A^2) >= mu1, which
// descends monotonically onto mu1 for a real-rooted polynomial.  The gate mu1 > cond*mu2 is decided WITHOUT solving
// for mu2: by Budan-Fourier (exact for real-rooted polynomials) the number of roots above c = mu1/cond equals the sign
// variations of (q, q', q''/2, q'''/6, 1) at c; the bin is rank-1 iff that count is exactly one.
// Eigenvector: a column of adj(A - mu1 I) = prod_{i>=2}(mu_i - mu1) u u^H, taking the column with the largest
// diagonal cofactor.  Exact for any spectral gap (also with tracking off, where no gate protects the solve).
template <typename T> SALSA_HD eig_result<T> herm4_gate_eigvec(const herm4<T> &R, T cond, bool need_vector_always)
{
    eig_result<T> res;
    res.rank1 = false;
    res.margin = 0;
    res.u[0] = {(T)1, (T)0};
    res.u[1] = res.u[2] = res.u[3] = {(T)0, (T)0};
    const T tr = R.d[0] + R.d[1] + R.d[2] + R.d[3];
    if (!(tr > (T)0)) return res; // zero matrix: s0 > s1*cond is 0 > 0 = False; LAPACK's U for it is the identity
    const T inv = (T)1 / tr;
    herm4<T> A;
#pragma unroll
    for (int i = 0; i < 4; i++) A.d[i] = R.d[i] * inv;
#pragma unroll
    for (int k = 0; k < 6; k++) A.o[k] = {R.o[k].re * inv, R.o[k].im * inv};
    const herm4<T> A2 = herm4_square(A);
    const T p1 = A.d[0] + A.d[1] + A.d[2] + A.d[3]; // == 1 up to rounding
    const T p2 = A2.d[0] + A2.d[1] + A2.d[2] + A2.d[3];
    const T p3 = herm4_trace_prod(A, A2);
    const T p4 = herm4_frob2(A2);
    const T e1 = p1;
    const T e2 = (e1 * p1 - p2) * (T)0.5;
    const T e3 = (e2 * p1 - e1 * p2 + p3) * (T)(1.0 / 3.0);
    const T e4 = (e3 * p1 - e2 * p2 + e1 * p3 - p4) * (T)0.25;
    const T a3 = -e1, a2 = e2, a1 = -e3, a0 = e4; // q(x) = x^4 + a3 x^3 + a2 x^2 + a1 x + a0
    T x = sqrt(p2);
    if (x > p1) x = p1;
    for (int it = 0; it < 48; it++) {
        T q = (((x + a3) * x + a2) * x + a1) * x + a0;
        T dq = (((T)4 * x + (T)3 * a3) * x + (T)2 * a2) * x + a1;
        if (!(dq > (T)0)) break;
        T step = q / dq;
        if (!(step > (T)0)) break; // at (or rounded past) the root
        T xn = x - step;
        if (!(xn < x)) break;
        x = xn;
        if (step <= (T)4 * (sizeof(T) == 8 ? (T)2.2e-16 : (T)1.2e-7) * x) break;
    }
    const T mu1 = x;
    if (cond <= (T)1) {
        res.rank1 = mu1 > (T)0; // mu2*cond < mu1 whenever mu1 > 0 (cond == 1: strict s0 > s1, a measure-zero tie)
    } else {
        const T c = mu1 / cond;
        T t0 = (((c + a3) * c + a2) * c + a1) * c + a0;
        T t1 = (((T)4 * c + (T)3 * a3) * c + (T)2 * a2) * c + a1;
        T t2 = ((T)6 * c + (T)3 * a3) * c + a2;
        T t3 = (T)4 * c + a3;
        int var = 0;
        T prev = (T)1; // q''''/24 = 1
        if (t3 != (T)0) { var += ((t3 < 0) != (prev < 0)); prev = t3; }
        if (t2 != (T)0) { var += ((t2 < 0) != (prev < 0)); prev = t2; }
        if (t1 != (T)0) { var += ((t1 < 0) != (prev < 0)); prev = t1; }
        if (t0 != (T)0) { var += ((t0 < 0) != (prev < 0)); prev = t0; }
        res.rank1 = (var == 1);
        res.margin = t0;
    }
    if (!res.rank1 && !need_vector_always) return res;
    // B = A - mu1 I ; adj(B) column with the largest |diagonal cofactor|
    herm4<T> B = A;
#pragma unroll
    for (int i = 0; i < 4; i++) B.d[i] -= mu1;
    const cplx<T> b00 = {B.d[0], 0}, b11 = {B.d[1], 0}, b22 = {B.d[2], 0}, b33 = {B.d[3], 0};
    const cplx<T> b01 = B.o[0], b02 = B.o[1], b03 = B.o[2], b12 = B.o[3], b13 = B.o[4], b23 = B.o[5];
    const cplx<T> b10 = cconj(b01), b20 = cconj(b02), b30 = cconj(b03), b21 = cconj(b12), b31 = cconj(b13), b32 = cconj(b23);
    // diagonal cofactors (real)
    T c00 = det3(b11, b12, b13, b21, b22, b23, b31, b32, b33).re;
    T c11 = det3(b00, b02, b03, b20, b22, b23, b30, b32, b33).re;
    T c22 = det3(b00, b01, b03, b10, b11, b13, b30, b31, b33).re;
    T c33 = det3(b00, b01, b02, b10, b11, b12, b20, b21, b22).re;
    // adj(B)_{ij} = cofactor_{ji} = (-1)^{i+j} det(B without row j, col i).  For Hermitian B adj is Hermitian.
    // upper entries adj01, adj02, adj03, adj12, adj13, adj23:
    cplx<T> m;
    m = det3(b01, b02, b03, b21, b22, b23, b31, b32, b33); const cplx<T> adj01 = {-m.re, -m.im}; // remove row1,col0
    m = det3(b01, b02, b03, b11, b12, b13, b31, b32, b33); const cplx<T> adj02 = m;              // remove row2,col0
    m = det3(b01, b02, b03, b11, b12, b13, b21, b22, b23); const cplx<T> adj03 = {-m.re, -m.im}; // remove row3,col0
    m = det3(b00, b02, b03, b10, b12, b13, b30, b32, b33); const cplx<T> adj12 = {-m.re, -m.im}; // remove row2,col1
    m = det3(b00, b02, b03, b10, b12, b13, b20, b22, b23); const cplx<T> adj13 = m;              // remove row3,col1
    m = det3(b00, b01, b03, b10, b11, b13, b20, b21, b23); const cplx<T> adj23 = {-m.re, -m.im}; // remove row3,col2
    T m0 = fabs(c00), m1 = fabs(c11), m2 = fabs(c22), m3 = fabs(c33);
    int j = 0;
    T best = m0;
    if (m1 > best) { best = m1; j = 1; }
    if (m2 > best) { best = m2; j = 2; }
    if (m3 > best) { best = m3; j = 3; }
    // column j of adj: u_i = adj_{ij}
    if (j == 0) { res.u[0] = {c00, 0}; res.u[1] = cconj(adj01); res.u[2] = cconj(adj02); res.u[3] = cconj(adj03); }
    else if (j == 1) { res.u[0] = adj01; res.u[1] = {c11, 0}; res.u[2] = cconj(adj12); res.u[3] = cconj(adj13); }
    else if (j == 2) { res.u[0] = adj02; res.u[1] = adj12; res.u[2] = {c22, 0}; res.u[3] = cconj(adj23); }
    else { res.u[0] = adj03; res.u[1] = adj13; res.u[2] = adj23; res.u[3] = {c33, 0}; }
    if (!(best > (T)0)) { // A == mu1 I numerically (fully degenerate): any vector; match LAPACK's identity column
        res.u[0] = {(T)1, (T)0};
        res.u[1] = res.u[2] = res.u[3] = {(T)0, (T)0};
    }
    return res;
}

// FOA: real(u[1:]/u[0]) then L2-normalise (salsa_feature_extraction.py:118-119; no guard, IEEE inf/nan propagate).
template <typename T> SALSA_HD void normalise_foa(const cplx<T> *u, T *e)
{
    const T den = u[0].re * u[0].re + u[0].im * u[0].im;
    T ss = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        e[i] = (u[i + 1].re * u[0].re + u[i + 1].im * u[0].im) / den;
        ss += e[i] * e[i];
    }
    ss = sqrt(ss);
#pragma unroll
    for (int i = 0; i < 3; i++) e[i] = e[i] / ss;
}

// MIC: angle(u[1:]*conj(u[0])) / (delta*k) (salsa_feature_extraction.py:121-123); dk = delta*(ibin+lower_bin).
template <typename T> SALSA_HD void normalise_mic(const cplx<T> *u, T dk, T *e)
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        cplx<T> w = cmulc(u[i + 1], u[0]);
        e[i] = atan2(w.im, w.re) / dk;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stockham autosort FFT addressing, radix R, N points, N/R threads per transform.  Pass with sub-transform length p
// (p = 1, R, R^2, ... < N), thread i in [0, N/R), k = i & (p-1):
//   reads   x[i + r*(N/R)]                      r = 0..R-1
//   scales  element r by W_N^(r * k * N/(p*R))  (W_N = exp(-2 pi i / N))
//   writes  y[(i-k)*R + k + r*p]
// After the last pass y is the DFT in natural order.
SALSA_HD int stockham_in(int i, int r, int N, int R) { return i + r * (N / R); }
SALSA_HD int stockham_tw(int i, int r, int p, int N, int R) { return r * (i & (p - 1)) * (N / (p * R)); }
SALSA_HD int stockham_out(int i, int r, int p, int R) { int k = i & (p - 1); return (i - k) * R + k + r * p; }

// ---------------------------------------------------------------------------------------------------------------
// Noise-floor tracker step (salsa_feature_extraction.py:65-87).  Returns indicator_sig.
struct tracker_state {
    double floor;
    int countdown;
};

SALSA_HD bool tracker_step(tracker_state &s, double mag)
{
    const double floor_up = 1 + 0.02, floor_up_slow = 1 + 0.1 * 0.02, floor_down = 1 - 0.02, snr_ratio = 1.5;
    if (mag > s.floor) {
        s.countdown -= 1;
        s.floor = (s.countdown < 0 ? floor_up_slow : floor_up) * s.floor;
    } else {
        s.countdown = 3;
        s.floor = floor_down * s.floor;
    }
    if (s.floor < 1e-6) s.floor = 1e-6;
    return mag > snr_ratio * s.floor;
}

} // namespace salsa
