// salsa_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the SALSA / SALSA-Lite feature extractor + the C ABI
// declared in include/salsa_hip.h.  Written for wave64 / 160 KiB LDS / HBM3E; no CUDA or multi-backend paths.
//
// Pipeline of salsa_extract_batch (full SALSA), all on the caller's stream:
//   K1 stft_kernel      one wave per packed 512-point complex FFT (two real channels), Stockham radix-8 through LDS;
//                       unpacks to the 4 channel spectra, writes the log-spectrogram channels 0-3 straight to the
//                       output and spills the DOA band of the spectra (float32-rounded, like the reference's
//                       complex64 STFT) to the workspace as Xs[b][t][channel pair][bin] (float4).
//   K2 tracker_kernel   one lane per (clip, bin): 3-frame RMS of channel 0 and the sequential noise-floor tracker
//                       in float64 -> valid32[b][32-bin group][t] (per-frame indicator mask of the group's bins).
//   K3 cov_eig_kernel   per tile of 8 frames x 256 bins the gated TF bins are compacted into an LDS work list (an item = two
//                       neighbouring frames of one bin, at least one gated in: their 7-frame windows share 6 frames);
//                       one lane per item: Hermitian covariances accumulated in registers, eigen-gate + principal
//                       eigenvector (salsa_math.h), FOA / MIC normalisation, writes channels 4-6 (zeros where gated).
// Further entry points: salsa_eigvec_batch (K2 + K3 on caller-supplied spectra), salsa_logspec_batch (K1 only), the scaler /
// normalise kernels, the contrib-surface variants (SALSA_FLAG_FLEX; salsa_extract_multichannel for 5 - 8 microphones: K1 over
// 3 / 4 channel pairs + cov_eig_n_kernel), salsa_to_freq_major, salsa_augment_batch, the pipelined schedules
// (salsa_plan_set_pipeline).
// SALSA-Lite / IPD is K1 alone (log-spectrogram + inter-channel phase fused into the unpack).
//
// Arithmetic types follow the reference (see DESIGN.md "Precision"): STFT evaluated in float64 and rounded to
// float32, log-spectrogram in float32, tracker / covariance / eigen-solve in float64.
#include "build_guard.h" // probe switches need -DSALSA_PROBE_BUILD; SALSA_BUILD_FLAGS (generated: tools/gen_build_guard.py)
#include <hip/hip_runtime.h>
#include <type_traits>
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
    int N;    // samples per channel (host checks 4*N < 2^31)
    int T;    // frames (host checks 7*T*F < 2^31 and the per-clip spill < 2^31 elements)
    int hop;
    int lower, upper, nd; // DOA band [lower, upper), nd = upper - lower
    int cutoff;           // lite: spectrogram band [lower, cutoff)
    int F;                // feature bins per frame
    int OC;               // output channels (7, or 4 for logspec-only)
    int ident;            // identity rows of W (192 | 96 | n_fft/2)
    int spec_lo, spec_hi; // bins [spec_lo, spec_hi) map to spectrogram rows k - spec_lo (1 .. ident+1 for the dataset scripts)
    int flex;             // contrib/salsa_flexible.py semantics (SALSA_FLAG_FLEX): raw-|X0| tracker, gate without tracking, ...
    int compress;
    int layout;
    int feature;          // SALSA_FEATURE_* ; 3 = logspec only
    int format;
    int tracking;
    int n_hop;
    int pair_sel;         // K1: -1 = both channel pairs of every frame in one launch; 0 / 1 = only channels {0,1} / {2,3}
    int nch;              // audio channels per clip (even): 4, or 6 / 8 on the multichannel contrib surface; OC = 2*nch - 1
    double cond;
    double inv_cond;      // 1/cond (0 when cond == 0: unused, cond <= 1 short-circuits the gate)
    double delta;         // 2 pi fs / (n_fft * 343)
    double snr_ratio;     // indicator_sig = mag > snr_ratio * floor (1.5, :36; contrib: floor_mask_ratio)
    const float *sc_mean; // optional fused normalise-on-load of the spectrogram channels: [4][F] mean / std, or NULL
    const float *sc_std;
    unsigned long long *stats; // optional solver counters (salsa_plan_set_stats), or NULL
    int force_f64;             // SALSA_FLAG_FORCE_F64: the float64 instantiation of the covariance / eigen kernel
    unsigned *doubt32;         // [B][32-bin group][T] bit mask of the TF bins whose coherence test the quartic could not decide
                               // (salsa_math.h SALSA_GATE_DOUBT; decided by gate_doubt_kernel after the launch), or NULL (ungated plans)
    unsigned *doubt_flag;      // one word per launch group: non-zero once ANY bin was flagged (zeroed by the tracker launch before the
                               // covariance / eigen launch, or by a memset on the tracker-less gated path): gate_doubt_kernel reads it and exits
};

constexpr int FEATURE_LOGSPEC_ONLY = 3;

// ------------------------------------------------------------------------------------------------------------ K1
// STFT + log-spectrogram (+ spill of the DOA band, or the SALSA-Lite phase features).
//
// Work item of ONE WAVE = one packed N-point complex FFT z = w*(y_c0 + i*y_c1) of a channel pair of one frame:
// 64 lanes x R points, Stockham radix-R passes exchanged through a wave-private LDS buffer (no workgroup barrier:
// the LDS serves a wave's DS instructions in order, so a wave-level scheduling fence is all the passes need).
// LDS element e lives at slot e ^ ((e>>3)&7): with 16-byte complex-float64 elements this XOR swizzle makes every
// pass's stride-R scatter (ds_write_b128, 8-lane groups) AND the unit-stride gather (ds_read_b128, its odd 16-lane
// groups) bank-conflict free without padding.  After the last pass lane L holds Z[L + 64 r] in registers, which is
// exactly the ownership the unpack wants: X_c0[k], X_c1[k] need Z[k] and Z[N-k], and Z[N-k] of k = L + 64 r sits in
// register R-1-r of lane 64-L, so the mirror is fetched with cross-lane shuffles (R/2 complex values) instead of a
// third trip through LDS.  The spectra are rounded to float32 like the reference's complex64 STFT and used straight
// from registers: 10*log10 of the power into output channels c0, c1 and the DOA band of both channels as ONE float4
// per bin into the spill.  A wave walks K1_NF consecutive frames x 2 pairs (no software prefetch: the registers it would
// need cost a wave per SIMD, measured slower); the first twiddle of each pass lives in registers, the window in LDS.
template <int N> struct fft_cfg {
    static constexpr int R = (N == 512) ? 8 : 4;                 // points per lane; 64 lanes per transform either way
    static constexpr int NP = (N == 512) ? 2 : 3;                // twiddled passes (p = R, R^2, ...)
};

__device__ __forceinline__ int swz(int e) { return e ^ ((e >> 3) & 7); }

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Store policy (round 5, profiles/r5_ab_nt_stores.txt): the 16-byte write-once streams -- K1's spill (K1_SPILL_NT) and K3's rows of
// channels 4 - 6 (K3_OUT_NT) -- leave as NON-TEMPORAL stores (nothing re-reads them before they leave the L2, and the audio lines
// consecutive frames share stay cached longer); the 4-byte spectrogram rows stay PLAIN stores: the write-back L2 merges them into
// full lines, which nt stores forgo (all-nt was measured slower, 0.54 vs 0.49 ms).  The fused kernel's ring / row stores are plain.
// |x|^2 of a complex64 spectrum value in float32, as ONE explicitly written FMA of an explicitly rounded product.  Written
// `x.x * x.x + x.y * x.y` the compiler is free to contract it either way round (or not at all), and did so differently in two
// unrolled instances of the STFT kernel once the code around it changed (session 3: the spectrogram of bins 128 - 191 moved by one
// ulp against the fused kernel's, which the bit-identity test of the two schedules caught).
__device__ __forceinline__ float power32(const float2 x)
{
    const float t = x.x * x.x;
    return __builtin_fmaf(x.y, x.y, t);
}
// max(a, b) as ONE v_max_f32: fmaxf() first canonicalises its operand (a second v_max_f32 v, v, v per value -- a quieting no-op for
// anything but a signalling NaN, which no arithmetic here produces); same result, NaN handling included (IEEE maxNum)
__device__ __forceinline__ float max_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float db10(float p) { return 3.01029995663981195f * __log2f(max_raw(1e-10f, p)); } // 10*log10(max(1e-10,p))

// Addressing: a wave-uniform base pointer (SGPR pair) + a 32-bit unsigned BYTE offset per lane lets the compiler use the
// "saddr" form of global loads/stores; `ptr[int_index]` instead costs two or three 64-bit VALU instructions per access
// (sign extension, shift, 64-bit add), and K1 / K3 are VALU-issue bound.  The host checks that a clip's arrays stay
// below 4 GiB.
template <typename V> __device__ __forceinline__ void st_off(V *base, unsigned byte_off, const V v) { *(V *)((char *)base + byte_off) = v; }
template <typename V> __device__ __forceinline__ V ld_off(const V *base, unsigned byte_off) { return *(const V *)((const char *)base + byte_off); }
// base + zero-extended lane offset + a COMPILE-TIME byte constant added after the extension: the constant lands in the load's
// immediate offset field (inside the 32-bit sum it cannot -- the unsigned addition may wrap as far as the compiler knows -- and
// costs a v_add_u32 per load)
template <typename V> __device__ __forceinline__ V ld_off_c(const V *base, unsigned byte_off, int const_bytes) { return *(const V *)((const char *)base + byte_off + const_bytes); }
// Production default since round 5 (0 = plain stores, for A/B): the spill's 16-byte stores as non-temporal stores ALONE (whole 1-KB
// wave stores that nothing re-reads before they leave the L2); the 4-byte spectrogram rows stay plain (see the store policy above).
#ifndef K1_SPILL_NT
#define K1_SPILL_NT 1
#endif
typedef float salsa_f4v __attribute__((ext_vector_type(4)));
#ifndef TR_NT_LD
#define TR_NT_LD 0 // probe: the tracker's read-once 8-byte loads of channel 0 as non-temporal loads
#endif
#ifndef K3_GATHER_NT
#define K3_GATHER_NT 0 // probe: the covariance kernel's 16-byte gathers as non-temporal loads (they ARE re-read: the +-3-frame halo)
#endif
typedef float salsa_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_off_nt(const float4 *base, unsigned byte_off)
{
    const salsa_f4v x = __builtin_nontemporal_load((const salsa_f4v *)((const char *)base + byte_off));
    return make_float4(x.x, x.y, x.z, x.w);
}
__device__ __forceinline__ void st_off_nt(float4 *base, unsigned byte_off, const float4 v)
{
    salsa_f4v x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, (salsa_f4v *)((char *)base + byte_off));
}

// frames per wave.  Full SALSA: 4 (8 reuses more overlap per wave but leaves a 29 %-full last round of workgroups: measured
// 2 % slower).  SALSA-Lite, whose items are frame-major and heavier (phase rows instead of the spill): 8 (4 measured 4 % slower)
// K1_INTERLEAVE (round-5 probe): the four waves of a workgroup take frames base + w, base + w + 4, ... instead of four consecutive
// frames each, so the 212 samples per channel two consecutive frames share are requested by sibling waves at about the same time.
#ifndef K1_INTERLEAVE
#define K1_INTERLEAVE 0
#endif
#ifndef K1_NF_FULL
#define K1_NF_FULL 4
#endif
#ifndef K1_STD
#define K1_STD 1 // the STD instantiations of stft_kernel for the dataset scripts' layout (0: always the general kernel; bit-identical)
#endif
// angle(X_c conj(X_0)) of SALSA-Lite / IPD (salsa_lite_feature_extraction.py:111) in float32, with the roundings spelled out: one
// product rounded, the other fused into the sum.  Left to the compiler, `a * b + c * d` was contracted one way in the general
// instantiation and the other way in the Lite STD one (1 % of the phases differed in the last bit); every instantiation calls this.
__device__ __forceinline__ float lite_phase(const float2 xc, const float2 x0)
{
#pragma clang fp contract(off)
    float wr = fmaf(xc.x, x0.x, xc.y * x0.y);
    float wi = fmaf(xc.y, x0.x, -(xc.x * x0.y));
    const float m = fmaxf(fabsf(wr), fabsf(wi));
    if (m < 1e-30f && m > 0.f) { // products of tiny spectra: redo the product scaled up (exact)
        const float sx = 0x1p60f;
        wr = fmaf(xc.x * sx, x0.x * sx, (xc.y * sx) * (x0.y * sx));
        wi = fmaf(xc.y * sx, x0.x * sx, -((xc.x * sx) * (x0.y * sx)));
    }
    return atan2f(wi, wr);
}

// SALSA-Lite STD instantiation: channel 0's spectrum of the wave's frame, kept from the pair-0 item for the pair-1 item, lives in LDS
// (16 KB per workgroup) instead of ten registers per lane.  A function-local __shared__ array of a template that only the Lite STD
// instantiation calls, so that no other instantiation's LDS layout moves (an 8-byte dummy array cost the full-SALSA kernel 2.5 %).
constexpr int K1_LITE_X0_SLOTS = 8; // per wave: one slot per bin register (frame-major order) or per frame of the wave (pair-major order, phase band in register 0)
template <bool ON> struct k1_x0_store {
    static __device__ __forceinline__ float2 *get()
    {
        __shared__ float2 a[4 * K1_LITE_X0_SLOTS * 64];
        return a;
    }
};
template <> struct k1_x0_store<false> {
    static __device__ __forceinline__ float2 *get() { return nullptr; }
};
#ifndef K1_LITE_STD
#define K1_LITE_STD 1 // the STD instantiation of the SALSA-Lite / IPD kernel (planar 4-channel audio, n_fft 512, no scaler); 0: the general kernel (bit-identical)
#endif
#ifndef K1_LITE_PAIR_MAJOR
#define K1_LITE_PAIR_MAJOR 1 // Lite STD: pair-major item order when the phase band fits bin register 0 (0: always frame-major; bit-identical)
#endif
#ifndef K1_LITE_STD_WAVES
#define K1_LITE_STD_WAVES 3 // workgroups per CU the Lite STD instantiation is compiled for
#endif
#ifndef K1_LITE_WAVES
#define K1_LITE_WAVES 1 // workgroups per CU the SALSA-Lite / IPD instantiations are compiled for (1: no register cap -> 184 VGPRs, 2 waves per SIMD)
#endif
#ifndef K1_NF_LITE
#define K1_NF_LITE 8 // frames per wave of the SALSA-Lite / IPD instantiations
#endif
template <bool LITE> struct k1_cfg {
    static constexpr int NF = LITE ? K1_NF_LITE : K1_NF_FULL;
};

// SC (round 4): the instantiation launched when a scaler is attached keeps the [4][F] mean / std tables in LDS.  It is a separate
// instantiation because the 8 KB of LDS and the 168-register cap cost the plain path 4 % (0.458 -> 0.477 ms) when they were
// unconditional; SC = false is the round-3 kernel, bit for bit (and still honours a scaler, through global loads).
// STD (round 6, the ISA audit of profiles/r6_k1_isa.txt): the dataset scripts' spectrogram layout known at COMPILE time -- n_fft 512,
// planar audio, both channel pairs in one launch, high-frequency compression on (rows 0..191 = bins 1..192, rows 192..199 = eight
// bins each), and no scaler unless SC holds it in LDS.  Of the ~900 instructions an item issued, ~330 were not the FFT: sixteen
// load addresses rebuilt per item (the stride was a run-time value: planar | interleaved), a run-time 7-iteration tail loop in the
// compressed rows, the Nyquist bin unpacked and tested although W never reads it (:163-171), a scaler branch + division sequence
// compiled into all eleven store sites, and exec-mask juggling around band tests whose answers are fixed per register.  Same
// arithmetic in the same order: outputs are bit-identical (tests: goldens, fused-vs-three-kernel identity).  The DOA band stays a
// run-time range (FOA 1..192, MIC 1..85, any fmin / fmax).
template <int N, typename T, bool LITE, int NF, int NPAIRS = 2, bool SC = false, bool STD = false>
__global__ __launch_bounds__(256, LITE ? (STD ? K1_LITE_STD_WAVES : K1_LITE_WAVES) : SC ? 3 : 1) void stft_kernel(const KParams kp, const float *__restrict__ audio,
                                                   const double *__restrict__ window,
                                                   const cplx<double> *__restrict__ tw, float *__restrict__ out,
                                                   float4 *__restrict__ Xs)
{
    constexpr int R = fft_cfg<N>::R;
    constexpr int NP = fft_cfg<N>::NP;
    constexpr int NB = N / 2 + 1;
    __shared__ cplx<T> buf[4][N];
    __shared__ __attribute__((aligned(16))) float pw[4][2][64]; // powers of the compressed band (<= 63 bins) of the wave's two channels (STD reads them 16 bytes at a time)

    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y, bx = blockIdx.x;
    const int Ns = kp.N, Tn = kp.T;
    constexpr int K1_NF = NF;
    constexpr int K1_TSTEP = K1_INTERLEAVE ? 4 : 1;
    const int t_begin = K1_INTERLEAVE ? bx * 4 * K1_NF + w : (bx * 4 + w) * K1_NF;
    cplx<T> *z = buf[w];

    // Register diet (occupancy): only the first twiddle of each pass stays in registers, its powers are rebuilt by a
    // multiplication chain of depth <= 3 (relative error ~4e-16, invisible after the float32 rounding of the spectra);
    // the window (pre-scaled by the unpack's exact 1/2) is shared by the block's waves through LDS.
    __shared__ T wins[N];
    for (int i = threadIdx.x; i < N; i += 256) wins[i] = (T)(0.5 * window[i]);
    // the fused scaler's [4][F] mean / std tables, read from LDS in the store path (round 4: as global loads next to every store
    // they were a `global_load; global_load; s_waitcnt vmcnt(0)` per value -- eight exposed L2 round trips per item)
    extern __shared__ float sct_dyn[]; // SC only: [2][4 * N / 2] floats of DYNAMIC LDS (the launch passes the size), so that the plain
    float *sct0 = sct_dyn, *sct1 = sct_dyn + 4 * (N / 2); // instantiation's static LDS layout is untouched: even an 8-byte dummy
    if (SC && kp.sc_mean) {                               // array here moved the other arrays and cost the STFT 2.5 %
        for (int i = threadIdx.x; i < 4 * kp.F; i += 256) {
            sct0[i] = kp.sc_mean[i];
            sct1[i] = kp.sc_std[i];
        }
    }
    __syncthreads(); // the only workgroup barrier, before any wave-uniform exit
    cplx<T> w1[NP];
    {
        int p = R;
#pragma unroll
        for (int q = 0; q < NP; q++, p *= R) {
            const cplx<double> wd = tw[salsa::stockham_tw(lane, 1, p, N, R)];
            w1[q] = {(T)wd.re, (T)wd.im};
        }
    }
    if (t_begin >= Tn) return; // wave-uniform; nothing below uses a workgroup barrier
    // channel pairs per clip: 2 (the dataset scripts), 3 / 4 on the contrib surface; NPAIRS == 0: any count, read from kp.nch
    // (9 - 16 microphones: one instantiation, the index arithmetic below is all that depends on it)
    const int npairs = NPAIRS > 0 ? NPAIRS : kp.nch / 2;
    const int nch = 2 * npairs;
    const float *clip = audio + (long)b * nch * Ns;
    // samples of one item: 2 channels x R strided points per lane.  Straight-line code on the (wave-uniform) interior
    // path -- no per-load branching; frames that overlap a clip end take the reflect path (np.pad(mode='reflect'); one
    // fold suffices because Ns > N/2, checked on the host).
    static_assert(!STD || (N == 512 && NPAIRS == 2 && !(LITE && SC)), "STD is the dataset scripts' configuration: n_fft 512, planar 4-channel audio");
    const bool planar = STD ? true : kp.layout == SALSA_LAYOUT_PLANAR;
    const int sstride = planar ? 1 : nch;
    // Item order.  Full SALSA: PAIR-major (all the wave's frames of channels 0/1, then of channels 2/3), so consecutive
    // items re-read the 41 % of samples that overlapping frames share while they are still in L2 (frame-major order
    // puts another 4 KiB item and a whole CU's worth of traffic in between: measured 1.7x audio over-fetch).  SALSA-Lite
    // needs channel 0 of the same frame when it processes pair 1, so it stays frame-major.
#if defined(K1_FRAME_MAJOR)
    constexpr bool PAIR_MAJOR = false;
#else
    constexpr bool PAIR_MAJOR = !LITE;
#endif
    const int nleft_ = (Tn - t_begin + K1_TSTEP - 1) / K1_TSTEP; // frames t_begin, t_begin + K1_TSTEP, ... below Tn
    const int nfr_ = STD ? __builtin_amdgcn_readfirstlane(nleft_ < K1_NF ? nleft_ : K1_NF) : (nleft_ < K1_NF ? nleft_ : K1_NF);
    const int psel = (LITE || STD) ? -1 : kp.pair_sel; // one channel pair per launch (the pipelined schedule): item = frame
    // Lite STD: when every bin whose phase survives :120 sits in bin register 0 (lower + upper <= 64: the dataset script's fmax_doa 2000
    // gives bins 1..42), channel 0 of a frame is 512 B per wave and the wave's whole run of frames fits in LDS: PAIR-major order like
    // full SALSA (the frame-major order re-read the samples neighbouring frames share from HBM: 1.6x the audio, profiles/r6_ab_notes.txt)
    const bool lite_pm = (LITE && STD && K1_LITE_PAIR_MAJOR) ? (64 - kp.lower >= kp.upper && K1_NF <= K1_LITE_X0_SLOTS) : false;
    auto item_frame = [&](int item) {
        if (NPAIRS != 2) return PAIR_MAJOR ? item % nfr_ : item / npairs;
        if (LITE && STD) return lite_pm ? (item >= nfr_ ? item - nfr_ : item) : item >> 1;
        return psel >= 0 ? item : PAIR_MAJOR ? (item >= nfr_ ? item - nfr_ : item) : item >> 1;
    };
    auto item_pair = [&](int item) {
        if (LITE && STD && NPAIRS == 2) return lite_pm ? (item >= nfr_ ? 1 : 0) : item & 1;
        if (NPAIRS != 2) return PAIR_MAJOR ? item / nfr_ : item % npairs;
        return psel >= 0 ? psel : PAIR_MAJOR ? (item >= nfr_ ? 1 : 0) : item & 1;
    };
    auto load_item = [&](int item, float *y0, float *y1) {
        const int t = t_begin + K1_TSTEP * item_frame(item);
        const int c0 = 2 * item_pair(item);
        const int base = t * kp.hop - N / 2;
        const unsigned ch0 = 4u * (unsigned)(planar ? c0 * Ns : c0), ch1 = ch0 + 4u * (unsigned)(planar ? Ns : 1); // byte offsets
        const unsigned step = 4u * (unsigned)sstride;
        if (base >= 0 && base + N <= Ns) {
            const unsigned q = (unsigned)(base + lane) * step;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (STD) { // planar, 4-byte stride: the eight strided points of a channel are one address + immediates r * 256 (tried for every
                           // planar plan behind a wave-uniform branch: the general instantiations went from 168 to 174 VGPRs = 2 waves per SIMD)
                    y0[r] = ld_off_c(clip, ch0 + q, r * (N / R) * 4);
                    y1[r] = ld_off_c(clip, ch1 + q, r * (N / R) * 4);
                } else {
                    y0[r] = ld_off(clip, ch0 + q + (unsigned)(r * (N / R)) * step);
                    y1[r] = ld_off(clip, ch1 + q + (unsigned)(r * (N / R)) * step);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) {
                int s = base + salsa::stockham_in(lane, r, N, R);
                s = s < 0 ? -s : s;
                s = s >= Ns ? 2 * (Ns - 1) - s : s;
                y0[r] = ld_off(clip, ch0 + (unsigned)s * step);
                y1[r] = ld_off(clip, ch1 + (unsigned)s * step);
            }
        }
    };

    const int nitems = psel >= 0 ? nfr_ : nfr_ * npairs;
    float y0[R], y1[R];
    float *o = out + (long)b * kp.OC * Tn * kp.F; // [OC][T][F] of this clip (int offsets below)
    float4 *xs = Xs + (long)b * Tn * npairs * kp.nd;
    const int mlane = (64 - lane) & 63;           // lane holding the mirror bins N-k of this lane's bins
    // log-spectrogram value of channel c, feature f; with a scaler attached also (x - mean) / std (database.py:197-202)
    auto spec = [&](const float p, const int c, const int f) -> float {
        const float v = db10(p);
        if (STD && !SC) return v;                 // (the launch guarantees: no scaler attached)
        if (SC) {
            const int i = c * kp.F + f;
            return kp.sc_mean ? (v - sct0[i]) / sct1[i] : v;
        }
        const unsigned off = 4u * (unsigned)(c * kp.F + f);
        return kp.sc_mean ? (v - ld_off(kp.sc_mean, off)) / ld_off(kp.sc_std, off) : v;
    };
    const unsigned plane = 4u * (unsigned)(Tn * kp.F); // bytes of one output channel of a clip
    float2 x0keep[(LITE && STD) ? 1 : R / 2 + 1]; // SALSA-Lite: channel-0 spectrum of this lane's bins, kept from pair 0 for pair 1 (Lite STD: in LDS)
    float2 *const x0s = k1_x0_store<LITE && STD>::get() + (LITE && STD ? w * K1_LITE_X0_SLOTS * 64 + lane : 0);

    for (int item = 0; item < nitems; item++) {
        const int t = t_begin + K1_TSTEP * item_frame(item);
        const int pr = item_pair(item);
        load_item(item, y0, y1);
        cplx<T> v[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const T wn = wins[salsa::stockham_in(lane, r, N, R)];
            v[r] = {wn * (T)y0[r], wn * (T)y1[r]};
        }
        // ---- Stockham passes, in place in the wave-private buffer
        salsa::dftR<R>(v); // pass p = 1 (no twiddles)
#pragma unroll
        for (int r = 0; r < R; r++) z[swz(salsa::stockham_out(lane, r, 1, R))] = v[r];
        {
            int p = R;
#pragma unroll
            for (int q = 0; q < NP; q++, p *= R) {
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < R; r++) v[r] = z[swz(salsa::stockham_in(lane, r, N, R))];
                wave_lds_fence();
                {
                    const cplx<T> a1 = w1[q], a2 = salsa::cmul(a1, a1), a3 = salsa::cmul(a2, a1);
                    v[1] = salsa::cmul(v[1], a1);
                    v[2] = salsa::cmul(v[2], a2);
                    v[3] = salsa::cmul(v[3], a3);
                    if (R == 8) {
                        const cplx<T> a4 = salsa::cmul(a2, a2);
                        v[4 % R] = salsa::cmul(v[4 % R], a4);
                        v[5 % R] = salsa::cmul(v[5 % R], salsa::cmul(a4, a1));
                        v[6 % R] = salsa::cmul(v[6 % R], salsa::cmul(a3, a3));
                        v[7 % R] = salsa::cmul(v[7 % R], salsa::cmul(a4, a3));
                    }
                }
                salsa::dftR<R>(v);
                if (q + 1 < NP) {
#pragma unroll
                    for (int r = 0; r < R; r++) z[swz(salsa::stockham_out(lane, r, p, R))] = v[r];
                }
            }
        }
        // lane now holds Z[lane + 64 r] in v[r] (the last pass writes y[i + r*64])

        // ---- unpack this pair's two spectra: bins k = lane + 64 r, r < R/2 (+ lane 0: the Nyquist bin)
        const int c0 = 2 * pr;
        auto emit_bin = [&](const int k, const cplx<T> a, const cplx<T> bm, float2 &x0k) {
            cplx<T> Xa, Xb;
            salsa::unpack_pair_prescaled(a, bm, Xa, Xb);
            const float2 xa = make_float2((float)Xa.re, (float)Xa.im); // the reference stores its STFT as complex64
            const float2 xb = make_float2((float)Xb.re, (float)Xb.im);
            const float pa = power32(xa), pb = power32(xb);
            if (!LITE) {
                if (kp.feature == SALSA_FEATURE_SALSA && k >= kp.lower && k < kp.upper)
                {
                    const unsigned so = 16u * (unsigned)((t * npairs + pr) * kp.nd + (k - kp.lower));
                    if (K1_SPILL_NT) st_off_nt(xs, so, make_float4(xa.x, xa.y, xb.x, xb.y));
                    else st_off(xs, so, make_float4(xa.x, xa.y, xb.x, xb.y));
                }
                if (k >= kp.spec_lo && k < kp.spec_hi) {
                    const unsigned off = 4u * (unsigned)((c0 * Tn + t) * kp.F + (k - kp.spec_lo));
                    st_off(o, off, spec(pa, c0, k - kp.spec_lo));
                    st_off(o, off + plane, spec(pb, c0 + 1, k - kp.spec_lo));
                } else if (kp.compress && k > kp.ident && k < N / 2) {
                    pw[w][0][k - kp.ident - 1] = pa;
                    pw[w][1][k - kp.ident - 1] = pb;
                }
            } else { // SALSA-Lite / SALSA-IPD (salsa_lite_feature_extraction.py:103-120)
                if (pr == 0) x0k = xa;
                if (k >= kp.lower && k < kp.cutoff) {
                    const int f = k - kp.lower;
                    const unsigned off = 4u * (unsigned)((c0 * Tn + t) * kp.F + f);
                    st_off(o, off, spec(pa, c0, f));
                    st_off(o, off + plane, spec(pb, c0 + 1, f));
                    const float2 x0 = x0k;
                    // angle(X_c conj(X_0)) / (delta*k) (lite :111-115) or / pi (ipd :113).  float32 throughout: the
                    // product's rounding moves the angle by <= 1e-7 rad and 1/(delta*k) is the float64 quotient rounded
                    // once, so the float32 result is within ~2 ulp of the reference's float64-then-cast value.
                    // (contrib's float32 frequency vector differs from delta*k by <= 6e-8 relative: same float32 result)
                    const float inv_scale = kp.feature == SALSA_FEATURE_IPD ? 0.318309886183790672f
                                                                           : (float)(1.0 / (kp.delta * (double)(k == 0 ? 1 : k)));
                    // pair 0 contributes channel 1 (phase vs channel 0); pair p >= 1 contributes channels 2p and 2p+1: the phase of
                    // channel c goes to output plane NCH + c - 1, i.e. NCH - 1 and NCH planes above this pair's spectrogram
                    auto phase = [&](const float2 xc) -> float {
                        if (!(f < kp.upper)) return 0.f; // ":120 phase_vector[:, :, upper_bin:] = 0" indexes the CROPPED axis
                        return lite_phase(xc, x0) * inv_scale;
                    };
                    if (pr >= 1) st_off(o, off + (unsigned)(nch - 1) * plane, phase(xa));
                    st_off(o, off + (unsigned)nch * plane, phase(xb));
                }
            }
        };
        // STD: register r holds bins 64 r .. 64 r + 63, so with spec_lo = 1, spec_hi = 193, ident = 192 the spectrogram-row / compressed-
        // band membership of a bin is fixed per register: r = 0 -- rows for every lane but lane 0 (bin 0: W has no row for it);
        // r = 1, 2 -- rows, all lanes; r = 3 -- lane 0 is bin 192 (row 191), lanes 1..63 are bins 193..255 of the compressed rows.
        auto emit_std = [&](auto rc, const cplx<T> a, const cplx<T> bm) {
            constexpr int r = decltype(rc)::value;
            const int k = lane + 64 * r;
            cplx<T> Xa, Xb;
            salsa::unpack_pair_prescaled(a, bm, Xa, Xb);
            const float2 xa = make_float2((float)Xa.re, (float)Xa.im);
            const float2 xb = make_float2((float)Xb.re, (float)Xb.im);
            const float pa = power32(xa), pb = power32(xb);
            if (k >= kp.lower && k < kp.upper) {
                const unsigned so = 16u * (unsigned)((t * 2 + pr) * kp.nd + (k - kp.lower));
                if (K1_SPILL_NT) st_off_nt(xs, so, make_float4(xa.x, xa.y, xb.x, xb.y));
                else st_off(xs, so, make_float4(xa.x, xa.y, xb.x, xb.y));
            }
            if (r < 3 ? (r > 0 || lane > 0) : lane == 0) {
                const unsigned off = 4u * (unsigned)((c0 * Tn + t) * 200 + (k - 1));
                st_off(o, off, spec(pa, c0, k - 1));
                st_off(o, off + plane, spec(pb, c0 + 1, k - 1));
            } else if (r == 3) {
                pw[w][0][lane - 1] = pa;
                pw[w][1][lane - 1] = pb;
            }
        };
        // Lite STD: same arithmetic as emit_bin's SALSA-Lite branch.  What is fixed per register here: register r holds bins 64 r .. 64 r + 63,
        // so whether it has any bin below the cutoff, and any bin whose phase is not zeroed by :120, are wave-uniform tests (scalar
        // branches around whole blocks instead of exec-masked code); channel 0's spectrum goes through LDS; no Nyquist item.
        auto emit_lite_std = [&](auto rc, const cplx<T> a, const cplx<T> bm) {
            constexpr int r = decltype(rc)::value;
            if (64 * r >= kp.cutoff) return;                       // wave-uniform: no bin of this register is written
            const bool any_phase = 64 * r - kp.lower < kp.upper;   // wave-uniform: some bin of this register keeps its phase (:120)
            const int k = lane + 64 * r;
            cplx<T> Xa, Xb;
            salsa::unpack_pair_prescaled(a, bm, Xa, Xb);
            const float2 xa = make_float2((float)Xa.re, (float)Xa.im); // the reference stores its STFT as complex64
            const float2 xb = make_float2((float)Xb.re, (float)Xb.im);
            const float pa = power32(xa), pb = power32(xb);
            float2 x0 = xa;
            if (any_phase) { // (pair-major order: only register 0 comes here, the slot is the frame's place in the wave's run)
                const int slot = lite_pm ? item_frame(item) : r;
                if (pr == 0) x0s[slot * 64] = xa;
                else x0 = x0s[slot * 64];
            }
            if (k >= kp.lower && k < kp.cutoff) {
                const int f = k - kp.lower;
                const unsigned off = 4u * (unsigned)((c0 * Tn + t) * kp.F + f);
                st_off(o, off, db10(pa));
                st_off(o, off + plane, db10(pb));
                float pha = 0.f, phb = 0.f;
                if (any_phase) {
                    const float inv_scale = kp.feature == SALSA_FEATURE_IPD ? 0.318309886183790672f
                                                                           : (float)(1.0 / (kp.delta * (double)(k == 0 ? 1 : k)));
                    auto phase = [&](const float2 xc) -> float {
                        if (!(f < kp.upper)) return 0.f; // ":120 phase_vector[:, :, upper_bin:] = 0" indexes the CROPPED axis
                        return lite_phase(xc, x0) * inv_scale;
                    };
                    if (pr >= 1) pha = phase(xa);
                    phb = phase(xb);
                }
                if (pr >= 1) st_off(o, off + 3u * plane, pha);
                st_off(o, off + 4u * plane, phb);
            }
        };
#pragma unroll
        for (int r = 0; r < R / 2; r++) {
            // mirror of k = lane + 64 r is N-k = (64-lane) + 64 (R-1-r): register R-1-r of lane 64-lane;
            // lane 0: N - 64 r = 64 (R-r), its own register (R-r) mod R
            cplx<T> bm = {__shfl(v[R - 1 - r].re, mlane), __shfl(v[R - 1 - r].im, mlane)};
            if (lane == 0) bm = v[(R - r) & (R - 1)];
            if constexpr (STD && LITE) {
                if (r == 0) emit_lite_std(std::integral_constant<int, 0>{}, v[r], bm);
                else if (r == 1) emit_lite_std(std::integral_constant<int, 1>{}, v[r], bm);
                else if (r == 2) emit_lite_std(std::integral_constant<int, 2>{}, v[r], bm);
                else emit_lite_std(std::integral_constant<int, 3>{}, v[r], bm);
            } else if constexpr (STD) {
                if (r == 0) emit_std(std::integral_constant<int, 0>{}, v[r], bm);
                else if (r == 1) emit_std(std::integral_constant<int, 1>{}, v[r], bm);
                else if (r == 2) emit_std(std::integral_constant<int, 2>{}, v[r], bm);
                else emit_std(std::integral_constant<int, 3>{}, v[r], bm);
            } else {
                emit_bin(lane + 64 * r, v[r], bm, x0keep[(LITE && STD) ? 0 : r]);
            }
            __builtin_amdgcn_sched_barrier(0); // one bin at a time: keeps the live set (and the VGPR count) small
        }
        if (!STD && lane == 0) emit_bin(N / 2, v[R / 2], v[R / 2], x0keep[(LITE && STD) ? 0 : R / 2]); // (STD: bin 256 is in no DOA band, row or compressed row)
        // ---- compressed high-frequency rows of W: sum of 8 (last row 7) bins times 1/8
        if constexpr (STD && !LITE) {
            // eight groups x two channels = lanes 0..15: two 16-byte LDS reads and a fixed chain of eight additions in the order of
            // the loop below (the seventh row's missing eighth term is skipped, not added as zero: pw[..][63] is never written)
            wave_lds_fence();
            const int h = lane & 1, gi = lane >> 1;
            if (gi < 8) {
                const float4 p0 = *(const float4 *)&pw[w][h][8 * gi], p1 = *(const float4 *)&pw[w][h][8 * gi + 4];
                float acc = 0.f;
                acc += 0.125f * p0.x; acc += 0.125f * p0.y; acc += 0.125f * p0.z; acc += 0.125f * p0.w;
                acc += 0.125f * p1.x; acc += 0.125f * p1.y; acc += 0.125f * p1.z;
                const float acc8 = acc + 0.125f * p1.w;
                acc = gi < 7 ? acc8 : acc;
                st_off(o, 4u * (unsigned)(((c0 + h) * Tn + t) * 200 + 192 + gi), spec(acc, c0 + h, 192 + gi));
            }
        } else if (!LITE && kp.compress) {
            wave_lds_fence();
            const int ng = kp.F - kp.ident;
            const int h = lane & 1, gi = lane >> 1;
            if (gi < ng) {
                const int cnt = gi < ng - 1 ? 8 : 7;
                float acc = 0.f;
                for (int q = 0; q < cnt; q++) acc += 0.125f * pw[w][h][8 * gi + q];
                st_off(o, 4u * (unsigned)(((c0 + h) * Tn + t) * kp.F + kp.ident + gi), spec(acc, c0 + h, kp.ident + gi));
            }
        }
        wave_lds_fence(); // this item's LDS reads are done before the next item's first pass overwrites z / pw
    }
}

// ------------------------------------------------------------------------------------------------------------ K2
// Noise-floor tracker.  The recurrence over time is strictly sequential per (clip, bin), but everything that feeds it
// (|X0|^2, the 3-frame mean, the float64 divide and square root) is not: producer waves stay one chunk of TR_CH frames
// ahead of the consumer wave, computing mag[t][bin] into an LDS ring.  Spill layout: Xs[b][t][pair][bin] as float4
// (c0.re, c0.im, c1.re, c1.im); channel 0 is the .xy of pair 0.  Output: valid32[b][32-bin group][t] = indicator_sig mask
// of frame t (bit j = bin 32*group + j).
constexpr int TR_CH = 64;       // frames per chunk
#ifndef TR_WAVES_N
#define TR_WAVES_N 8
#endif
constexpr int TR_WAVES = TR_WAVES_N; // wave 0 = consumer; with more than 4 waves, wave 4 idles (see tracker_kernel); the rest produce

template <int COUNT>
__device__ __forceinline__ void tracker_load(const KParams &kp, const float4 *__restrict__ x0, int stride, int c0,
                                             int first, bool active, float2 *x)
{
    // |X0| samples of frames c0+first-2 .. c0+first+COUNT-1 (wrap on the time axis; beyond the clip: unused)
    const int Tn = kp.T;
#pragma unroll
    for (int i = 0; i < COUNT + 2; i++) {
        int t = c0 + first + i - 2;
        if (t >= Tn) t = Tn - 1;
        while (t < 0) t += Tn;
        if (TR_NT_LD) {
            const salsa_f2v v = active ? __builtin_nontemporal_load((const salsa_f2v *)&x0[t * stride]) : salsa_f2v{0.f, 0.f};
            x[i] = make_float2(v.x, v.y);
        } else
            x[i] = active ? *(const float2 *)&x0[t * stride] : make_float2(0.f, 0.f); // channel 0 = .xy of pair 0
    }
}

// x / 3 correctly rounded (= the reference's float64 division, :53-55) in three instructions instead of the ~14 of the compiler's
// IEEE division sequence (two v_div_scale, a quarter-rate v_rcp_f64, five FMAs, v_div_fmas, v_div_fixup): with y = RN(1/3) and
// q = RN(x y) faithful, the residual r = x - 3 q is exact in one FMA and RN(q + r y) is the correctly rounded quotient (Markstein's
// theorem; checked against exact rational arithmetic on 7e5 values incl. subnormals, tools/probes/div3_check.py).  The tracker's
// producer waves do one division and one square root per (frame, bin): 29 M of each per 32-clip batch.
#ifndef TR_DIV3_EXACT
#define TR_DIV3_EXACT 1
#endif
__device__ __forceinline__ double div3_exact(const double x)
{
#if TR_DIV3_EXACT
    const double y = 1.0 / 3.0; // RN(1/3), a compile-time constant
    const double q = x * y;
    const double r = __builtin_fma(-3.0, q, x);
    return __builtin_fma(r, y, q);
#else
    return x / 3;
#endif
}

template <int COUNT, int BINS>
__device__ __forceinline__ void tracker_mag(const float2 *x, int first, double *dst /*[TR_CH][BINS]*/, int col, bool raw)
{
    double p[COUNT + 2];
#pragma unroll
    for (int i = 0; i < COUNT + 2; i++) {
        const double re = x[i].x, im = x[i].y;
        p[i] = re * re + im * im;
    }
#pragma unroll
    for (int i = 0; i < COUNT; i++) {
        if (first + i < TR_CH) // :53-55 in the reference's order ; contrib :326-328 tracks the raw |X0| instead
            dst[(first + i) * BINS + col] = raw ? sqrt(p[i + 2]) : sqrt(div3_exact(((0.0 + p[i + 2]) + p[i + 1]) + p[i]));
    }
}

// One workgroup = 32 adjacent bins of one clip, TR_WAVES waves: wave 0 is the CONSUMER (the recurrence), the others are
// producers.  What bounds this kernel is the consumer: 4801 strictly sequential steps per clip of ~16 instructions each, issued
// by ONE wave -- so everything is arranged for that wave's issue rate and dependent-chain latency:
//  * a wave issues at most one instruction per 4-cycle slot of its SIMD, so the step costs ~4.5 cycles x its instruction count
//    (17 here, 23 in round 1) -- PROVIDED the consumer has its SIMD to itself.  A workgroup's waves go to the four SIMDs
//    cyclically, so wave 4 would share the consumer's: it does nothing but the barriers (a wave parked at s_barrier takes no
//    issue slots) and the six producers are waves 1-3 and 5-7.  They need ~440 cycles per (2 frames x 32 bins) item -- the
//    float64 divide and square root are quarter-rate -- i.e. ~2700 cycles per chunk against the consumer's ~4900;
//  * the step's dependent chain is  multiply -> select -> max  (salsa::tracker_step forms both candidate products first);
//  * indicator_sig never becomes a per-lane value: the compare writes a scalar lane mask (one bit per bin), which is exactly
//    the per-frame mask the covariance kernel wants; v_writelane drops it into lane `frame` of one VGPR and the chunk's 64
//    masks leave as ONE coalesced 256-byte store: valid32[b][32-bin group][t].  (Round 1 shifted a bit into a per-bin history
//    word -- three VALU instructions per step -- and K3 had to ballot the words back into frame masks.)
//  * all 64 lanes run the recurrence (lanes 32-63 mirror 0-31), so the consumer has no divergent region around the writelanes;
//  * producers keep TWO register sets of spectra and alternate them (the chunk loop is unrolled by two): the loads issued in
//    one iteration are first used in the next, a whole chunk later.  (Round 1 copied "next" into "current" at the end of every
//    iteration, which made each iteration wait for the loads it had just issued.)
constexpr int TR_BINS = 32;
#ifndef TR_IDLE4
#define TR_IDLE4 1
#endif
#ifndef TR_MASK_HISTORY
#define TR_MASK_HISTORY 1 // the countdown as three scalar `above` masks (round 3); 0: the per-lane countdown of round 2
#endif
#ifndef TR_CHUNK_CLAMP_SKIP
#define TR_CHUNK_CLAMP_SKIP 0 // round 5: chunks whose floors cannot reach the 1e-6 clamp run the step without it (see the consumer).
                              // Masks bit-identical (20 GPU parity tests), tracker 0.175 - 0.183 (off) vs 0.180 - 0.188 ms (on): no gain, off
#endif
#ifndef TR_LAZY_CLAMP
#define TR_LAZY_CLAMP 0 // measured: 0.35 ms with the lazy clamp against 0.22 ms without -- the chain is not what bounds the step
#endif
// v_writelane_b32 through the LLVM intrinsic (this clang has no __builtin for it).  Not inline asm: a VALU compare that writes
// VCC needs two wait states before v_writelane may read it, and only the compiler's hazard recogniser inserts them -- the
// hand-written form read stale masks in 0.7 % of the frames.
extern "C" __device__ int salsa_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
// TR_WIDE (round-3 review: "give the consumer 64 distinct bins per wave and report what that does"): a workgroup serves 64 bins,
// lanes 32-63 run bins 32-63 instead of mirroring lanes 0-31, a frame's ballot is two mask words.  Measured (profiles/r4_ab_notes.txt):
// see there; the mask layout valid32[b][32-bin group][t] is unchanged.
#ifndef TR_WIDE
#define TR_WIDE 0
#endif
constexpr int TR_WG_BINS = TR_WIDE ? 64 : TR_BINS;
static unsigned tracker_grid(const KParams &kp) { return (unsigned)(kp.B * ((kp.nd + TR_WG_BINS - 1) / TR_WG_BINS)); }

__global__ __launch_bounds__(64 * TR_WAVES) void tracker_kernel(const KParams kp, const float4 *__restrict__ Xs,
                                                                unsigned *__restrict__ valid32)
{
    if (kp.doubt_flag && blockIdx.x == 0 && threadIdx.x == 0) *kp.doubt_flag = 0u; // (consumed two launches later, same stream)
    constexpr int BINS = TR_WG_BINS, FS = 64 / BINS;                                      // FS frames per producer instruction
    constexpr bool IDLE4 = TR_WAVES > 4 && TR_IDLE4;                                     // wave 4 shares the consumer's SIMD: keep it idle
    constexpr int NPROD = IDLE4 ? TR_WAVES - 2 : TR_WAVES - 1;
    constexpr int PER_ALL = (TR_CH + TR_WAVES * FS - 1) / (TR_WAVES * FS);               // prologue: all waves produce chunk 0
    constexpr int PER_PROD = (TR_CH + NPROD * FS - 1) / (NPROD * FS);                    // frames per producer lane per chunk
    __shared__ double ring[2][TR_CH * BINS];
    const int ng32 = (kp.nd + BINS - 1) / BINS;
    const int b = blockIdx.x / ng32, g = blockIdx.x % ng32;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int col = lane % BINS, fsub = lane / BINS;            // producers: lane = (frame sub-slot, bin column)
    const int bin = g * BINS + col;
    const bool active = bin < kp.nd;
    const int Tn = kp.T;
    const int stride = (kp.nch / 2) * kp.nd; // float4 elements per frame (one per channel pair and bin)
    const float4 *x0 = Xs + (long)b * Tn * stride + (active ? bin : 0);
    const int nchunks = (Tn + TR_CH - 1) / TR_CH;
    const bool raw = kp.flex != 0;
    {
        float2 x[PER_ALL + 2];
        const int first = (w * FS + fsub) * PER_ALL;
        tracker_load<PER_ALL>(kp, x0, stride, 0, first, active, x);
        tracker_mag<PER_ALL, BINS>(x, first, ring[0], col, raw);
    }
    float2 xa[PER_PROD + 2], xb[PER_PROD + 2];
    const bool producer = w > 0 && !(IDLE4 && w == 4);
    const int pidx = (IDLE4 && w > 4) ? w - 2 : w - 1;           // producer number 0 .. NPROD-1
    const int pfirst = (pidx * FS + fsub) * PER_PROD;
    if (producer && nchunks > 1) tracker_load<PER_PROD>(kp, x0, stride, TR_CH, pfirst, active, xa);
    __syncthreads();
    // Consumer state: noise floor + countdown (salsa_feature_extraction.py:30, :58), evaluated with exactly the reference's
    // operations (one float64 multiply by 1.02 / 1.002 / 0.98, the 1e-6 clamp, the two strict compares).
    double fl = 0.0;
    int cd = 3;
    const double snr = kp.snr_ratio;
    const int n32 = (kp.nd + TR_BINS - 1) / TR_BINS;      // 32-bin mask groups of a clip
    unsigned *vout = valid32 + ((long)b * n32 + (TR_WIDE ? 2 * g : g)) * Tn; // [b][32-bin group][t]
    unsigned *vout_hi = (TR_WIDE && 2 * g + 1 < n32) ? vout + Tn : nullptr;     // TR_WIDE: the group of bins 32-63 of this workgroup
#if TR_MASK_HISTORY
    // Round 3: the countdown never becomes a per-lane value either.  "countdown < 1 before this step's decrement" (:68-69: the
    // slow rise) holds exactly when the three steps before this one were all `above` (the countdown starts at 3, every `above`
    // takes one off, anything else puts it back to 3), and `above` is already a scalar lane mask -- the float64 compare writes
    // one.  So the state is three 64-bit masks in SGPRs, the test is two s_and_b64 on the scalar unit, and
    // __builtin_amdgcn_inverse_ballot hands the result to v_cndmask as its mask operand: the three VALU instructions of the
    // per-lane countdown (compare, reset-select, decrement) are gone, and with them the reason to form BOTH candidate products --
    // the factor (1.02 | 1.002 | 0.98) is selected and multiplied once, the same single multiplication the reference does.
    // 14 -> 10 vector instructions per step.  Measured (tools/probes/f64_latency_probe.hip, the probe builds TR_PROBE_NO_*): the
    // consumer wave alone is 0.158 of the kernel's 0.177 ms, and its step time follows its INSTRUCTION COUNT (~7.6 cycles per
    // vector instruction of this mix: VOP3 compares writing SGPR pairs, selects reading them), not the dependent chain -- a
    // hand-pipelined order that puts the indicator of step t-1 into the stall slots of step t's chain ran no faster (0.179).
    unsigned long long h1 = 0ull, h2 = 0ull, h3 = 0ull; // `above` masks of the previous three steps (wave-uniform)
    // CLAMPED = false (round 5): the 1e-6 clamp (:85) left out.  The floor falls by at most 0.98 per step, so a chunk that STARTS
    // with every lane's floor >= 1e-6 / 0.98^64 (3.65e-6; 4e-6 is tested) cannot bring any floor below 1e-6 within its 64 steps:
    // max(x, 1e-6) == x bit for bit there, and the chunk runs 9 instead of 10 vector instructions per step with the float64 max
    // off the dependent chain.  Near-silent bins (a floor at the clamp) take the clamped step; the test is one ballot per chunk.
    auto step = [&](const double m, auto clamped) -> unsigned long long {
        const bool slow = __builtin_amdgcn_inverse_ballot_w64(h1 & h2 & h3);
        const double up = slow ? 1.0 + 0.1 * 0.02 : 1.0 + 0.02;
        const bool above = m > fl;
        const double f = above ? up : 1.0 - 0.02;
        if (decltype(clamped)::value) fl = fmax(f * fl, 1e-6); // (a product of finite numbers is canonical: one v_max_f64)
        else fl = f * fl;
        h3 = h2;
        h2 = h1;
        h1 = __ballot(above);
        return __ballot(m > snr * fl); // :87
    };
#else
    auto step = [&](const double m, auto) -> unsigned long long { return __ballot(salsa::tracker_step(fl, cd, m, snr)); }; // :65-87
#endif
    auto consume = [&](const int c) {
        const double *cur = ring[c & 1] + col;
        if (c == 0) { // noise_floor = 0.5 * mean(mag[0:5])  (:58)
            const int n0 = Tn < 5 ? Tn : 5;
            double acc = 0.0;
            for (int t = 0; t < n0; t++) acc += cur[t * BINS];
            fl = 0.5 * (acc / (double)n0);
            if (kp.flex && fl < 1e-6) fl = 1e-6; // contrib's tracker clamps its initial floor (:118-120)
        }
        unsigned word = 0, word_hi = 0; // lane i: indicator_sig mask (bit j = bin 32 g + j) of frame 64 c + i (TR_WIDE: + bins 32-63)
        const int nfr = Tn - c * TR_CH < TR_CH ? Tn - c * TR_CH : TR_CH; // wave-uniform
        if (nfr == TR_CH) { // every chunk but the last: straight-line code, no per-frame conditionals
#if TR_MASK_HISTORY
            const bool noclamp = __ballot(!(fl >= 4e-6)) == 0ull; // wave-uniform: no lane's floor can reach the clamp in this chunk
#else
            const bool noclamp = false;
#endif
#pragma unroll
            for (int i0 = 0; i0 < TR_CH; i0 += 16) {
                double m[16]; // one LDS round trip per 16 frames, not per frame
#pragma unroll
                for (int i = 0; i < 16; i++) m[i] = cur[(i0 + i) * BINS];
#if TR_LAZY_CLAMP && !TR_MASK_HISTORY
                // The 1e-6 clamp (:85) is a float64 max on the step's dependent chain (multiply -> select -> max), yet it only
                // ever acts on near-silent bins.  So a block of 16 steps runs WITHOUT it, a running minimum of the floor rides
                // along off the chain, and only if some lane's floor dipped below 1e-6 (wave-uniform test) the block is redone
                // from its saved state with the clamp.  Where no clamp acts max(x, 1e-6) == x, so the result is bit-identical.
                const double fl0 = fl;
                const int cd0 = cd;
                const unsigned word0 = word;
                double mn = fl;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const unsigned long long bal = __ballot(salsa::tracker_step<false>(fl, cd, m[i], snr)); // :65-87
                    asm("v_min_f64 %0, %0, %1" : "+v"(mn) : "v"(fl)); // (fmin() would canonicalise its operand first: one more float64 op)
                    word = (unsigned)salsa_writelane_i32((int)(unsigned)bal, i0 + i, (int)word);
                }
                if (__ballot(mn < 1e-6) != 0ull) {
                    fl = fl0;
                    cd = cd0;
                    word = word0;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const unsigned long long bal = __ballot(salsa::tracker_step<true>(fl, cd, m[i], snr));
                        word = (unsigned)salsa_writelane_i32((int)(unsigned)bal, i0 + i, (int)word);
                    }
                }
#else
                if (TR_CHUNK_CLAMP_SKIP && noclamp) {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const unsigned long long bal = step(m[i], std::false_type{});
                        word = (unsigned)salsa_writelane_i32((int)(unsigned)bal, i0 + i, (int)word);
                        if (TR_WIDE) word_hi = (unsigned)salsa_writelane_i32((int)(unsigned)(bal >> 32), i0 + i, (int)word_hi);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const unsigned long long bal = step(m[i], std::true_type{});
                        word = (unsigned)salsa_writelane_i32((int)(unsigned)bal, i0 + i, (int)word);
                        if (TR_WIDE) word_hi = (unsigned)salsa_writelane_i32((int)(unsigned)(bal >> 32), i0 + i, (int)word_hi);
                    }
                }
#endif
            }
        } else {
            for (int i = 0; i < nfr; i++) {
                const unsigned long long bal = step(cur[i * BINS], std::true_type{});
                word = lane == i ? (unsigned)bal : word; // (ragged last chunk only)
                if (TR_WIDE) word_hi = lane == i ? (unsigned)(bal >> 32) : word_hi;
            }
        }
        if (lane < nfr) vout[c * TR_CH + lane] = word;
        if (TR_WIDE && vout_hi && lane < nfr) vout_hi[c * TR_CH + lane] = word_hi;
    };
    // producers, iteration c: issue the loads of chunk c+2 into `nxt`, turn `now` (chunk c+1, loaded an iteration ago) into
    // magnitudes in the ring half the consumer is not reading
    auto produce = [&](const int c, const float2 *now, float2 *nxt) {
        if (c + 1 >= nchunks) return;
        if (c + 2 < nchunks) tracker_load<PER_PROD>(kp, x0, stride, (c + 2) * TR_CH, pfirst, active, nxt);
        tracker_mag<PER_PROD, BINS>(now, pfirst, ring[(c + 1) & 1], col, raw);
    };
    // (probe builds: TR_PROBE_NO_CONSUME / TR_PROBE_NO_PRODUCE drop one role -- wrong results, the other role's time)
    // Round 4 experiment (TR_RAW_BARRIER 1): the chunk barriers order LDS traffic only (the ring), so they could be
    // `s_waitcnt lgkmcnt(0); s_barrier` instead of __syncthreads() -- a fence + barrier = `s_waitcnt vmcnt(0)` as well, which makes
    // every producer sit out the loads it has just issued for the chunk after next and the consumer the acknowledgement of its
    // 256-byte mask store.  Bit-identical masks, and no faster (see below): kept off.
#ifndef TR_RAW_BARRIER
#define TR_RAW_BARRIER 0 // measured: tracker 0.1736 -> 0.1747 ms (event pair), 0.096 -> 0.101 (prefix): no gain -- the chunk time is the consumer's
#endif                   // 64 dependent steps; the drains it skips were hidden behind them

    auto chunk_barrier = [&]() {
#if TR_RAW_BARRIER
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#else
        __syncthreads();
#endif
    };
    for (int c = 0; c < nchunks; c += 2) {
#ifndef TR_PROBE_NO_CONSUME
        if (w == 0) consume(c);
#endif
#ifndef TR_PROBE_NO_PRODUCE
        if (w != 0 && producer) produce(c, xa, xb);
#endif
        chunk_barrier();
        if (c + 1 < nchunks) {
#ifndef TR_PROBE_NO_CONSUME
            if (w == 0) consume(c + 1);
#endif
#ifndef TR_PROBE_NO_PRODUCE
            if (w != 0 && producer) produce(c + 1, xb, xa);
#endif
        }
        chunk_barrier();
    }
}

// ------------------------------------------------------------------------------------------------------------ K3
// Covariance + eigen-gate + eigenvector.  Only TF bins that pass the noise gate need the (float64, ~700 instruction)
// solve, and they are scattered: a wave that owns 64 fixed bins runs the solve if ANY lane is valid.  So each workgroup
// takes a tile of K3_FT frames x (<=256) bins of one clip, compacts the valid (frame, bin) pairs into an LDS work list
// (lane order preserved, so neighbouring lanes still read neighbouring bins), writes zeros for the rest, and then the
// 256 lanes walk the dense list.  FEAT: write float32 channels 4-6 of the feature array (zeros above the DOA band up
// to F); otherwise write the float64 (3, n_bins, n_frames) array of extract_normalized_eigenvector (+ gate codes).
#ifndef K3_FT_N
#define K3_FT_N 8 // measured 2/4/8/16/32/64: 8 is fastest (tiles in bursts do ~10x the work of quiet ones: small tiles balance)
#endif
constexpr int K3_FT = K3_FT_N; // frames per tile; divides TR_CH so a tile's gate words sit in one chunk
#ifndef K3_GROUP
#define K3_GROUP 2
#endif
#ifndef K3_NT_N
#define K3_NT_N 128 // threads = bins per workgroup (a multiple of 64: every wave owns one 64-bin group of the tracker's masks).  Measured 256 / 128 / 64:
                    // cov_eig 0.365 / 0.340 / 0.348 ms, step 1.023 / 1.000 / 1.013 (profiles/r4_ab_notes.txt): the work list of a tile is ~1.5 x 256
                    // items, so with 256 threads half the waves sat out the second pass at the barrier; two waves share evenly
#endif
constexpr int K3_NT = K3_NT_N;
constexpr int K3_OW = K3_NT + 8; // columns of the LDS output tile: a block's bins + the zero band above them when it fits
// Tile order.  Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD b % 8, each with its own L2), so with
// tile = blockIdx.x two neighbouring 8-frame tiles -- which share 6 of the 14 spill frames they read -- always sit on
// different XCDs and the shared frames are fetched from HBM twice.  Runs of K3_XCD_CHUNK consecutive tiles are instead given to
// ONE XCD (the j-th workgroup of XCD x takes tile (j / CHUNK * 8 + x) * CHUNK + j % CHUNK): neighbours run on the same L2 at
// about the same time, while the runs themselves still rotate over the XCDs (bursts of heavy tiles spread over all eight).
#ifndef K3_XCD_CHUNK
#define K3_XCD_CHUNK 16
#endif

// FAST (round 3; tracking on, compile-time window): the work-list loop compiles ONLY the gate and the column-0 eigenvector
// (salsa_math.h, PATH 1); a gated bin whose column-0 pivot is too small (u_0 ~ 0: rare) is pushed onto a second LDS list and
// solved after the loop by the general arg-max path (PATH 2), one frame at a time.
//
// PK (round 4; FEAT + FAST, FOA / MIC, cond > 1): the two frames of a work item are solved TOGETHER as one packed-float32 pair
// (salsa_math.h: herm4_gate_eigvec_pk -- every v_pk_*_f32 does useful work in both halves), on covariances accumulated in float32
// in the (re, im) packing the spectra are loaded in (cov4pk_rank1: 16 packed FMAs per frame against 40 float64 instructions).
// A frame whose gate margin, pivot or feature conditioning is inside the float32 error bound comes back `unsure` and joins the
// float64 cold list, which recomputes its covariance in float64 from the spill (~1 % of the gated frames of the bench clips:
// tools/pk_study.py), so every gate decision the packed solve keeps equals the float64 one.
#ifndef SALSA_PK
#define SALSA_PK 1
#endif
// The packed solve's gate certificate needs |q'(c)| * (error of c = mu1 / cond) well below SALSA_PK_GATE_TOL: both
// grow as cond -> 1 (c -> mu1, where q' = prod(mu1 - mu_i)), so plans with cond_num below 2 take the float64 instantiation
// (the dataset scripts use 5; the goldens 5 and 2).
#define SALSA_PK_COND_MIN 2.0
#ifndef K3_STAGE_LDS
#define K3_STAGE_LDS 0
#endif
#ifndef K3_OUT_NT
#define K3_OUT_NT 1
#endif
#ifndef K3_PK_WAVES
#define K3_PK_WAVES 4 // waves per SIMD the register allocation is held to (4: 128 VGPRs, 3: 168)
#endif
template <bool FEAT, int NHOP, bool FAST = false, bool PK = false>
__global__ __launch_bounds__(K3_NT, PK ? K3_PK_WAVES : 1) void cov_eig_kernel(const KParams kp, const float4 *__restrict__ Xs,
                                                      const unsigned *__restrict__ valid32,
                                                      float *__restrict__ out_feat, double *__restrict__ out_eig,
                                                      unsigned char *__restrict__ gate)
{
    constexpr int G = NHOP >= 0 ? K3_GROUP : 1; // frames per work item (compile-time window only)
    constexpr bool PAIRED = G > 1;
    static_assert(K3_FT % G == 0 && G <= 4 && K3_FT / G <= 16, "work-list entry layout");
    __shared__ unsigned short list[K3_FT * K3_NT];
    __shared__ unsigned short slow[FAST ? K3_FT * K3_NT : 1]; // (frame in tile) << 8 | bin in block
    __shared__ int count, nslow;
    // byte offsets (into the clip's spill) of frames t0 - NHOP .. t0 + K3_FT - 1 + NHOP with np.pad's 'wrap' on the time axis
    // (:43) applied, computed once per workgroup: done per lane and per frame in the work-list loop, the wrap compiled to an
    // integer-division sequence plus a loop, ~40 instructions x 8 frames per item -- a quarter of the loop's issue slots
    __shared__ __attribute__((aligned(16))) unsigned rowoff[NHOP >= 0 ? K3_FT + 2 * NHOP + 2 : 2];
    // FEAT: the tile's channels 4-6 are assembled in LDS (zeros + the gated bins' results) and written out as whole rows with
    // 16-byte stores at the end, instead of one 4-byte store per lane per (channel, frame) for the zeros plus three scattered
    // 4-byte stores per result
    __shared__ __attribute__((aligned(16))) float otile[FEAT ? 3 * K3_FT * K3_OW : 4];
#if K3_STAGE_LDS
    // Round-5 experiment (review item 6): the tile's 14 frames x 2 pairs x K3_NT bins copied into LDS with coalesced 16-byte loads
    // once per workgroup, the work items' sixteen gathers answered from there.  Measured slower (profiles/r5_ab_notes.txt): off.
    __shared__ __attribute__((aligned(16))) float4 stage[PK ? (K3_FT + 6) * 2 * K3_NT : 1];
#endif
    const int tid = threadIdx.x;
    const int Tn = kp.T;
    int tile = blockIdx.x, b = blockIdx.y;
    if (K3_XCD_CHUNK > 0) {
        const unsigned ntile = gridDim.x, total = ntile * gridDim.y, lin = blockIdx.x + ntile * blockIdx.y;
        constexpr unsigned span = 8u * (K3_XCD_CHUNK > 0 ? K3_XCD_CHUNK : 1);
        if (lin < total / span * span) { // (the ragged tail keeps the identity order)
            const unsigned xcd = lin & 7u, j = lin >> 3;
            const unsigned tl = ((j / (span / 8)) * 8u + xcd) * (span / 8) + j % (span / 8);
            b = (int)(tl / ntile);
            tile = (int)(tl - (unsigned)b * ntile);
        }
    }
    const int t0 = tile * K3_FT;
    const int nft = Tn - t0 < K3_FT ? Tn - t0 : K3_FT;
    const int bin0 = blockIdx.z * K3_NT;
    const int nbc = kp.nd - bin0 < K3_NT ? kp.nd - bin0 : K3_NT; // bins of this tile
    if (tid == 0) count = 0, nslow = 0;
    if (kp.doubt32 && tid < K3_FT * (K3_NT / 32)) { // this tile's words of the doubt mask (no other workgroup touches them): zero before any atomicOr below
        const int ft = tid % K3_FT, g32 = bin0 / 32 + tid / K3_FT, n32 = (kp.nd + TR_BINS - 1) / TR_BINS;
        if (ft < nft && g32 < n32) kp.doubt32[((long)b * n32 + g32) * Tn + t0 + ft] = 0u;
    }
    if (NHOP >= 0 && tid < K3_FT + 2 * NHOP) {
        int tt = t0 - NHOP + tid;
        while (tt < 0) tt += Tn;
        while (tt >= Tn) tt -= Tn;
        rowoff[tid] = (unsigned)tt * (16u * 2u * (unsigned)kp.nd);
    }
    if (FEAT) {
        for (int i = tid; i < 3 * K3_FT * K3_OW / 4; i += K3_NT) ((float4 *)otile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    float *of = FEAT ? out_feat + ((long)b * kp.OC + 4) * Tn * kp.F : nullptr; // channels 4-6 of this clip, [3][T][F]
    double *oe = FEAT ? nullptr : out_eig + (long)b * 3 * kp.nd * Tn;
    unsigned char *og = (!FEAT && gate) ? gate + (long)b * kp.nd * Tn : nullptr;
    auto emit = [&](int t, int bin, const double *e, unsigned char g) {
        if (FEAT) {
#pragma unroll
            for (int i = 0; i < 3; i++) otile[(i * K3_FT + (t - t0)) * K3_OW + (bin - bin0)] = (float)e[i];
        } else {
#pragma unroll
            for (int i = 0; i < 3; i++) oe[((long)i * kp.nd + bin) * Tn + t] = e[i];
            if (og) og[(long)bin * Tn + t] = g;
        }
    };
    const double zero3[3] = {0.0, 0.0, 0.0};
    {
        // Compaction.  lane = bin; a wave covers exactly one 64-bin group, so the gate mask of (frame, group) -- one ballot
        // over the lanes' history bits -- is a wave-uniform scalar whose bits ARE the lanes to keep.  Each wave counts its
        // K3_FT masks with scalar popcounts, reserves its slice of the work list with ONE LDS atomic, and every kept lane
        // drops its (frame, bin) at slice + (bits below it): lane order survives, so neighbouring list entries are
        // neighbouring bins.
        const int bl = tid;
        const bool in = bl < nbc;
        const int ng32 = (kp.nd + TR_BINS - 1) / TR_BINS;
        const int bin = bin0 + bl;
        const int lane = tid & 63;
        const int grp = __builtin_amdgcn_readfirstlane(bin >> 6); // the wave's 64-bin group = two of the tracker's 32-bin groups
        const unsigned long long inmask = __ballot(in);
        // the tracker's per-frame masks of this wave's bins (two 32-bin groups x K3_FT frames): ONE vector load -- lane l takes
        // frame l % K3_FT of half l / K3_FT -- then readlane; the masks are wave-uniform scalars and no ballot is needed.
        // (Scalar loads here were measured 40 % slower for the whole kernel: sixteen serialised scalar-cache misses per wave.)
        static_assert(2 * K3_FT <= 64, "one lane per (half, frame)");
        unsigned myw = 0u;
        if (kp.tracking && lane < 2 * K3_FT) {
            const int ft = lane % K3_FT, half = lane / K3_FT;
            if (ft < nft && 2 * grp + half < ng32) myw = valid32[((long)b * ng32 + 2 * grp + half) * Tn + t0 + ft];
        }
        unsigned long long words[K3_FT];
        int total = 0;
#pragma unroll
        for (int ft = 0; ft < K3_FT; ft++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)myw, ft), hi = (unsigned)__builtin_amdgcn_readlane((int)myw, K3_FT + ft);
            const unsigned long long tr = (((unsigned long long)hi << 32) | lo) & inmask;
            words[ft] = kp.tracking ? tr : (ft < nft ? inmask : 0ull);
        }
        // A work item is a GROUP of G neighbouring frames of one bin with at least one of them gated in.
        auto any_of = [&](int ft) {
            unsigned long long wd = 0ull;
#pragma unroll
            for (int j = 0; j < G; j++) wd |= words[ft + j];
            return wd;
        };
#pragma unroll
        for (int ft = 0; ft < K3_FT; ft += G) total += __popcll(any_of(ft));
        int base = 0;
        if (lane == 0 && total) base = atomicAdd(&count, total);
        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
        for (int ft = 0; ft < K3_FT; ft += G) {
            const unsigned long long wd = any_of(ft);
            if (in) {
                unsigned v = 0; // which frames of the group are gated in
#pragma unroll
                for (int j = 0; j < G; j++) v |= ((unsigned)(words[ft + j] >> lane) & 1u) << j;
                if (v) list[base + __popcll(wd & ((1ull << lane) - 1))] = (unsigned short)((v << 12) | ((ft / G) << 8) | bl);
#pragma unroll
                for (int j = 0; j < G; j++)
                    if (!FEAT && ft + j < nft && !((v >> j) & 1)) emit(t0 + ft + j, bin, zero3, 0);
            }
            base += __popcll(wd);
        }
    }
    // columns of this block's rows: its bins, plus -- in the last block -- the zeros above the DOA band up to F (:373-374)
    const int seg = (FEAT && blockIdx.z == gridDim.z - 1) ? (kp.F - bin0 < K3_OW ? kp.F - bin0 : K3_OW) : nbc;
    if (FEAT && blockIdx.z == gridDim.z - 1 && kp.F - bin0 > K3_OW) { // (wider than the LDS tile: the rest directly)
        const int first = bin0 + K3_OW, tail = kp.F - first;
        for (int i = tid; i < nft * tail; i += K3_NT) {
            const int ft = i / tail, f = first + (i - ft * tail);
#pragma unroll
            for (int c = 0; c < 3; c++) st_off(of, 4u * (unsigned)((c * Tn + t0 + ft) * kp.F + f), 0.f);
        }
    }
    __syncthreads();
    const int n = count;
    const int stride = 2 * kp.nd;
    const int nhop = NHOP >= 0 ? NHOP : kp.n_hop;
    const float4 *xclip = Xs + (long)b * Tn * stride;
    // :111-112 the coherence test only gates when tracking -- except in contrib, whose test always gates (:352-354)
    const bool ungated = !kp.tracking && !kp.flex;
    auto solve_emit = [&](const salsa::herm4<double> &R, int t, int bin, auto path) {
        constexpr int PATH = decltype(path)::value;
        const bool foa = kp.format == SALSA_FORMAT_FOA;
        const salsa::eig_result<double> er = salsa::herm4_gate_eigvec<PATH>(R, kp.cond, kp.inv_cond, ungated, !foa);
        if (PATH == 1 && er.fallback) { // passed the gate, pivot too small: the cold loop below takes it
            slow[atomicAdd(&nslow, 1)] = (unsigned short)(((t - t0) << 8) | (bin - bin0));
            return;
        }
        if (er.doubt && kp.doubt32 && !ungated) { // the threshold sits ON a root of the quartic: what is emitted below is provisional,
            atomicOr(&kp.doubt32[((long)b * ((kp.nd + TR_BINS - 1) / TR_BINS) + (bin >> 5)) * Tn + t], 1u << (bin & 31)); // gate_doubt_kernel decides
            atomicOr(kp.doubt_flag, 1u);
        }
        double e[3] = {0.0, 0.0, 0.0};
        unsigned char g = er.rank1 ? 2 : 1;
        if (er.rank1 || ungated) {
            const int k = bin + kp.lower;
            // delta*k (:121-123); contrib divides by a float32 frequency vector with [0] = 1 (:188-190)
            const double den = kp.flex ? (double)((float)(k == 0 ? 1 : k) * (float)kp.delta) : kp.delta * (double)k;
            if (PATH != 2 && er.col0) { // gated fast path: column 0 of the adjugate, real pivot (salsa_math.h)
                if (foa) salsa::normalise_foa_col0(er.u, e);
                else salsa::normalise_mic_col0(er.u, den, e);
            } else if (PATH != 1) {
                if (foa) salsa::normalise_foa(er.u, e, ungated);
                else salsa::normalise_mic(er.u, den, e);
            }
            g = 2;
        } else if (FEAT && kp.flex && !kp.tracking) {
            e[0] = __builtin_nan(""); // marks "failed the test" for flex_allpass_kernel (a passing bin can be exactly 0)
        }
        emit(t, bin, e, g);
    };
    using hot_path = std::integral_constant<int, FAST ? 1 : 2>;
    using cold_path = std::integral_constant<int, 2>;
    if constexpr (PK) {
        static_assert(!PK || (G == 2 && FEAT && FAST && NHOP >= 0), "the packed solve takes the two frames of a pair");
        // Float32 all the way.  (Tried and dropped: issuing the NEXT item's sixteen gathers between this item's covariance and
        // its solve.  The solve needs ~130 registers by itself, so the prefetched 64 push the loop to 195 VGPRs -- 2 waves per
        // SIMD -- or, held to 128 / 168, into scratch: 0.50 - 0.88 ms against 0.38, profiles/r4_k3_pk_ab.txt.)
        constexpr int NW = 2 * NHOP + 2;
        const unsigned half = 16u * (unsigned)kp.nd;
#if K3_STAGE_LDS
        if (n > 0) { // (a tile with nothing gated stages nothing)
            for (int i = tid; i < (K3_FT + 6) * 2 * K3_NT; i += K3_NT) {
                const int fr = i / (2 * K3_NT), rem = i - fr * 2 * K3_NT, pr = rem / K3_NT, bl = rem - pr * K3_NT;
                if (bl < nbc) stage[i] = ld_off(xclip, rowoff[fr] + 16u * (unsigned)(bin0 + bl) + (pr ? half : 0u));
            }
        }
        __syncthreads();
#endif
        for (int s = tid; s < n; s += K3_NT) {
            const int i = list[s];
            const int ft = 2 * ((i >> 8) & 15), bl = i & 255; // entry = validity of the pair's frames << 12 | pair << 8 | bin
            float4 xa[NW], xc[NW];
            {
                const unsigned *ro = rowoff + ft; // frames t - NHOP .. t + 1 + NHOP (np.pad(..., 'wrap') on the time axis, :43)
                const unsigned boff = 16u * (unsigned)(bin0 + bl);
#pragma unroll
                for (int k = 0; k < NW; k++) {
#if K3_STAGE_LDS
                    xa[k] = stage[((ft + k) * 2 + 0) * K3_NT + bl];
                    xc[k] = stage[((ft + k) * 2 + 1) * K3_NT + bl];
                    (void)ro; (void)boff;
#else
                    const unsigned r = ro[k];
                    xa[k] = K3_GATHER_NT ? ld_off_nt(xclip, r + boff) : ld_off(xclip, r + boff);
                    xc[k] = K3_GATHER_NT ? ld_off_nt(xclip, r + boff + half) : ld_off(xclip, r + boff + half);
#endif
                }
            }
            auto chans = [&](int k, salsa::pk2f *v) {
                v[0] = salsa::pk2f{xa[k].x, xa[k].y};
                v[1] = salsa::pk2f{xa[k].z, xa[k].w};
                v[2] = salsa::pk2f{xc[k].x, xc[k].y};
                v[3] = salsa::pk2f{xc[k].z, xc[k].w};
            };
            // the (re, im)-packed covariance of the six shared frames, one more frame for each window (both windows, also when
            // only one frame is gated in: the other half of every packed instruction is free)
            salsa::cov4pk Cc = {}, C0, C1;
            salsa::pk2f v[4];
#pragma unroll
            for (int k = 1; k <= 2 * NHOP; k++) {
                chans(k, v);
                salsa::cov4pk_rank1(Cc, Cc, v);
            }
            chans(0, v);
            salsa::cov4pk_rank1(C0, Cc, v);
            chans(2 * NHOP + 1, v);
            salsa::cov4pk_rank1(C1, Cc, v);
            const int live = (i >> 12) & 3;
            int odd;
            const salsa::herm4<salsa::pk2f> A = salsa::herm4_pk_from_windows(C0, C1, odd);
            salsa::pk2f e[3];
            salsa::pk_eig r;
            if (kp.format == SALSA_FORMAT_FOA) {
                r = salsa::herm4_gate_eigvec_pk<false>(A, (float)kp.cond, (float)kp.inv_cond, live & ~odd);
                if (r.pass) salsa::normalise_foa_pk(r, e);
            } else {
                r = salsa::herm4_gate_eigvec_pk<true>(A, (float)kp.cond, (float)kp.inv_cond, live & ~odd);
                if (r.pass) salsa::normalise_mic_pk(r, (float)(kp.delta * (double)(bin0 + bl + kp.lower)), e);
            }
            r.unsure |= odd & live;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if ((r.unsure >> j) & 1) // float64 decides: the cold loop below
                    slow[atomicAdd(&nslow, 1)] = (unsigned short)(((ft + j) << 8) | bl);
                else if ((r.pass >> j) & 1) {
#pragma unroll
                    for (int q = 0; q < 3; q++) otile[(q * K3_FT + ft + j) * K3_OW + bl] = e[q][j];
                }
            }
        }
    }
    for (int s = tid; !PK && s < n; s += K3_NT) {
        const int i = list[s];
        const int t = t0 + G * ((i >> 8) & 15); // entry = validity of the group's frames << 12 | group << 8 | bin
        const int bin = bin0 + (i & 255);
        const float4 *xb = xclip + bin;
        const unsigned row = 16u * (unsigned)stride, boff = 16u * (unsigned)bin, half = 16u * (unsigned)kp.nd; // bytes
        if (PAIRED) {
            // frames t-NHOP .. t+1+NHOP as independent 16-B loads issued together: the two windows share 2*NHOP frames, so a
            // pair costs 2*NHOP+2 gathers instead of 2*(2*NHOP+1) (the gate mask is bursty in time: 1.8 of 2 frames of a
            // listed pair are gated in on the bench clips).  Splitting the loads into batches to save VGPRs was measured
            // slower: their latency is what this kernel hides.
            constexpr int NW = NHOP >= 0 ? 2 * NHOP + G : G;
            float4 xa[NW], xc[NW];
            const unsigned *ro = rowoff + (t - t0); // frames t - NHOP .. (np.pad(..., 'wrap') on the time axis, :43: the table)
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const unsigned r = ro[k];
                xa[k] = ld_off(xclip, r + boff);
                xc[k] = ld_off(xclip, r + boff + half);
            }
            auto frame = [&](int k, salsa::herm4<double> &A) {
                const cplx<double> x[4] = {{(double)xa[k].x, (double)xa[k].y}, {(double)xa[k].z, (double)xa[k].w},
                                           {(double)xc[k].x, (double)xc[k].y}, {(double)xc[k].z, (double)xc[k].w}};
                salsa::herm4_rank1_add(A, x);
            };
            salsa::herm4<double> Rc = {}; // the frames every window of the group contains
#pragma unroll
            for (int k = G - 1; k <= 2 * NHOP; k++) frame(k, Rc);
#pragma unroll
            for (int j = 0; j < G; j++) {
                if ((i >> (12 + j)) & 1) {
                    salsa::herm4<double> R = Rc;
#pragma unroll
                    for (int k = j; k < G - 1; k++) frame(k, R);
#pragma unroll
                    for (int k = 2 * NHOP + 1; k <= 2 * NHOP + j; k++) frame(k, R);
                    solve_emit(R, t + j, bin, hot_path{});
                }
            }
        } else {
            salsa::herm4<double> R = {};
            for (int k = -nhop; k <= nhop; k++) {
                int tt = t + k;
                while (tt < 0) tt += Tn;
                while (tt >= Tn) tt -= Tn;
                const float4 a = xb[tt * stride], c = xb[tt * stride + kp.nd];
                const cplx<double> x[4] = {{(double)a.x, (double)a.y}, {(double)a.z, (double)a.w},
                                           {(double)c.x, (double)c.y}, {(double)c.z, (double)c.w}};
                salsa::herm4_rank1_add(R, x);
            }
            solve_emit(R, t, bin, hot_path{});
        }
    }
    if (FAST) { // cold loop: the few gated bins the hot loop could not finish, general path, one frame per item
        __syncthreads();
        const int ns = nslow;
        if (kp.stats && tid == 0) { // (verification counters; NULL in production)
            int live = 0;
            for (int s = 0; s < n; s++) live += __popc((list[s] >> 12) & 15);
            atomicAdd(&kp.stats[0], (unsigned long long)n);
            atomicAdd(&kp.stats[1], (unsigned long long)live);
            atomicAdd(&kp.stats[2], (unsigned long long)ns);
            atomicAdd(&kp.stats[3], 1ull);
        }
        for (int s = tid; s < ns; s += K3_NT) {
            const int i = slow[s];
            const int t = t0 + (i >> 8), bin = bin0 + (i & 255);
            const float4 *xb = xclip + bin;
            salsa::herm4<double> R = {};
            for (int k = -nhop; k <= nhop; k++) {
                int tt = t + k;
                while (tt < 0) tt += Tn;
                while (tt >= Tn) tt -= Tn;
                const float4 a = xb[tt * stride], c = xb[tt * stride + kp.nd];
                const cplx<double> x[4] = {{(double)a.x, (double)a.y}, {(double)a.z, (double)a.w},
                                           {(double)c.x, (double)c.y}, {(double)c.z, (double)c.w}};
                salsa::herm4_rank1_add(R, x);
            }
            solve_emit(R, t, bin, cold_path{});
        }
    }
    if (FEAT) { // write the tile out: 3 channels x nft frames, `seg` consecutive floats each
        __syncthreads();
        const bool vec = !(kp.F & 3) && !(seg & 3) && !(bin0 & 3); // rows start and end on 16-byte boundaries
        if (vec) {
            const int q = seg >> 2;
            for (int i = tid; i < 3 * nft * q; i += K3_NT) {
                const int row = i / q, col = i - row * q, c = row / nft, ft = row - c * nft;
                float4 *dst = (float4 *)(of + ((long)(c * Tn + t0 + ft) * kp.F + bin0 + 4 * col));
                const float4 val = *(const float4 *)(otile + (c * K3_FT + ft) * K3_OW + 4 * col);
                if (K3_OUT_NT) st_off_nt(dst, 0u, val); // (production default since round 5: rows written once, never re-read by this kernel)
                else *dst = val;
            }
        } else {
            for (int i = tid; i < 3 * nft * seg; i += K3_NT) {
                const int row = i / seg, col = i - row * seg, c = row / nft, ft = row - c * nft;
                of[(long)(c * Tn + t0 + ft) * kp.F + bin0 + col] = otile[(c * K3_FT + ft) * K3_OW + col];
            }
        }
    }
}

// The TF bins cov_eig_kernel flagged in kp.doubt32 -- the threshold mu1 / cond numerically ON a root of the characteristic quartic, which
// then cannot decide "s0 > s1 * cond" (:106): multiple eigenvalues at the threshold lose float64 to sqrt / cube-root precision -- decided
// on the matrix: float64 covariance from the spill, eigenvalues by Jacobi rotations (salsa_math.h herm4_rank1_by_jacobi, the oracle's
// method), eigenvector by the general adjugate path, and the bin's three values rewritten.  On natural signals nothing is flagged:
// the launch (64 workgroups) reads the group's flag word and exits; only a flagged launch scans the mask.  Round 6.
template <bool FEAT>
__global__ __launch_bounds__(256) void gate_doubt_kernel(const KParams kp, const float4 *__restrict__ Xs, float *__restrict__ out_feat,
                                                         double *__restrict__ out_eig, unsigned char *__restrict__ gate)
{
    if (*(const volatile unsigned *)kp.doubt_flag == 0u) return; // nothing flagged (every natural signal): one load per wave
    const int Tn = kp.T, n32 = (kp.nd + TR_BINS - 1) / TR_BINS, stride = 2 * kp.nd;
    const long nwords = (long)kp.B * n32 * Tn;
    const bool foa = kp.format == SALSA_FORMAT_FOA;
    for (long wi = (long)blockIdx.x * blockDim.x + threadIdx.x; wi < nwords; wi += (long)gridDim.x * blockDim.x) {
        unsigned word = kp.doubt32[wi];
        if (!word) continue;
        const int t = (int)(wi % Tn), g32 = (int)((wi / Tn) % n32), b = (int)(wi / ((long)Tn * n32));
        while (word) {
            const int j = __ffs((int)word) - 1;
            word &= word - 1u;
            const int bin = 32 * g32 + j;
            const float4 *xb = Xs + (long)b * Tn * stride + bin;
            salsa::herm4<double> R = {};
            for (int k = -kp.n_hop; k <= kp.n_hop; k++) { // the cold loop's covariance, term for term
                int tt = t + k;
                while (tt < 0) tt += Tn;
                while (tt >= Tn) tt -= Tn;
                const float4 a = xb[tt * stride], c = xb[tt * stride + kp.nd];
                const cplx<double> x[4] = {{(double)a.x, (double)a.y}, {(double)a.z, (double)a.w},
                                           {(double)c.x, (double)c.y}, {(double)c.z, (double)c.w}};
                salsa::herm4_rank1_add(R, x);
            }
            const bool rank1 = salsa::herm4_rank1_by_jacobi(R, kp.cond);
            double e[3] = {0.0, 0.0, 0.0};
            if (rank1) {
                const salsa::eig_result<double> er = salsa::herm4_gate_eigvec<2>(R, kp.cond, kp.inv_cond, true, !foa);
                const int k = bin + kp.lower;
                const double den = kp.flex ? (double)((float)(k == 0 ? 1 : k) * (float)kp.delta) : kp.delta * (double)k;
                if (foa) salsa::normalise_foa(er.u, e, false);
                else salsa::normalise_mic(er.u, den, e);
            } else if (FEAT && kp.flex && !kp.tracking) {
                e[0] = __builtin_nan(""); // "failed the test" for flex_allpass_kernel, as cov_eig_kernel marks it
            }
            if (FEAT) {
#pragma unroll
                for (int i = 0; i < 3; i++) out_feat[(((long)b * kp.OC + 4 + i) * Tn + t) * kp.F + bin] = (float)e[i];
            } else {
#pragma unroll
                for (int i = 0; i < 3; i++) out_eig[(((long)b * 3 + i) * kp.nd + bin) * Tn + t] = e[i];
                if (gate) gate[((long)b * kp.nd + bin) * Tn + t] = rank1 ? 2 : 1;
            }
        }
    }
}

template <bool FEAT>
static void launch_gate_doubt(const KParams &kp, hipStream_t s, const float4 *Xs, float *out_feat, double *out_eig, unsigned char *gate)
{
    if (!kp.doubt32) return;
    const long nwords = (long)kp.B * ((kp.nd + TR_BINS - 1) / TR_BINS) * kp.T;
    const unsigned blocks = (unsigned)(nwords < 256L * 64 ? (nwords + 255) / 256 : 64);
    hipLaunchKernelGGL(gate_doubt_kernel<FEAT>, dim3(blocks ? blocks : 1u), dim3(256), 0, s, kp, Xs, out_feat, out_eig, gate);
}

template <bool FEAT>
static void launch_cov_eig(const KParams &kp, dim3 grid, hipStream_t s, const float4 *Xs, const unsigned *valid,
                           float *out_feat, double *out_eig, unsigned char *gate)
{
    const bool gated = kp.tracking || kp.flex; // (!ungated: the coherence test decides, so passing bins have a spectral gap)
    if (kp.doubt32 && !kp.tracking) (void)hipMemsetAsync(kp.doubt_flag, 0, sizeof(unsigned), s); // (no tracker launch zeroed it: contrib's gate without tracking)
    // (the packed pair solve: feature output only -- salsa_eigvec_batch keeps float64 results -- and never for contrib's variant)
    if (FEAT && SALSA_PK && K3_GROUP == 2 && kp.n_hop == 3 && gated && SALSA_COL0 && !kp.flex && kp.cond >= SALSA_PK_COND_MIN && kp.cond < 1e6 && !kp.force_f64)
        hipLaunchKernelGGL((cov_eig_kernel<FEAT, 3, true, FEAT && K3_GROUP == 2>), grid, dim3(K3_NT), 0, s, kp, Xs, valid, out_feat, out_eig, gate);
    else if (kp.n_hop == 3 && gated && SALSA_COL0)
        hipLaunchKernelGGL((cov_eig_kernel<FEAT, 3, true>), grid, dim3(K3_NT), 0, s, kp, Xs, valid, out_feat, out_eig, gate);
    else if (kp.n_hop == 3)
        hipLaunchKernelGGL((cov_eig_kernel<FEAT, 3>), grid, dim3(K3_NT), 0, s, kp, Xs, valid, out_feat, out_eig, gate);
    else
        hipLaunchKernelGGL((cov_eig_kernel<FEAT, -1>), grid, dim3(K3_NT), 0, s, kp, Xs, valid, out_feat, out_eig, gate);
    launch_gate_doubt<FEAT>(kp, s, Xs, out_feat, out_eig, gate); // (a no-op unless kp.doubt32: gated plans)
}

// ------------------------------------------------------------------------------------------------------------ fused K1 + K3
// Round 5: the STFT spill removed from the path.  One workgroup WALKS a segment of L consecutive frames of one clip in steps of
// FZ_S = 8 frames: its eight waves FFT the step's eight frames (both channel pairs each, exactly stft_kernel's arithmetic), write the
// log-spectrogram rows of the segment's own frames to the output as before -- and the DOA band of the spectra NOT to a 0.94-GB
// spill in HBM but into a ring of FZ_RING = 14 frames in LDS ([frame][pair][bin] float4, the spill's own layout); then the same
// eight waves compact the gated TF bins of the eight frames whose +-3-frame windows are now complete into a work list and run
// cov_eig_kernel's packed-float32 pair solve with the sixteen 16-byte gathers of an item answered by LDS instead of L2 / HBM.
// What is recomputed is the 3-frame halo at either end of a SEGMENT (6 / L of the FFT work: 4 % at L = 152), not of an 8-frame
// tile.  The tracker's masks come from the separate launch before it.
//  * A wave's audio for the NEXT step is loaded into registers before the solve phase of this one (the barriers between the
//    phases order LDS traffic only), so the FFT phase starts on data that has arrived.
//  * The frames the packed solve hands back to float64 (~0.3 % FOA, ~3 % MIC) are not solved inside the step -- one lane's float64
//    solve would hold the whole workgroup at the step's barrier, 7 k cycles for 1.8 frames per step -- but written out as records
//    (the 7-frame window, 240 bytes) into a per-workgroup slice of the workspace and solved together at the end of the segment;
//    a slice that is full (FZ_COLD_CAP records: never on natural signals) falls back to the in-step cold loop.
//
// LDS (dynamic, <= 160 KB; one workgroup per CU): [0, 64 KB) the eight waves' FFT buffers -- reused by the solve phase for the
// output tile of channels 4-6, the work list and the cold list; then the ring (14 x 2 x nd x 16 B: 85.6 KB at nd = 191), the
// window, the row-offset table and counters, the optional scaler tables.
constexpr int FZ_S = 8;
constexpr int FZ_W = 8;
constexpr int FZ_NT = 64 * FZ_W;
constexpr int FZ_RING = FZ_S + 6;
constexpr int FZ_COLD_CAP = 256;    // deferred float64 records per workgroup
constexpr unsigned FZ_OFF_OTILE = 0, FZ_OFF_LIST = 24576, FZ_OFF_SLOW = FZ_OFF_LIST + 2048, FZ_OFF_RING = 65536;
struct fz_cold_rec {
    float4 x[14];                   // window frame k: x[2k] = channel pair 0, x[2k+1] = pair 1
    int t, bl, pad0, pad1;
};
static size_t fused_lds_bytes(int nd, int F, bool scaler) { return FZ_OFF_RING + (size_t)FZ_RING * 32 * nd + 4096 + 128 + (scaler ? 2 * 4 * (size_t)F * 4 : 0); }

template <bool MIC>
__global__ __launch_bounds__(FZ_NT, 1) void fused_kernel(const KParams kp, const float *__restrict__ audio,
                                                        const double *__restrict__ window, const cplx<double> *__restrict__ tw,
                                                        float *__restrict__ out, const unsigned *__restrict__ valid32, const int L,
                                                        fz_cold_rec *__restrict__ cold_all, const int cold_cap)
{
    constexpr int N = 512, R = 8, NP = 2;
    using T = double;
    extern __shared__ __attribute__((aligned(16))) unsigned char fz_lds[];
    const int nd = kp.nd, F = kp.F, Tn = kp.T, Ns = kp.N;
    const int OWs = (F + 3) & ~3;                                    // columns of the output tile (the zero band above nd included)
    const unsigned frame_bytes = 32u * (unsigned)nd, half = 16u * (unsigned)nd;
    cplx<T> *zall = (cplx<T> *)fz_lds;
    float *otile = (float *)(fz_lds + FZ_OFF_OTILE);
    unsigned short *list = (unsigned short *)(fz_lds + FZ_OFF_LIST), *slow = (unsigned short *)(fz_lds + FZ_OFF_SLOW);
    unsigned char *ringb = fz_lds + FZ_OFF_RING;
    T *wins = (T *)(ringb + (size_t)FZ_RING * frame_bytes);
    unsigned *rowoff = (unsigned *)(wins + N);                       // [14] + counters (never aliased by the FFT buffers)
    int *count = (int *)(rowoff + 16), *nslow = count + 1, *ncold = count + 2;
    float *sct0 = (float *)(rowoff + 32), *sct1 = sct0 + 4 * F;

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * L, f1 = f0 + L < Tn ? f0 + L : Tn;   // own frames [f0, f1)
    fz_cold_rec *cold = cold_all + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * FZ_COLD_CAP;
    for (int i = tid; i < N; i += FZ_NT) wins[i] = (T)(0.5 * window[i]);
    if (kp.sc_mean)
        for (int i = tid; i < 4 * F; i += FZ_NT) sct0[i] = kp.sc_mean[i], sct1[i] = kp.sc_std[i];
    if (tid == 0) *count = 0, *nslow = 0, *ncold = 0;
    cplx<T> w1[NP];
    {
        int p = R;
#pragma unroll
        for (int q = 0; q < NP; q++, p *= R) {
            const cplx<double> wd = tw[salsa::stockham_tw(lane, 1, p, N, R)];
            w1[q] = {(T)wd.re, (T)wd.im};
        }
    }
    __syncthreads();
    cplx<T> *z = zall + w * N;
    float *pw = (float *)z;                                          // the compressed band's powers: in the wave's FFT buffer, free after the last pass
    const float *clip = audio + (long)b * 4 * Ns;
    const bool planar = kp.layout == SALSA_LAYOUT_PLANAR;
    const int sstride = planar ? 1 : 4;
    float *o = out + (long)b * kp.OC * Tn * F;
    const int mlane = (64 - lane) & 63;
    const unsigned plane = 4u * (unsigned)(Tn * F);
    auto spec = [&](const float p, const int c, const int f) -> float {
        const float v = db10(p);
        const int i = c * F + f;
        return kp.sc_mean ? (v - sct0[i]) / sct1[i] : v;
    };
    auto wrap_frame = [&](const int vf) { // np.pad(..., 'wrap') on the time axis (:43): the halo of the clip's first / last segment
        int t = vf;
        t = t < 0 ? t + Tn : t;
        return t >= Tn ? t - Tn : t;
    };
    // the 2 x 2 x 8 samples a lane contributes to the two packed FFTs of one frame
    auto load_frame = [&](const int vf, float (&y)[2][2][R]) {
        const int t = wrap_frame(vf);
        const int base = t * kp.hop - N / 2;
        const unsigned step = 4u * (unsigned)sstride;
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {
            const int c0 = 2 * pr;
            const unsigned ch0 = 4u * (unsigned)(planar ? c0 * Ns : c0), ch1 = ch0 + 4u * (unsigned)(planar ? Ns : 1);
#ifdef FZ_PROBE_NOLOAD
            if (true) {
#pragma unroll
                for (int r = 0; r < R; r++) y[pr][0][r] = (float)(lane + r + t) * 1e-3f, y[pr][1][r] = (float)(lane - r + c0) * 1e-3f;
            } else
#endif
            if (base >= 0 && base + N <= Ns) {
                const unsigned q = (unsigned)(base + lane) * step;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    y[pr][0][r] = ld_off(clip, ch0 + q + (unsigned)(r * (N / R)) * step);
                    y[pr][1][r] = ld_off(clip, ch1 + q + (unsigned)(r * (N / R)) * step);
                }
            } else { // a frame that overlaps a clip end: np.pad(mode='reflect')
#pragma unroll
                for (int r = 0; r < R; r++) {
                    int sidx = base + salsa::stockham_in(lane, r, N, R);
                    sidx = sidx < 0 ? -sidx : sidx;
                    sidx = sidx >= Ns ? 2 * (Ns - 1) - sidx : sidx;
                    y[pr][0][r] = ld_off(clip, ch0 + (unsigned)sidx * step);
                    y[pr][1][r] = ld_off(clip, ch1 + (unsigned)sidx * step);
                }
            }
        }
    };
    // one frame: both channel pairs (stft_kernel's item, twice)
    auto fft_frame = [&](const int vf, const float (&y)[2][2][R]) {
        const int t = wrap_frame(vf);
        const bool own = vf >= f0 && vf < f1;
        const int sl = (vf - (f0 - 3)) % FZ_RING;
        float4 *slot = (float4 *)(ringb + (unsigned)sl * frame_bytes);
#pragma unroll 1
        for (int pr = 0; pr < 2; pr++) { // (rolled: unrolled, the two transforms' live ranges overlap and spill)
            const int c0 = 2 * pr;
            cplx<T> v[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const T wn = wins[salsa::stockham_in(lane, r, N, R)];
                const float ya = pr ? y[1][0][r] : y[0][0][r], yb = pr ? y[1][1][r] : y[0][1][r]; // (wave-uniform selects)
                v[r] = {wn * (T)ya, wn * (T)yb};
            }
            salsa::dftR<R>(v);
#pragma unroll
            for (int r = 0; r < R; r++) z[swz(salsa::stockham_out(lane, r, 1, R))] = v[r];
            {
                int p = R;
#pragma unroll
                for (int q = 0; q < NP; q++, p *= R) {
                    wave_lds_fence();
#pragma unroll
                    for (int r = 0; r < R; r++) v[r] = z[swz(salsa::stockham_in(lane, r, N, R))];
                    wave_lds_fence();
                    {
                        const cplx<T> a1 = w1[q], a2 = salsa::cmul(a1, a1), a3 = salsa::cmul(a2, a1), a4 = salsa::cmul(a2, a2);
                        v[1] = salsa::cmul(v[1], a1);
                        v[2] = salsa::cmul(v[2], a2);
                        v[3] = salsa::cmul(v[3], a3);
                        v[4] = salsa::cmul(v[4], a4);
                        v[5] = salsa::cmul(v[5], salsa::cmul(a4, a1));
                        v[6] = salsa::cmul(v[6], salsa::cmul(a3, a3));
                        v[7] = salsa::cmul(v[7], salsa::cmul(a4, a3));
                    }
                    salsa::dftR<R>(v);
                    if (q + 1 < NP) {
#pragma unroll
                        for (int r = 0; r < R; r++) z[swz(salsa::stockham_out(lane, r, p, R))] = v[r];
                    }
                }
            }
            auto emit_bin = [&](const int k, const cplx<T> a, const cplx<T> bm) {
                cplx<T> Xa, Xb;
                salsa::unpack_pair_prescaled(a, bm, Xa, Xb);
                const float2 xa = make_float2((float)Xa.re, (float)Xa.im); // the reference stores its STFT as complex64
                const float2 xb = make_float2((float)Xb.re, (float)Xb.im);
                if (k >= kp.lower && k < kp.upper) slot[pr * nd + (k - kp.lower)] = make_float4(xa.x, xa.y, xb.x, xb.y);
                if (own) {
                    const float pa = power32(xa), pb = power32(xb);
                    if (k >= kp.spec_lo && k < kp.spec_hi) {
                        const unsigned off = 4u * (unsigned)((c0 * Tn + t) * F + (k - kp.spec_lo));
                        st_off(o, off, spec(pa, c0, k - kp.spec_lo));
                        st_off(o, off + plane, spec(pb, c0 + 1, k - kp.spec_lo));
                    } else if (kp.compress && k > kp.ident && k < N / 2) {
                        pw[k - kp.ident - 1] = pa;
                        pw[64 + k - kp.ident - 1] = pb;
                    }
                }
            };
#pragma unroll
            for (int r = 0; r < R / 2; r++) {
                cplx<T> bm = {__shfl(v[R - 1 - r].re, mlane), __shfl(v[R - 1 - r].im, mlane)};
                if (lane == 0) bm = v[(R - r) & (R - 1)];
                emit_bin(lane + 64 * r, v[r], bm);
                __builtin_amdgcn_sched_barrier(0);
            }
            // (the Nyquist bin belongs to no spectrogram row and no DOA band of the default configuration: a wave-uniform skip)
            if ((kp.spec_hi > N / 2 || kp.upper > N / 2) && lane == 0) emit_bin(N / 2, v[R / 2], v[R / 2]);
            if (own && kp.compress) {
                wave_lds_fence();
                const int ng = F - kp.ident;
                const int h = lane & 1, gi = lane >> 1;
                if (gi < ng) {
                    const int cnt = gi < ng - 1 ? 8 : 7;
                    float acc = 0.f;
                    for (int q = 0; q < cnt; q++) acc += 0.125f * pw[64 * h + 8 * gi + q];
                    st_off(o, 4u * (unsigned)(((c0 + h) * Tn + t) * F + kp.ident + gi), spec(acc, c0 + h, kp.ident + gi));
                }
            }
            wave_lds_fence();
        }
    };
    auto lds_barrier = [&]() { // orders LDS traffic only: global loads / stores stay in flight across it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // float64 covariance of one 7-frame window + the general solve (cov_eig_kernel's cold path); e[] = the feature or zeros
    auto cold_solve = [&](auto getx, const int bl, double *e) -> bool { // getx(k, a, c): the two float4 of window frame k
        salsa::herm4<double> Rm = {};
#pragma unroll 1
        for (int k = 0; k < 7; k++) {
            float4 a, c;
            getx(k, a, c);
            const cplx<double> x[4] = {{(double)a.x, (double)a.y}, {(double)a.z, (double)a.w},
                                       {(double)c.x, (double)c.y}, {(double)c.z, (double)c.w}};
            salsa::herm4_rank1_add(Rm, x);
        }
        salsa::eig_result<double> er = salsa::herm4_gate_eigvec<2>(Rm, kp.cond, kp.inv_cond, false, MIC);
        if (er.doubt) { // the threshold sits ON a root of the quartic (salsa_math.h SALSA_GATE_DOUBT): decided on the matrix, in place --
                        // the three-kernel path's gate_doubt_kernel reads the spill, which this kernel's records have overwritten
            const bool r1 = salsa::herm4_rank1_by_jacobi(Rm, kp.cond);
            if (r1) er = salsa::herm4_gate_eigvec<2>(Rm, kp.cond, kp.inv_cond, true, MIC);
            er.rank1 = r1;
        }
        if (!er.rank1) return false;
        if (MIC) salsa::normalise_mic(er.u, kp.delta * (double)(bl + kp.lower), e);
        else salsa::normalise_foa(er.u, e, false);
        return true;
    };

    float y[2][2][R];
    // prologue: the six frames around the segment's first step that no step produces (v = f0-3 .. f0+2)
#ifndef FZ_PROBE_NOFFT
    if (w < 6) {
        load_frame(f0 - 3 + w, y);
        fft_frame(f0 - 3 + w, y);
    }
    if (f0 + 3 + w < f1 + 3) load_frame(f0 + 3 + w, y);
#endif
    const int ng32 = (nd + TR_BINS - 1) / TR_BINS, ng64 = (nd + 63) >> 6;
    float *of = out + ((long)b * kp.OC + 4) * Tn * F;                 // channels 4-6 of this clip, [3][T][F]
    const float fcond = (float)kp.cond, finv = (float)kp.inv_cond;
    for (int fs0 = f0; fs0 < f1; fs0 += FZ_S) {
        const int nft = f1 - fs0 < FZ_S ? f1 - fs0 : FZ_S;
        // the tracker's masks of this step's frames (two 32-bin groups x 8 frames per wave of the compaction): issued before the
        // FFTs, first used after them
        unsigned myw = 0u;
        if (w < ng64 && lane < 2 * FZ_S) {
            const int ft = lane % FZ_S, hf = lane / FZ_S;
            if (ft < nft && 2 * w + hf < ng32) myw = valid32[((long)b * ng32 + 2 * w + hf) * Tn + fs0 + ft];
        }
        // ---- FFT phase: frames fs0+3 .. fs0+10 (as far as the segment's window reaches), then the next step's audio on its way
#ifndef FZ_PROBE_NOFFT
        {
            const int vf = fs0 + 3 + w;
            if (vf < f1 + 3) fft_frame(vf, y);
            if (fs0 + FZ_S < f1 && vf + FZ_S < f1 + 3) load_frame(vf + FZ_S, y);
        }
#endif
        lds_barrier(); // B1: the ring holds fs0-3 .. fs0+10; the FFT buffers are free
#ifdef FZ_PROBE_NOSOLVE
        if (myw == 0xdeadbeefu) otile[tid] = 1.f;
        continue;
#endif
        // ---- output tile zeroed, row offsets, compaction of the gated TF bins (waves 0 .. ng64-1: one 64-bin group each)
        for (int i = tid; i < 3 * FZ_S * OWs / 4; i += FZ_NT) ((float4 *)otile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < FZ_RING) rowoff[tid] = (unsigned)((fs0 - f0 + tid) % FZ_RING) * frame_bytes;
        if (w < ng64) {
            const int bl = tid;
            const bool in = bl < nd;
            const unsigned long long inmask = __ballot(in);
            unsigned long long words[FZ_S];
            int total = 0;
#pragma unroll
            for (int ft = 0; ft < FZ_S; ft++) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)myw, ft), hi = (unsigned)__builtin_amdgcn_readlane((int)myw, FZ_S + ft);
                words[ft] = (((unsigned long long)hi << 32) | lo) & inmask;
            }
#pragma unroll
            for (int ft = 0; ft < FZ_S; ft += 2) total += __popcll(words[ft] | words[ft + 1]);
            int base = 0;
            if (lane == 0 && total) base = atomicAdd(count, total);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int ft = 0; ft < FZ_S; ft += 2) {
                const unsigned long long wd = words[ft] | words[ft + 1];
                if (in) {
                    const unsigned v = ((unsigned)(words[ft] >> lane) & 1u) | (((unsigned)(words[ft + 1] >> lane) & 1u) << 1);
                    if (v) list[base + __popcll(wd & ((1ull << lane) - 1))] = (unsigned short)((v << 12) | ((ft / 2) << 8) | bl);
                }
                base += __popcll(wd);
            }
        }
        lds_barrier(); // B2
        const int n = *count;
        // ---- hot loop: the packed-float32 pair solve of cov_eig_kernel<true, 3, true, true>, windows read from the ring
        for (int s = tid; s < n; s += FZ_NT) {
            const int i = list[s];
            const int ft = 2 * ((i >> 8) & 15), bl = i & 255;
            constexpr int NW = 8;
            const unsigned *ro = rowoff + ft;
            const unsigned boff = 16u * (unsigned)bl;
            float4 xa[NW], xc[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const unsigned r = ro[k];
                xa[k] = *(const float4 *)(ringb + r + boff);
                xc[k] = *(const float4 *)(ringb + r + boff + half);
            }
            auto chans = [&](int k, salsa::pk2f *v) {
                v[0] = salsa::pk2f{xa[k].x, xa[k].y};
                v[1] = salsa::pk2f{xa[k].z, xa[k].w};
                v[2] = salsa::pk2f{xc[k].x, xc[k].y};
                v[3] = salsa::pk2f{xc[k].z, xc[k].w};
            };
            salsa::cov4pk Cc = {}, C0, C1;
            salsa::pk2f v[4];
#pragma unroll
            for (int k = 1; k <= 6; k++) {
                chans(k, v);
                salsa::cov4pk_rank1(Cc, Cc, v);
            }
            chans(0, v);
            salsa::cov4pk_rank1(C0, Cc, v);
            chans(7, v);
            salsa::cov4pk_rank1(C1, Cc, v);
            const int live = (i >> 12) & 3;
            int odd;
            const salsa::herm4<salsa::pk2f> A = salsa::herm4_pk_from_windows(C0, C1, odd);
            salsa::pk2f e[3];
            salsa::pk_eig r = salsa::herm4_gate_eigvec_pk<MIC>(A, fcond, finv, live & ~odd);
            if (r.pass) {
                if (MIC) salsa::normalise_mic_pk(r, (float)(kp.delta * (double)(bl + kp.lower)), e);
                else salsa::normalise_foa_pk(r, e);
            }
            r.unsure |= odd & live;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if ((r.unsure >> j) & 1) { // float64 decides: a record for the end of the segment (or, slice full, the in-step cold list)
                    const int ci = atomicAdd(ncold, 1);
                    if (ci < cold_cap) {
                        fz_cold_rec *rec = cold + ci;
#pragma unroll
                        for (int k = 0; k < 7; k++) {
                            const unsigned rr = ro[j + k];
                            rec->x[2 * k] = *(const float4 *)(ringb + rr + boff);
                            rec->x[2 * k + 1] = *(const float4 *)(ringb + rr + boff + half);
                        }
                        rec->t = fs0 + ft + j;
                        rec->bl = bl;
                    } else slow[atomicAdd(nslow, 1)] = (unsigned short)(((ft + j) << 8) | bl);
                } else if ((r.pass >> j) & 1) {
#pragma unroll
                    for (int q = 0; q < 3; q++) otile[(q * FZ_S + ft + j) * OWs + bl] = e[q][j];
                }
            }
        }
        lds_barrier(); // B3
        const int ns = *nslow;
        if (kp.stats && tid == 0) {
            int lv = 0;
            for (int s = 0; s < n; s++) lv += __popc((list[s] >> 12) & 3);
            atomicAdd(&kp.stats[0], (unsigned long long)n);
            atomicAdd(&kp.stats[1], (unsigned long long)lv);
            atomicAdd(&kp.stats[3], 1ull);
        }
        if (ns) { // (wave-uniform; only when the record slice overflowed) the in-step cold loop
            for (int s = tid; s < ns; s += FZ_NT) {
                const int i = slow[s];
                const int ft = i >> 8, bl = i & 255;
                double e[3] = {0.0, 0.0, 0.0};
                if (cold_solve([&](int k, float4 &a, float4 &c) {
                        const unsigned rr = rowoff[ft + k] + 16u * (unsigned)bl;
                        a = *(const float4 *)(ringb + rr);
                        c = *(const float4 *)(ringb + rr + half);
                    }, bl, e)) {
#pragma unroll
                    for (int q = 0; q < 3; q++) otile[(q * FZ_S + ft) * OWs + bl] = (float)e[q];
                }
            }
            lds_barrier(); // B4
        }
        // ---- channels 4-6 of the step's frames: whole rows (the zeros above the DOA band included, :373-374)
        if (tid == 0) *count = 0, *nslow = 0;
        if (!(F & 3)) {
            const int q = F >> 2;
            for (int i = tid; i < 3 * nft * q; i += FZ_NT) {
                const int row = i / q, col = i - row * q, c = row / nft, ft = row - c * nft;
                *(float4 *)(of + ((long)(c * Tn + fs0 + ft) * F + 4 * col)) = *(const float4 *)(otile + (c * FZ_S + ft) * OWs + 4 * col);
            }
        } else {
            for (int i = tid; i < 3 * nft * F; i += FZ_NT) {
                const int row = i / F, col = i - row * F, c = row / nft, ft = row - c * nft;
                of[(long)(c * Tn + fs0 + ft) * F + col] = otile[(c * FZ_S + ft) * OWs + col];
            }
        }
        lds_barrier(); // B5: the tile has been read; the next step's FFTs may overwrite it (and the ring's oldest frames)
    }
    // ---- the segment's deferred float64 frames: their rows (zeros there) were stored above by this workgroup; __syncthreads()
    // waits for those stores and for the records, then the passing frames' three values go straight to the output
    __syncthreads();
    const int nc = *ncold < cold_cap ? *ncold : cold_cap;
    if (kp.stats && tid == 0) atomicAdd(&kp.stats[2], (unsigned long long)*ncold);
#ifndef FZ_PROBE_NOCOLD
    for (int s = tid; s < nc; s += FZ_NT) {
        const fz_cold_rec *rec = cold + s;
        const int t = rec->t, bl = rec->bl;
        double e[3] = {0.0, 0.0, 0.0};
        if (cold_solve([&](int k, float4 &a, float4 &c) { a = rec->x[2 * k]; c = rec->x[2 * k + 1]; }, bl, e)) {
#pragma unroll
            for (int q = 0; q < 3; q++) of[(long)(q * Tn + t) * F + bl] = (float)e[q];
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------ K3, N channels
// contrib/salsa_flexible.py takes ANY number of microphones (stacked_covmat_eigh :52-77: an N x N Hermitian eigenproblem per
// gated TF bin).  The 4 x 4 closed form above does not generalise, so 5 - 8 microphones (padded to an even count NCH = 6 | 8
// with a silent channel, which only adds a zero eigenvalue) take this kernel: one lane per (frame, bin), summed covariance of
// the 2*n_hop+1 frames in float64, cyclic complex Jacobi with the rotations accumulated (eigenvalues = the diagonal, eigen-
// vectors = the accumulated columns), gate "largest > second largest * ew_thresh" (:353), feature angle(conj(u_0) u_c) / f
// (:360-362).  A completeness path, not a tuned one: the 2 x NCH^2 float64 matrices spill to scratch.
// NCH > 0: compile-time size, fully unrolled (6 | 8).  NCH == 0: any even count up to MAXN = 16 read from kp.nch -- the same code
// with run-time loop bounds and dynamically indexed scratch arrays (9 - 16 microphones: slower still, and as rare).
constexpr int HERMN_MAX = SALSA_MAX_MICS;
template <int NCH> struct hermn {
    static constexpr int S = NCH > 0 ? NCH : HERMN_MAX;
    double ar[S][S], ai[S][S];
};

template <int NCH>
__device__ __forceinline__ void hermn_rotate(hermn<NCH> &A, hermn<NCH> &V, const int p, const int q, const int n)
{
    const double xr = A.ar[p][q], xi = A.ai[p][q];
    const double r2 = xr * xr + xi * xi;
    if (r2 == 0.0) return;
    const double r = sqrt(r2);
    // a_pq = r e^{i phi}.  U = D G: D_qq = e^{-i phi} makes the pivot real, G is the real Jacobi rotation that zeroes it.
    const double er = xr / r, ei = xi / r;
    const double tau = (A.ar[q][q] - A.ar[p][p]) / (2.0 * r);
    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
    // U_pp = c, U_qp = -sn e^{-i phi}, U_pq = sn, U_qq = c e^{-i phi}
    const double uqp_r = -sn * er, uqp_i = sn * ei, uqq_r = c * er, uqq_i = -c * ei;
    auto cols = [&](hermn<NCH> &M) { // M <- M U (columns p and q)
#pragma unroll
        for (int i = 0; i < n; i++) {
            const double pr = M.ar[i][p], pi = M.ai[i][p], qr = M.ar[i][q], qi = M.ai[i][q];
            M.ar[i][p] = pr * c + (qr * uqp_r - qi * uqp_i);
            M.ai[i][p] = pi * c + (qr * uqp_i + qi * uqp_r);
            M.ar[i][q] = pr * sn + (qr * uqq_r - qi * uqq_i);
            M.ai[i][q] = pi * sn + (qr * uqq_i + qi * uqq_r);
        }
    };
    cols(A);
#pragma unroll
    for (int j = 0; j < n; j++) { // A <- U^H A (rows p and q): conj(U_pp) = c, conj(U_qp), conj(U_pq) = sn, conj(U_qq)
        const double pr = A.ar[p][j], pi = A.ai[p][j], qr = A.ar[q][j], qi = A.ai[q][j];
        A.ar[p][j] = c * pr + (uqp_r * qr + uqp_i * qi);
        A.ai[p][j] = c * pi + (uqp_r * qi - uqp_i * qr);
        A.ar[q][j] = sn * pr + (uqq_r * qr + uqq_i * qi);
        A.ai[q][j] = sn * pi + (uqq_r * qi - uqq_i * qr);
    }
    A.ar[p][q] = A.ai[p][q] = A.ar[q][p] = A.ai[q][p] = 0.0; // exactly, as the algebra says
    A.ai[p][p] = A.ai[q][q] = 0.0;
    cols(V);
}

template <int NCH>
__global__ __launch_bounds__(64) void cov_eig_n_kernel(const KParams kp, const float4 *__restrict__ Xs,
                                                       const unsigned *__restrict__ valid32, float *__restrict__ out)
{
    constexpr int S = hermn<NCH>::S;
    const int n = NCH > 0 ? NCH : kp.nch, NP = n / 2;
    const int t = blockIdx.x, b = blockIdx.y, Tn = kp.T;
    const int ng32 = (kp.nd + TR_BINS - 1) / TR_BINS;
    float *of = out + ((long)b * kp.OC + n) * Tn * kp.F + (long)t * kp.F; // first spatial plane, this frame's row
    const long plane = (long)Tn * kp.F;
    const float4 *xclip = Xs + (long)b * Tn * NP * kp.nd;
    for (int bin = threadIdx.x; bin < kp.F; bin += 64) {
        float e[S - 1];
#pragma unroll
        for (int c = 0; c < S - 1; c++) e[c] = 0.f;
        bool gated = bin < kp.nd;
        if (gated && kp.tracking) gated = (valid32[((long)b * ng32 + (bin >> 5)) * Tn + t] >> (bin & 31)) & 1u;
        if (gated) {
            hermn<NCH> A, V;
#pragma unroll
            for (int i = 0; i < n; i++)
#pragma unroll
                for (int j = 0; j < n; j++) {
                    A.ar[i][j] = A.ai[i][j] = 0.0;
                    V.ar[i][j] = i == j ? 1.0 : 0.0;
                    V.ai[i][j] = 0.0;
                }
            for (int k = -kp.n_hop; k <= kp.n_hop; k++) { // summed covariance, wrap on the time axis (:316-318, :347-349)
                int tt = t + k;
                while (tt < 0) tt += Tn;
                while (tt >= Tn) tt -= Tn;
                double xr[S], xi[S];
#pragma unroll
                for (int pr = 0; pr < NP; pr++) {
                    const float4 v = xclip[((long)tt * NP + pr) * kp.nd + bin];
                    xr[2 * pr] = v.x; xi[2 * pr] = v.y; xr[2 * pr + 1] = v.z; xi[2 * pr + 1] = v.w;
                }
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < n; j++) { // x_i conj(x_j)
                        A.ar[i][j] += xr[i] * xr[j] + xi[i] * xi[j];
                        A.ai[i][j] += xi[i] * xr[j] - xr[i] * xi[j];
                    }
            }
            double tr = 0.0;
#pragma unroll
            for (int i = 0; i < n; i++) tr += A.ar[i][i];
            bool good = false;
            if (tr > 0.0) {
                for (int sweep = 0; sweep < 16; sweep++) {
                    double off = 0.0;
#pragma unroll
                    for (int p = 0; p < n; p++)
#pragma unroll
                        for (int q = p + 1; q < n; q++) off += A.ar[p][q] * A.ar[p][q] + A.ai[p][q] * A.ai[p][q];
                    if (off <= 1e-34 * tr * tr) break;
#pragma unroll
                    for (int p = 0; p < n; p++)
#pragma unroll
                        for (int q = p + 1; q < n; q++) hermn_rotate<NCH>(A, V, p, q, n);
                }
                int i1 = 0;
                double l1 = A.ar[0][0];
#pragma unroll
                for (int i = 1; i < n; i++)
                    if (A.ar[i][i] > l1) { l1 = A.ar[i][i]; i1 = i; }
                double l2 = -1e300;
#pragma unroll
                for (int i = 0; i < n; i++)
                    if (i != i1 && A.ar[i][i] > l2) l2 = A.ar[i][i];
                good = l1 > l2 * kp.cond; // ews[:, -1] > ews[:, -2] * ew_thresh (:353)
                if (good) {
                    double ur[S], ui[S];
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        ur[i] = ui[i] = 0.0;
#pragma unroll
                        for (int j = 0; j < n; j++)
                            if (j == i1) { ur[i] = V.ar[i][j]; ui[i] = V.ai[i][j]; }
                    }
                    const int kb = bin + kp.lower;
                    const double den = (double)((float)(kb == 0 ? 1 : kb) * (float)kp.delta); // float32 norm_freq (:188-190)
#pragma unroll
                    for (int c = 1; c < n; c++) { // angle(conj(u_0) u_c) / f   (:360-362)
                        const double wr = ur[0] * ur[c] + ui[0] * ui[c], wi = ur[0] * ui[c] - ui[0] * ur[c];
                        e[c - 1] = (float)(atan2(wi, wr) / den);
                    }
                }
            }
            if (!good && !kp.tracking) e[0] = __builtin_nanf(""); // marks "failed the test" for flex_allpass_kernel
        }
#pragma unroll
        for (int c = 0; c < n - 1; c++) of[c * plane + bin] = e[c];
    }
}

// reference layout (n_bins, n_frames, 4) complex64 -> internal Xs[b][t][pair][bin] float4
__global__ void relayout_kernel(const float4 *__restrict__ X, float4 *__restrict__ Xs, int B, int nb, int Tn)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * nb * Tn * 2;
    if (idx >= total) return;
    const int bin = (int)(idx % nb);
    long r = idx / nb;
    const int pr = (int)(r % 2);
    r /= 2;
    const int t = (int)(r % Tn);
    const int b = (int)(r / Tn);
    Xs[idx] = X[(((long)b * nb + bin) * Tn + t) * 2 + pr];
}

// ------------------------------------------------------------------------------------------------------------ scaler
// compute_scaler (:204-262) on device: float64 sum and sum of squares over time of the first n_sc channels, per
// frequency.  One block per (clip, channel, tile of 64 frames); lane = frequency (coalesced rows); one float64 atomic
// pair per lane per block.  sums: [2][n_sc][F] (sum, sumsq), accumulated into (caller zeroes it once).
__global__ __launch_bounds__(256) void scaler_accumulate_kernel(const float *__restrict__ feat, int C, int T, int F,
                                                                int n_sc, double *__restrict__ sums)
{
    const int f = threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    const int t0 = blockIdx.x * 64, t1 = t0 + 64 < T ? t0 + 64 : T;
    if (f >= F) return;
    const float *p = feat + (((long)b * C + c) * T) * F + f;
    double s = 0.0, ss = 0.0;
    for (int t = t0; t < t1; t++) {
        const double v = (double)p[(long)t * F];
        s += v;
        ss += v * v;
    }
    atomicAdd(&sums[(long)c * F + f], s);
    atomicAdd(&sums[((long)n_sc + c) * F + f], ss);
}

// normalise-on-load (dataset/database.py:197-202): feature[:n_sc] = (feature[:n_sc] - mean) / std, in place;
// mean/std: [n_sc][F] float32.  Channels >= n_sc (the spatial channels) are left untouched.
__global__ __launch_bounds__(256) void normalize_kernel(float *__restrict__ feat, long rows, int C, int T, int F, int n_sc,
                                                        const float *__restrict__ mean, const float *__restrict__ std)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); // one wave per (b, c, t) row of F floats
    if (row >= rows) return;
    const int t = (int)(row % T);
    const int c = (int)((row / T) % n_sc);
    const long b = row / ((long)T * n_sc);
    float *p = feat + ((b * C + c) * T + t) * F;
    for (int f = threadIdx.x & 63; f < F; f += 64) p[f] = (p[f] - mean[c * F + f]) / std[c * F + f];
}

// contrib/salsa_flexible.py with is_tracking=False: ONE all-pass mask array is created (:336-337) and then narrowed in
// place by "mask[mask] = good_coherence_mask" (:354), so a bin that fails the coherence test once is never looked at
// again in that clip.  cov_eig_kernel marks failures with NaN in channel 4; this pass (lane = bin, coalesced rows,
// sequential in time) zeroes everything from a bin's first failure on.
__global__ __launch_bounds__(256) void flex_allpass_kernel(const KParams kp, float *__restrict__ out)
{
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= kp.nd) return;
    float *of = out + ((long)blockIdx.y * kp.OC + kp.nch) * kp.T * kp.F + bin; // first spatial plane
    bool dead = false;
    for (int t = 0; t < kp.T; t++) {
        const float v = of[t * kp.F];
        dead = dead || (v != v);
        if (dead) {
            for (int c = 0; c < kp.nch - 1; c++) of[(c * kp.T + t) * kp.F] = 0.f;
        }
    }
}

// [rows][T][F] float32 (time-major, what the extract kernels write) -> [rows][F][T] float64 (the freq-major float64
// array contrib/salsa_flexible.py returns, :264).  64 x 64 tiles through LDS (+1 padding), both sides coalesced.
__global__ __launch_bounds__(256) void to_freq_major_kernel(const float *__restrict__ in, double *__restrict__ out, int T, int F)
{
    __shared__ float tile[64][65];
    const long row = blockIdx.z;
    const int t0 = blockIdx.y * 64, f0 = blockIdx.x * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const float *src = in + row * (long)T * F;
    double *dst = out + row * (long)T * F;
    for (int i = ly; i < 64; i += 4)
        if (t0 + i < T && f0 + lx < F) tile[i][lx] = src[(long)(t0 + i) * F + f0 + lx];
    __syncthreads();
    for (int i = ly; i < 64; i += 4)
        if (f0 + i < F && t0 + lx < T) dst[(long)(f0 + i) * T + t0 + lx] = (double)tile[lx][i];
}

// ------------------------------------------------------------------------------------------------------------- resampling
// The resampling step of the reference's loader: librosa.load(path, sr=fs) (salsa_feature_extraction.py:353, lite :93) on a
// file of another native rate calls librosa 0.8.0 core/audio.py::resample -> resampy 0.2.2 (requirements.yml:181)
// resample(x, sr_orig, sr_new, filter='kaiser_best'): a windowed-sinc interpolator whose inner loop (resampy/interpn.py
// ::resample_f, numba) walks the filter's left wing from sample n = int(time_register) downwards and its right wing from
// n + 1 upwards, with the filter linearly interpolated between table entries and the float32 output element updated in
// place -- i.e. every tap is `y = float32(float64(y) + weight * float64(x))`, left wing first.  One thread per output
// sample does exactly that sequence (no FMA contraction), so the result is the sequential loop's bit for bit.  The
// filter table, its first differences and the time registers (a sequential float64 accumulation in the reference) are
// the caller's: salsa_amd/resample.py builds them once per (rate pair, length).  Bandwidth is irrelevant here (n_out x
// ~2 * 64 / scale taps from L2-resident tables): it is a loader step, not the hot path.
__global__ __launch_bounds__(256) void resample_kernel(const float *__restrict__ x, float *__restrict__ y, long n_in, long n_out,
                                                       long n_fix, const double *__restrict__ win, const double *__restrict__ delta,
                                                       int nwin, int num_table, double scale, int index_step,
                                                       const double *__restrict__ treg)
{
#pragma clang fp contract(off)
    const long t = blockIdx.x * 256L + threadIdx.x;
    if (t >= n_fix) return;
    const long row = blockIdx.y;
    const float *xr = x + row * n_in;
    float acc = 0.f;                                     // (t >= n_out: librosa's fix_length pads with zeros)
    if (t < n_out) {
        const double tr = treg[t];
        const long n = (long)tr;
        double frac = scale * (tr - (double)n);
        double index_frac = frac * (double)num_table;
        int offset = (int)index_frac;
        double eta = index_frac - (double)offset;
        long m = (nwin - offset) / index_step;
        const long i_max = n + 1 < m ? n + 1 : m;
        for (long i = 0; i < i_max; i++) {
            const long idx = offset + i * index_step;
            const double w = win[idx] + eta * delta[idx];
            acc = (float)((double)acc + w * (double)xr[n - i]);
        }
        frac = scale - frac;
        index_frac = frac * (double)num_table;
        offset = (int)index_frac;
        eta = index_frac - (double)offset;
        m = (nwin - offset) / index_step;
        const long k_max = n_in - n - 1 < m ? n_in - n - 1 : m;
        for (long k = 0; k < k_max; k++) {
            const long idx = offset + k * index_step;
            const double w = win[idx] + eta * delta[idx];
            acc = (float)((double)acc + w * (double)xr[n + k + 1]);
        }
    }
    y[row * n_fix + t] = acc;
}

// ---------------------------------------------------------------------------------------------------------- PCM -> planar float32
// What librosa.load(path, sr=fs, mono=False, dtype=np.float32) (salsa_feature_extraction.py:353) does to a WAV file's samples before
// anything else: soundfile reads the interleaved PCM frames as float32 (libsndfile's normalisation: int16 / 2^15, int32 / 2^31,
// uint8 (x - 128) / 2^7, float32 as is -- all exact in float32 up to the one rounding of a 32-bit integer) and librosa transposes to
// (channels, samples).  The file pipeline uploads the file's data chunk as it is (half the PCIe bytes for 16-bit clips, no host
// arithmetic) and this kernel converts + de-interleaves: one thread per frame, one vector load of the frame's samples, one
// coalesced 4-byte store per channel plane.
template <typename S, int NCH> __device__ __forceinline__ void pcm_frame(const void *pcm, long n, float *v)
{
    struct alignas(sizeof(S) * NCH) vec { S x[NCH]; };
    const vec f = ((const vec *)pcm)[n];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if constexpr (sizeof(S) == 2) v[c] = (float)f.x[c] * (1.0f / 32768.0f);
        else if constexpr (sizeof(S) == 1) v[c] = ((float)f.x[c] - 128.0f) * (1.0f / 128.0f);
        else if constexpr (std::is_same<S, int>::value) v[c] = (float)((double)f.x[c] * (1.0 / 2147483648.0));
        else v[c] = f.x[c];
    }
}
template <typename S> __global__ __launch_bounds__(256) void pcm_to_planar_kernel(const void *__restrict__ pcm, float *__restrict__ out, long n_frames, int nch)
{
    const long n = blockIdx.x * 256L + threadIdx.x;
    if (n >= n_frames) return;
    if (nch == 4) {
        float v[4];
        pcm_frame<S, 4>(pcm, n, v);
#pragma unroll
        for (int c = 0; c < 4; c++) out[c * n_frames + n] = v[c];
    } else {
        for (int c = 0; c < nch; c++) {
            float v[1];
            pcm_frame<S, 1>(pcm, n * nch + c, v);
            out[c * n_frames + n] = v[0];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ augmentation
// The reference's SALSA training augmentation (utilities/transforms.py; recipe in dataset/datamodule.py:45-52, :73-82) as
// ONE gather pass over a feature batch [B][7][T][F]: channel swap (FOA :394-437 / MIC :469-523, applied in the reference's
// order with its float32 arithmetic: the MIC swap re-references the three phase rows by differences), frequency shift with
// reflect padding (:298-320), then the cutout rectangles (:87-121, :149-194, :223-254; last rectangle wins; the spatial rows
// get zeros).  One thread = all 7 channels of one (clip, frame, bin).  par: int32 [B][AUG_NPAR] = m0..m3, shift, up, 0, 0,
// top[8], h[8], left[8], w[8] ; uval: float32 [B][8] in [0,1) ; minmax: float32 [B][2] -> fill = min + (max - min) * u.
constexpr int AUG_NPAR = 40;
__global__ __launch_bounds__(256) void augment_kernel(const float *__restrict__ in, long in_batch, long in_chan,
                                                      float *__restrict__ out, int T, int F,
                                                      int format, int n_zero, const int *__restrict__ par,
                                                      const float *__restrict__ uval, const float *__restrict__ minmax)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T * F) return;
    const int t = i / F, f = i - t * F;
    const int *p = par + b * AUG_NPAR;
    const long plane = (long)T * F;
    const float *src = in + (long)b * in_batch; // the input may be a time-cropped view: its own batch / channel strides
    float *dst = out + (long)b * 7 * plane + i;
    int hit = -1;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int top = p[8 + r], h = p[16 + r], left = p[24 + r], w = p[32 + r];
        if (t >= top && t < top + h && f >= left && f < left + w) hit = r;
    }
    if (hit >= 0) {
#pragma clang fp contract(off) // min + (max - min) * u as three rounded operations, like the host-side restatement (no FMA)
        const float lo = minmax[2 * b], hi = minmax[2 * b + 1];
        const float v = lo + (hi - lo) * uval[b * 8 + hit];
#pragma unroll
        for (int c = 0; c < 7; c++) dst[c * plane] = c < 7 - n_zero ? v : 0.f;
        return;
    }
    const int s = p[4];
    int fs = f;
    if (s > 0) {
        if (p[5]) fs = f - s < 0 ? s - f : f - s;                    // shifted up: pad s bins at the front (reflect at bin 0)
        else fs = f + s > F - 1 ? 2 * (F - 1) - (f + s) : f + s;       // shifted down: pad at the back (reflect at bin F-1)
    }
    float x[7];
#pragma unroll
    for (int c = 0; c < 7; c++) x[c] = src[c * in_chan + (long)t * F + fs];
    if (format == SALSA_FORMAT_FOA) { // W Y Z X | Iy Iz Ix : swap x<->y, negate x, y, z
        if (p[0]) { float a = x[1]; x[1] = x[3]; x[3] = a; a = x[4]; x[4] = x[6]; x[6] = a; }
        if (p[1]) x[6] = -x[6];
        if (p[2]) x[4] = -x[4];
        if (p[3]) x[5] = -x[5];
    } else {                          // M1 M2 M3 M4 | p12 p13 p14
        if (p[0]) { float a = x[1]; x[1] = x[2]; x[2] = a; a = x[4]; x[4] = x[5]; x[5] = a; }
        if (p[1]) {
            const float c0 = x[0], c3 = x[3], c4 = x[4], c5 = x[5], c6 = x[6];
            x[0] = c3; x[3] = c0;
            x[6] = -c6; x[5] = c5 - c6; x[4] = c4 - c6;
        }
        if (p[2]) {
            const float c0 = x[0], c1 = x[1], c2 = x[2], c3 = x[3], c4 = x[4], c5 = x[5], c6 = x[6];
            x[0] = c1; x[1] = c0; x[2] = c3; x[3] = c2;
            x[4] = -c4; x[5] = c6 - c4; x[6] = c5 - c4;
        }
    }
#pragma unroll
    for (int c = 0; c < 7; c++) dst[c * plane] = x[c];
}

__global__ __launch_bounds__(256) void db10_kernel(const float *__restrict__ p, float *__restrict__ o, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = db10(p[i]);
}

} // namespace

// ================================================================================================== plan + C ABI
constexpr int SALSA_MAX_GROUPS = 16;

struct salsa_plan {
    salsa_params p;
    int device;
    int lower, upper, cutoff, nd, F, ident;
    int spec_lo, spec_hi, flex;
    double delta, snr_ratio;
    double *d_window;
    cplx<double> *d_tw;
    const float *sc_mean, *sc_std; // caller-owned device arrays set by salsa_plan_set_scaler (or NULL)
    unsigned long long *stats;     // caller-owned device counters set by salsa_plan_set_stats (or NULL)
    int fused;                     // salsa_plan_set_fused: 0 = three kernels; 1 = STFT -> tracker -> fused STFT + covariance / eigen (stage a)
    int timing;
    int stop_after; // measurement only: 1 = issue the STFT launch alone, 2 = STFT + tracker, 0 = the whole path (salsa_plan_set_timing(plan, -1 | -2))
    int n_kernels;
    hipEvent_t ev0[SALSA_MAX_KERNELS], ev1[SALSA_MAX_KERNELS]; // start/stop of each launch (timing mode only)
    const char *names[SALSA_MAX_KERNELS];
    // clip-group pipeline (salsa_plan_set_pipeline): stream 0 runs the STFT kernels of all groups back to back; group g's
    // tracker and covariance/eigen kernels run on stream 1+g, so the latency-bound tracker of one group hides under the
    // STFT / eigen work of its neighbours.  With SALSA_PIPE_SPLIT_PAIRS the STFT of a group is two launches (channels 0/1,
    // then 2/3) and the tracker -- which only needs channel 0 -- starts after the first.  With SALSA_PIPE_GRAPH the whole
    // fork/join is captured ONCE per (buffers, sizes) into a hipGraph and replayed with a single hipGraphLaunch.
    int n_groups;
    int pipe_flags;
    hipStream_t streams[SALSA_MAX_GROUPS + 1];
    hipEvent_t ev_fork, ev_stft[SALSA_MAX_GROUPS], ev_stft2[SALSA_MAX_GROUPS], ev_join[SALSA_MAX_GROUPS + 1];
    hipStream_t cap_stream;
    hipGraphExec_t gexec;
    struct {
        const float *audio;
        float *out;
        void *ws;
        const float *sc_mean, *sc_std;
        int batch, n_groups, flags;
        int64_t n_samples;
    } gkey;
};

// the fused kernel serves the dataset scripts' main configuration (what cov_eig_kernel's packed instantiation serves) at sizes
// whose ring fits the LDS; everything else keeps the three-kernel path
static bool fused_eligible(const salsa_plan *pl, const KParams &kp)
{
    return SALSA_PK && pl->p.n_fft == 512 && kp.feature == SALSA_FEATURE_SALSA && kp.nch == 4 && kp.n_hop == 3 && kp.tracking &&
           !kp.flex && kp.cond >= SALSA_PK_COND_MIN && kp.cond < 1e6 && !kp.force_f64 && kp.nd >= 1 && kp.T >= 4 * FZ_S && kp.F <= 256 &&
           fused_lds_bytes(kp.nd, kp.F, kp.sc_mean != nullptr) <= 160 * 1024;
}
// frames per segment: enough workgroups for every CU, whole rounds of them when the batch allows, segments long enough that
// the 6-frame halo stays a few percent
static int fused_segment_frames(const KParams &kp)
{
    const long total = (long)kp.B * kp.T;
    long rounds = (total + 256L * 160 - 1) / (256L * 160);
    long nseg = (256 * rounds + kp.B - 1) / kp.B;
    if (nseg < 1) nseg = 1;
    long Lf = (kp.T + nseg - 1) / nseg;
    Lf = (Lf + FZ_S - 1) / FZ_S * FZ_S;
    if (Lf < 4 * FZ_S) Lf = 4 * FZ_S;
    return (int)Lf;
}
static size_t fused_cold_bytes(const KParams &kp) { return (size_t)kp.B * ((kp.T + fused_segment_frames(kp) - 1) / fused_segment_frames(kp)) * FZ_COLD_CAP * sizeof(fz_cold_rec); }
static int launch_fused(salsa_plan *pl, const KParams &kp, const float *d_audio, float *d_out, const unsigned *valid, void *cold, hipStream_t s)
{
    const size_t lds = fused_lds_bytes(kp.nd, kp.F, kp.sc_mean != nullptr);
    static bool attr_set[64][2] = {};
    const bool mic = kp.format == SALSA_FORMAT_MIC;
    if (pl->device >= 0 && pl->device < 64 && !attr_set[pl->device][mic]) {
        if (mic) HIP_TRY(hipFuncSetAttribute((const void *)fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        else HIP_TRY(hipFuncSetAttribute((const void *)fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[pl->device][mic] = true;
    }
    const int L = fused_segment_frames(kp);
    dim3 grid((unsigned)((kp.T + L - 1) / L), (unsigned)kp.B);
    if (mic) hipLaunchKernelGGL(fused_kernel<true>, grid, dim3(FZ_NT), lds, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, valid, L, (fz_cold_rec *)cold, pl->fused == 2 ? 0 : FZ_COLD_CAP);
    else hipLaunchKernelGGL(fused_kernel<false>, grid, dim3(FZ_NT), lds, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, valid, L, (fz_cold_rec *)cold, pl->fused == 2 ? 0 : FZ_COLD_CAP);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

template <int NPAIRS>
static int launch_stft_multi(salsa_plan *pl, const KParams &kp, const float *d_audio, float *d_out, float4 *Xs, hipStream_t s)
{
    const bool lite = kp.feature == SALSA_FEATURE_LITE;
    constexpr int NF = 4;
    dim3 grid((unsigned)((kp.T + 4 * NF - 1) / (4 * NF)), (unsigned)kp.B);
    if (pl->p.n_fft == 512) {
        if (lite) hipLaunchKernelGGL((stft_kernel<512, double, true, NF, NPAIRS>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
        else hipLaunchKernelGGL((stft_kernel<512, double, false, NF, NPAIRS>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
    } else {
        if (lite) hipLaunchKernelGGL((stft_kernel<256, double, true, NF, NPAIRS>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
        else hipLaunchKernelGGL((stft_kernel<256, double, false, NF, NPAIRS>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
    }
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

extern "C" {

int salsa_abi_version(void) { return SALSA_ABI_VERSION; }
const char *salsa_build_flags(void) { return SALSA_BUILD_FLAGS; }
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
    pl->flex = (p.flags & SALSA_FLAG_FLEX) != 0;
    pl->snr_ratio = p.floor_mask_ratio > 0 ? p.floor_mask_ratio : 1.5;
    if (pl->flex) {
        // contrib/salsa_flexible.py SpatialFeaturesAbstract.__init__ (:177-184) + __call__ (:252-263): no fs/2 clamp on
        // fmax_doa, the spectrogram cutoff comes from fmax_spec, spectrogram and spatial features share ONE band
        // [lo, hi) (the cropped axis, or all n_fft/2+1 bins), spatial rows >= upper_bin OF THAT AXIS optionally zeroed.
        if (p.audio_format != SALSA_FORMAT_MIC || p.feature_type == SALSA_FEATURE_IPD) {
            delete pl;
            return fail(SALSA_EFORMAT, "the contrib (flex) surface has the MIC-style SALSA and SALSA-Lite features only%s");
        }
        pl->upper = (int)floor((double)((int64_t)p.fmax_doa * p.n_fft) / (double)p.fs);
        pl->cutoff = (int)floor((double)((int64_t)(p.fmax_spec > 0 ? p.fmax_spec : 9000) * p.n_fft) / (double)p.fs);
        if (pl->upper > pl->cutoff) {
            delete pl;
            return fail(SALSA_EBINS, "Upper bin for spatial feature is higher than cutoff bin for spectrogram!%s");
        }
        const bool crop = !(p.flags & SALSA_FLAG_NO_CLIP_FREQS);
        const int lo = crop ? pl->lower : 0, hi = crop ? (pl->cutoff < nbins ? pl->cutoff : nbins) : nbins;
        const int zero_from = (p.flags & SALSA_FLAG_CLIP_SPATIAL_ALIAS) ? pl->upper : hi - lo; // index into the band
        pl->lower = lo;
        pl->cutoff = hi;
        pl->F = hi - lo;
        if (pl->F <= 0) {
            delete pl;
            return fail(SALSA_EBINS, "empty spectrogram band%s");
        }
        pl->ident = p.n_fft / 2;
        pl->spec_lo = lo;
        pl->spec_hi = hi;
        if (p.feature_type == SALSA_FEATURE_SALSA) {
            pl->nd = zero_from < pl->F ? zero_from : pl->F; // bins above it are never evaluated: cov_eig zero-fills them
            pl->upper = lo + pl->nd;
        } else {
            pl->nd = 0;
            pl->upper = zero_from;                           // the lite kernel zeroes band rows >= kp.upper
        }
    } else if (p.feature_type == SALSA_FEATURE_SALSA) {
        pl->F = freq_dim(p.n_fft, p.is_compress_high_freq);
        pl->ident = p.is_compress_high_freq ? (p.n_fft == 512 ? 192 : 96) : p.n_fft / 2;
        pl->spec_lo = 1;
        pl->spec_hi = pl->ident + 1;
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
    pl->n_groups = 1; // measured on ROCm 7.2: multi-stream issue costs more host time than the overlap returns
    *out_plan = pl;
    return SALSA_OK;
}

int salsa_plan_destroy(salsa_plan *pl)
{
    if (!pl) return SALSA_OK;
    if (pl->d_window) (void)hipFree(pl->d_window);
    if (pl->d_tw) (void)hipFree(pl->d_tw);
    for (int i = 0; i < SALSA_MAX_KERNELS; i++) {
        if (pl->ev0[i]) (void)hipEventDestroy(pl->ev0[i]);
        if (pl->ev1[i]) (void)hipEventDestroy(pl->ev1[i]);
    }
    for (int i = 0; i <= SALSA_MAX_GROUPS; i++) {
        if (pl->streams[i]) (void)hipStreamDestroy(pl->streams[i]);
        if (pl->ev_join[i]) (void)hipEventDestroy(pl->ev_join[i]);
        if (i < SALSA_MAX_GROUPS && pl->ev_stft[i]) (void)hipEventDestroy(pl->ev_stft[i]);
        if (i < SALSA_MAX_GROUPS && pl->ev_stft2[i]) (void)hipEventDestroy(pl->ev_stft2[i]);
    }
    if (pl->gexec) (void)hipGraphExecDestroy(pl->gexec);
    if (pl->cap_stream) (void)hipStreamDestroy(pl->cap_stream);
    if (pl->ev_fork) (void)hipEventDestroy(pl->ev_fork);
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
// one [B][32-bin group][T] uint32 bit mask (the tracker's gate masks; the doubt mask of the coherence test), rounded up to whole
// 64-frame chunks and 64-bin groups
static size_t mask_bytes(int batch, size_t T, int nd) { return align256((size_t)batch * ((T + 63) / 64) * ((nd + 63) / 64) * 64 * 8); }

size_t salsa_workspace_bytes(const salsa_plan *pl, int batch, int64_t n_samples)
{
    if (!pl || batch <= 0 || n_samples <= 0 || pl->p.feature_type != SALSA_FEATURE_SALSA) return 0;
    const size_t T = 1 + n_samples / pl->p.hop_len;
    return align256((size_t)batch * T * 4 * pl->nd * sizeof(float2)) + 2 * mask_bytes(batch, T, pl->nd) + align256(sizeof(unsigned) * (size_t)batch) + 256; // spill, gate masks, doubt mask, doubt flags
}

size_t salsa_eigvec_workspace_bytes(const salsa_plan *pl, int batch, int n_bins, int64_t n_frames)
{
    if (!pl || batch <= 0 || n_bins <= 0 || n_frames <= 0) return 0;
    return align256((size_t)batch * n_frames * 4 * n_bins * sizeof(float2)) + 2 * mask_bytes(batch, (size_t)n_frames, n_bins) + align256(sizeof(unsigned) * (size_t)batch) + 256;
}

static KParams make_kparams(const salsa_plan *pl, int batch, int64_t n_samples)
{
    KParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.B = batch;
    kp.N = (int)n_samples;
    kp.T = (int)(1 + n_samples / pl->p.hop_len);
    kp.hop = pl->p.hop_len;
    kp.lower = pl->lower;
    kp.upper = pl->upper;
    kp.nd = pl->nd;
    kp.cutoff = pl->cutoff;
    kp.F = pl->F;
    kp.OC = 7;
    kp.ident = pl->ident;
    kp.spec_lo = pl->spec_lo;
    kp.spec_hi = pl->spec_hi;
    kp.flex = pl->flex;
    kp.snr_ratio = pl->snr_ratio;
    kp.compress = pl->flex ? 0 : pl->p.is_compress_high_freq;
    kp.layout = pl->p.audio_layout;
    kp.feature = pl->p.feature_type;
    kp.format = pl->p.audio_format;
    kp.tracking = pl->p.is_tracking;
    kp.n_hop = pl->p.n_hopframes;
    kp.pair_sel = -1;
    kp.nch = 4;
    kp.cond = pl->p.cond_num;
    kp.inv_cond = pl->p.cond_num > 0 ? 1.0 / pl->p.cond_num : 0.0;
    kp.delta = pl->delta;
    kp.sc_mean = pl->sc_mean;
    kp.sc_std = pl->sc_std;
    kp.stats = pl->stats;
    kp.force_f64 = (pl->p.flags & SALSA_FLAG_FORCE_F64) != 0;
    return kp;
}

// timing mode: bracket one launch with events on ITS stream
static int mark_begin(salsa_plan *pl, hipStream_t s, const char *name)
{
    if (!pl->timing || pl->n_kernels >= SALSA_MAX_KERNELS) return -1;
    const int i = pl->n_kernels++;
    pl->names[i] = name;
    if (!pl->ev0[i]) (void)hipEventCreate(&pl->ev0[i]);
    if (!pl->ev1[i]) (void)hipEventCreate(&pl->ev1[i]);
    (void)hipEventRecord(pl->ev0[i], s);
    return i;
}
static void mark_end(salsa_plan *pl, hipStream_t s, int i)
{
    if (i >= 0) (void)hipEventRecord(pl->ev1[i], s);
}

static int launch_stft(salsa_plan *pl, const KParams &kp, const float *d_audio, float *d_out, float4 *Xs, hipStream_t s)
{
    const bool lite = kp.feature == SALSA_FEATURE_LITE || kp.feature == SALSA_FEATURE_IPD;
    const bool single = !lite && kp.pair_sel >= 0; // one channel pair per launch: twice the frames per wave, same work per wave
    constexpr int NF_FULL = k1_cfg<false>::NF, NF_LITE = k1_cfg<true>::NF, NF_PAIR = 2 * k1_cfg<false>::NF;
    const int fpb = 4 * (lite ? NF_LITE : single ? NF_PAIR : NF_FULL); // frames per workgroup
    const unsigned nblk = (unsigned)((kp.T + fpb - 1) / fpb);
    dim3 grid(nblk, (unsigned)kp.B);
    // a scaler is attached: the instantiation with the tables in LDS -- which hold [2][4 * 256] floats, so only while F <= 256 (the
    // contrib plan with SALSA_FLAG_NO_CLIP_FREQS and a lite band from bin 0 have F = 257: they take the plain kernel, whose
    // store path reads the tables from global memory at any F)
    // the dataset scripts' layout as a compile-time fact (stft_kernel's STD instantiations; K1_STD 0: the general kernel, for A/B)
    const bool std_layout = K1_STD && pl->p.n_fft == 512 && !lite && !single && kp.feature == SALSA_FEATURE_SALSA && kp.compress &&
                            kp.spec_lo == 1 && kp.spec_hi == 193 && kp.ident == 192 && kp.F == 200 && kp.nch == 4 &&
                            kp.layout == SALSA_LAYOUT_PLANAR && kp.pair_sel < 0;
    const bool lite_std = K1_LITE_STD && pl->p.n_fft == 512 && lite && kp.nch == 4 && kp.layout == SALSA_LAYOUT_PLANAR && !kp.sc_mean &&
                          kp.cutoff <= 256 && kp.pair_sel < 0;
    if (lite_std) {
        hipLaunchKernelGGL((stft_kernel<512, double, true, NF_LITE, 2, false, true>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
    } else if (std_layout) {
        constexpr size_t SCT_BYTES = 2 * 4 * 256 * sizeof(float);
        if (kp.sc_mean) hipLaunchKernelGGL((stft_kernel<512, double, false, NF_FULL, 2, true, true>), grid, dim3(256), SCT_BYTES, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
        else hipLaunchKernelGGL((stft_kernel<512, double, false, NF_FULL, 2, false, true>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
    } else if (pl->p.n_fft == 512 && kp.sc_mean && !single && kp.F <= 256) {
        constexpr size_t SCT_BYTES = 2 * 4 * 256 * sizeof(float);
        if (lite) hipLaunchKernelGGL((stft_kernel<512, double, true, NF_LITE, 2, true>), grid, dim3(256), SCT_BYTES, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
        else hipLaunchKernelGGL((stft_kernel<512, double, false, NF_FULL, 2, true>), grid, dim3(256), SCT_BYTES, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
    } else if (pl->p.n_fft == 512) {
        if (lite) hipLaunchKernelGGL((stft_kernel<512, double, true, NF_LITE>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
        else if (single) hipLaunchKernelGGL((stft_kernel<512, double, false, NF_PAIR>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
        else hipLaunchKernelGGL((stft_kernel<512, double, false, NF_FULL>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
    } else {
        if (lite) hipLaunchKernelGGL((stft_kernel<256, double, true, NF_LITE>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
        else if (single) hipLaunchKernelGGL((stft_kernel<256, double, false, NF_PAIR>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
        else hipLaunchKernelGGL((stft_kernel<256, double, false, NF_FULL>), grid, dim3(256), 0, s, kp, d_audio, pl->d_window, pl->d_tw, d_out, Xs);
    }
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
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != pl->device)
            return fail(SALSA_EINVAL, "the plan's tables live on the device that was current at salsa_plan_create; make it current%s");
    }
    {   // kernels index inside one clip with 32-bit offsets
        const int64_t T64 = 1 + n_samples / pl->p.hop_len;
        if (n_samples * 16 >= INT32_MAX || T64 * 7 * pl->F >= INT32_MAX / 2 || T64 * 2 * (pl->nd > 0 ? pl->nd : 1) >= INT32_MAX / 8)
            return fail(SALSA_EINVAL, "clip too long for 32-bit per-clip indexing (split it)%s");
        if ((T64 + K3_FT - 1) / K3_FT > 65535) return fail(SALSA_EINVAL, "clip too long for one launch (split it)%s");
    }
    hipStream_t s = (hipStream_t)hip_stream;
    KParams kp = make_kparams(pl, batch, n_samples);
    const bool full = pl->p.feature_type == SALSA_FEATURE_SALSA;
    float4 *Xs = nullptr;
    unsigned *valid = nullptr;
    if (full) {
        const size_t need = salsa_workspace_bytes(pl, batch, n_samples);
        if (!d_workspace || workspace_bytes < need) return fail(SALSA_EWORKSPACE, "workspace too small%s (need %ld bytes)", "", (long)need);
        Xs = (float4 *)d_workspace;
        valid = (unsigned *)((unsigned char *)d_workspace + align256((size_t)batch * kp.T * 4 * kp.nd * sizeof(float2)));
        if ((kp.tracking || kp.flex) && kp.cond > 1.0 && kp.nd > 0) {
            kp.doubt32 = (unsigned *)((unsigned char *)valid + mask_bytes(batch, (size_t)kp.T, kp.nd));
            kp.doubt_flag = (unsigned *)((unsigned char *)kp.doubt32 + mask_bytes(batch, (size_t)kp.T, kp.nd)); // [batch]: one per launch group
        }
    }
    pl->n_kernels = 0;
    const long T = kp.T;
    const size_t nchunks = (size_t)((T + TR_CH - 1) / TR_CH);
    // group g = clips [g0, g1): every buffer is clip-major, so a group is just a pointer offset.  s1 runs the STFT launch(es);
    // s2 the tracker and the covariance/eigen kernel (s1 == s2: plain in-order issue on one stream).
    auto run_group = [&](int g0, int g1, hipStream_t s1, hipStream_t s2, hipEvent_t after_first, hipEvent_t after_second,
                         bool split) -> int {
        KParams gp = kp;
        gp.B = g1 - g0;
        const float *a = d_audio + (size_t)g0 * 4 * kp.N;
        float *o = d_out + (size_t)g0 * 7 * T * kp.F;
        float4 *xs = Xs ? Xs + (size_t)g0 * T * 2 * kp.nd : nullptr;
        unsigned *vm = valid ? valid + (size_t)g0 * ((kp.nd + TR_BINS - 1) / TR_BINS) * T : nullptr; // [b][32-bin group][t]
        if (kp.doubt32) gp.doubt32 = kp.doubt32 + (size_t)g0 * ((kp.nd + TR_BINS - 1) / TR_BINS) * T, gp.doubt_flag = kp.doubt_flag + g0;
        const bool two = split && full && gp.nd > 0;
        gp.pair_sel = two ? 0 : -1;
        // timing mode with a repeat count (salsa_plan_set_timing(plan, K > 1)): every kernel is launched K times back to
        // back between ONE event pair -- all of them are idempotent on (audio, spill, masks) -- so the per-launch figure is
        // elapsed / K with no event between the launches (an event pair around a single launch adds ~12 % to it)
        const int reps = pl->timing > 1 ? pl->timing : 1;
        int m = mark_begin(pl, s1, "stft_logspec");
        int rc = SALSA_OK;
        for (int r = 0; r < reps && !rc; r++) rc = launch_stft(pl, gp, a, o, xs, s1);
        mark_end(pl, s1, m);
        if (rc || !full) return rc;
        if (pl->stop_after == 1) return SALSA_PARTIAL; // (measurement mode: the caller is told the outputs are NOT complete)
        if (s1 != s2) {
            HIP_TRY(hipEventRecord(after_first, s1));
            HIP_TRY(hipStreamWaitEvent(s2, after_first, 0));
        }
        if (gp.nd == 0) { // empty DOA band: channels 4-6 are all zero (:373-374)
            HIP_TRY(hipMemset2DAsync(o + 4 * T * kp.F, sizeof(float) * 7 * T * kp.F, 0, sizeof(float) * 3 * T * kp.F, (size_t)gp.B, s2));
            return SALSA_OK;
        }
        if (two) { // channels 2/3 (the tracker below only needs channel 0 and may run beside this launch)
            gp.pair_sel = 1;
            m = mark_begin(pl, s1, "stft_logspec");
            rc = launch_stft(pl, gp, a, o, xs, s1);
            mark_end(pl, s1, m);
            if (rc) return rc;
            gp.pair_sel = -1;
            if (s1 != s2) HIP_TRY(hipEventRecord(after_second, s1));
        }
        if (gp.tracking) {
            m = mark_begin(pl, s2, "noise_floor_tracker");
            for (int r = 0; r < reps; r++)
                hipLaunchKernelGGL(tracker_kernel, dim3(tracker_grid(gp)), dim3(64 * TR_WAVES), 0, s2, gp, xs, vm);
            mark_end(pl, s2, m);
            HIP_TRY(hipGetLastError());
        }
        if (pl->stop_after == 2) return SALSA_PARTIAL;
        if (two && s1 != s2) HIP_TRY(hipStreamWaitEvent(s2, after_second, 0));
        // stage (a): the fused kernel on the masks of the launches above; its float64 records go where the spill was (dead
        // once the tracker has read it: same stream)
        if (pl->fused >= 1 && !two && fused_eligible(pl, gp) && fused_cold_bytes(gp) <= (size_t)gp.B * T * 4 * gp.nd * sizeof(float2)) {
            m = mark_begin(pl, s2, "fused_stft_cov_eig");
            for (int r = 0; r < reps && !rc; r++) rc = launch_fused(pl, gp, a, o, vm, (void *)xs, s2);
            mark_end(pl, s2, m);
            return rc;
        }
        m = mark_begin(pl, s2, "cov_eig");
        const unsigned ntile = (unsigned)((gp.T + K3_FT - 1) / K3_FT);
        dim3 grid(ntile, (unsigned)gp.B, (unsigned)((gp.nd + K3_NT - 1) / K3_NT));
        for (int r = 0; r < reps; r++) launch_cov_eig<true>(gp, grid, s2, xs, vm, o, (double *)nullptr, (unsigned char *)nullptr);
        mark_end(pl, s2, m);
        HIP_TRY(hipGetLastError());
        if (gp.flex && !gp.tracking && gp.nd > 0) {
            m = mark_begin(pl, s2, "flex_allpass");
            hipLaunchKernelGGL(flex_allpass_kernel, dim3((unsigned)((gp.nd + 255) / 256), (unsigned)gp.B), dim3(256), 0, s2, gp, o);
            mark_end(pl, s2, m);
            HIP_TRY(hipGetLastError());
        }
        return SALSA_OK;
    };
    const bool piped = full && !pl->timing && !pl->stop_after && kp.nd > 0 && (pl->n_groups > 1 || (pl->pipe_flags & SALSA_PIPE_SPLIT_PAIRS));
    if (!piped) return run_group(0, batch, s, s, nullptr, nullptr, false);
    const int G = batch < pl->n_groups ? batch : pl->n_groups;
    const bool split = (pl->pipe_flags & SALSA_PIPE_SPLIT_PAIRS) != 0;
    // fork from `origin`, run the groups on the plan's streams, join back into `origin`
    auto issue = [&](hipStream_t origin) -> int {
        HIP_TRY(hipEventRecord(pl->ev_fork, origin));
        HIP_TRY(hipStreamWaitEvent(pl->streams[0], pl->ev_fork, 0));
        for (int g = 0; g < G; g++) {
            HIP_TRY(hipStreamWaitEvent(pl->streams[1 + g], pl->ev_fork, 0)); // orders this call after the caller's earlier work
            const int g0 = (int)((long)batch * g / G), g1 = (int)((long)batch * (g + 1) / G);
            const int rc = run_group(g0, g1, pl->streams[0], pl->streams[1 + g], pl->ev_stft[g], pl->ev_stft2[g], split);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(pl->ev_join[1 + g], pl->streams[1 + g]));
            HIP_TRY(hipStreamWaitEvent(origin, pl->ev_join[1 + g], 0));
        }
        HIP_TRY(hipEventRecord(pl->ev_join[0], pl->streams[0]));
        HIP_TRY(hipStreamWaitEvent(origin, pl->ev_join[0], 0));
        return SALSA_OK;
    };
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s) (void)hipStreamIsCapturing(s, &cap);
    if (!(pl->pipe_flags & SALSA_PIPE_GRAPH) || cap != hipStreamCaptureStatusNone)
        return issue(s); // eager fork/join (inside a caller's capture it becomes part of the caller's graph)
    // one hipGraphLaunch per call: the fork/join above captured once for these buffers and sizes
    const bool hit = pl->gexec && pl->gkey.audio == d_audio && pl->gkey.out == d_out && pl->gkey.ws == d_workspace &&
                     pl->gkey.sc_mean == pl->sc_mean && pl->gkey.sc_std == pl->sc_std && pl->gkey.batch == batch &&
                     pl->gkey.n_samples == n_samples && pl->gkey.n_groups == G && pl->gkey.flags == pl->pipe_flags;
    if (!hit) {
        if (pl->gexec) {
            (void)hipGraphExecDestroy(pl->gexec);
            pl->gexec = nullptr;
        }
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(pl->cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = issue(pl->cap_stream);
        const hipError_t e = hipStreamEndCapture(pl->cap_stream, &graph);
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        HIP_TRY(e);
        const hipError_t ei = hipGraphInstantiate(&pl->gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIP_TRY(ei);
        pl->gkey.audio = d_audio;
        pl->gkey.out = d_out;
        pl->gkey.ws = d_workspace;
        pl->gkey.sc_mean = pl->sc_mean;
        pl->gkey.sc_std = pl->sc_std;
        pl->gkey.batch = batch;
        pl->gkey.n_samples = n_samples;
        pl->gkey.n_groups = G;
        pl->gkey.flags = pl->pipe_flags;
    }
    HIP_TRY(hipGraphLaunch(pl->gexec, s));
    return SALSA_OK;
}

size_t salsa_multichannel_workspace_bytes(const salsa_plan *pl, int n_channels, int batch, int64_t n_samples)
{
    if (!pl || batch <= 0 || n_samples <= 0 || n_channels < 4 || n_channels > SALSA_MAX_MICS || (n_channels & 1)) return 0;
    if (pl->p.feature_type != SALSA_FEATURE_SALSA) return 256;
    const size_t T = 1 + n_samples / pl->p.hop_len;
    return align256((size_t)batch * T * n_channels * pl->nd * sizeof(float2)) +
           align256((size_t)batch * ((pl->nd + TR_BINS - 1) / TR_BINS) * T * sizeof(unsigned)) + 256;
}

int salsa_extract_multichannel(salsa_plan *pl, const float *d_audio, int n_channels, int batch, int64_t n_samples, float *d_out,
                               void *d_workspace, size_t workspace_bytes, void *hip_stream)
{
    if (!pl || !d_audio || !d_out || batch <= 0 || n_samples <= 0) return fail(SALSA_EINVAL, "salsa_extract_multichannel: bad argument%s");
    if (!pl->flex || pl->p.audio_layout != SALSA_LAYOUT_PLANAR)
        return fail(SALSA_EINVAL, "salsa_extract_multichannel is the contrib (SALSA_FLAG_FLEX) surface, planar audio%s");
    if (n_channels < 6 || n_channels > SALSA_MAX_MICS || (n_channels & 1))
        return fail(SALSA_EINVAL, "salsa_extract_multichannel takes an even number of channels from 6 to 16 (pad an odd count with a silent channel; <= 4: salsa_extract_batch)%s");
    if (n_samples <= pl->p.n_fft / 2) return fail(SALSA_EINVAL, "clip shorter than n_fft/2 samples cannot be reflect-padded%s");
    const int64_t T64 = 1 + n_samples / pl->p.hop_len;
    const int OC = 2 * n_channels - 1;
    if (n_samples * 4 * n_channels >= INT32_MAX || T64 * OC * pl->F >= INT32_MAX / 2 || T64 * n_channels * (pl->nd > 0 ? pl->nd : 1) >= INT32_MAX / 16 ||
        T64 > 65535 * 16)
        return fail(SALSA_EINVAL, "clip too long for 32-bit per-clip indexing (split it)%s");
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != pl->device)
            return fail(SALSA_EINVAL, "the plan's tables live on the device that was current at salsa_plan_create; make it current%s");
    }
    hipStream_t s = (hipStream_t)hip_stream;
    KParams kp = make_kparams(pl, batch, n_samples);
    kp.nch = n_channels;
    kp.OC = OC;
    kp.sc_mean = kp.sc_std = nullptr;
    const bool full = pl->p.feature_type == SALSA_FEATURE_SALSA;
    float4 *Xs = nullptr;
    unsigned *valid = nullptr;
    if (full) {
        const size_t need = salsa_multichannel_workspace_bytes(pl, n_channels, batch, n_samples);
        if (!d_workspace || workspace_bytes < need) return fail(SALSA_EWORKSPACE, "workspace too small%s (need %ld bytes)", "", (long)need);
        Xs = (float4 *)d_workspace;
        valid = (unsigned *)((unsigned char *)d_workspace + align256((size_t)batch * kp.T * n_channels * kp.nd * sizeof(float2)));
    }
    int rc = n_channels == 6 ? launch_stft_multi<3>(pl, kp, d_audio, d_out, Xs, s)
           : n_channels == 8 ? launch_stft_multi<4>(pl, kp, d_audio, d_out, Xs, s)
                             : launch_stft_multi<0>(pl, kp, d_audio, d_out, Xs, s); // 10 - 16: channel count at run time
    if (rc || !full) return rc;
    if (kp.tracking && kp.nd > 0) {
        hipLaunchKernelGGL(tracker_kernel, dim3(tracker_grid(kp)), dim3(64 * TR_WAVES), 0, s, kp, Xs, valid);
        HIP_TRY(hipGetLastError());
    }
    dim3 grid((unsigned)kp.T, (unsigned)kp.B);
    if (n_channels == 6) hipLaunchKernelGGL(cov_eig_n_kernel<6>, grid, dim3(64), 0, s, kp, Xs, valid, d_out);
    else if (n_channels == 8) hipLaunchKernelGGL(cov_eig_n_kernel<8>, grid, dim3(64), 0, s, kp, Xs, valid, d_out);
    else hipLaunchKernelGGL(cov_eig_n_kernel<0>, grid, dim3(64), 0, s, kp, Xs, valid, d_out);
    HIP_TRY(hipGetLastError());
    if (!kp.tracking && kp.nd > 0) {
        hipLaunchKernelGGL(flex_allpass_kernel, dim3((unsigned)((kp.nd + 255) / 256), (unsigned)kp.B), dim3(256), 0, s, kp, d_out);
        HIP_TRY(hipGetLastError());
    }
    return SALSA_OK;
}

int salsa_logspec_batch(salsa_plan *pl, const float *d_audio, int batch, int n_channels, int64_t n_samples,
                        float *d_out, void *hip_stream)
{
    if (!pl || !d_audio || !d_out || batch <= 0 || n_samples <= 0) return fail(SALSA_EINVAL, "salsa_logspec_batch: bad argument%s");
    if (n_channels != 4) return fail(SALSA_EINVAL, "salsa_logspec_batch: n_channels must be 4 (pad with silent channels)%s");
    if (n_samples <= pl->p.n_fft / 2) return fail(SALSA_EINVAL, "clip shorter than n_fft/2 samples cannot be reflect-padded%s");
    if (n_samples * 16 >= INT32_MAX || (1 + n_samples / pl->p.hop_len) * 7 * 256 >= INT32_MAX / 2)
        return fail(SALSA_EINVAL, "clip too long for 32-bit per-clip indexing (split it)%s");
    KParams kp = make_kparams(pl, batch, n_samples);
    kp.feature = FEATURE_LOGSPEC_ONLY;
    kp.OC = 4;
    kp.sc_mean = kp.sc_std = nullptr; // MagStftExtractor.extract returns raw dB
    kp.layout = SALSA_LAYOUT_PLANAR;
    kp.F = freq_dim(pl->p.n_fft, pl->p.is_compress_high_freq);
    kp.ident = pl->p.is_compress_high_freq ? (pl->p.n_fft == 512 ? 192 : 96) : pl->p.n_fft / 2;
    kp.compress = pl->p.is_compress_high_freq;
    kp.spec_lo = 1;
    kp.spec_hi = kp.ident + 1;
    return launch_stft(pl, kp, d_audio, d_out, nullptr, (hipStream_t)hip_stream);
}

int salsa_eigvec_batch(salsa_plan *pl, const float *d_X, int batch, int n_bins, int64_t n_frames, int lower_bin,
                       double *d_out, unsigned char *d_gate, void *d_workspace, size_t workspace_bytes,
                       void *hip_stream)
{
    if (!pl || !d_X || !d_out || batch <= 0 || n_bins <= 0 || n_frames <= 0)
        return fail(SALSA_EINVAL, "salsa_eigvec_batch: bad argument%s");
    if ((int64_t)n_frames * 2 * n_bins >= INT32_MAX) return fail(SALSA_EINVAL, "block too large for 32-bit per-clip indexing%s");
    const size_t need = salsa_eigvec_workspace_bytes(pl, batch, n_bins, n_frames);
    if (!d_workspace || workspace_bytes < need) return fail(SALSA_EWORKSPACE, "workspace too small%s (need %ld bytes)", "", (long)need);
    hipStream_t s = (hipStream_t)hip_stream;
    KParams kp = make_kparams(pl, batch, 0);
    kp.T = (int)n_frames;
    kp.nd = n_bins;
    kp.lower = lower_bin;
    kp.upper = lower_bin + n_bins;
    kp.F = n_bins;
    kp.feature = SALSA_FEATURE_SALSA;
    float4 *Xs = (float4 *)d_workspace;
    unsigned *valid = (unsigned *)((unsigned char *)d_workspace + align256((size_t)batch * n_frames * 4 * n_bins * sizeof(float2)));
    if ((kp.tracking || kp.flex) && kp.cond > 1.0) {
        kp.doubt32 = (unsigned *)((unsigned char *)valid + mask_bytes(batch, (size_t)n_frames, n_bins));
        kp.doubt_flag = (unsigned *)((unsigned char *)kp.doubt32 + mask_bytes(batch, (size_t)n_frames, n_bins));
    }
    const long total = (long)batch * n_bins * n_frames * 2;
    hipLaunchKernelGGL(relayout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float4 *)d_X, Xs, batch, n_bins, (int)n_frames);
    HIP_TRY(hipGetLastError());
    if (kp.tracking) {
        hipLaunchKernelGGL(tracker_kernel, dim3(tracker_grid(kp)), dim3(64 * TR_WAVES), 0, s, kp, Xs, valid);
        HIP_TRY(hipGetLastError());
    }
    const unsigned ntile = (unsigned)((kp.T + K3_FT - 1) / K3_FT);
    dim3 grid(ntile, (unsigned)kp.B, (unsigned)((n_bins + K3_NT - 1) / K3_NT));
    launch_cov_eig<false>(kp, grid, s, Xs, valid, (float *)nullptr, d_out, d_gate);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

int salsa_eigvec_feature_batch(salsa_plan *pl, const float *d_X, int batch, int n_bins, int64_t n_frames, int lower_bin,
                               float *d_feat, void *d_workspace, size_t workspace_bytes, void *hip_stream)
{
    if (!pl || !d_X || !d_feat || batch <= 0 || n_bins <= 0 || n_frames <= 0)
        return fail(SALSA_EINVAL, "salsa_eigvec_feature_batch: bad argument%s");
    if ((int64_t)n_frames * 2 * n_bins >= INT32_MAX / 8 || (int64_t)n_frames * 7 * n_bins >= INT32_MAX / 2)
        return fail(SALSA_EINVAL, "block too large for 32-bit per-clip indexing%s");
    if ((n_frames + K3_FT - 1) / K3_FT > 65535) return fail(SALSA_EINVAL, "block too long for one launch%s");
    const size_t need = salsa_eigvec_workspace_bytes(pl, batch, n_bins, n_frames);
    if (!d_workspace || workspace_bytes < need) return fail(SALSA_EWORKSPACE, "workspace too small%s (need %ld bytes)", "", (long)need);
    hipStream_t s = (hipStream_t)hip_stream;
    KParams kp = make_kparams(pl, batch, 0);
    kp.T = (int)n_frames;
    kp.nd = n_bins;
    kp.lower = lower_bin;
    kp.upper = lower_bin + n_bins;
    kp.F = n_bins;
    kp.OC = 7;
    kp.feature = SALSA_FEATURE_SALSA;
    float4 *Xs = (float4 *)d_workspace;
    unsigned *valid = (unsigned *)((unsigned char *)d_workspace + align256((size_t)batch * n_frames * 4 * n_bins * sizeof(float2)));
    if ((kp.tracking || kp.flex) && kp.cond > 1.0) {
        kp.doubt32 = (unsigned *)((unsigned char *)valid + mask_bytes(batch, (size_t)n_frames, n_bins));
        kp.doubt_flag = (unsigned *)((unsigned char *)kp.doubt32 + mask_bytes(batch, (size_t)n_frames, n_bins));
    }
    const long total = (long)batch * n_bins * n_frames * 2;
    hipLaunchKernelGGL(relayout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float4 *)d_X, Xs, batch, n_bins, (int)n_frames);
    HIP_TRY(hipGetLastError());
    if (kp.tracking) {
        hipLaunchKernelGGL(tracker_kernel, dim3(tracker_grid(kp)), dim3(64 * TR_WAVES), 0, s, kp, Xs, valid);
        HIP_TRY(hipGetLastError());
    }
    const unsigned ntile = (unsigned)((kp.T + K3_FT - 1) / K3_FT);
    dim3 grid(ntile, (unsigned)kp.B, (unsigned)((n_bins + K3_NT - 1) / K3_NT));
    launch_cov_eig<true>(kp, grid, s, Xs, valid, d_feat, (double *)nullptr, (unsigned char *)nullptr);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

int salsa_plan_set_fused(salsa_plan *pl, int mode)
{
    if (!pl || mode < 0 || mode > 2) return fail(SALSA_EINVAL, "salsa_plan_set_fused: bad argument%s");
    pl->fused = mode;
    return SALSA_OK;
}

int salsa_plan_set_stats(salsa_plan *pl, unsigned long long *d_counters)
{
    if (!pl) return fail(SALSA_EINVAL, "salsa_plan_set_stats: NULL plan%s");
    pl->stats = d_counters;
    return SALSA_OK;
}

int salsa_plan_set_timing(salsa_plan *pl, int enable)
{
    if (!pl) return fail(SALSA_EINVAL, "salsa_plan_set_timing: NULL plan%s");
    pl->timing = enable > 0 ? enable : 0; // 1: an event pair around every launch; K > 1: K launches per event pair
    pl->stop_after = enable < 0 ? (enable >= -2 ? -enable : 0) : 0; // -1 / -2: plain issue of a PREFIX of the path (no events)
    pl->n_kernels = 0;
    return SALSA_OK;
}

int salsa_plan_read_timing(salsa_plan *pl, float *ms, const char **names, int *n_out)
{
    if (!pl || !ms || !n_out) return fail(SALSA_EINVAL, "salsa_plan_read_timing: NULL argument%s");
    *n_out = 0;
    if (!pl->timing || pl->n_kernels == 0) return SALSA_OK;
    for (int i = 0; i < pl->n_kernels; i++) {
        HIP_TRY(hipEventSynchronize(pl->ev1[i]));
        HIP_TRY(hipEventElapsedTime(&ms[i], pl->ev0[i], pl->ev1[i]));
        if (pl->timing > 1) ms[i] /= (float)pl->timing; // per launch
        if (names) names[i] = pl->names[i];
    }
    *n_out = pl->n_kernels;
    return SALSA_OK;
}

int salsa_scaler_accumulate(const float *d_feat, int batch, int n_channels, int64_t n_frames, int n_freq,
                            int n_scaler_channels, double *d_sums, void *hip_stream)
{
    if (!d_feat || !d_sums || batch <= 0 || n_channels <= 0 || n_frames <= 0 || n_freq <= 0 || n_freq > 256 ||
        n_scaler_channels <= 0 || n_scaler_channels > n_channels || n_frames >= INT32_MAX)
        return fail(SALSA_EINVAL, "salsa_scaler_accumulate: bad argument%s");
    dim3 grid((unsigned)((n_frames + 63) / 64), (unsigned)n_scaler_channels, (unsigned)batch);
    hipLaunchKernelGGL(scaler_accumulate_kernel, grid, dim3(256), 0, (hipStream_t)hip_stream, d_feat, n_channels,
                       (int)n_frames, n_freq, n_scaler_channels, d_sums);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

int salsa_normalize_batch(float *d_feat, int batch, int n_channels, int64_t n_frames, int n_freq, int n_scaler_channels,
                          const float *d_mean, const float *d_std, void *hip_stream)
{
    if (!d_feat || !d_mean || !d_std || batch <= 0 || n_channels <= 0 || n_frames <= 0 || n_freq <= 0 ||
        n_scaler_channels <= 0 || n_scaler_channels > n_channels || n_frames >= INT32_MAX)
        return fail(SALSA_EINVAL, "salsa_normalize_batch: bad argument%s");
    const long rows = (long)batch * n_scaler_channels * n_frames;
    if ((rows + 3) / 4 >= INT32_MAX) return fail(SALSA_EINVAL, "salsa_normalize_batch: too many rows for one launch%s");
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)hip_stream, d_feat,
                       rows, n_channels, (int)n_frames, n_freq, n_scaler_channels, d_mean, d_std);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

static int ensure_group_streams(salsa_plan *pl)
{
    if (pl->streams[0]) return SALSA_OK;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi); // hi = numerically lowest = highest priority
    bool ok = true;
    for (int i = 0; i <= SALSA_MAX_GROUPS && ok; i++)
        ok = hipStreamCreateWithPriority(&pl->streams[i], hipStreamNonBlocking, i == 0 ? lo : hi) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&pl->cap_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < SALSA_MAX_GROUPS && ok; i++)
        ok = hipEventCreateWithFlags(&pl->ev_stft[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&pl->ev_stft2[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i <= SALSA_MAX_GROUPS && ok; i++) ok = hipEventCreateWithFlags(&pl->ev_join[i], hipEventDisableTiming) == hipSuccess;
    return ok ? SALSA_OK : fail(SALSA_EHIP, "stream / event creation failed%s");
}

int salsa_plan_set_scaler(salsa_plan *pl, const float *d_mean, const float *d_std)
{
    if (!pl || ((d_mean == nullptr) != (d_std == nullptr))) return fail(SALSA_EINVAL, "salsa_plan_set_scaler: bad argument%s");
    pl->sc_mean = d_mean;
    pl->sc_std = d_std;
    return SALSA_OK;
}

int salsa_plan_set_pipeline(salsa_plan *pl, int n_groups, int flags)
{
    if (!pl || n_groups < 1 || (flags & ~(SALSA_PIPE_SPLIT_PAIRS | SALSA_PIPE_GRAPH)))
        return fail(SALSA_EINVAL, "salsa_plan_set_pipeline: bad argument%s");
    if (n_groups > 1 || (flags & SALSA_PIPE_SPLIT_PAIRS)) { // the plan-owned streams are only created when a pipeline is requested
        const int rc = ensure_group_streams(pl);
        if (rc) return rc;
    }
    pl->n_groups = n_groups > SALSA_MAX_GROUPS ? SALSA_MAX_GROUPS : n_groups;
    pl->pipe_flags = flags;
    return SALSA_OK;
}

int salsa_plan_set_groups(salsa_plan *pl, int n_groups)
{
    if (!pl) return fail(SALSA_EINVAL, "salsa_plan_set_groups: bad argument%s");
    return salsa_plan_set_pipeline(pl, n_groups, pl->pipe_flags);
}

int salsa_augment_batch(const float *d_in, int64_t in_batch_stride, int64_t in_channel_stride, float *d_out, int batch,
                        int64_t n_frames, int n_freq, int audio_format, int n_zero_channels, const int *d_params,
                        const float *d_uval, const float *d_minmax, void *hip_stream)
{
    if (in_channel_stride < n_frames * n_freq || in_batch_stride < 7 * in_channel_stride)
        return fail(SALSA_EINVAL, "salsa_augment_batch: input strides smaller than the [7][T][F] block%s");
    if (!d_in || !d_out || d_in == d_out || !d_params || !d_uval || !d_minmax || batch <= 0 || batch > 65535 || n_frames <= 0 ||
        n_freq <= 1 || n_frames * n_freq >= INT32_MAX || n_zero_channels < 0 || n_zero_channels > 7)
        return fail(SALSA_EINVAL, "salsa_augment_batch: bad argument%s");
    if (audio_format != SALSA_FORMAT_FOA && audio_format != SALSA_FORMAT_MIC) return fail(SALSA_EFORMAT, "Unknown audio format%s");
    const long n = (long)n_frames * n_freq;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, (hipStream_t)hip_stream,
                       d_in, (long)in_batch_stride, (long)in_channel_stride, d_out, (int)n_frames, n_freq, audio_format,
                       n_zero_channels, d_params, d_uval, d_minmax);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

int salsa_selftest_decibel(const float *d_power, float *d_db, int64_t n, void *hip_stream)
{
    if (!d_power || !d_db || n <= 0 || (n + 255) / 256 >= INT32_MAX) return fail(SALSA_EINVAL, "salsa_selftest_decibel: bad argument%s");
    hipLaunchKernelGGL(db10_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, d_power, d_db, (long)n);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

int salsa_to_freq_major(const float *d_feat, int64_t n_rows, int64_t n_frames, int n_freq, double *d_out, void *hip_stream)
{
    if (!d_feat || !d_out || n_rows <= 0 || n_frames <= 0 || n_freq <= 0 || n_rows > 65535 || (n_frames + 63) / 64 > 65535)
        return fail(SALSA_EINVAL, "salsa_to_freq_major: bad argument%s");
    dim3 grid((unsigned)((n_freq + 63) / 64), (unsigned)((n_frames + 63) / 64), (unsigned)n_rows);
    hipLaunchKernelGGL(to_freq_major_kernel, grid, dim3(256), 0, (hipStream_t)hip_stream, d_feat, d_out, (int)n_frames, n_freq);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

int salsa_resample_batch(const float *d_x, int n_rows, int64_t n_in, float *d_y, int64_t n_out, int64_t n_out_fixed, double sample_ratio,
                         const double *d_interp_win, const double *d_interp_delta, int n_win, int num_table,
                         const double *d_time_register, void *hip_stream)
{
    if (!d_x || !d_y || !d_interp_win || !d_interp_delta || !d_time_register || n_rows <= 0 || n_rows > 65535 || n_in <= 0 ||
        n_out < 0 || n_out_fixed < n_out || n_out_fixed <= 0 || !(sample_ratio > 0.0) || n_win <= 0 || num_table <= 0 ||
        (n_out_fixed + 255) / 256 >= INT32_MAX)
        return fail(SALSA_EINVAL, "salsa_resample_batch: bad argument%s");
    const double scale = sample_ratio < 1.0 ? sample_ratio : 1.0;          // resampy/interpn.py: scale = min(1.0, sample_ratio)
    const int index_step = (int)(scale * (double)num_table);              //                     index_step = int(scale * num_table)
    if (index_step < 1) return fail(SALSA_EINVAL, "salsa_resample_batch: sample_ratio * num_table < 1%s");
    dim3 grid((unsigned)((n_out_fixed + 255) / 256), (unsigned)n_rows);
    hipLaunchKernelGGL(resample_kernel, grid, dim3(256), 0, (hipStream_t)hip_stream, d_x, d_y, (long)n_in, (long)n_out, (long)n_out_fixed,
                       d_interp_win, d_interp_delta, n_win, num_table, scale, index_step, d_time_register);
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

int salsa_pcm_to_planar(const void *d_pcm, int sample_format, int n_channels, int64_t n_frames, float *d_out, void *hip_stream)
{
    if (!d_pcm || !d_out || n_channels <= 0 || n_channels > 64 || n_frames <= 0 || (n_frames + 255) / 256 >= INT32_MAX)
        return fail(SALSA_EINVAL, "salsa_pcm_to_planar: bad argument%s");
    const size_t fb = (size_t)n_channels * (sample_format == SALSA_PCM_S16 ? 2 : sample_format == SALSA_PCM_U8 ? 1 : 4);
    if (n_channels == 4 && ((uintptr_t)d_pcm % fb)) return fail(SALSA_EINVAL, "salsa_pcm_to_planar: d_pcm must be aligned to one frame%s");
    dim3 grid((unsigned)((n_frames + 255) / 256));
    hipStream_t s = (hipStream_t)hip_stream;
    switch (sample_format) {
    case SALSA_PCM_S16: hipLaunchKernelGGL(pcm_to_planar_kernel<short>, grid, dim3(256), 0, s, d_pcm, d_out, (long)n_frames, n_channels); break;
    case SALSA_PCM_S32: hipLaunchKernelGGL(pcm_to_planar_kernel<int>, grid, dim3(256), 0, s, d_pcm, d_out, (long)n_frames, n_channels); break;
    case SALSA_PCM_U8: hipLaunchKernelGGL(pcm_to_planar_kernel<unsigned char>, grid, dim3(256), 0, s, d_pcm, d_out, (long)n_frames, n_channels); break;
    case SALSA_PCM_F32: hipLaunchKernelGGL(pcm_to_planar_kernel<float>, grid, dim3(256), 0, s, d_pcm, d_out, (long)n_frames, n_channels); break;
    default: return fail(SALSA_EINVAL, "salsa_pcm_to_planar: unknown sample format%s");
    }
    HIP_TRY(hipGetLastError());
    return SALSA_OK;
}

} // extern "C"
