// gru_scan.hip -- fused GRU time scan for the SELD CRNN decoder on gfx950 (C ABI in include/salsa_gru.h).
//
// Why: the decoder's 2-layer BiGRU is 0.4 % of the model's FLOPs, but through torch it is ~2000 tiny kernels per
// training step (MIOpen's RNN: Op2dTensorLite x1936 ...), 40 % of the step.  The recurrence is latency-bound, so it
// becomes ONE launch per layer: a workgroup owns one (sample, direction) sequence, thread j owns hidden unit j, h lives
// in LDS, and W_hh (768 KB fp32 per direction) is streamed from L2 every step with coalesced rows -- every workgroup of a
// direction reads the same weights, so they stay L2-resident.  float32 throughout (same arithmetic as nn.GRU fp32).
#include "build_guard.h" // probe switches need -DSALSA_PROBE_BUILD; SALSA_BUILD_FLAGS (generated: tools/gen_build_guard.py)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/salsa_gru.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// the register-resident kernels' gate math is the serial tail of every step: hardware reciprocal (1 ulp) instead of the IEEE
// division sequence, tanh through the same exponential (absolute error ~1e-7, far below the float16 operands' rounding)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// whh_t: [D][H(k)][3H] (weight_hh transposed by the caller) so that for a fixed k consecutive threads read
// consecutive addresses.
template <int H>
__global__ __launch_bounds__(H) void gru_fwd_kernel(const float *__restrict__ gi, const float *__restrict__ whh_t,
                                                    const float *__restrict__ bhh, float *__restrict__ hs,
                                                    float *__restrict__ saved, int T, int B, int D)
{
    __shared__ float h[H];
    const int j = threadIdx.x;
    const int b = blockIdx.x, d = blockIdx.y;
    const float *w = whh_t + (long)d * H * 3 * H;
    const float br = bhh[d * 3 * H + j], bz = bhh[d * 3 * H + H + j], bn = bhh[d * 3 * H + 2 * H + j];
    float hj = 0.f;
    h[j] = 0.f;
    __syncthreads();
    for (int s = 0; s < T; s++) {
        const int t = d == 0 ? s : T - 1 - s;
        const long base = ((long)t * B + b) * D + d;
        float ar = br, az = bz, an = bn;
#pragma unroll 8
        for (int k = 0; k < H; k++) {
            const float hk = h[k];
            const float *wk = w + (long)k * 3 * H;
            ar = fmaf(wk[j], hk, ar);
            az = fmaf(wk[H + j], hk, az);
            an = fmaf(wk[2 * H + j], hk, an);
        }
        const float *g = gi + base * 3 * H;
        const float r = sigmoidf_(g[j] + ar);
        const float z = sigmoidf_(g[H + j] + az);
        const float n = tanhf(g[2 * H + j] + r * an);
        hj = (1.f - z) * n + z * hj;
        hs[base * H + j] = hj;
        if (saved) {
            float *sv = saved + base * 4 * H;
            sv[j] = r;
            sv[H + j] = z;
            sv[2 * H + j] = n;
            sv[3 * H + j] = an;
        }
        __syncthreads(); // everyone has finished reading h of the previous step
        h[j] = hj;
        __syncthreads();
    }
}

// Backward scan (BPTT).  whh: [D][3H][H] PyTorch layout: row-major rows of length H, so for a fixed row consecutive
// threads (k) read consecutive addresses when forming dh_prev[k] = sum_rows whh[row][k] * dgh[row].
template <int H>
__global__ __launch_bounds__(H) void gru_bwd_kernel(const float *__restrict__ dhs, const float *__restrict__ whh,
                                                    const float *__restrict__ hs, const float *__restrict__ saved,
                                                    float *__restrict__ dgi, float *__restrict__ dgh, int T, int B, int D)
{
    __shared__ float g[3 * H];
    const int j = threadIdx.x;
    const int b = blockIdx.x, d = blockIdx.y;
    const float *w = whh + (long)d * 3 * H * H;
    float carry = 0.f;
    for (int s = T - 1; s >= 0; s--) { // reverse of the forward scan order
        const int t = d == 0 ? s : T - 1 - s;
        const long base = ((long)t * B + b) * D + d;
        float hprev = 0.f;
        if (s > 0) {
            const int tp = d == 0 ? t - 1 : t + 1;
            hprev = hs[(((long)tp * B + b) * D + d) * H + j];
        }
        const float *sv = saved + base * 4 * H;
        const float r = sv[j], z = sv[H + j], n = sv[2 * H + j], hn = sv[3 * H + j];
        const float dh = dhs[base * H + j] + carry;
        const float dn = dh * (1.f - z);
        const float dz = dh * (hprev - n);
        const float dn_pre = dn * (1.f - n * n);
        const float dr_pre = dn_pre * hn * r * (1.f - r);
        const float dz_pre = dz * z * (1.f - z);
        float *o = dgi + base * 3 * H;
        o[j] = dr_pre;
        o[H + j] = dz_pre;
        o[2 * H + j] = dn_pre;
        float *q = dgh + base * 3 * H;
        const float dhn = dn_pre * r;
        q[j] = dr_pre;
        q[H + j] = dz_pre;
        q[2 * H + j] = dhn;
        __syncthreads(); // previous step's reads of g are done
        g[j] = dr_pre;
        g[H + j] = dz_pre;
        g[2 * H + j] = dhn;
        __syncthreads();
        float acc = dh * z;
#pragma unroll 8
        for (int row = 0; row < 3 * H; row++) acc = fmaf(w[(long)row * H + j], g[row], acc);
        carry = acc;
    }
}

// Inference variant: W_hh RESIDENT IN REGISTERS.  With few sequences (8 clips x 2 directions = 16 workgroups) and many steps
// (300 label-rate frames) the streaming kernel above spends ~10 us per step pulling 768 KB of weights through L2 into one
// CU: 39 % of an inference sub-batch.  A 1024-thread workgroup's register file holds 512 KB, enough for the direction's
// 3H x H matrix as float16 (384 KB): thread (q, part) keeps, for hidden unit q, the three gate rows' columns
// part*64 .. +63 (192 weights = 96 registers), multiplies them with its 64-element slice of h (LDS, float16 copy: 8 reads
// of 16 B) with v_dot2_f32_f16 -- two multiply-adds per instruction, float32 accumulation -- and a 4-lane butterfly
// completes the three dot products, so lane part 0 of every group has r, z, n of ITS unit.  Gates, the carried h of each
// unit (z * h), the outputs and the accumulation stay float32; only the matrix-vector operands are rounded, to float16
// (11 significant bits, finer than the bf16 that autocast would give an nn.GRU) -- used without gradients (inference)
// and for training under bf16 autocast, with gru_bwd_regw_kernel differentiating the SAME rounded weights; float32
// training keeps the float32 kernels above.  One barrier per step (h is double-buffered); nothing but the weights lives
// in registers across steps (biases are re-read with the step's input projections: kept, they were spilled).
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
#ifndef GRU_FILL
#define GRU_FILL 16 // floats per register-fill chunk of the forward scan
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int H>
__global__ __launch_bounds__(1024) void gru_fwd_regw_kernel(const float *__restrict__ gi, const float *__restrict__ whh,
                                                            const float *__restrict__ bhh, float *__restrict__ hs,
                                                            float *__restrict__ saved, int T, int B, int D)
{
    static_assert(H == 256, "1024 threads = 256 units x 4 column slices");
    constexpr int KS = H / 4; // columns per thread
    __shared__ __attribute__((aligned(16))) _Float16 h[2][H]; // the matrix-vector operand; each unit's own h stays float32 (hq)
    const int q = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int b = blockIdx.x, d = blockIdx.y;
    half2_t wp[3][KS / 2];
    // register fill: 192 float32 weights -> 96 packed float16 pairs, in chunks of 16 floats that are converted before the next
    // chunk's loads go out (all 48 16-byte loads at once need 192 registers next to the 96 being filled: the compiler spilled 84
    // of them to scratch, ~45 us of a 103-us launch at T = 40)
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const float *row = whh + ((long)d * 3 * H + g * H + q) * H + part * KS;
#pragma unroll
        for (int c = 0; c < KS / GRU_FILL; c++) {
            float4 v[GRU_FILL / 4];
#pragma unroll
            for (int u = 0; u < GRU_FILL / 4; u++) v[u] = *(const float4 *)(row + GRU_FILL * c + 4 * u);
#pragma unroll
            for (int u = 0; u < GRU_FILL / 4; u++) {
                half2_t lo = half2_t{(_Float16)v[u].x, (_Float16)v[u].y}, hi = half2_t{(_Float16)v[u].z, (_Float16)v[u].w};
                asm volatile("" : "+v"(lo), "+v"(hi)); // the conversion happens HERE (left alone the compiler keeps the float32
                wp[g][(GRU_FILL / 2) * c + 2 * u] = lo; // values alive and converts at first use: every load then waits for itself
                wp[g][(GRU_FILL / 2) * c + 2 * u + 1] = hi; // and goes to scratch)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (threadIdx.x < H) h[0][threadIdx.x] = (_Float16)0.f;
    float hq = 0.f;
    __syncthreads();
    for (int s = 0; s < T; s++) {
        const int t = d == 0 ? s : T - 1 - s;
        const long base = ((long)t * B + b) * D + d;
        // lane `part` of a unit's quad owns one gate: 0 = r, 1 = z, 2 = n and the unit's state; each loads ITS input projection
        // and bias (in flight during the dot products).  The biases are re-read every step: 96 of the 128 registers hold weights,
        // values kept across steps were being spilled; re-read (L1 hits) they ride with the projections.
        float gx = 0.f, bx = 0.f;
        if (part < 3) {
            gx = gi[base * 3 * H + part * H + q];
            const float *bp = bhh + d * 3 * H + part * H + q;
            asm volatile("" : "+v"(bp)); // (not loop-invariant as far as the compiler can tell)
            bx = bp[0];
        }
        const uint4 *hp = (const uint4 *)&h[s & 1][part * KS];
        float ar = 0.f, az = 0.f, an = 0.f;
        // LDS reads one ahead of the dot products that consume them, issued and waited for by hand (the compiler cannot count
        // LDS returns across the asm statements below and would wait for the read it has just issued)
        const unsigned haddr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)hp;
        u32x4 hb[2];
        asm volatile("ds_read_b128 %0, %1" : "=v"(hb[0]) : "v"(haddr));
#pragma unroll
        for (int i = 0; i < KS / 8; i++) {
            if (i + 1 < KS / 8) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hb[(i + 1) & 1]) : "v"(haddr), "n"(16 * (i + 1)));
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(hb[i & 1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hb[i & 1]));
            }
            const unsigned hv[4] = {hb[i & 1].x, hb[i & 1].y, hb[i & 1].z, hb[i & 1].w}; // 8 consecutive elements of h
            // v_dot2c_f32_f16 acc, w, h written out: the weights are plain INPUT operands.  (Round 2 had the builtin behind an
            // empty asm with the weight as an in/out operand to keep each pair packed in its register -- which made the compiler
            // copy every weight to a scratch register first: v_mov + s_nop + v_dot2 per pair, 2.0 of the step's 2.3 us.)  The three
            // gates' accumulators alternate, so no dot product waits for the one before it.
#pragma unroll
            for (int j = 0; j < 4; j++) {
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(ar) : "v"(wp[0][4 * i + j]), "v"(hv[j]));
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(az) : "v"(wp[1][4 * i + j]), "v"(hv[j]));
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(an) : "v"(wp[2][4 * i + j]), "v"(hv[j]));
            }
        }
        // the four column slices of a unit are the four lanes of a DPP quad: two quad permutes instead of two LDS shuffles
        auto quad_sum = [](float x) {
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true)); // quad_perm [1,0,3,2]
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true)); // quad_perm [2,3,0,1]
            return x;
        };
        ar = quad_sum(ar);
        az = quad_sum(az);
        an = quad_sum(an);
        // gates, one per lane of the quad: r and z through ONE sigmoid (lanes 0 and 1), broadcast to the quad; lane 2 finishes n
        // and the state.  (Round 2: lane 0 did all three -- every wave issued two sigmoids and a tanh for 16 active lanes.)
        const float a = part == 0 ? ar : part == 1 ? az : an;
        const float hn = a + bx;                                  // lane 2: W_hn h + b_hn
        const float sg = sigmoid_fast(gx + a + bx);               // lane 0: r, lane 1: z
        const float r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0x00, 0xF, 0xF, true)); // quad_perm [0,0,0,0]
        const float z = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0x55, 0xF, 0xF, true)); // quad_perm [1,1,1,1]
        if (saved && part < 2) saved[base * 4 * H + part * H + q] = sg; // for the backward scan: r, z, n and W_hn h + b_hn
        if (part == 2) {
            const float n = tanh_fast(gx + r * hn);
            hq = (1.f - z) * n + z * hq;
            hs[base * H + q] = hq;
            h[(s + 1) & 1][q] = (_Float16)hq;
            if (saved) {
                float *sv = saved + base * 4 * H;
                sv[2 * H + q] = n;
                sv[3 * H + q] = hn;
            }
        }
        __syncthreads();
    }
}

// Backward scan with W_hh resident in registers (float16), the counterpart of gru_fwd_regw_kernel.  dh_prev[q] = sum over the 3H
// rows of W_hh[row][q] * dgh[row].  Thread (cg, rs) keeps the 4 columns 4 cg .. 4 cg + 3 of the 48 rows rs * 48 .. + 47 (192 weights,
// two rows of one column per register).  Per step: lane rs = c < 4 of group cg forms the gate derivatives of unit 4 cg + c and
// publishes (dr_pre, dz_pre, dhn) in LDS; after a barrier every thread multiplies its weights with ITS 48-row slice of that
// vector (12 reads of 16 B) and a 16-lane DPP reduction completes the four columns.
// (Round 2's layout -- one column x 192 rows per thread -- had every thread read 768 B of LDS per step: 16 waves x 48
// ds_read_b128 = 6 144 LDS cycles, 5 of the step's 6.2 us; the LDS moves a full 1 KiB per wave-read whether or not lanes share
// addresses.  Four columns per thread need a quarter of the reads for the same 192 multiply-adds.)
template <int H>
__global__ __launch_bounds__(1024) void gru_bwd_regw_kernel(const float *__restrict__ dhs, const float *__restrict__ whh,
                                                            const float *__restrict__ hs, const float *__restrict__ saved,
                                                            float *__restrict__ dgi, float *__restrict__ dgh, int T, int B, int D)
{
    static_assert(H == 256, "1024 threads = 64 groups of 4 columns x 16 row slices");
    constexpr int RS = 3 * H / 16; // rows per thread: 48
    __shared__ __attribute__((aligned(16))) float g[2][3 * H];
    const int cg = threadIdx.x >> 4, rs = threadIdx.x & 15;
    const int q = 4 * cg + (rs & 3); // the unit whose gate derivatives this lane forms (lanes rs < 4 only)
    const bool gate_lane = rs < 4;
    const int b = blockIdx.x, d = blockIdx.y;
    half2_t wp[4][RS / 2];
#pragma unroll
    for (int i = 0; i < RS / 2; i++) {
        const float *w0 = whh + ((long)d * 3 * H + rs * RS + 2 * i) * H + 4 * cg;
        const float4 r0 = *(const float4 *)w0, r1 = *(const float4 *)(w0 + H);
        half2_t c0 = half2_t{(_Float16)r0.x, (_Float16)r1.x}, c1 = half2_t{(_Float16)r0.y, (_Float16)r1.y};
        half2_t c2 = half2_t{(_Float16)r0.z, (_Float16)r1.z}, c3 = half2_t{(_Float16)r0.w, (_Float16)r1.w};
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)); // convert HERE (see the forward kernel's register fill)
        wp[0][i] = c0;
        wp[1][i] = c1;
        wp[2][i] = c2;
        wp[3][i] = c3;
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0); // four row pairs (eight 16-byte loads) in flight at a time
    }
    float carry = 0.f;
    for (int s = T - 1; s >= 0; s--) { // reverse of the forward scan order
        const int t = d == 0 ? s : T - 1 - s;
        const long base = ((long)t * B + b) * D + d;
        float dhz = 0.f;
        float *gs = g[s & 1];
        if (gate_lane) {
            float hprev = 0.f;
#ifdef GRU_PROBE_BWD_NO_LOAD
            if (false) {
#else
            if (s > 0) {
#endif
                const int tp = d == 0 ? t - 1 : t + 1;
                hprev = hs[(((long)tp * B + b) * D + d) * H + q];
            }
            const float *sv = saved + base * 4 * H;
#ifdef GRU_PROBE_BWD_NO_LOAD
            const float r = 0.3f, z = 0.4f, n = 0.1f, hn = 0.2f;
            const float dh = 0.01f + carry;
            (void)sv;
#else
            const float r = sv[q], z = sv[H + q], n = sv[2 * H + q], hn = sv[3 * H + q];
            const float dh = dhs[base * H + q] + carry;
#endif
            const float dn = dh * (1.f - z);
            const float dz = dh * (hprev - n);
            const float dn_pre = dn * (1.f - n * n);
            const float dr_pre = dn_pre * hn * r * (1.f - r);
            const float dz_pre = dz * z * (1.f - z);
            const float dhn = dn_pre * r;
#ifndef GRU_PROBE_BWD_NO_STORE
            float *o = dgi + base * 3 * H;
            o[q] = dr_pre;
            o[H + q] = dz_pre;
            o[2 * H + q] = dn_pre;
            float *qq = dgh + base * 3 * H;
            qq[q] = dr_pre;
            qq[H + q] = dz_pre;
            qq[2 * H + q] = dhn;
#endif
            gs[q] = dr_pre;
            gs[H + q] = dz_pre;
            gs[2 * H + q] = dhn;
            dhz = dh * z;
        }
        __syncthreads(); // g of this step is complete (the other buffer may still be read by slower waves: double-buffered)
        // v_fma_mix_f32 acc, w.lo|hi (float16), g (float32), acc written out with the weights as plain INPUT operands (round 2: the
        // compiler's own fma_mix behind an empty asm with in/out weights = a v_mov per pair) and one accumulator per column; the
        // LDS reads run one ahead of the multiply-adds that consume them, issued and counted by hand (see the forward kernel)
        const unsigned gaddr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)&gs[rs * RS];
        f32x4 gb[2];
        asm volatile("ds_read_b128 %0, %1" : "=v"(gb[0]) : "v"(gaddr));
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#ifdef GRU_PROBE_BWD_NO_MATVEC
#pragma unroll
        for (int i = RS / 4 - 1; i < RS / 4; i++) {
#else
#pragma unroll
        for (int i = 0; i < RS / 4; i++) {
#endif
            if (i + 1 < RS / 4) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(gb[(i + 1) & 1]) : "v"(gaddr), "n"(16 * (i + 1)));
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(gb[i & 1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gb[i & 1]));
            }
            const f32x4 gv = gb[i & 1];
#define GRU_MIX_LO(acc, w, x) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(w), "v"(x))
#define GRU_MIX_HI(acc, w, x) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(w), "v"(x))
            GRU_MIX_LO(a0, wp[0][2 * i], gv.x); GRU_MIX_LO(a1, wp[1][2 * i], gv.x); GRU_MIX_LO(a2, wp[2][2 * i], gv.x); GRU_MIX_LO(a3, wp[3][2 * i], gv.x);
            GRU_MIX_HI(a0, wp[0][2 * i], gv.y); GRU_MIX_HI(a1, wp[1][2 * i], gv.y); GRU_MIX_HI(a2, wp[2][2 * i], gv.y); GRU_MIX_HI(a3, wp[3][2 * i], gv.y);
            GRU_MIX_LO(a0, wp[0][2 * i + 1], gv.z); GRU_MIX_LO(a1, wp[1][2 * i + 1], gv.z); GRU_MIX_LO(a2, wp[2][2 * i + 1], gv.z); GRU_MIX_LO(a3, wp[3][2 * i + 1], gv.z);
            GRU_MIX_HI(a0, wp[0][2 * i + 1], gv.w); GRU_MIX_HI(a1, wp[1][2 * i + 1], gv.w); GRU_MIX_HI(a2, wp[2][2 * i + 1], gv.w); GRU_MIX_HI(a3, wp[3][2 * i + 1], gv.w);
#undef GRU_MIX_LO
#undef GRU_MIX_HI
        }
        // sum over the 16 row slices (= the 16 lanes of a DPP row): quad butterflies, then the two mirrors; every lane ends with
        // the four totals and lane rs = c keeps column c's
        auto row_sum = [](float x) {
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true)); // row_half_mirror
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true)); // row_mirror
            return x;
        };
        a0 = row_sum(a0);
        a1 = row_sum(a1);
        a2 = row_sum(a2);
        a3 = row_sum(a3);
        const int c = rs & 3;
        carry = (c == 0 ? a0 : c == 1 ? a1 : c == 2 ? a2 : a3) + dhz; // only read by the gate lanes
    }
}

} // namespace

extern "C" {

int salsa_gru_scan_fwd(const float *gi, const float *whh_t, const float *bhh, float *hs, float *saved, int T, int B, int D,
                       int H, void *hip_stream)
{
    if (!gi || !whh_t || !bhh || !hs || T <= 0 || B <= 0 || D <= 0 || (H != 256 && H != 128 && H != 64)) return -1;
    hipStream_t s = (hipStream_t)hip_stream;
    dim3 grid((unsigned)B, (unsigned)D);
    if (H == 256) hipLaunchKernelGGL((gru_fwd_kernel<256>), grid, dim3(256), 0, s, gi, whh_t, bhh, hs, saved, T, B, D);
    else if (H == 128) hipLaunchKernelGGL((gru_fwd_kernel<128>), grid, dim3(128), 0, s, gi, whh_t, bhh, hs, saved, T, B, D);
    else hipLaunchKernelGGL((gru_fwd_kernel<64>), grid, dim3(64), 0, s, gi, whh_t, bhh, hs, saved, T, B, D);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_gru_scan_fwd_regw(const float *gi, const float *whh, const float *bhh, float *hs, float *saved, int T, int B, int D,
                            int H, void *hip_stream)
{
    if (!gi || !whh || !bhh || !hs || T <= 0 || B <= 0 || D <= 0 || H != 256) return -1;
    hipLaunchKernelGGL((gru_fwd_regw_kernel<256>), dim3((unsigned)B, (unsigned)D), dim3(1024), 0, (hipStream_t)hip_stream, gi, whh,
                       bhh, hs, saved, T, B, D);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_gru_scan_bwd_regw(const float *dhs, const float *whh, const float *hs, const float *saved, float *dgi, float *dgh,
                            int T, int B, int D, int H, void *hip_stream)
{
    if (!dhs || !whh || !hs || !saved || !dgi || !dgh || T <= 0 || B <= 0 || D <= 0 || H != 256) return -1;
    hipLaunchKernelGGL((gru_bwd_regw_kernel<256>), dim3((unsigned)B, (unsigned)D), dim3(1024), 0, (hipStream_t)hip_stream, dhs, whh,
                       hs, saved, dgi, dgh, T, B, D);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_gru_scan_bwd(const float *dhs, const float *whh, const float *hs, const float *saved, float *dgi, float *dgh,
                       int T, int B, int D, int H, void *hip_stream)
{
    if (!dhs || !whh || !hs || !saved || !dgi || !dgh || T <= 0 || B <= 0 || D <= 0 || (H != 256 && H != 128 && H != 64)) return -1;
    hipStream_t s = (hipStream_t)hip_stream;
    dim3 grid((unsigned)B, (unsigned)D);
    if (H == 256) hipLaunchKernelGGL((gru_bwd_kernel<256>), grid, dim3(256), 0, s, dhs, whh, hs, saved, dgi, dgh, T, B, D);
    else if (H == 128) hipLaunchKernelGGL((gru_bwd_kernel<128>), grid, dim3(128), 0, s, dhs, whh, hs, saved, dgi, dgh, T, B, D);
    else hipLaunchKernelGGL((gru_bwd_kernel<64>), grid, dim3(64), 0, s, dhs, whh, hs, saved, dgi, dgh, T, B, D);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

} // extern "C"
