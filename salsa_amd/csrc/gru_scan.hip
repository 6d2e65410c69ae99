// gru_scan.hip -- fused GRU time scan for the SELD CRNN decoder on gfx950 (C ABI in include/salsa_gru.h).
//
// Why: the decoder's 2-layer BiGRU is 0.4 % of the model's FLOPs, but through torch it is ~2000 tiny kernels per
// training step (MIOpen's RNN: Op2dTensorLite x1936 ...), 40 % of the step.  The recurrence is latency-bound, so it
// becomes ONE launch per layer: a workgroup owns one (sample, direction) sequence, thread j owns hidden unit j, h lives
// in LDS, and W_hh (768 KB fp32 per direction) is streamed from L2 every step with coalesced rows -- every workgroup of a
// direction reads the same weights, so they stay L2-resident.  float32 throughout (same arithmetic as nn.GRU fp32).
#include <hip/hip_runtime.h>
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
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const float *row = whh + ((long)d * 3 * H + g * H + q) * H + part * KS;
#pragma unroll
        for (int i = 0; i < KS / 2; i++) wp[g][i] = half2_t{(_Float16)row[2 * i], (_Float16)row[2 * i + 1]};
    }
    if (threadIdx.x < H) h[0][threadIdx.x] = (_Float16)0.f;
    float hq = 0.f;
    __syncthreads();
    for (int s = 0; s < T; s++) {
        const int t = d == 0 ? s : T - 1 - s;
        const long base = ((long)t * B + b) * D + d;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, br = 0.f, bz = 0.f, bn = 0.f;
        if (part == 0) { // this step's input projections, in flight during the dot products
            const float *g = gi + base * 3 * H;
            g0 = g[q];
            g1 = g[H + q];
            g2 = g[2 * H + q];
            // the biases too: 96 of the 128 registers hold weights, so values kept across steps were being spilled and
            // reloaded from scratch one by one in the middle of the gate math; re-read (L1 hits) they ride with the loads above
            const float *bp = bhh + d * 3 * H + q;
            asm volatile("" : "+v"(bp)); // (not loop-invariant as far as the compiler can tell)
            br = bp[0];
            bz = bp[H];
            bn = bp[2 * H];
        }
        typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2))); // the operand type of the dot-product builtin
        const uint4 *hp = (const uint4 *)&h[s & 1][part * KS];
        float ar = 0.f, az = 0.f, an = 0.f;
#pragma unroll
        for (int i = 0; i < KS / 8; i++) {
            const uint4 raw = hp[i]; // 8 consecutive elements of h
            const unsigned hv[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int g = 0; g < 3; g++) {
                float acc = g == 0 ? ar : g == 1 ? az : an;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    half2_t w = wp[g][4 * i + j];
                    asm volatile("" : "+v"(w)); // keeps each weight pair packed in its one register
                    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(fp16x2_t, w), __builtin_bit_cast(fp16x2_t, hv[j]), acc, false); // v_dot2_f32_f16
                }
                if (g == 0) ar = acc;
                else if (g == 1) az = acc;
                else an = acc;
            }
            __builtin_amdgcn_sched_barrier(0); // 96 of the 128 registers hold weights: one h vector in flight at a time
        }
        ar += __shfl_xor(ar, 1); az += __shfl_xor(az, 1); an += __shfl_xor(an, 1);
        ar += __shfl_xor(ar, 2); az += __shfl_xor(az, 2); an += __shfl_xor(an, 2);
        if (part == 0) {
            const float r = sigmoid_fast(g0 + ar + br);
            const float z = sigmoid_fast(g1 + az + bz);
            const float n = tanh_fast(g2 + r * (an + bn));
            hq = (1.f - z) * n + z * hq;
            hs[base * H + q] = hq;
            h[(s + 1) & 1][q] = (_Float16)hq;
            if (saved) { // for the backward scan: r, z, n and W_hn h + b_hn
                float *sv = saved + base * 4 * H;
                sv[q] = r;
                sv[H + q] = z;
                sv[2 * H + q] = n;
                sv[3 * H + q] = an + bn;
            }
        }
        __syncthreads();
    }
}

// Backward scan with W_hh resident in registers (float16), the counterpart of gru_fwd_regw_kernel: thread (q, part) keeps
// COLUMN q of W_hh for the rows part*192 .. +191 (192 weights), i.e. its share of dh_prev[q] = sum_rows W_hh[row][q] * dgh[row].
// Per step: the part-0 lane of unit q forms the gate derivatives of its unit and publishes (dr_pre, dz_pre, dhn) in LDS; after
// a barrier every thread multiplies its 192 weights with its slice of that vector and a 4-lane butterfly completes dh_prev.
template <int H>
__global__ __launch_bounds__(1024) void gru_bwd_regw_kernel(const float *__restrict__ dhs, const float *__restrict__ whh,
                                                            const float *__restrict__ hs, const float *__restrict__ saved,
                                                            float *__restrict__ dgi, float *__restrict__ dgh, int T, int B, int D)
{
    static_assert(H == 256, "1024 threads = 256 units x 4 row slices");
    constexpr int RS = 3 * H / 4; // rows per thread
    __shared__ __attribute__((aligned(16))) float g[2][3 * H];
    const int q = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int b = blockIdx.x, d = blockIdx.y;
    half2_t wp[RS / 2];
#pragma unroll
    for (int i = 0; i < RS / 2; i++) {
        const float *w0 = whh + ((long)d * 3 * H + part * RS + 2 * i) * H + q;
        wp[i] = half2_t{(_Float16)w0[0], (_Float16)w0[H]};
    }
    float carry = 0.f;
    for (int s = T - 1; s >= 0; s--) { // reverse of the forward scan order
        const int t = d == 0 ? s : T - 1 - s;
        const long base = ((long)t * B + b) * D + d;
        float dhz = 0.f;
        float *gs = g[s & 1];
        if (part == 0) {
            float hprev = 0.f;
            if (s > 0) {
                const int tp = d == 0 ? t - 1 : t + 1;
                hprev = hs[(((long)tp * B + b) * D + d) * H + q];
            }
            const float *sv = saved + base * 4 * H;
            const float r = sv[q], z = sv[H + q], n = sv[2 * H + q], hn = sv[3 * H + q];
            const float dh = dhs[base * H + q] + carry;
            const float dn = dh * (1.f - z);
            const float dz = dh * (hprev - n);
            const float dn_pre = dn * (1.f - n * n);
            const float dr_pre = dn_pre * hn * r * (1.f - r);
            const float dz_pre = dz * z * (1.f - z);
            const float dhn = dn_pre * r;
            float *o = dgi + base * 3 * H;
            o[q] = dr_pre;
            o[H + q] = dz_pre;
            o[2 * H + q] = dn_pre;
            float *qq = dgh + base * 3 * H;
            qq[q] = dr_pre;
            qq[H + q] = dz_pre;
            qq[2 * H + q] = dhn;
            gs[q] = dr_pre;
            gs[H + q] = dz_pre;
            gs[2 * H + q] = dhn;
            dhz = dh * z;
        }
        __syncthreads(); // g of this step is complete (the other buffer may still be read by slower waves: double-buffered)
        const float4 *gp = (const float4 *)&gs[part * RS];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < RS / 4; i++) {
            const float4 gv = gp[i];
            half2_t w01 = wp[2 * i], w23 = wp[2 * i + 1];
            asm volatile("" : "+v"(w01), "+v"(w23)); // keep the loop-invariant half -> float conversions out of registers
            acc = fmaf((float)w01.x, gv.x, acc);
            acc = fmaf((float)w01.y, gv.y, acc);
            acc = fmaf((float)w23.x, gv.z, acc);
            acc = fmaf((float)w23.y, gv.w, acc);
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        carry = acc + dhz; // only read by the part-0 lane
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
