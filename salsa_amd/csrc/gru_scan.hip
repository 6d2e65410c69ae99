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
