// nn_ops.hip -- gfx950 kernels for the memory-bound layers of the SELD CRNN (C ABI: include/salsa_nn.h).
// Everything here is a streaming pass: 16-byte accesses, channels-last so that a thread's vector is contiguous.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include "../../include/salsa_nn.h"

namespace {

struct f32x4 {
    float v[4];
};
struct bf16x8 {
    unsigned short v[8];
};

__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) // round to nearest even (inputs are finite sums of bf16 values)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40); // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <typename V, int L> struct vec_io;
template <> struct vec_io<f32x4, 4> {
    static __device__ __forceinline__ void load(const void *p, float *f)
    {
        const float4 t = *(const float4 *)p;
        f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
    }
    static __device__ __forceinline__ void store(void *p, const float *f) { *(float4 *)p = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct vec_io<bf16x8, 8> {
    static __device__ __forceinline__ void load(const void *p, float *f)
    {
        const uint4 t = *(const uint4 *)p;
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f[2 * i] = bf2f((unsigned short)(w[i] & 0xffffu));
            f[2 * i + 1] = bf2f((unsigned short)(w[i] >> 16));
        }
    }
    static __device__ __forceinline__ void store(void *p, const float *f)
    {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = (unsigned)f2bf(f[2 * i]) | ((unsigned)f2bf(f[2 * i + 1]) << 16);
        *(uint4 *)p = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// one thread = one output pixel x L channels (16 bytes); x, y channels-last
template <typename V, int L>
__global__ __launch_bounds__(256) void avgpool2x2_fwd_kernel(const char *__restrict__ x, char *__restrict__ y, long n_vec,
                                                             int H, int W, int C)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_vec) return;
    const int cv = C / L, Ho = H / 2, Wo = W / 2;
    const int c = (int)(i % cv);
    long r = i / cv;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long n = r / Ho;
    const long esz = 16 / L; // bytes per element
    const char *p = x + (((n * H + 2 * ho) * W + 2 * wo) * C + (long)c * L) * esz;
    float a[L], b[L], cc[L], d[L], o[L];
    vec_io<V, L>::load(p, a);
    vec_io<V, L>::load(p + (long)C * esz, b);
    vec_io<V, L>::load(p + (long)W * C * esz, cc);
    vec_io<V, L>::load(p + ((long)W * C + C) * esz, d);
#pragma unroll
    for (int k = 0; k < L; k++) o[k] = (((a[k] + b[k]) + cc[k]) + d[k]) / 4.0f; // the reference implementation's order
    vec_io<V, L>::store(y + i * 16, o);
}

// one thread = one INPUT pixel x L channels: grad_x = grad_y[h/2][w/2] / 4, zero in a dropped odd row / column
template <typename V, int L>
__global__ __launch_bounds__(256) void avgpool2x2_bwd_kernel(const char *__restrict__ gy, char *__restrict__ gx, long n_vec,
                                                             int H, int W, int C)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_vec) return;
    const int cv = C / L, Ho = H / 2, Wo = W / 2;
    const int c = (int)(i % cv);
    long r = i / cv;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const long n = r / H;
    float g[L];
    if (h < 2 * Ho && w < 2 * Wo) {
        vec_io<V, L>::load(gy + ((((n * Ho + h / 2) * Wo + w / 2) * cv + c) * 16), g);
#pragma unroll
        for (int k = 0; k < L; k++) g[k] = g[k] / 4.0f;
    } else {
#pragma unroll
        for (int k = 0; k < L; k++) g[k] = 0.f;
    }
    vec_io<V, L>::store(gx + i * 16, g);
}

} // namespace

extern "C" {

int salsa_nn_avgpool2x2_fwd(const void *x, void *y, int dtype, int64_t N, int H, int W, int C, void *hip_stream)
{
    const int L = dtype == 1 ? 8 : 4;
    if (!x || !y || N <= 0 || H < 2 || W < 2 || C <= 0 || C % L || (dtype != 0 && dtype != 1)) return -1;
    const long n_vec = (long)N * (H / 2) * (W / 2) * (C / L);
    const dim3 grid((unsigned)((n_vec + 255) / 256));
    if (dtype == 1)
        hipLaunchKernelGGL((avgpool2x2_fwd_kernel<bf16x8, 8>), grid, dim3(256), 0, (hipStream_t)hip_stream, (const char *)x, (char *)y, n_vec, H, W, C);
    else
        hipLaunchKernelGGL((avgpool2x2_fwd_kernel<f32x4, 4>), grid, dim3(256), 0, (hipStream_t)hip_stream, (const char *)x, (char *)y, n_vec, H, W, C);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_nn_avgpool2x2_bwd(const void *grad_y, void *grad_x, int dtype, int64_t N, int H, int W, int C, void *hip_stream)
{
    const int L = dtype == 1 ? 8 : 4;
    if (!grad_y || !grad_x || N <= 0 || H < 2 || W < 2 || C <= 0 || C % L || (dtype != 0 && dtype != 1)) return -1;
    const long n_vec = (long)N * H * W * (C / L);
    const dim3 grid((unsigned)((n_vec + 255) / 256));
    if (dtype == 1)
        hipLaunchKernelGGL((avgpool2x2_bwd_kernel<bf16x8, 8>), grid, dim3(256), 0, (hipStream_t)hip_stream, (const char *)grad_y, (char *)grad_x, n_vec, H, W, C);
    else
        hipLaunchKernelGGL((avgpool2x2_bwd_kernel<f32x4, 4>), grid, dim3(256), 0, (hipStream_t)hip_stream, (const char *)grad_y, (char *)grad_x, n_vec, H, W, C);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

} // extern "C"
