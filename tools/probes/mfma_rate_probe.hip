// mfma_rate_probe.hip -- how fast does v_mfma_f32_32x32x16_bf16 stream from N waves per SIMD, with the accumulators in
// architectural VGPRs (what the compiler picks for the 64 -> 64 convolution kernels: 256 VGPRs, 0 AGPRs) or in AGPRs, and with
// 1 / 2 / 4 alternating accumulators per wave?  (round 3: the convolution kernel's tile loop WITHOUT any memory or LDS work
// reaches only 62 - 68 % of the matrix pipe: is that the instruction stream or the hardware?)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe mfma_rate_probe.hip && ./mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITERS 2048

template <int NACC, bool AGPR> __global__ __launch_bounds__(256) void k(float *out, const float *in)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)in[threadIdx.x + i]; b[i] = (__bf16)in[64 + threadIdx.x + i]; }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; j++) acc[j] = f32x16{};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) {
            if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
    for (int j = 0; j < NACC; j++) for (int i = 0; i < 16; i++) s += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool AGPR> double run(int wgs_per_cu, float *out, const float *in)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    k<NACC, AGPR><<<grid, 256>>>(out, in);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<NACC, AGPR><<<grid, 256>>>(out, in);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * grid * 4 /*waves*/ * (double)ITERS * NACC * 32768.0;
    return flops / (ms * 1e-3) / 1e12;
}

int main()
{
    float *out, *in; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 4096);
    float h[1024]; for (int i = 0; i < 1024; i++) h[i] = (float)((i * 37) % 17) * 0.11f - 0.8f; // non-trivial data: realistic clock
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    printf("TFLOP/s of a bare MFMA stream (2500 = the dense bf16 peak at 2.4 GHz), one 256-thread workgroup = 1 wave per SIMD\n");
    printf("%-44s %10s %10s\n", "", "1 WG/CU", "2 WG/CU");
    printf("%-44s %10.0f %10.0f\n", "1 accumulator,  VGPR", run<1, false>(1, out, in), run<1, false>(2, out, in));
    printf("%-44s %10.0f %10.0f\n", "2 accumulators, VGPR", run<2, false>(1, out, in), run<2, false>(2, out, in));
    printf("%-44s %10.0f %10.0f\n", "4 accumulators, VGPR", run<4, false>(1, out, in), run<4, false>(2, out, in));
    printf("%-44s %10.0f %10.0f\n", "1 accumulator,  AGPR", run<1, true>(1, out, in), run<1, true>(2, out, in));
    printf("%-44s %10.0f %10.0f\n", "2 accumulators, AGPR", run<2, true>(1, out, in), run<2, true>(2, out, in));
    printf("%-44s %10.0f %10.0f\n", "4 accumulators, AGPR", run<4, true>(1, out, in), run<4, true>(2, out, in));
    return 0;
}
