// dot2_rate_probe.hip -- issue rate of the instructions a register-resident matrix-vector product can be built from (gfx950):
// v_dot2_f32_f16, v_fma_f32, v_pk_fma_f32, v_pk_fma_f16, v_dot2c (VOP2 form).  One workgroup of 1024 threads per CU (4 waves per
// SIMD), 8 independent accumulators per thread, N instructions each; prints cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/dot2_rate_probe tools/probes/dot2_rate_probe.hip && tools/probes/dot2_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters, float seed)
{
    float a[8];
    float2_t p[8];
    half2_t hp[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * i; p[i] = float2_t{seed, seed * i}; hp[i] = half2_t{(_Float16)seed, (_Float16)(seed * i)}; }
    half2_t w = half2_t{(_Float16)(seed + threadIdx.x), (_Float16)1.0f}, h = half2_t{(_Float16)0.5f, (_Float16)seed};
    float2_t pw = float2_t{seed, 1.f}, ph = float2_t{0.5f, seed};
    float fw = seed + 1.f, fh = 0.999f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[u]) : "v"(w), "v"(h));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[u]) : "v"(fw), "v"(fh));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[u]) : "v"(pw), "v"(ph));
            if (MODE == 3) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(hp[u]) : "v"(w), "v"(h));
            if (MODE == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[u]) : "v"(w), "v"(h));
            if (MODE == 5) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[u]) : "v"(w), "v"(fh));
            if (MODE == 6) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[u]) : "v"(w), "v"(fh));
            if (MODE == 7) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[u]) : "v"(w), "v"(h));
            if (MODE == 8) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[u]) : "v"(w));
            if (MODE == 9) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[u]) : "v"(w), "v"(h));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)hp[i].x + (float)hp[i].y;
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name)
{
    float *out;
    long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, cyc, 100, 0.001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, cyc, iters, 0.001f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c[256];
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    // per SIMD: 4 waves x 8 x iters instructions
    const double n = 4.0 * 8 * iters;
    printf("%-16s %8.3f ms   %.2f ns per wave-instruction per SIMD   (s_memtime ticks per instruction: %.2f at 100 MHz)\n", name, ms, ms * 1e6 / n,
           (double)c[0] / n);
}

int main()
{
    run<0>("v_dot2_f32_f16");
    run<1>("v_fma_f32");
    run<2>("v_pk_fma_f32");
    run<3>("v_pk_fma_f16");
    run<4>("v_dot2c_f32_f16");
    run<5>("v_fma_mix_f32 lo");
    run<6>("v_fma_mix_f32 hi");
    run<7>("v_dot2_f32_bf16");
    run<8>("v_cvt_f32_f16");
    run<9>("v_dot2c_f32_bf16");
    return 0;
}
