// Probe: does a latency-bound dependent-chain kernel keep its speed while a throughput kernel runs, (a) on plain streams,
// (b) on streams with disjoint CU masks (hipExtStreamCreateWithCUMask)?   hipcc --offload-arch=gfx950 -O3 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void chain(double *out, int steps, unsigned long long *ts)
{
    const unsigned long long t0 = wall_clock64();
    __builtin_amdgcn_s_setprio(3);
    double f = 1.0 + threadIdx.x * 1e-9, m = 1.00001;
    int cd = 3;
    for (int i = 0; i < steps; i++) {
        const double pa = f * 1.02, pb = f * 0.98;
        const bool above = m > f;
        cd = above ? cd - 1 : 3;
        f = fmax(above ? pa : pb, 1e-6);
        m = m * 1.0000001 + (cd < 0 ? 1e-9 : 0.0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = f;
    if (threadIdx.x == 0) { ts[2 * blockIdx.x] = t0; ts[2 * blockIdx.x + 1] = wall_clock64(); }
}

__global__ void heavy(const float4 *in, float4 *out, long n, int reps, unsigned long long *ts)
{
    if (threadIdx.x == 0) atomicMin(&ts[0], (unsigned long long)wall_clock64());
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 v = in[i];
        double a = v.x, b = v.y, c = v.z, d = v.w;
        for (int r = 0; r < reps; r++) { a = a * b + c; b = b * c + d; c = c * d + a; d = d * a + b; }
        out[i] = make_float4((float)a, (float)b, (float)c, (float)d);
    }
    if (threadIdx.x == 0) atomicMax(&ts[1], (unsigned long long)wall_clock64());
}

int main()
{
    const long n = 64l << 20; // 1 GiB in + 1 GiB out
    float4 *in, *out; double *co; unsigned long long *ts, *hts;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&co, 96 * 64 * 8)); CK(hipMalloc(&ts, 96 * 16)); CK(hipMalloc(&hts, 16));
    CK(hipMemset(in, 0, n * 16));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("CUs %d\n", ncu);
    hipStream_t s1, s2, m1, m2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    // masks: bit i = CU i (interleaved over XCDs/SEs by the driver's numbering); chain gets every 8th group of 4... keep simple: first 32 CUs
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> ma(words, 0), mb(words, 0);
    for (int i = 0; i < ncu; i++) { if (i % 8 == 0) ma[i / 32] |= 1u << (i % 32); else mb[i / 32] |= 1u << (i % 32); }
    hipError_t e1 = hipExtStreamCreateWithCUMask(&m1, words, ma.data()), e2 = hipExtStreamCreateWithCUMask(&m2, words, mb.data());
    printf("cu-mask streams: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    hipEvent_t a, b, c, d; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c)); CK(hipEventCreate(&d));
    const int steps = 4801 * 4;
    auto run = [&](hipStream_t sc, hipStream_t sh, bool with_heavy, const char *tag) -> int {
        for (int it = 0; it < 3; it++) {
            CK(hipDeviceSynchronize());
            { unsigned long long init[2] = {~0ull, 0}; CK(hipMemcpy(hts, init, 16, hipMemcpyHostToDevice)); }
            if (with_heavy) { CK(hipEventRecord(c, sh)); hipLaunchKernelGGL(heavy, dim3(ncu * 8), dim3(256), 0, sh, in, out, n, 24, hts); CK(hipEventRecord(d, sh)); }
            CK(hipEventRecord(a, sc)); hipLaunchKernelGGL(chain, dim3(96), dim3(64), 0, sc, co, steps, ts); CK(hipEventRecord(b, sc));
            CK(hipDeviceSynchronize());
            float tc = 0, th = 0; CK(hipEventElapsedTime(&tc, a, b)); if (with_heavy) CK(hipEventElapsedTime(&th, c, d));
            unsigned long long h[192], hh[2]; CK(hipMemcpy(h, ts, 96 * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hh, hts, 16, hipMemcpyDeviceToHost));
            unsigned long long cs = ~0ull, ce = 0, longest = 0; for (int k = 0; k < 96; k++) { if (h[2*k] < cs) cs = h[2*k]; if (h[2*k+1] > ce) ce = h[2*k+1]; if (h[2*k+1]-h[2*k] > longest) longest = h[2*k+1]-h[2*k]; }
            const double tick = 1e-5; // wall_clock64: 100 MHz -> ms per tick
            if (it == 2) printf("%-28s chain ev %.3f ms | in-kernel: span %.3f, longest block %.3f, start vs heavy start %+.3f, heavy span %.3f (ev %.3f)\n", tag, tc,
                                (ce - cs) * tick, longest * tick, with_heavy ? ((double)cs - (double)hh[0]) * tick : 0.0, with_heavy ? (hh[1] - hh[0]) * tick : 0.0, th);
        }
        return 0;
    };
    if (run(s1, s2, false, "chain alone")) return 1;
    if (run(s1, s2, true, "plain streams, concurrent")) return 1;
    if (e1 == hipSuccess && e2 == hipSuccess) {
        if (run(m1, m2, false, "chain alone (masked 1/8)")) return 1;
        if (run(m1, m2, true, "disjoint CU masks")) return 1;
        if (run(m1, s2, true, "chain masked, heavy unmasked")) return 1;
    }
    return 0;
}
