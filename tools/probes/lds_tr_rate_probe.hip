// Probe: sustained rate of ds_read_b64_tr_b16 / ds_read_b64 / ds_read_b128 per CU on gfx950, for the address patterns of the
// weight-gradient kernels (conv_wide.hip / conv_mfma.hip) and a linear one; 8 or 4 waves per CU, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -o lds_tr_rate_probe lds_tr_rate_probe.hip && ./lds_tr_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP, int PAT>
__global__ __launch_bounds__(512) void probe(unsigned long long *out, int iters)
{
    extern __shared__ unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32768 / 4; i += blockDim.x) ((unsigned *)lds)[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)lds;
    unsigned addr;
    const int i16 = lane & 15, cb = (lane >> 4) & 1, kh = lane >> 5;
    if (PAT == 0) addr = base + wave * 2048 + lane * (OP == 2 ? 16 : 8);                                       // linear
    else if (PAT == 1) addr = base + wave * 2048 + (8 * kh + (i16 >> 2)) * 64 + cb * 32 + (i16 & 3) * 8;       // wide wrw fragment (unswizzled 64-byte slots)
    else addr = base + wave * 2048 + (lane & 31) * 64 + ((lane >> 5) * 16);                                    // ds_read_b128 of the forward kernel without its swizzle
    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 b0 = {0, 0, 0, 0}, b1 = b0, b2 = b0, b3 = b0;
    const long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:256\n\tds_read_b64_tr_b16 %2, %4 offset:512\n\tds_read_b64_tr_b16 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                                  : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr));
        if (OP == 1) asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:256\n\tds_read_b64 %2, %4 offset:512\n\tds_read_b64 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                                  : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr));
        if (OP == 2) asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:512\n\tds_read_b128 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                                  : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(addr));
    }
    const long t1 = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.x == 0) out[0] = (unsigned long long)(t1 - t0);
    if (a0 + a1 + a2 + a3 + b0.x + b1.x + b2.x + b3.x == 0x123456789ull) out[1] = 1;
}
template <int OP, int PAT> void run(const char *name, int waves)
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const int iters = 4000;
    hipLaunchKernelGGL((probe<OP, PAT>), dim3(256), dim3(64 * waves), 32768 + 16384, 0, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<OP, PAT>), dim3(256), dim3(64 * waves), 32768 + 16384, 0, d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double reads = (double)iters * 4 * waves, bytes = reads * 64 * (OP == 2 ? 16 : 8);
    printf("%-28s %d waves/CU: %.2f us, %.1f cycles (s_memtime-style counter %llu) per wave-read, %.1f B/clk/CU at 2.4 GHz\n", name, waves, ms * 1e3,
           ms * 1e-3 * 2.4e9 / reads, h[0], bytes / (ms * 1e-3 * 2.4e9));
    hipFree(d);
}
int main()
{
    for (int waves = 4; waves <= 8; waves += 4) {
        if (waves == 4) {
            run<0, 0>("b64_tr linear", 4); run<0, 1>("b64_tr wrw pattern", 4); run<1, 0>("b64 linear", 4); run<1, 1>("b64 wrw pattern", 4);
            run<2, 0>("b128 linear", 4); run<2, 2>("b128 rows of 64 B", 4);
        } else {
            run<0, 0>("b64_tr linear", 8); run<0, 1>("b64_tr wrw pattern", 8); run<1, 0>("b64 linear", 8); run<1, 1>("b64 wrw pattern", 8);
            run<2, 0>("b128 linear", 8); run<2, 2>("b128 rows of 64 B", 8);
        }
    }
    return 0;
}
