// Probe: HBM read rate of LDS-direct loads (global_load_lds_dwordx4) against ordinary global_load_dwordx4 into registers, at the
// occupancies of the convolution kernels (256-thread workgroups, 1 / 2 / 4 per CU, 8 or 16 wave-loads in flight per wave).
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_rate_probe lds_dma_rate_probe.hip && ./lds_dma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void probe(const uint4 *__restrict__ src, uint4 *__restrict__ sink, long n_vec, int lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const long per_wg = (long)DEPTH * 256;                       // 16-byte vectors per round and workgroup
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (long base = (long)blockIdx.x * per_wg; base + per_wg <= n_vec; base += (long)gridDim.x * per_wg) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < DEPTH; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + base + j * 256 + tid),
                                                 (__attribute__((address_space(3))) void *)(lds + ((j * 4 + wave) * 1024) % lds_bytes), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; j++) v[j] = src[base + j * 256 + tid];
#pragma unroll
            for (int j = 0; j < DEPTH; j++) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        }
    }
    if (MODE == 0) { __syncthreads(); acc = *(uint4 *)(lds + tid * 16); }
    if (acc.x == 0x12345678u) sink[tid] = acc;
}
template <int MODE, int DEPTH> void run(const char *name, const uint4 *src, uint4 *sink, long n_vec, int wgs_per_cu)
{
    const int lds = wgs_per_cu == 1 ? 64 * 1024 : wgs_per_cu == 2 ? 64 * 1024 : 32 * 1024;
    hipFuncSetAttribute((const void *)probe<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(256 * wgs_per_cu), dim3(256), lds, 0, src, sink, n_vec, lds);
        hipEventRecord(e1);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s depth %2d  %d workgroups/CU: %7.1f us  %6.2f TB/s\n", name, DEPTH, wgs_per_cu, ms * 1e3, n_vec * 16.0 / (ms * 1e-3) / 1e12);
}
int main()
{
    const long bytes = 2L << 30, n_vec = bytes / 16; // 2 GiB: beyond the Infinity Cache
    uint4 *src, *sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 4096);
    hipMemset(src, 1, bytes);
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 8>("LDS-direct  global_load_lds x4", src, sink, n_vec, w);
        run<0, 16>("LDS-direct  global_load_lds x4", src, sink, n_vec, w);
        run<1, 8>("registers   global_load_dwordx4", src, sink, n_vec, w);
        run<1, 16>("registers   global_load_dwordx4", src, sink, n_vec, w);
    }
    return 0;
}
