// Probe: lane->element mapping of ds_read_b64_tr_b16 on gfx950 for a few per-lane address patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned short *out, int mode)
{
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)lds;
    if (mode == 0) addr = base + l * 8;                     // each lane: its own 4 consecutive elements
    else if (mode == 1) addr = base + (l & 15) * 128 + (l >> 4) * 8; // lane = row (pitch 64 elems), 16-lane group = 4-column block
    else addr = base + (l & 15) * 2 + (l >> 4) * 128;       // lane = column, group = row block
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    for (int j = 0; j < 4; j++) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main()
{
    unsigned short *d, h[256];
    hipMalloc(&d, 512);
    for (int mode = 0; mode < 3; mode++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; l += (l < 18 ? 1 : 15)) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
