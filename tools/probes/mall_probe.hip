// mall_probe.hip -- does the 256-MiB Infinity Cache keep WRITTEN data for a reader that follows?  (round 3: decides whether a
// time-chunked schedule of the three feature kernels could keep the STFT spill on-die.)
// For a working set of X MiB: kernel W streams X MiB of float4 stores, kernel R streams the same X MiB of loads (sum into a
// sink), repeated; also W alone, R alone, and "ring" = W then R on the SAME X MiB region while a second, large stream (Y MiB
// read + Y MiB write, standing in for audio in / features out) passes through between them.
//   hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip && ./mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void wr(float4 *p, size_t n, float v)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void rd(const float4 *p, size_t n, float *sink)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = p[i];
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ __launch_bounds__(256) void cp(const float4 *a, float4 *b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main()
{
    const size_t MiB = 1 << 20;
    float4 *buf, *big_a, *big_b;
    float *sink;
    CK(hipMalloc(&buf, 4096 * MiB));
    CK(hipMalloc(&big_a, 1024 * MiB));
    CK(hipMalloc(&big_b, 1024 * MiB));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, 4096 * MiB));
    CK(hipMemset(big_a, 0, 1024 * MiB));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    const int reps = 20;
    printf("%8s %12s %12s %12s %14s %14s\n", "X MiB", "W GB/s", "R GB/s", "W+R GB/s", "W,R ms/iter", "W,cp,R GB/s(spill)");
    for (size_t X : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048, 4096}) {
        const size_t n = X * MiB / 16;
        float ms_w, ms_r, ms_wr, ms_ring;
        for (int i = 0; i < 3; i++) { wr<<<grid, 256>>>(buf, n, 1.f); rd<<<grid, 256>>>(buf, n, sink); }
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) wr<<<grid, 256>>>(buf, n, (float)i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_w, e0, e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) rd<<<grid, 256>>>(buf, n, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_r, e0, e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) { wr<<<grid, 256>>>(buf, n, (float)i); rd<<<grid, 256>>>(buf, n, sink); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_wr, e0, e1));
        // W, then a pass-through stream of the same size (copy X MiB -> X MiB elsewhere), then R: does the spill survive?
        const size_t nc = (X > 1024 ? 1024 : X) * MiB / 16;
        float ms_cp;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) cp<<<grid, 256>>>(big_a, big_b, nc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_cp, e0, e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) { wr<<<grid, 256>>>(buf, n, (float)i); cp<<<grid, 256>>>(big_a, big_b, nc); rd<<<grid, 256>>>(buf, n, sink); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ring, e0, e1));
        const double gb = X * (double)MiB / 1e9;
        printf("%8zu %12.0f %12.0f %12.0f %14.4f %14.0f   (cp alone %.0f GB/s; W,cp,R %.4f ms vs W+R %.4f + cp %.4f)\n", X, gb * reps / (ms_w * 1e-3), gb * reps / (ms_r * 1e-3),
               2 * gb * reps / (ms_wr * 1e-3), ms_wr / reps, 2 * gb * reps / ((ms_ring - ms_cp) * 1e-3), 2 * (nc * 16.0 / 1e9) * reps / (ms_cp * 1e-3),
               ms_ring / reps, ms_wr / reps, ms_cp / reps);
    }
    return 0;
}
