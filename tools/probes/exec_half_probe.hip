// exec_half_probe.hip -- round 5: does a wave64 whose EXEC mask has only lanes 0-31 set issue its float64 VALU instructions in
// fewer cycles than with all 64 lanes?  (The tracker's consumer wave runs 32 bins mirrored into 64 lanes; if half an EXEC mask
// halved the issue time its ~78-cycle step would shrink.)  One wave, nothing else on the SIMD; clock64 ticks per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o exec_half_probe exec_half_probe.hip && ./exec_half_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int MODE> __global__ void probe(double *out, long long *cyc, double seed, double f, int nact)
{
    double a = seed + threadIdx.x * 1e-9, b = seed * 1.5, c = seed * 0.7, d = seed * 0.3;
    const double lim = 1e-6;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < nact) { // EXEC = the low nact lanes for everything inside
        t0 = clock64();
        for (int i = 0; i < N / 64; i++) {
            if (MODE == 0) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(f));) }
            if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 64; k++) {
                    const bool above = b > a;
                    const double fsel = above ? c : d;
                    a = fmax(a * fsel, lim);
                    asm volatile("" : "+v"(a));
                }
            }
            if (MODE == 2) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4" : "+v"(*(float *)&a), "+v"(*(float *)&b), "+v"(*(float *)&c), "+v"(*(float *)&d) : "v"(*(float *)&f));) }
        }
        t1 = clock64();
    }
    out[threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const char *names[3] = {"4 independent v_mul_f64 chains (per instr)", "tracker chain cmp->2 cndmask->mul->max (per step)", "4 independent v_mul_f32 chains (per instr)"};
    for (int m = 0; m < 3; m++)
        for (int nact : {64, 32, 16}) {
            long long h = 0;
            for (int rep = 0; rep < 2; rep++) {
                if (m == 0) probe<0><<<1, 64>>>(out, cyc, 1.0, 1.0000001, nact);
                if (m == 1) probe<1><<<1, 64>>>(out, cyc, 1.0, 1.0000001, nact);
                if (m == 2) probe<2><<<1, 64>>>(out, cyc, 1.0, 1.0000001, nact);
                hipDeviceSynchronize();
            }
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-52s EXEC = low %2d lanes: %7.2f clock64 ticks\n", names[m], nact, (double)h / N);
        }
    return 0;
}
