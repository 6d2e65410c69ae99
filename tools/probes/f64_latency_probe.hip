// f64_latency_probe.hip -- dependent-issue latency and single-wave issue interval of the float64 VALU operations the noise-floor
// tracker's recurrence is made of (round 3: is its ~90-cycle step bound by instruction issue or by the dependent chain?).
// One wave per workgroup, one workgroup: nothing else competes for the SIMD.  Cycles from s_memtime (100 MHz? no: the shader
// clock counter clock64()).
//   hipcc --offload-arch=gfx950 -O3 -o f64_latency_probe f64_latency_probe.hip && ./f64_latency_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE> __global__ void probe(double *out, long long *cyc, double seed, double f)
{
    double a = seed + threadIdx.x * 1e-9, b = seed * 1.5, c = seed * 0.7, d = seed * 0.3;
    const double lim = 1e-6;
    long long t0 = clock64();
    for (int i = 0; i < N / 64; i++) {
        if (MODE == 0) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(f));) }                        // dependent mul chain
        if (MODE == 1) { REP64(asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(lim));) }                      // dependent max chain
        if (MODE == 2) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(f));) } // 4 independent chains: issue interval
        if (MODE == 3) { // compare -> select (both halves) -> back into the compare: cmp + 2 cndmask per step, compiler-generated
#pragma unroll
            for (int k = 0; k < 64; k++) {
                const bool gt = b > a;
                a = gt ? c : d;
                asm volatile("" : "+v"(a));
            }
        }
        if (MODE == 4) { REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(*(float *)&a) : "v"(*(float *)&f));) } // f32 dependent chain (reference)
        if (MODE == 5) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4" : "+v"(*(float *)&a), "+v"(*(float *)&b), "+v"(*(float *)&c), "+v"(*(float *)&d) : "v"(*(float *)&f));) }
        if (MODE == 6) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(float *)&a) : "v"(*(float *)&b) : "vcc");) } // dependent cndmask chain
        if (MODE == 7) { // the tracker's step: compare -> select the factor -> multiply -> clamp (compiler-generated, as in the kernel)
#pragma unroll
            for (int k = 0; k < 64; k++) {
                const bool above = b > a;
                const double fsel = above ? c : d;
                a = fmax(a * fsel, lim);
                asm volatile("" : "+v"(a));
            }
        }
    }
    long long t1 = clock64();
    out[threadIdx.x + 64 * MODE] = a + b + c + d;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8 * 8); hipMalloc(&cyc, 8 * 8);
    const char *names[8] = {"dependent v_mul_f64", "dependent v_max_f64", "4 independent v_mul_f64 chains (per instr)", "v_cmp_gt_f64 -> s_nop 1 -> v_cndmask_b32 (per pair)",
                            "dependent v_mul_f32", "4 independent v_mul_f32 chains (per instr)", "dependent v_cndmask_b32", "tracker chain cmp->2 cndmask->mul->max (per step)"};
    for (int rep = 0; rep < 2; rep++) {
        probe<0><<<1, 64>>>(out, cyc, 1.0, 1.0000001); probe<1><<<1, 64>>>(out, cyc, 1.0, 1.0); probe<2><<<1, 64>>>(out, cyc, 1.0, 1.0000001);
        probe<3><<<1, 64>>>(out, cyc, 1.0, 1.0); probe<4><<<1, 64>>>(out, cyc, 1.0, 1.0000001); probe<5><<<1, 64>>>(out, cyc, 1.0, 1.0000001);
        probe<6><<<1, 64>>>(out, cyc, 1.0, 1.0); probe<7><<<1, 64>>>(out, cyc, 1.0, 1.0000001);
        hipDeviceSynchronize();
    }
    long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    for (int m = 0; m < 8; m++) printf("%-55s %8.2f clock64 ticks per instruction / unit\n", names[m], (double)h[m] / N);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("(device clock %d kHz; clock64 = s_memtime ticks; compare the f32 rows: a dependent v_mul_f32 chain is 4+ shader cycles per instruction)\n", clk);
    return 0;
}
