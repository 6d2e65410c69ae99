// conv_1x1.hip -- the 1x1 / stride 1 convolutions of the SELD CRNN's residual shortcuts (models/model_utils.py:340-349:
// 64 -> 128 on 160x50, 128 -> 256 on 80x25, 256 -> 512 on 40x12 maps), channels-last bf16, on the gfx950 matrix cores:
// forward, data gradient (the same kernel with the transposed filter) and weight gradient.  Over the flattened pixel axis a
// 1x1 convolution is a plain GEMM, Y[p][co] = sum_ci X[p][ci] W[co][ci], with 12.6 GFLOP per direction in the whole network --
// these kernels are bound by the HBM traffic of their activations (98 / 49 / 15 MB per layer), not by the matrix pipe, and are
// written for few instructions and enough waves in flight rather than for MFMA utilisation.  (MIOpen's implicit-GEMM path with
// its cast / transpose helper kernels took 0.44 ms of a 13.4 ms training step for them.)
#include "build_guard.h" // probe switches need -DSALSA_PROBE_BUILD; SALSA_BUILD_FLAGS (generated: tools/gen_build_guard.py)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nn_det.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// ---------------------------------------------------------------------------------------------- forward / data gradient
// x [M][K], w [N][K] (both K-contiguous: an MFMA operand fragment is ONE 16-byte load per lane, no LDS), y [M][N].
// Computed transposed like the 3x3 kernels -- A = filter rows (32 output channels x 16 k), B = pixels (16 k x 32 pixels) --
// so a lane ends up with 4 consecutive output channels of ONE pixel per accumulator group: 8-byte stores.
// Workgroup = 4 waves x 32 pixels = 128 pixels x (32 NT) output channels; every wave walks K in 16-channel steps with the
// next step's NT + 1 fragments requested before the current step's NT MFMAs issue.  The filter block (<= 64 KB) is read by
// all four waves and by every workgroup: L1 / L2 traffic, not HBM.
template <int NT>
__global__ __launch_bounds__(256) void conv1x1_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ w,
                                                      unsigned short *__restrict__ y, long M, int K, int N)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l32 = lane & 31, kg = lane >> 5;
    const long p = (long)blockIdx.x * 128 + wv * 32 + l32; // this lane's pixel (B operand column, output row)
    const long pl = p < M ? p : M - 1;
    const int n0 = blockIdx.y * (32 * NT);
    const bf16x8 *xr = (const bf16x8 *)(x + pl * K + 8 * kg);
    const bf16x8 *wr = (const bf16x8 *)(w + (long)(n0 + l32) * K + 8 * kg);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = f32x16{};
    const int steps = K / 16;
    bf16x8 b_cur = xr[0], a_cur[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) a_cur[t] = wr[(long)t * 32 * K / 8];
    for (int s = 0; s < steps; s++) {
        bf16x8 b_nxt = b_cur, a_nxt[NT];
        const int sn = s + 1 < steps ? s + 1 : s; // (the last step re-requests itself: no branch in the loop)
        b_nxt = xr[2 * sn];
#pragma unroll
        for (int t = 0; t < NT; t++) a_nxt[t] = wr[(long)t * 32 * K / 8 + 2 * sn];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[t], b_cur, acc[t], 0, 0, 0);
        b_cur = b_nxt;
#pragma unroll
        for (int t = 0; t < NT; t++) a_cur[t] = a_nxt[t];
    }
    {   // 16-byte stores: the partner lane (same pixel, lane +- 32) holds the other half of every 8-channel run; v_permlane32_swap
        // trades runs so that each lane owns whole ones (round 3, as in conv_mfma.hip conv64_epilogue)
        const bool inside = p < M;                         // (the same for both lanes of a pair)
        unsigned short *o = y + (inside ? p : 0) * N + n0 + 8 * kg;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const unsigned ax = pack_bf16(acc[t][4 * g], acc[t][4 * g + 1]), ay = pack_bf16(acc[t][4 * g + 2], acc[t][4 * g + 3]);
                const unsigned bx = pack_bf16(acc[t][4 * g + 4], acc[t][4 * g + 5]), by = pack_bf16(acc[t][4 * g + 6], acc[t][4 * g + 7]);
                const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false); // lanes 32-63 of a <-> lanes 0-31 of b
                const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                if (inside) *(uint4 *)(o + 32 * t + 8 * g) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
    }
}

// ------------------------------------------------------------------------------------------------------ weight gradient
// dW[co][ci] += sum_p dy[p][co] x[p][ci]: the reduction runs over PIXELS, so both operands need 8 consecutive pixels of one
// channel per lane while memory is pixel-major -- gfx950's transposing LDS read (ds_read_b64_tr_b16; lane mapping in
// conv_mfma.hip, "weight gradient") builds such fragments from pixel-major LDS tiles.  A workgroup owns a 128 co x 64 ci block
// of dW and a contiguous share of the pixel tiles (64 pixels each; grid.z splits the pixels so that ~512 workgroups exist);
// wave w owns co 32w.. x both ci halves = 2 accumulator tiles, added to the float32 result with atomics at the end.
constexpr int WP = 64;                  // pixels per tile
constexpr int ROW_G = 128 + 8, ROW_X = 64 + 8; // padded LDS rows (bf16 elements): 272- and 144-byte pitches

struct tr_frag {
    unsigned long long lo, hi;
};
__device__ __forceinline__ bf16x8 tr_value(const tr_frag &f)
{
    union { unsigned long long q[2]; bf16x8 v; } u;
    u.q[0] = f.lo;
    u.q[1] = f.hi;
    return u.v;
}
// one fragment (16 pixels x 32 channels) = two transposing reads: pixel rows +0..3 and +4..7 of the lane's 8-pixel half
#define TR_ISSUE(f, addr, imm, row_bytes)                                                                                  \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                              \
                 : "=&v"((f).lo), "=&v"((f).hi)                                                                            \
                 : "v"(addr), "n"(imm), "n"((imm) + 4 * (row_bytes)))

template <int KS> __device__ __forceinline__ void wrw1_steps(f32x16 (&acc)[2], const unsigned ga, const unsigned xa)
{
    if constexpr (KS < WP / 16) {
        tr_frag fa, fb0, fb1;
        TR_ISSUE(fa, ga, 2 * (16 * KS * ROW_G), 2 * ROW_G);
        TR_ISSUE(fb0, xa, 2 * (16 * KS * ROW_X), 2 * ROW_X);
        TR_ISSUE(fb1, xa, 2 * (16 * KS * ROW_X + 32), 2 * ROW_X);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa.lo), "+v"(fa.hi), "+v"(fb0.lo), "+v"(fb0.hi), "+v"(fb1.lo), "+v"(fb1.hi));
        const bf16x8 a = tr_value(fa);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_value(fb0), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_value(fb1), acc[1], 0, 0, 0);
        wrw1_steps<KS + 1>(acc, ga, xa);
    }
}

__global__ __launch_bounds__(256) void conv1x1_wrw_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy,
                                                          float *__restrict__ dw, long M, int K, int N, int tiles_per_wg,
                                                          float *__restrict__ part /* deterministic mode: [gridDim.z][N*K] */)
{
    __shared__ __attribute__((aligned(16))) unsigned short gl[WP * ROW_G];
    __shared__ __attribute__((aligned(16))) unsigned short xl[WP * ROW_X];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int co0 = blockIdx.x * 128, ci0 = blockIdx.y * 64;
    const int i16 = lane & 15, cb = (lane >> 4) & 1, kh = lane >> 5;
    const unsigned ga = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)gl +
                        2u * (unsigned)((8 * kh + (i16 >> 2)) * ROW_G + 32 * wv + 16 * cb + 4 * (i16 & 3));
    const unsigned xa = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)xl +
                        2u * (unsigned)((8 * kh + (i16 >> 2)) * ROW_X + 16 * cb + 4 * (i16 & 3));
    f32x16 acc[2] = {f32x16{}, f32x16{}};
    const long n_tiles = (M + WP - 1) / WP;
    const long t_begin = (long)blockIdx.z * tiles_per_wg, t_end = t_begin + tiles_per_wg < n_tiles ? t_begin + tiles_per_wg : n_tiles;
    // 16-byte pieces per thread: dy tile 64 x 128 bf16 = 1024 pieces (4 per thread), x tile 64 x 64 = 512 (2 per thread)
    uint4 pg[4], px[2];
    auto fetch = [&](long tile) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = tid + 256 * j, row = i >> 4, piece = i & 15;
            const long p = tile * WP + row;
            pg[j] = p < M ? *(const uint4 *)(dy + p * N + co0 + 8 * piece) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = tid + 256 * j, row = i >> 3, piece = i & 7;
            const long p = tile * WP + row;
            px[j] = p < M ? *(const uint4 *)(x + p * K + ci0 + 8 * piece) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if (t_begin < t_end) fetch(t_begin);
    for (long tile = t_begin; tile < t_end; tile++) {
        __syncthreads(); // the previous tile's LDS reads are done
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = tid + 256 * j;
            *(uint4 *)(gl + (i >> 4) * ROW_G + (i & 15) * 8) = pg[j];
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = tid + 256 * j;
            *(uint4 *)(xl + (i >> 3) * ROW_X + (i & 7) * 8) = px[j];
        }
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1); // in flight during the multiply below
        wrw1_steps<0>(acc, ga, xa);
    }
    // D[m = co][n = ci]: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (t_begin < t_end) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int co = co0 + 32 * wv + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), ci = ci0 + 32 * t + (lane & 31);
                salsa_nn_accumulate(dw, part, (long)N * K, (int)blockIdx.z, (long)co * K + ci, acc[t][reg]);
            }
    }
}

} // namespace

extern "C" {

/* 1 when salsa_nn_conv1x1 takes the shape: x [M][Cin] -> y [M][Cout], Cin a multiple of 16, Cout a multiple of 64 */
int salsa_nn_conv1x1_supported(int64_t M, int Cin, int Cout)
{
    return M > 0 && Cin >= 16 && Cin % 16 == 0 && Cout >= 64 && Cout % 64 == 0 && M * (Cin > Cout ? Cin : Cout) < ((int64_t)1 << 40);
}

/* y[p][co] = sum_ci x[p][ci] * w[co][ci]: x bf16 [M][Cin] (channels-last pixels), w bf16 [Cout][Cin], y bf16 [M][Cout].  The
 * data gradient is the same call on dy with the transposed filter [Cin][Cout]. */
int salsa_nn_conv1x1(const void *x, const void *w, void *y, int64_t M, int Cin, int Cout, void *hip_stream)
{
    if (!x || !w || !y || x == y || !salsa_nn_conv1x1_supported(M, Cin, Cout)) return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    const unsigned gx = (unsigned)((M + 127) / 128);
    if (Cout % 128 == 0)
        hipLaunchKernelGGL(conv1x1_kernel<4>, dim3(gx, (unsigned)(Cout / 128)), dim3(256), 0, st, (const unsigned short *)x,
                           (const unsigned short *)w, (unsigned short *)y, (long)M, Cin, Cout);
    else
        hipLaunchKernelGGL(conv1x1_kernel<2>, dim3(gx, (unsigned)(Cout / 64)), dim3(256), 0, st, (const unsigned short *)x,
                           (const unsigned short *)w, (unsigned short *)y, (long)M, Cin, Cout);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

/* 1 when salsa_nn_conv1x1_wrw takes the shape: Cout a multiple of 128, Cin a multiple of 64 */
int salsa_nn_conv1x1_wrw_supported(int64_t M, int Cin, int Cout)
{
    return M > 0 && Cin >= 64 && Cin % 64 == 0 && Cout >= 128 && Cout % 128 == 0 && M * (Cin > Cout ? Cin : Cout) < ((int64_t)1 << 40);
}

/* dw[co][ci] += sum_p dy[p][co] * x[p][ci]: float32 [Cout][Cin], ADDED to (zero it first); x bf16 [M][Cin], dy bf16 [M][Cout] */
int salsa_nn_conv1x1_wrw(const void *x, const void *dy, float *dw, int64_t M, int Cin, int Cout, void *hip_stream)
{
    if (!x || !dy || !dw || !salsa_nn_conv1x1_wrw_supported(M, Cin, Cout)) return -1;
    const long n_tiles = (long)((M + WP - 1) / WP);
    const unsigned bx = (unsigned)(Cout / 128), by = (unsigned)(Cin / 64);
    long split = 512 / (long)(bx * by); // ~512 workgroups in all, at least two tiles each
    if (split < 1) split = 1;
    if (split > (n_tiles + 1) / 2) split = (n_tiles + 1) / 2;
    if (split < 1) split = 1;
    const int per = (int)((n_tiles + split - 1) / split);
    const unsigned bz = (unsigned)((n_tiles + per - 1) / per);
    int rc = 0;
    float *part = salsa_nn_det_begin((int)bz, (long)Cout * Cin, (hipStream_t)hip_stream, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(conv1x1_wrw_kernel, dim3(bx, by, bz), dim3(256), 0, (hipStream_t)hip_stream, (const unsigned short *)x,
                       (const unsigned short *)dy, dw, (long)M, Cin, Cout, per, part);
    if (part) return salsa_nn_det_finish(part, (int)bz, (long)Cout * Cin, dw, (hipStream_t)hip_stream);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

} // extern "C"
