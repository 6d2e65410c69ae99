// conv_mfma.hip -- 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, channels-last bf16, on the gfx950 matrix cores.
// Implicit GEMM computed transposed, C^T[co][pixel] = sum_{tap,ci} W[tap][co][ci] * X[pixel+tap][ci], so that
//   A (32 co x 16 ci)     = 16 contiguous bytes of the weight row [co][ci..ci+7]       per lane (lane&31 = co, lane>>5 = k half)
//   B (16 ci x 32 pixels) = 16 contiguous bytes of the input pixel [pixel][ci..ci+7]    per lane (lane&31 = pixel)
//   D (32 co x 32 pixels) = for ONE pixel per lane, 4 groups of 4 consecutive co        -> 8-byte bf16 stores
// i.e. neither operand needs a transpose and the result is written pixel-major (NHWC) with vector stores.
// A persistent workgroup (4 waves) walks output tiles of 4 rows x 32 pixels: wave = one half of the output channels x one
// pair of rows, its slice of the filter (9 taps x 32 co x 64 ci bf16 = 36 fragments) held in REGISTERS for the whole launch;
// only the input halo tile goes through LDS (rows padded 128 -> 144 bytes so the 16-byte operand reads of 32 consecutive
// pixels spread over all banks), the next tile's loads in flight during the current tile's MFMAs.
// Also here: the first layer's kernel (Cin <= 8 -> 64, reads the extractor's planar float32 output; "stem layer" below) and
// the 64 -> 64 weight gradient (transposing LDS reads; "weight gradient" below).
#include "build_guard.h" // probe switches need -DSALSA_PROBE_BUILD; SALSA_BUILD_FLAGS (generated: tools/gen_build_guard.py)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "nn_det.h"
#include "nn_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef CONV_WIDE_STORE
#define CONV_WIDE_STORE 1 // 16-byte epilogue stores through v_permlane32_swap (0: the 8-byte stores of rounds 1-2)
#endif
#ifndef STEM_OUT_NT
#define STEM_OUT_NT 0
#endif
#ifndef CONV64_OUT_NT
#define CONV64_OUT_NT 0 // round-5 probe: the plain epilogue's 16-byte output stores as non-temporal stores (written once; the input's halo rows are what the L2 should keep)
#endif
constexpr int CH = 64;            // channels in = out
constexpr int ROW = 72;           // padded channel stride in LDS (bf16 elements): 144 bytes
#ifndef CONV_RPW
#define CONV_RPW 2 // measured: 2 rows per wave at 2 waves per SIMD (0.44 ms for 32x640x200) beats 4 rows at 1 wave per SIMD (0.50)
#endif
#ifndef CONV_WPS
#define CONV_WPS 2
#endif
constexpr int RPW = CONV_RPW;      // output rows per wave
constexpr int TH = 2 * RPW, TW = 32; // output tile of a workgroup: waves = 2 row groups x 2 halves of the 64 output channels
constexpr int HALO_W = TW + 2, HALO_H = TH + 2;
constexpr int HALO_PIECES = HALO_H * HALO_W * 8;              // 16-byte pieces of one input tile
constexpr int PREF = (HALO_PIECES + 255) / 256;               // pieces per thread

// two float32 -> packed bf16 (low half = a), round to nearest even: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// Addressing of a map in KERNEL coordinates: pixel (n, h, w) lives at element n * sn + h * sh + w * sw.  Normal: sh = W * 64, sw = 64.
// TRANSPOSED (tap_t = 1): the kernel is handed the map with its axes swapped (its H = the image's width, its W = the image's
// height, sh = 64, sw = image width * 64) and the filter taps swapped to match (conv(x)^T = conv_{w^T}(x^T)) -- nothing moves in
// memory.  Why: a tile is 4 rows x 32 columns of kernel pixels (the MFMA's 32 columns), so a 100-column map wastes 28 of every
// 128 columns and a 200-column one 24 of 224, while the CRNN's heights (640, 320, 4800, 2400) are multiples of 32 and its
// widths of 4: transposed, every tile is full (320 x 100: 250 tiles per image instead of 320).  A pixel's 64 channels are one
// contiguous 128-byte line either way.  psn / psh / psw: the same for the 2x2-pooled output of the POOL epilogue.
struct Geo {
    long sn;
    int sh, sw, tap_t;
    long psn;
    int psh, psw;
};
__device__ __forceinline__ long geo_off(const Geo &g, long n, int h, int w) { return n * g.sn + (long)h * g.sh + (long)w * g.sw; }

// x: [N][H][W][64] bf16, w: [64 co][3][3][64 ci] bf16 (torch's channels-last weight layout), y: [N][H][W][64] bf16.
// Epilogue of one tile, shared by the two forward kernels: D -> (shift, residual, ReLU) -> bf16 stores, or the pooled variant.
template <bool POOL>
__device__ __forceinline__ void conv64_epilogue(const f32x16 (&acc)[RPW], unsigned short *__restrict__ y, const float *__restrict__ shift,
                                                const float4 (&shv)[4], const unsigned short *__restrict__ residual, int relu, long n,
                                                int th, int tw, int H, int W, int lane, int px, int mb, int rg, const Geo &geo,
                                                const uint2 (*pre)[4] = nullptr /* RES: the tile's residual pieces, loaded at the tile's top */)
{
    if (POOL) { // inference, RPW == 2: the 2x2 average pool that follows (stem, model_utils.py:224) taken on the float32 values
        // before the single rounding -- the wave's two rows are the vertical pair, the neighbouring lane the horizontal one;
        // the full-resolution activation is never written (H and W even: a tile holds whole 2x2 cells)
        const int h0 = th * TH + rg * RPW, wcol = tw * TW + px;
        const bool inside = h0 < H && wcol < W;
        uint2 pk[4];
        uint2 rvp[RPW][4];
        if (residual && pre) {
#pragma unroll
            for (int rr = 0; rr < RPW; rr++)
#pragma unroll
                for (int g = 0; g < 4; g++) rvp[rr][g] = pre[rr][g];
        } else if (residual) { // (all eight pieces requested together: see the plain epilogue below)
#pragma unroll
            for (int rr = 0; rr < RPW; rr++) {
                const long roff = geo_off(geo, n, inside ? h0 + rr : 0, inside ? wcol : 0) + 32 * mb + 4 * (lane >> 5);
#pragma unroll
                for (int g = 0; g < 4; g++) rvp[rr][g] = *(const uint2 *)(residual + roff + 8 * g);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            const float4 sh = shv[g];
#pragma unroll
            for (int rr = 0; rr < RPW; rr++) {
                float v4[4] = {acc[rr][4 * g] + sh.x, acc[rr][4 * g + 1] + sh.y, acc[rr][4 * g + 2] + sh.z, acc[rr][4 * g + 3] + sh.w};
                if (residual) {
                    const uint2 rv = rvp[rr][g];
                    v4[0] += __uint_as_float(rv.x << 16); v4[1] += __uint_as_float(rv.x & 0xffff0000u);
                    v4[2] += __uint_as_float(rv.y << 16); v4[3] += __uint_as_float(rv.y & 0xffff0000u);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) s4[j] += relu ? fmaxf(v4[j], 0.f) : v4[j];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) s4[j] = 0.25f * (s4[j] + __shfl_xor(s4[j], 1));
            pk[g].x = pack_bf16(s4[0], s4[1]);
            pk[g].y = pack_bf16(s4[2], s4[3]);
        }
        // 16-byte stores: runs traded with the partner lane (same pixel, lane +- 32), as in the plain epilogue below
        unsigned short *o = y + (n * geo.psn + (long)(inside ? h0 / 2 : 0) * geo.psh + (long)(inside ? wcol / 2 : 0) * geo.psw) + 32 * mb + 8 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            const auto sx = __builtin_amdgcn_permlane32_swap(pk[g].x, pk[g + 1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(pk[g].y, pk[g + 1].y, false, false);
            if (inside && !(px & 1)) *(uint4 *)(o + 8 * g) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
        return;
    }
    // A lane holds, of its pixel, four runs of 4 consecutive channels (8 g + 4 khalf + 0..3) and its partner lane (the same
    // pixel, lane +- 32) the other half of each 8-channel run.  Written as they lie that is four 8-byte stores per row: the
    // epilogue was store-ISSUE-bound (probe builds, 8 x 2400 x 100: 0.229 ms, 0.148 without the epilogue and its stores).
    // v_permlane32_swap (gfx950) trades runs between the two lanes -- the low lane gives away its runs 1 and 3 for the partner's
    // halves of runs 0 and 2 -- so that every lane owns two whole 8-channel runs: two 16-byte stores per row (round 3).
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
        const int h = th * TH + rg * RPW + rr, wcol = tw * TW + px;
        const bool inside = h < H && wcol < W;      // (the same for both lanes of a pair: they share the pixel)
        const long off = geo_off(geo, n, inside ? h : 0, inside ? wcol : 0) + 32 * mb + 4 * (lane >> 5);
        uint2 pk[4];
        // (round 4: the row's four residual pieces are requested together -- fetched inside the g loop the compiler waited
        // `vmcnt(0)` after each one: four dependent round trips per row, in inference and in the data gradients that add a skip branch)
        uint2 rvq[4] = {make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u)};
        if (shift && residual && pre) {
#pragma unroll
            for (int g = 0; g < 4; g++) rvq[g] = pre[rr][g];
        } else if (shift && residual) {
#pragma unroll
            for (int g = 0; g < 4; g++) rvq[g] = *(const uint2 *)(residual + off + 8 * g);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            float v4[4] = {acc[rr][4 * g], acc[rr][4 * g + 1], acc[rr][4 * g + 2], acc[rr][4 * g + 3]};
            if (shift) { // inference epilogue: folded BatchNorm shift (+ residual) (+ ReLU) before the single rounding
                const float4 sh = shv[g];
                v4[0] += sh.x; v4[1] += sh.y; v4[2] += sh.z; v4[3] += sh.w;
                if (residual) {
                    const uint2 rv = rvq[g];
                    v4[0] += __uint_as_float(rv.x << 16); v4[1] += __uint_as_float(rv.x & 0xffff0000u);
                    v4[2] += __uint_as_float(rv.y << 16); v4[3] += __uint_as_float(rv.y & 0xffff0000u);
                }
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v4[j] = fmaxf(v4[j], 0.f);
                }
            }
            pk[g].x = pack_bf16(v4[0], v4[1]);
            pk[g].y = pack_bf16(v4[2], v4[3]);
        }
#if CONV_WIDE_STORE
        // swap(A, B): lanes 32-63 of A <-> lanes 0-31 of B.  With A = run g (even), B = run g + 1: the low lane ends with
        // (own half of run g, partner's half of run g) = channels 8 g .. 8 g + 7, the high lane with the two halves of run g + 1
        unsigned short *o = y + geo_off(geo, n, inside ? h : 0, inside ? wcol : 0) + 32 * mb + 8 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            const auto sx = __builtin_amdgcn_permlane32_swap(pk[g].x, pk[g + 1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(pk[g].y, pk[g + 1].y, false, false);
            const uint4 v = make_uint4(sx[0], sy[0], sx[1], sy[1]);
#ifdef CONV_NO_STORE // probe: everything computed, nothing written
            if (v.x == 0x12345678u && v.y == 0x9abcdef0u)
#endif
            if (inside) {
#if CONV64_OUT_NT
                typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(u4v_t{v.x, v.y, v.z, v.w}, (u4v_t *)(o + 8 * g));
#else
                *(uint4 *)(o + 8 * g) = v;
#endif
            }
        }
#else
        if (inside) {
            unsigned short *o = y + off;
#pragma unroll
            for (int g = 0; g < 4; g++) {
#ifdef CONV_NO_STORE
                if (pk[g].x == 0x12345678u && pk[g].y == 0x9abcdef0u)
#endif
                *(uint2 *)(o + 8 * g) = pk[g];
            }
        }
#endif
    }
}

// Training: per-channel sum and sum of squares of the tile's bf16-ROUNDED outputs (what the BatchNorm that follows would read
// back from HBM in its own statistics pass).  A lane holds 16 channels x 2 rows of ONE pixel column.  No LDS in the tile loop
// -- LDS float atomics, even pre-reduced to 4-way conflicts, DOUBLED the kernel (0.45 -> 0.96 ms at 32 x 640 x 200) -- and no
// 32 running sums per lane either (the filter owns the register file): two DPP swaps sum each quad of neighbouring pixel
// columns, after which all four lanes of the quad hold the same 16 totals, and lane q of the quad keeps the running sums of
// channel group q only: 8 registers.  The quads are combined once, after the last tile.
__device__ __forceinline__ float dpp_xor1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_xor2(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true));
}
#ifndef CONV_STATS_DOT2
#define CONV_STATS_DOT2 0 // round 5 experiment: the two rows of a channel as ONE bf16 pair, sum and sum of squares by v_dot2_f32_bf16: 144 instead of 240 vector
                          // instructions per tile-wave and NO faster (with statistics 0.442 / 0.446 -> 0.459 / 0.460 ms at 32 x 640 x 200; 36 B of scratch); off
#endif
__device__ __forceinline__ void conv64_stats(const f32x16 (&acc)[RPW], float (&rs)[4], float (&rq)[4], int th, int tw, int H, int W,
                                             int px, int rg)
{
    const int h0 = th * TH + rg * RPW, wcol = tw * TW + px;
    const int quad = px & 3;
#if CONV_STATS_DOT2 && CONV_RPW == 2
    // Round 5.  The epilogue cost what a separate statistics pass would (tools/probes/conv64_stats_probe.py: +60 - 80 us on the 311-us
    // kernel at 32 x 640 x 200): 9 vector instructions per bf16 PAIR of neighbouring channels (pack, two unpacks, two masks, two
    // adds, two FMAs) + the quad sums.  A lane's two output ROWS of one channel make a pair as well: packed once (the same
    // round-to-nearest-even as the store's), `v_dot2_f32_bf16 pair, (1, 1)` is the channel's sum over both rows and
    // `v_dot2_f32_bf16 pair, pair` its sum of squares -- 3 instructions per channel where there were 9, and the row mask of an edge tile
    // is one AND on the pair (skipped, wave-uniformly, on interior tiles).
    const bool row1 = h0 + 1 < H;                                    // wave-uniform
    const unsigned keep_rows = !(h0 < H) ? 0u : row1 ? 0xffffffffu : 0x0000ffffu;
    const unsigned lane_mask = wcol < W ? keep_rows : 0u;            // per lane: the pixel column may lie outside the map
    const bool edge = !row1 || tw * TW + TW > W;                     // wave-uniform: some lane or row of this wave's tile part is outside
    const unsigned ones = 0x3f803f80u;                               // bf16 (1, 1)
#pragma unroll
    for (int g = 0; g < 4; g++) {
        float t[4], q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned pr = pack_bf16(acc[0][4 * g + j], acc[1][4 * g + j]);
            if (edge) pr &= lane_mask;
            asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(t[j]) : "v"(pr), "v"(ones));
            asm("v_dot2_f32_bf16 %0, %1, %1, 0" : "=v"(q[j]) : "v"(pr));
        }
        const float keep = quad == g ? 1.f : 0.f; // lane `quad` of each quad keeps channel group g = quad
#pragma unroll
        for (int j = 0; j < 4; j++) {
            t[j] += dpp_xor1(t[j]); q[j] += dpp_xor1(q[j]);
            t[j] += dpp_xor2(t[j]); q[j] += dpp_xor2(q[j]);
            rs[j] = fmaf(keep, t[j], rs[j]);
            rq[j] = fmaf(keep, q[j], rq[j]);
        }
    }
#else
    float m[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) m[rr] = (h0 + rr < H && wcol < W) ? 1.f : 0.f;
#pragma unroll
    for (int g = 0; g < 4; g++) { // accumulator elements 4 g .. 4 g + 3 = channels 8 g + j (+ the lane's base); 8 temporaries live
        float t[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int rr = 0; rr < RPW; rr++) {
                const unsigned u = pack_bf16(acc[rr][4 * g + 2 * k], acc[rr][4 * g + 2 * k + 1]);
                const float r0 = __uint_as_float(u << 16), r1 = __uint_as_float(u & 0xffff0000u);
                const float a0 = r0 * m[rr], a1 = r1 * m[rr];
                t[2 * k] += a0; t[2 * k + 1] += a1;
                q[2 * k] = fmaf(a0, r0, q[2 * k]); q[2 * k + 1] = fmaf(a1, r1, q[2 * k + 1]);
            }
        const float keep = quad == g ? 1.f : 0.f; // lane `quad` of each quad keeps channel group g = quad
#pragma unroll
        for (int j = 0; j < 4; j++) {
            t[j] += dpp_xor1(t[j]); q[j] += dpp_xor1(q[j]);
            t[j] += dpp_xor2(t[j]); q[j] += dpp_xor2(q[j]);
            rs[j] = fmaf(keep, t[j], rs[j]);
            rq[j] = fmaf(keep, q[j], rq[j]);
        }
    }
#endif
}

// The WHOLE filter lives in registers (72 A fragments per lane: a wave per SIMD may use all 512 VGPR+AGPR), so LDS only
// serves the input tile: one 16-byte B read per two MFMAs.  The next tile's halo is prefetched into registers while the
// current tile is multiplied.
template <bool POOL>
__global__ __launch_bounds__(256, CONV_WPS) void conv3x3_c64_fwd_kernel(const unsigned short *__restrict__ x,
                                                              const unsigned short *__restrict__ w,
                                                              unsigned short *__restrict__ y, int N, int H, int W,
                                                              const float *__restrict__ shift,
                                                              const unsigned short *__restrict__ residual, int relu,
                                                              double *__restrict__ /* statistics: the LDS-direct kernel only */)
{
    __shared__ __attribute__((aligned(16))) unsigned short xl[HALO_H * HALO_W * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int px = lane & 31, kh = (lane >> 5) * 8;
    const int mb = wv & 1, rg = wv >> 1; // this wave: output channels 32*mb .. +31, tile rows RPW*rg .. +RPW-1
    bf16x8 af[9][4];    // [tap][16-channel step]: filter row co = 32*mb + (lane&31), input channels kc*16 + kh .. +7
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int kc = 0; kc < 4; kc++) af[tap][kc] = *(const bf16x8 *)(w + ((long)((mb * 32 + px) * 9 + tap) * CH + kc * 16 + kh));
    float4 shv[4]; // the lane's 16 folded-BatchNorm shifts (inference), loaded once: ordinary loads inside the tile loop would
                   // make the compiler drain every load in flight, the next tiles' included
#pragma unroll
    for (int g = 0; g < 4; g++) shv[g] = shift ? *(const float4 *)(shift + 32 * mb + 4 * (lane >> 5) + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int tiles_w = (W + TW - 1) / TW, tiles_h = (H + TH - 1) / TH;
    const long n_tiles = (long)N * tiles_h * tiles_w;
    uint4 pre[PREF];
    auto fetch = [&](long tile) {
        const int tw = (int)(tile % tiles_w);
        const int th = (int)((tile / tiles_w) % tiles_h);
        const long n = tile / ((long)tiles_w * tiles_h);
#pragma unroll
        for (int j = 0; j < PREF; j++) {
            const int i = tid + j * 256;
            const int piece = i & 7, p = i >> 3;
            const int hh = p / HALO_W, ww = p - hh * HALO_W;
            const int h = th * TH + hh - 1, wcol = tw * TW + ww - 1;
            pre[j] = make_uint4(0u, 0u, 0u, 0u);
#ifndef CONV_NO_FETCH // probe: what the kernel costs without its input stream
            if (i < HALO_PIECES && h >= 0 && h < H && wcol >= 0 && wcol < W)
                pre[j] = *(const uint4 *)(x + (((n * H + h) * W + wcol) * CH + piece * 8));
#endif
        }
    };
    long tile = blockIdx.x;
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads(); // the previous tile's LDS reads are done
#pragma unroll
        for (int j = 0; j < PREF; j++) {
            const int i = tid + j * 256;
            if (i < HALO_PIECES) *(uint4 *)(xl + (long)(i >> 3) * ROW + (i & 7) * 8) = pre[j];
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x); // in flight during the multiply below
        const int tw = (int)(tile % tiles_w);
        const int th = (int)((tile / tiles_w) % tiles_h);
        const long n = tile / ((long)tiles_w * tiles_h);
        f32x16 acc[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) acc[r] = f32x16{};
        // 36 K-steps (9 taps x 4 channel blocks), software-pipelined BY HAND: the compiler sinks plain LDS loads down to their
        // first use (ds_read ; s_waitcnt ; mfma), which leaves the matrix pipe idle for the LDS latency every step.  So the
        // B reads are volatile asm with the step's constant byte offset as the instruction's immediate, and the wait is an
        // asm that takes the fragments as operands: step q+1's fragments are requested, step q's four MFMAs issue, then wait.
        const unsigned lbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)xl +
                               2u * (unsigned)((rg * RPW * HALO_W + px) * ROW + kh);
#if CONV_RPW == 2 && !defined(CONV_NO_ROW_SHARE)
        // Output rows 0 and 1 of the wave read input rows 0-2 and 1-3: a fragment of input row ir (tap column s, channel block
        // kc) serves output row 0 with filter row ir and output row 1 with filter row ir-1.  Walking the 4 x 3 x 4 = 48 distinct
        // fragments instead of the 2 x 36 (row, step) pairs cuts the LDS reads by a third for the same 72 MFMAs (the LDS pipe,
        // shared by the CU's 8 waves, was as busy as the matrix pipe).
#define LDS_B1(dst, u)                                                                                                 \
    asm volatile("ds_read_b128 %0, %1 offset:%2"                                                                       \
                 : "=v"(dst)                                                                                           \
                 : "v"(lbase), "n"(2 * ((((u) / 12) * HALO_W + ((u) / 4) % 3) * ROW + ((u) & 3) * 16)))
#define LDS_WAIT1(f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f))
#ifndef CONV_DEPTH
#define CONV_DEPTH 1 // fragments in flight ahead of the one being multiplied (1, 2, 3 measured the same)
#endif
        constexpr int DEPTH = CONV_DEPTH, RING = DEPTH + 1;
        bf16x8 bb[RING]; // fragment u multiplies while u+1 .. u+DEPTH are in flight (one MFMA pair is shorter than the LDS latency)
#pragma unroll
        for (int u = 0; u < DEPTH; u++) LDS_B1(bb[u % RING], u);
#pragma unroll
        for (int u = 0; u < 48; u++) {
            if (u + DEPTH < 48) LDS_B1(bb[(u + DEPTH) % RING], u + DEPTH);
            // wait until fragment u has landed: at most min(DEPTH, 47 - u) younger reads may still be outstanding
            if (u + DEPTH < 48) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bb[u % RING]) : "n"(DEPTH));
            else if (47 - u == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bb[u % RING]));
            else if (47 - u == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bb[u % RING]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb[u % RING]));
            __builtin_amdgcn_sched_barrier(0);
            const int ir = u / 12, sx = (u / 4) % 3, kc = u & 3;
            if (ir <= 2) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ir * 3 + sx][kc], bb[u % RING], acc[0], 0, 0, 0);
            if (ir >= 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(ir - 1) * 3 + sx][kc], bb[u % RING], acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef LDS_B1
#undef LDS_WAIT1
#else
#define LDS_B(dst, q, rr)                                                                                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2"                                                                       \
                 : "=v"(dst)                                                                                           \
                 : "v"(lbase), "n"(2 * ((((rr) + ((q) >> 2) / 3) * HALO_W + ((q) >> 2) % 3) * ROW + ((q) & 3) * 16)))
#if CONV_RPW == 4
#define LDS_WAIT(f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]))
#else
#define LDS_WAIT(f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]))
#endif
        bf16x8 bb[2][RPW]; // ping-pong: step q multiplies bb[q&1] while bb[(q+1)&1] is in flight
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) LDS_B(bb[0][rr], 0, rr);
        LDS_WAIT(bb[0]);
#pragma unroll
        for (int q = 0; q < 36; q++) {
            if (q + 1 < 36) {
#pragma unroll
                for (int rr = 0; rr < RPW; rr++) LDS_B(bb[(q + 1) & 1][rr], q + 1, rr);
            }
            __builtin_amdgcn_sched_barrier(0); // the four reads go out BEFORE this step's MFMAs (which then cover their latency)
#pragma unroll
            for (int rr = 0; rr < RPW; rr++) acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q >> 2][q & 3], bb[q & 1][rr], acc[rr], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < 36) LDS_WAIT(bb[(q + 1) & 1]);
        }
#undef LDS_B
#undef LDS_WAIT
#endif
        // D: column = lane&31 = pixel, row (= co within this wave's 32) = (reg&3) + 8*(reg>>2) + 4*(lane>>5): a lane holds, for ONE
        // pixel, four groups of four consecutive output channels -> 8-byte stores.  (Transposing through a wave-private LDS
        // strip to get 16-byte stores of 64 contiguous bytes per pixel was measured SLOWER here, 1.59 -> 2.03 ms per training
        // step, twice: this kernel's epilogue competes with the next tile's MFMAs for LDS and registers.  The stem kernel
        // below, which has next to no MFMA work, does gain from it.)
        conv64_epilogue<POOL>(acc, y, shift, shv, residual, relu, n, th, tw, H, W, lane, px, mb, rg,
                              Geo{(long)H * W * CH, W * CH, CH, 0, (long)(H / 2) * (W / 2) * CH, (W / 2) * CH, CH});
    }
}


// ---- XCD-aware walk of a persistent workgroup over the (image, tile row, tile column) space (round 3).
// Workgroup b runs on XCD b % 8, each XCD with its own L2.  Dealing the tiles round-robin (tile = b + i * grid) puts vertically
// neighbouring tiles -- which share two of a tile's six halo rows -- on different XCDs, so every halo row came from HBM twice:
// rocprofv3 FETCH_SIZE of the forward kernel at 8 x 2400 x 100 was 392 MB for a 246-MB input (1.59x = the 6/4 x 34/32 halo), of
// the weight-gradient kernel 1.34 GB for 1.05 GB.  Here the flattened (image, tile row) axis is cut into bands of TW_BAND rows,
// band k belongs to XCD k % 8, and the G = grid / 8 workgroups of an XCD walk ITS bands' tiles in order: at any moment an XCD works
// on ~G consecutive tiles of a few neighbouring rows, and the shared halo rows are L2 hits.  nx = 1 (grids that are not a
// multiple of 8) degenerates to the plain round-robin order.
constexpr int TW_BAND = 4;
struct TileWalk {
    int tiles_w, tiles_h, nx, xcd, G, n_local, d_tw, d_lr;
};
struct TilePos {
    int tw, lr; // tile column, row index among this XCD's rows
};
__device__ __forceinline__ TileWalk tile_walk(int N, int tiles_h, int tiles_w, int grid, int b)
{
    TileWalk w;
    w.tiles_w = tiles_w;
    w.tiles_h = tiles_h;
    w.nx = (grid % 8 == 0 && grid >= 64) ? 8 : 1;
    w.xcd = b % w.nx;
    w.G = grid / w.nx;
    const int rows = N * tiles_h, full = rows / TW_BAND, rem = rows % TW_BAND;
    const int mine = full / w.nx + (w.xcd < full % w.nx ? 1 : 0);
    w.n_local = (mine * TW_BAND + (full % w.nx == w.xcd ? rem : 0)) * tiles_w;
    w.d_tw = w.G % tiles_w;
    w.d_lr = w.G / tiles_w;
    return w;
}
__device__ __forceinline__ TilePos tile_pos(const TileWalk &w, int s) { return {s % w.tiles_w, s / w.tiles_w}; }
__device__ __forceinline__ void tile_advance(const TileWalk &w, TilePos &p) // s += G
{
    p.tw += w.d_tw;
    const int carry = p.tw >= w.tiles_w ? 1 : 0;
    p.tw -= carry ? w.tiles_w : 0;
    p.lr += w.d_lr + carry;
}
__device__ __forceinline__ void tile_coords(const TileWalk &w, const TilePos &p, int &n, int &th)
{
    const int r = ((p.lr / TW_BAND) * w.nx + w.xcd) * TW_BAND + p.lr % TW_BAND; // global (image, tile row) index
    n = r / w.tiles_h;
    th = r - n * w.tiles_h;
}

// ---- variant with the input tiles loaded STRAIGHT INTO LDS (global_load_lds_dwordx4), two tiles ahead.
// Probe builds of the kernel above: 0.242 ms for 8 x 2400 x 100, 0.143 ms without its input loads -- the next tile's halo,
// prefetched into registers ONE tile ahead, arrives later than one tile's multiply takes, and 256 VGPRs leave no room for a
// second staging set.  Direct-to-LDS loads need no staging registers: three LDS tiles (2 x 78 KB per CU) rotate, the loads
// for tile i+2 are issued when tile i starts, and the LDS write pass and one of the two barriers per tile disappear.  A wave's
// load instruction fills 64 consecutive 16-byte LDS slots, so the pixel stride in LDS is exactly 128 bytes; the bank
// conflicts that the padded layout avoided are avoided here by a rotation: channel block c of pixel p sits in slot
// (c + p) & 7, applied to the GLOBAL address when loading and to the LDS address when reading (where pixel and block offsets
// of a fragment are compile-time constants: eight per-lane base registers, one per value of the constant part mod 8).  Halo pixels outside the
// image load from a 16-byte zero constant.
__device__ uint4 conv_zero16; // zero-initialised
// its address pinned in a scalar register pair (used in place it is re-loaded from the GOT -- s_getpc, s_load_dwordx2, s_waitcnt
// lgkmcnt(0) -- in front of the halo loads of every tile; see conv_wide.hip wide_zero_ptr)
__device__ __forceinline__ const unsigned short *conv_zero_ptr()
{
    const unsigned short *z = (const unsigned short *)&conv_zero16;
    asm volatile("" : "+s"(z));
    return z;
}

constexpr int APIX = 64;                             // bf16 per pixel in LDS (no padding)
constexpr int ABUF = HALO_H * HALO_W * APIX;         // one tile
constexpr int NBUF = 3;
constexpr int AFETCH = (HALO_PIECES + 255) / 256;    // load instructions per thread per tile (the last one partial)

// The 48 fragment steps of one tile of the LDS-direct kernel as a compile-time recursion (every index a constant: written as a
// loop over a permuted step number the filter array was indexed "dynamically" and the compiler moved it to scratch memory).
// Step S multiplies fragment FRAG(S) = (input row ir, tap column sx, channel block kc) while the next AD fragments are in
// flight; rb[] are the lane's eight rotated base addresses (see the kernel).
#ifndef CONV_FETCH_NO_HOIST
#define CONV_FETCH_NO_HOIST 1
#endif
#ifndef CONV_RES_EARLY
#define CONV_RES_EARLY 1
#endif
#ifndef CONV_ORDER
#define CONV_ORDER 1 // 1: fragments ordered so that consecutive MFMAs never share an accumulator (0: row by row, round 1)
#endif
constexpr int conv64_frag(int s) { return CONV_ORDER ? (s < 24 ? ((s & 1) ? 36 + (s >> 1) : (s >> 1)) : 12 + (s - 24)) : s; }

template <int U>
__device__ __forceinline__ void conv64_lds_read(bf16x8 &dst, const unsigned (&rb)[8])
{
    constexpr int ir = U / 12, sx = (U / 4) % 3, kc = U & 3;
#ifdef CONV_NO_LDSREAD // probe: the fragment is whatever the register holds
    asm volatile("" : "=v"(dst) : "v"(rb[(2 * kc + sx + 2 * ir) & 7]));
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2"
                 : "=v"(dst)
                 : "v"(rb[(2 * kc + sx + 2 * ir) & 7]), "n"(2 * ((ir * (TW + 2) + sx) * 64)));
#endif
}

#ifdef CONV_NO_SCHED // (probe: let the compiler order the step stream)
#define CONV_SCHED_BARRIER() do {} while (0)
#else
#define CONV_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
// HOOK: called with std::integral_constant<int, j> before step CONV_HOOK_FIRST + CONV_HOOK_EVERY * j, j = 0 .. AFETCH - 1: the
// kernel issues one wave-load of the tile-after-next there (see "spread fetch" in the kernel)
#ifndef CONV_HOOK_FIRST
#define CONV_HOOK_FIRST 2
#endif
#ifndef CONV_HOOK_EVERY
#define CONV_HOOK_EVERY 6
#endif
struct conv64_no_hook {
    template <class J> __device__ __forceinline__ void operator()(J) const {}
};
template <int S, int AD, class HOOK = conv64_no_hook>
__device__ __forceinline__ void conv64_steps(const bf16x8 (&af)[9][4], bf16x8 (&bb)[AD + 1], f32x16 (&acc)[RPW], const unsigned (&rb)[8],
                                             const HOOK &hook = HOOK())
{
    constexpr int RING = AD + 1;
    if constexpr (S >= CONV_HOOK_FIRST && (S - CONV_HOOK_FIRST) % CONV_HOOK_EVERY == 0 && (S - CONV_HOOK_FIRST) / CONV_HOOK_EVERY < 8 && S < 48)
        hook(std::integral_constant<int, (S - CONV_HOOK_FIRST) / CONV_HOOK_EVERY>{});
    if constexpr (S == 0) {
#pragma unroll
        for (int s = 0; s < AD; s++) { // (AD is 1 or 2: spelled out so that the fragment numbers stay template constants)
            if (s == 0) conv64_lds_read<conv64_frag(0)>(bb[0], rb);
            if (s == 1) conv64_lds_read<conv64_frag(1)>(bb[1 % RING], rb);
            if (s == 2) conv64_lds_read<conv64_frag(2)>(bb[2 % RING], rb);
        }
    }
    if constexpr (S < 48) {
        if constexpr (S + AD < 48) conv64_lds_read<conv64_frag(S + AD < 48 ? S + AD : 0)>(bb[(S + AD) % RING], rb);
        // fragment S has landed when at most the younger reads are outstanding
        if constexpr (S + AD < 48) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bb[S % RING]) : "n"(AD));
        else if constexpr (47 - S == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bb[S % RING]));
        else if constexpr (47 - S == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bb[S % RING]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb[S % RING]));
        CONV_SCHED_BARRIER();
        constexpr int u = conv64_frag(S), ir = u / 12, sx = (u / 4) % 3, kc = u & 3;
        // (the first product into each accumulator takes a ZERO C operand -- an inline constant of the instruction -- instead of
        // 32 v_mov clearing the accumulators every tile; with CONV_ORDER these are steps 0 (row 0 -> acc 0) and 1 (row 3 -> acc 1))
        constexpr bool first0 = CONV_ORDER && S == 0, first1 = CONV_ORDER && S == 1;
        if constexpr (ir <= 2) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ir * 3 + sx][kc], bb[S % RING], first0 ? f32x16{} : acc[0], 0, 0, 0);
        if constexpr (ir >= 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(ir - 1) * 3 + sx][kc], bb[S % RING], first1 ? f32x16{} : acc[1], 0, 0, 0);
        CONV_SCHED_BARRIER();
        conv64_steps<S + 1, AD, HOOK>(af, bb, acc, rb, hook);
    }
}

// XF (round 4): the input is the RAW output of the convolution before (x1) and this kernel's real operand is
// a = dropout(relu(batch_norm(x1))) -- the normalised activation is never written to HBM nor read back (524 MB each way for the
// stem's first BatchNorm).  The tiles still arrive by LDS-direct loads; once a thread's own pieces of tile i+1 have landed (the
// counted wait that already exists) it rewrites them in place -- a = relu(x * scale[c] + shift[c]), the dropout mask regenerated
// from the element index exactly as bn_apply_kernel draws it (nn_common.h) -- and the barrier that opens tile i+1 publishes them.
// A thread's pieces always hold the same eight channels, so its 16 coefficients come from a 512-byte LDS table once per tile;
// halo pieces outside the image stay the zeros they were loaded as (the convolution pads the ACTIVATION with zeros).
struct XformArgs {
    const float *mean, *invstd, *gamma, *beta; // the BatchNorm's batch statistics and affine parameters, float32 [64]
    DropArgs drop;
};

// RES (round 5): the launch adds a residual (inference: the block's shortcut; training: the skip branch's gradient in a data
// gradient).  Loaded inside the epilogue, the residual pieces were YOUNGER than the LDS-direct loads of the tile after next, so the
// `s_waitcnt vmcnt(0)` in front of their first use drained that prefetch and exposed the residual's own latency twice per tile (once
// per output row).  Here the tile's eight pieces are requested at the tile's TOP, before the prefetch is issued: they are older,
// the counted wait that lets the prefetch stay in flight covers them, and they land during the multiply.  16 registers, which
// the instantiation has once the load offsets are recomputed per tile (CONV_FETCH_NO_HOIST).
template <bool POOL, bool STATS = false, bool XF = false, bool RES = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_fwd_async_kernel(const unsigned short *__restrict__ x,
                                                                       const unsigned short *__restrict__ w,
                                                                       unsigned short *__restrict__ y, int N, int H, int W,
                                                                       const float *__restrict__ shift,
                                                                       const unsigned short *__restrict__ residual, int relu,
                                                                       double *__restrict__ stats_part, const Geo geo,
                                                                       const XformArgs xf = XformArgs{})
{
    static_assert(RPW == 2, "row sharing below is written for two rows per wave");
    static_assert(!(POOL && STATS), "statistics are a training feature, the fused pool an inference one");
    __shared__ __attribute__((aligned(16))) unsigned short xl[NBUF * ABUF];
    __shared__ float lstats[STATS ? 2 * 128 : 1]; // [row-group wave][sum | sum of squares][64 channels], used once after the last tile
    __shared__ __attribute__((aligned(16))) float xtab[XF ? 2 * 64 : 4]; // XF: scale[64] = invstd * gamma, shift[64] = beta - mean * scale
    float rs[4] = {0.f, 0.f, 0.f, 0.f}, rq[4] = {0.f, 0.f, 0.f, 0.f}; // STATS: this lane's running sums (conv64_stats)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (XF) {
        if (tid < 64) {
            const float sc = xf.invstd[tid] * xf.gamma[tid];
            xtab[tid] = sc;
            xtab[64 + tid] = xf.beta[tid] - xf.mean[tid] * sc;
        }
        __syncthreads(); // (before any LDS-direct load is in flight)
    }
    const int px = lane & 31, khalf = lane >> 5, kh = khalf * 8;
    const int mb = wv & 1, rg = wv >> 1;
    const unsigned short *const zero16 = conv_zero_ptr();
    bf16x8 af[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int kc = 0; kc < 4; kc++) // (transposed geometry: tap (ky, kx) multiplies the filter's (kx, ky))
            af[tap][kc] = *(const bf16x8 *)(w + ((long)((mb * 32 + px) * 9 + (geo.tap_t ? (tap % 3) * 3 + tap / 3 : tap)) * CH + kc * 16 + kh));
    float4 shv[4]; // the lane's 16 folded-BatchNorm shifts (inference), loaded once: ordinary loads inside the tile loop would
                   // make the compiler drain every load in flight, the next tiles' included
#pragma unroll
    for (int g = 0; g < 4; g++)
        shv[g] = (!STATS && shift) ? *(const float4 *)(shift + 32 * mb + 4 * (lane >> 5) + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);

    // the filter has landed BEFORE the tile loop: left to its first use, the compiler's wait for these ordinary loads sits in
    // front of the loop's first MFMA as vmcnt(0) and drains the LDS-direct loads in flight with it, every tile
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int kc = 0; kc < 4; kc++) asm volatile("" ::"v"(af[tap][kc]));
#pragma unroll
    for (int g = 0; g < 4; g++) asm volatile("" ::"v"(shv[g].x), "v"(shv[g].y), "v"(shv[g].z), "v"(shv[g].w));
    const int tiles_w = (W + TW - 1) / TW, tiles_h = (H + TH - 1) / TH;
    // Tile cursors.  Probe builds with neither loads nor stores still ran at 2x the MFMA time: the per-tile integer work
    // (two divisions per decode, a division by 34 and 64-bit address arithmetic per load) was the bound.  So a cursor keeps
    // (tw, th, n) and steps by the grid size with carries, and a thread's halo pixel walks rows incrementally (its channel
    // block never changes: the slot index advances by 256 = 0 mod 8 and the pixel by 32 = 0 mod 8 per load).
    struct Cursor { int tw, th, n; };
    const TileWalk walk = tile_walk(N, tiles_h, tiles_w, (int)gridDim.x, (int)blockIdx.x);
    const int stride = walk.G;                       // step of this workgroup's position in its XCD's tile sequence
    auto cursor_of = [&](const TilePos &p) {
        Cursor c;
        c.tw = p.tw;
        tile_coords(walk, p, c.n, c.th);
        return c;
    };
    const int p0 = tid >> 3, hh0 = p0 / HALO_W, ww0 = p0 - hh0 * HALO_W;
    const int blk8 = (((tid & 7) - p0) & 7) * 8; // channel offset of this thread's 16 bytes: slot (block + pixel) & 7
    auto fetch = [&](const Cursor &c, int buf) {
        const unsigned short *origin = x + geo_off(geo, c.n, c.th * TH, c.tw * TW); // pixel (0, 0) of the tile
        const int h_lo = -c.th * TH, h_hi = H - c.th * TH, w_lo = -c.tw * TW, w_hi = W - c.tw * TW; // valid (row, col) - origin
        int hh = hh0 - 1, ww = ww0 - 1; // halo pixel relative to the tile origin
#if CONV_FETCH_NO_HOIST
        // Round 5.  hh0 / ww0 never change, so the compiler hoisted the seven pieces' sign-extended element offsets out of the tile
        // loop as seven 64-bit register pairs -- and in the instantiations that need a few registers more (the statistics epilogue:
        // 8 running sums) it SPILLED three of them: every tile then reloaded them from scratch with `s_waitcnt vmcnt(0)` in front of
        // the LDS-direct load that uses them, i.e. three times per tile a wave sat out everything it had in flight, the next two
        // tiles' loads included (+70 us on the 311-us kernel at 32 x 640 x 200; found in the ISA, the counters only showed
        // "waiting").  Laundering the two start values makes the offsets loop-variant: ~5 integer instructions per piece per tile,
        // recomputed, and nothing to spill.
        // (Only where it spilled: the plain forward / data-gradient instantiation fits with the offsets hoisted and is 2 - 7 % faster so.)
        if (STATS || POOL || XF || RES) asm volatile("" : "+v"(hh), "+v"(ww));
#endif
#pragma unroll
        for (int j = 0; j < AFETCH; j++) {
            const bool inside = hh >= h_lo && hh < h_hi && ww >= w_lo && ww < w_hi;
            const unsigned short *src = inside ? origin + (hh * geo.sh + ww * geo.sw + blk8) : zero16;
#ifdef CONV_ZERO_SRC // (probe: every halo load answered from one cached line)
            src = zero16;
#endif
#ifdef CONV_PROBE_ROLL // (TIMING probe, wrong results: two of the six halo rows answered from one cached line -- the traffic of a tile
            if (hh < 1) src = zero16; // walk that keeps the rows consecutive tiles share in LDS)
#endif
#ifdef CONV_FETCH_COLMAJOR // (TIMING probe, wrong results: neighbouring lanes fetch along the tile's SHORT axis -- contiguous 768-byte
            {              // runs in the transposed geometry -- instead of along the long one, 34 lines a map row apart)
                const int q = (tid >> 3) + 32 * j, h2 = q % HALO_H - 1, w2 = q / HALO_H - 1;
                const bool in2 = q < HALO_H * HALO_W && h2 >= h_lo && h2 < h_hi && w2 >= w_lo && w2 < w_hi;
                src = in2 ? origin + (h2 * geo.sh + w2 * geo.sw + blk8) : zero16;
            }
#endif
            unsigned short *dst = xl + buf * ABUF + (j * 256 + wv * 64) * 8; // the wave's 64 slots (lane l -> + 16 l bytes)
#ifndef CONV_NO_FETCH
            if (tid + j * 256 < HALO_PIECES)
#else
            if (tid < 0)
#endif
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            ww += 32; // the next load of this thread: 32 pixels on
            const int wrap = ww >= HALO_W - 1 ? 1 : 0;
            ww -= wrap ? HALO_W : 0;
            hh += wrap;
        }
    };
    // Spread fetch (round 4 experiment, OFF).  Counters of this kernel at 32 x 640 x 200 (tools/probes/pmc_mem.sh): the CU's L1 is
    // stalled on pending requests 73 % of the launch and on its translation queue 30 %, while an L2 read is answered in ~800 cycles
    // on average; it sees 9.5e7 accesses per launch -- one per LANE of every 16-byte load and store, halo re-reads included -- i.e.
    // 373 k accesses per CU in a 551 k-cycle launch, and with every load answered from one cached line (CONV_ZERO_SRC) the same
    // accesses pass at 0.96 per cycle in 388 k cycles.  So the L1 path (about one 16-byte lane access per cycle and CU = 8 TB/s over
    // the chip) is this kernel's second roofline, next to the matrix pipe, and misses make each access dearer.  Hypothesis tested
    // here: the seven wave-loads of a tile, issued in one burst, block their waves in the ISSUE of the loads; so piece j of the
    // tile-after-next is issued from inside the multiply (conv64_steps' hook, address formed from the compile-time piece number).
    // Measured: plain forward 341 -> 349 us, with statistics 411 -> 399, 320 x 100 the same: it is the L1's RATE, not the burst.
#ifndef CONV_SPREAD_FETCH
#define CONV_SPREAD_FETCH 0
#endif
    struct FetchCtx { const unsigned short *origin; int h_lo, h_hi, w_lo, w_hi, buf; bool on; };
    auto fetch_ctx = [&](const Cursor &c, int buf, bool on) {
        FetchCtx f;
        f.origin = x + geo_off(geo, c.n, c.th * TH, c.tw * TW);
        f.h_lo = -c.th * TH; f.h_hi = H - c.th * TH; f.w_lo = -c.tw * TW; f.w_hi = W - c.tw * TW;
        f.buf = buf; f.on = on;
        return f;
    };
    auto fetch_piece = [&](auto jc, const FetchCtx &f) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < AFETCH) {
            if (f.on && (j * 256 + wv * 64 < HALO_PIECES)) { // (wave-uniform)
                const int q = p0 + 32 * j;                                    // this thread's halo pixel of piece j, row-major in 6 x 34
                const int hq = (q * 241) >> 13, wq = q - hq * HALO_W;         // q / 34 for q < 236
                const int hh = hq - 1, ww = wq - 1;
                const bool inside = hh >= f.h_lo && hh < f.h_hi && ww >= f.w_lo && ww < f.w_hi && tid + j * 256 < HALO_PIECES;
                const unsigned short *src = inside ? f.origin + (hh * geo.sh + ww * geo.sw + blk8) : zero16;
                unsigned short *dst = xl + f.buf * ABUF + (j * 256 + wv * 64) * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            }
        }
    };
    static_assert(HALO_H * HALO_W + 31 < 236 && AFETCH <= 8, "q / 34 == (q * 241) >> 13; eight hook slots");
    // XF: rewrite this thread's own (landed) pieces of a tile in place
    auto transform = [&](const Cursor &c, int buf) {
        const int h_lo = -c.th * TH, h_hi = H - c.th * TH, w_lo = -c.tw * TW, w_hi = W - c.tw * TW;
        const long origin = geo_off(geo, c.n, c.th * TH, c.tw * TW); // element offset of the tile's pixel (0, 0) in the tensor
        const float4 s0 = *(const float4 *)(xtab + blk8), s1 = *(const float4 *)(xtab + blk8 + 4);
        const float4 t0 = *(const float4 *)(xtab + 64 + blk8), t1 = *(const float4 *)(xtab + 64 + blk8 + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        int hh = hh0 - 1, ww = ww0 - 1;
#ifndef C64_XF_UNROLL
#define C64_XF_UNROLL 1 // pieces rewritten at a time: the filter owns the register file (unrolled by 7: 256 bytes of scratch per lane)
#endif
#pragma unroll C64_XF_UNROLL
        for (int j = 0; j < AFETCH; j++) {
            const bool inside = hh >= h_lo && hh < h_hi && ww >= w_lo && ww < w_hi;
            if (inside && tid + j * 256 < HALO_PIECES) {
                uint4 *slot = (uint4 *)(xl + buf * ABUF + (j * 256 + tid) * 8);
                const uint4 v = *slot;
                const unsigned in[4] = {v.x, v.y, v.z, v.w};
                unsigned out[4];
                const long e0 = origin + (long)hh * geo.sh + (long)ww * geo.sw + blk8; // the piece's first element in the tensor
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float a = fmaxf(__uint_as_float(in[k] << 16) * sc[2 * k] + sh[2 * k], 0.f);
                    float b = fmaxf(__uint_as_float(in[k] & 0xffff0000u) * sc[2 * k + 1] + sh[2 * k + 1], 0.f);
                    if (xf.drop.thresh) { // one 32-bit hash per pair of neighbouring elements, 16 bits each (drop_keep)
                        const unsigned hsh = drop_hash((unsigned)(e0 >> 1) + k, xf.drop.seed);
                        a = (hsh & 0xFFFFu) >= xf.drop.thresh ? a * xf.drop.scale : 0.f;
                        b = (hsh >> 16) >= xf.drop.thresh ? b * xf.drop.scale : 0.f;
                    }
                    out[k] = pack_bf16(a, b);
                }
                *slot = make_uint4(out[0], out[1], out[2], out[3]);
            }
            ww += 32;
            const int wrap = ww >= HALO_W - 1 ? 1 : 0;
            ww -= wrap ? HALO_W : 0;
            hh += wrap;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the rewritten pieces are in LDS before this wave reaches the next barrier
    };
    // Loads of ONE wave return in order, but its stores retire independently, so a counted wait is only exact where nothing
    // but loads is younger than the loads waited for.  Order per tile: barrier -> issue tile+2 -> multiply tile -> wait until
    // only tile+2's loads are outstanding (tile+1 has landed; the stores of the tile before retired during the multiply)
    // -> store tile.  The barrier then makes every wave's part of tile+1 visible and frees the buffer tile+3 goes into.
#define WAIT_ALL_BUT_LAST_FETCH()                                                        \
    do {                                                                                 \
        if (wv < (HALO_PIECES % 256 + 63) / 64) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AFETCH) : "memory");     \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AFETCH - 1) : "memory");             \
    } while (0)
    static_assert(HALO_PIECES % 256 != 0, "the last fetch instruction is issued by the first (HALO_PIECES % 256 + 63) / 64 waves only");
    const int n_tiles = walk.n_local;                // tiles of this workgroup's XCD; `tile` = position in that sequence
    int tile = (int)blockIdx.x / walk.nx;
    TilePos pcur = tile_pos(walk, tile), pahead = pcur;
    if (tile < n_tiles) fetch(cursor_of(pahead), 0);
    tile_advance(walk, pahead);
    TilePos pnext = pahead;                         // XF: one tile ahead of `pcur`
    if (tile + stride < n_tiles) {
        fetch(cursor_of(pahead), 1);
        WAIT_ALL_BUT_LAST_FETCH();
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (XF && tile < n_tiles) transform(cursor_of(pcur), 0);
    tile_advance(walk, pahead); // two tiles ahead of `pcur` from here on
    for (int it = 0; tile < n_tiles; tile += stride, it++, tile_advance(walk, pcur), tile_advance(walk, pahead), tile_advance(walk, pnext)) {
        const Cursor cur = cursor_of(pcur);
#ifndef CONV_NO_BARRIER // (probe)
        __builtin_amdgcn_s_barrier(); // (a raw barrier: __syncthreads() would drain the loads in flight)
#endif
        const int tw = cur.tw, th = cur.th;
        const long n = cur.n;
        uint2 rpre[RPW][4];
        if (RES) { // this tile's residual pieces, requested BEFORE the prefetch below (same addresses as conv64_epilogue's own loads)
#pragma unroll
            for (int rr = 0; rr < RPW; rr++) {
                const int h0 = th * TH + rg * RPW, wcol = tw * TW + px;
                const int h = POOL ? h0 + rr : h0 + rr;
                const bool inside = POOL ? (h0 < H && wcol < W) : (h < H && wcol < W);
                const unsigned short *rp = residual + (geo_off(geo, n, inside ? h : 0, inside ? wcol : 0) + 32 * mb + 4 * (lane >> 5));
                // (asm: as ordinary loads the compiler waits for them itself at their first use -- with `vmcnt(0)`, because the
                // prefetch behind them is issued conditionally and it merges the paths -- and drains the prefetch after all; the
                // counted wait after the multiply is what covers them)
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rpre[rr][0]) : "v"(rp));
                asm volatile("global_load_dwordx2 %0, %1, off offset:16" : "=v"(rpre[rr][1]) : "v"(rp));
                asm volatile("global_load_dwordx2 %0, %1, off offset:32" : "=v"(rpre[rr][2]) : "v"(rp));
                asm volatile("global_load_dwordx2 %0, %1, off offset:48" : "=v"(rpre[rr][3]) : "v"(rp));
            }
        }
#if CONV_SPREAD_FETCH
        const FetchCtx fctx = fetch_ctx(cursor_of(pahead), (it + 2) % NBUF, tile + 2 * stride < n_tiles); // issued inside the multiply
#else
        if (tile + 2 * stride < n_tiles) fetch(cursor_of(pahead), (it + 2) % NBUF); // into the buffer of the tile before
#endif
        f32x16 acc[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) acc[r] = f32x16{};
        // 48 distinct fragments (4 input rows x 3 tap columns x 4 channel blocks) for 72 MFMAs, as in the kernel above
        // fragment u = (input row ir, tap column sx, channel block kc): pixel (rg*RPW + ir) * 34 + px + sx, block 2*kc + khalf ->
        // slot (block + pixel) & 7 = (L + K) & 7 with L = the lane's part and K = 2*kc + sx + 2*ir (34 = 2 mod 8) compile-time
        unsigned rb[8];
        {
            const unsigned bbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)xl +
                                   2u * (unsigned)((it % NBUF) * ABUF + (rg * RPW * HALO_W + px) * APIX);
            const int L = 2 * RPW * rg + px + khalf;
#pragma unroll
            for (int m = 0; m < 8; m++) rb[m] = bbase + 16u * (unsigned)((L + m) & 7);
        }
#define LDS_BA(dst, u)                                                                                                 \
    asm volatile("ds_read_b128 %0, %1 offset:%2"                                                                       \
                 : "=v"(dst)                                                                                           \
                 : "v"(rb[(2 * ((u) & 3) + ((u) / 4) % 3 + 2 * ((u) / 12)) & 7]), "n"(2 * ((((u) / 12) * HALO_W + ((u) / 4) % 3) * APIX)))
#ifndef CONV_ADEPTH
#define CONV_ADEPTH 1 // 1 and 2 measured the same, 3 slower (spills): LDS latency is not what this loop waits for
#endif
        constexpr int AD = CONV_ADEPTH, RING = AD + 1; // fragments in flight ahead of the one being multiplied
        bf16x8 bb[RING];
        // Fragment ORDER.  Input rows 0 and 3 serve one output row each, rows 1 and 2 both: walked row by row, the first and
        // the last 12 MFMAs of a tile all accumulate into the SAME register block back to back, and a dependent
        // v_mfma_f32_32x32x16_bf16 cannot start until its predecessor's 8 passes have drained (with the ds_read / s_waitcnt /
        // s_nop issue slots between them the gap is the microarchitecture guide's "+43 cycles" case): a third of the MFMAs
        // ran at half rate.  Interleaving row 0 with row 3 alternates the two accumulators on EVERY step.
#if CONV_SPREAD_FETCH
        conv64_steps<0, AD>(af, bb, acc, rb, [&](auto jc) { fetch_piece(jc, fctx); });
#else
        conv64_steps<0, AD>(af, bb, acc, rb);
#endif
#undef LDS_BA
        if (tile + 2 * stride < n_tiles) WAIT_ALL_BUT_LAST_FETCH();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (RES) // the residual pieces are older than everything the counted wait above lets stay in flight: they have landed
            asm volatile("" : "+v"(rpre[0][0]), "+v"(rpre[0][1]), "+v"(rpre[0][2]), "+v"(rpre[0][3]), "+v"(rpre[1][0]), "+v"(rpre[1][1]),
                              "+v"(rpre[1][2]), "+v"(rpre[1][3])
                         :: "memory");
#ifdef CONV_NO_EPI // probe: the accumulators are kept alive, nothing is converted or stored
        asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
#else
        conv64_epilogue<POOL>(acc, y, STATS ? nullptr : shift, shv, residual, relu, n, th, tw, H, W, lane, px, mb, rg, geo, RES ? rpre : nullptr);
        if (STATS) conv64_stats(acc, rs, rq, th, tw, H, W, px, rg);
#endif
        // XF: tile i+1 -- this thread's pieces landed at the wait above; rewritten here, after the epilogue, where the accumulators
        // are dead (before it: 108 bytes of scratch per lane)
        if (XF && tile + stride < n_tiles) transform(cursor_of(pnext), (it + 1) % NBUF);
    }
#undef WAIT_ALL_BUT_LAST_FETCH
    if (STATS) { // this workgroup's partial sums: [2][64] float64, row blockIdx.x of the BatchNorm kernels' partial table
        // the 8 quads of a half-wave hold the same channel groups: three butterfly steps over pixel columns +4, +8, +16 (once per
        // launch; LDS atomics here cost 16 us of a 110-us launch), then the two row-group waves of a channel half meet in LDS
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int d = 4; d <= 16; d <<= 1) {
                rs[j] += __shfl_xor(rs[j], d);
                rq[j] += __shfl_xor(rq[j], d);
            }
        if (px < 4) {
            const int cbase = rg * 128 + 32 * mb + 4 * khalf + 8 * px;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                lstats[cbase + j] = rs[j];
                lstats[cbase + 64 + j] = rq[j];
            }
        }
        __syncthreads();
        if (tid < 128) stats_part[(long)blockIdx.x * 128 + tid] = (double)lstats[tid] + (double)lstats[128 + tid];
    }
}

#ifndef CONV_ASYNC
#define CONV_ASYNC 1
#endif
#ifndef CONV_BIG_GRID
#define CONV_BIG_GRID 1024 // persistent workgroups for large inputs (two resident per CU)
#endif
#if CONV_ASYNC
#define CONV_FWD_KERNEL conv3x3_c64_fwd_async_kernel
#else
#define CONV_FWD_KERNEL conv3x3_c64_fwd_kernel
#endif

} // namespace

#ifndef C64_TRANSPOSE
#define C64_TRANSPOSE 1 // hand the 64 -> 64 kernels the map with its axes swapped when that needs >= 5 % fewer tiles (see struct Geo)
#endif
#if CONV_ASYNC
#define C64_GEO_ARG(pl) , (pl).geo
#else
#define C64_GEO_ARG(pl)
#endif
namespace {
struct C64Plan {
    int Hk, Wk; // the map's extent in kernel coordinates
    long tiles;
    Geo geo;
};
C64Plan c64_plan(int64_t N, int H, int W, int th, int tw)
{
    const long tn = (long)N * ((H + th - 1) / th) * ((W + tw - 1) / tw), tt = (long)N * ((W + th - 1) / th) * ((H + tw - 1) / tw);
    const bool tr = C64_TRANSPOSE && CONV_ASYNC && tt * 100 < tn * 95;
    C64Plan p;
    p.Hk = tr ? W : H;
    p.Wk = tr ? H : W;
    p.tiles = tr ? tt : tn;
    p.geo.sn = (long)H * W * CH;
    p.geo.sh = tr ? CH : W * CH;
    p.geo.sw = tr ? W * CH : CH;
    p.geo.tap_t = tr ? 1 : 0;
    p.geo.psn = (long)(H / 2) * (W / 2) * CH;
    p.geo.psh = tr ? CH : (W / 2) * CH;
    p.geo.psw = tr ? (W / 2) * CH : CH;
    return p;
}
} // namespace

extern "C" int salsa_nn_conv3x3_c64(const void *x, const void *w, void *y, int64_t N, int H, int W, void *hip_stream)
{
    if (!x || !w || !y || x == y || N <= 0 || H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH) return -1;
    const C64Plan pl = c64_plan(N, H, W, TH, TW);
    // persistent workgroups, two resident per CU: 512 or 1024 of them (multiples of 512 measured best), fewer for tiny inputs
    const unsigned nb = (unsigned)(pl.tiles >= 16384 ? CONV_BIG_GRID : pl.tiles >= 512 ? 512 : pl.tiles);
    hipLaunchKernelGGL(CONV_FWD_KERNEL<false>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, (const unsigned short *)x,
                       (const unsigned short *)w, (unsigned short *)y, (int)N, pl.Hk, pl.Wk, (const float *)nullptr,
                       (const unsigned short *)nullptr, 0, (double *)nullptr C64_GEO_ARG(pl));
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

/* number of partial rows ([2][64] float64 each) salsa_nn_conv3x3_c64_stats writes = its workgroup count */
extern "C" int salsa_nn_conv3x3_c64_stats_blocks(int64_t N, int H, int W)
{
    if (N <= 0 || H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH || !CONV_ASYNC) return 0;
    const long tiles = c64_plan(N, H, W, TH, TW).tiles;
    return (int)(tiles >= 16384 ? CONV_BIG_GRID : tiles >= 512 ? 512 : tiles);
}

// training: the plain convolution, which also leaves the per-channel sum / sum of squares of its (bf16-rounded) output as
// per-workgroup float64 partial rows stats_part[blocks][2][64] -- the BatchNorm that follows then needs no statistics pass
// over the tensor (salsa_nn_bn_train_fwd's stats_part / stats_blocks)
extern "C" int salsa_nn_conv3x3_c64_stats(const void *x, const void *w, void *y, double *stats_part, int64_t N, int H, int W,
                                          void *hip_stream)
{
    if (!x || !w || !y || !stats_part || x == y || !salsa_nn_conv3x3_c64_stats_blocks(N, H, W)) return -1;
    const unsigned nb = (unsigned)salsa_nn_conv3x3_c64_stats_blocks(N, H, W);
#if CONV_ASYNC
    const C64Plan pl = c64_plan(N, H, W, TH, TW);
    hipLaunchKernelGGL((conv3x3_c64_fwd_async_kernel<false, true>), dim3(nb), dim3(256), 0, (hipStream_t)hip_stream,
                       (const unsigned short *)x, (const unsigned short *)w, (unsigned short *)y, (int)N, pl.Hk, pl.Wk, (const float *)nullptr,
                       (const unsigned short *)nullptr, 0, stats_part, pl.geo);
#endif
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// training, round 4: y = conv(a, w) with a = dropout(relu(batch_norm(x1))) formed on the fly from the RAW output x1 of the
// convolution before (XformArgs above) + the statistics epilogue of salsa_nn_conv3x3_c64_stats for the BatchNorm that follows
extern "C" int salsa_nn_conv3x3_c64_xform_stats(const void *x1, const void *w, void *y, double *stats_part, const float *mean,
                                                const float *invstd, const float *gamma, const float *beta, float drop_p,
                                                uint32_t drop_seed, int64_t N, int H, int W, void *hip_stream)
{
    if (!x1 || !w || !y || !stats_part || x1 == y || !mean || !invstd || !gamma || !beta || drop_p < 0.f || drop_p >= 1.f ||
        !salsa_nn_conv3x3_c64_stats_blocks(N, H, W))
        return -1;
    const unsigned nb = (unsigned)salsa_nn_conv3x3_c64_stats_blocks(N, H, W);
#if CONV_ASYNC
    const C64Plan pl = c64_plan(N, H, W, TH, TW);
    const XformArgs xf = {mean, invstd, gamma, beta, drop_args(drop_p, drop_seed)};
    hipLaunchKernelGGL((conv3x3_c64_fwd_async_kernel<false, true, true>), dim3(nb), dim3(256), 0, (hipStream_t)hip_stream,
                       (const unsigned short *)x1, (const unsigned short *)w, (unsigned short *)y, (int)N, pl.Hk, pl.Wk, (const float *)nullptr,
                       (const unsigned short *)nullptr, 0, stats_part, pl.geo, xf);
#endif
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// inference: y = [relu](conv(x, w) + shift[co] [+ residual]) -- w pre-scaled by the folded BatchNorm factor gamma / sigma
extern "C" int salsa_nn_conv3x3_c64_bias_act(const void *x, const void *w, const float *shift, const void *residual, void *y,
                                             int relu, int64_t N, int H, int W, void *hip_stream)
{
    if (!x || !w || !shift || !y || x == y || N <= 0 || H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH) return -1;
    const C64Plan pl = c64_plan(N, H, W, TH, TW);
    const unsigned nb = (unsigned)(pl.tiles >= 16384 ? CONV_BIG_GRID : pl.tiles >= 512 ? 512 : pl.tiles);
#if CONV_ASYNC && CONV_RES_EARLY
    if (residual)
        hipLaunchKernelGGL((conv3x3_c64_fwd_async_kernel<false, false, false, true>), dim3(nb), dim3(256), 0, (hipStream_t)hip_stream,
                           (const unsigned short *)x, (const unsigned short *)w, (unsigned short *)y, (int)N, pl.Hk, pl.Wk, shift,
                           (const unsigned short *)residual, relu, (double *)nullptr, pl.geo);
    else
#endif
    hipLaunchKernelGGL(CONV_FWD_KERNEL<false>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, (const unsigned short *)x,
                       (const unsigned short *)w, (unsigned short *)y, (int)N, pl.Hk, pl.Wk, shift, (const unsigned short *)residual, relu,
                       (double *)nullptr C64_GEO_ARG(pl));
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// the same with the 2x2 average pool that follows fused in: y is [N][H/2][W/2][64]; H and W must be even
extern "C" int salsa_nn_conv3x3_c64_bias_act_pool(const void *x, const void *w, const float *shift, const void *residual, void *y,
                                                  int relu, int64_t N, int H, int W, void *hip_stream)
{
    if (!x || !w || !shift || !y || x == y || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || RPW != 2 ||
        N * H * W >= INT32_MAX / CH)
        return -1;
    const C64Plan pl = c64_plan(N, H, W, TH, TW);
    const unsigned nb = (unsigned)(pl.tiles >= 16384 ? CONV_BIG_GRID : pl.tiles >= 512 ? 512 : pl.tiles);
#if CONV_ASYNC && CONV_RES_EARLY
    if (residual)
        hipLaunchKernelGGL((conv3x3_c64_fwd_async_kernel<true, false, false, true>), dim3(nb), dim3(256), 0, (hipStream_t)hip_stream,
                           (const unsigned short *)x, (const unsigned short *)w, (unsigned short *)y, (int)N, pl.Hk, pl.Wk, shift,
                           (const unsigned short *)residual, relu, (double *)nullptr, pl.geo);
    else
#endif
    hipLaunchKernelGGL(CONV_FWD_KERNEL<true>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, (const unsigned short *)x,
                       (const unsigned short *)w, (unsigned short *)y, (int)N, pl.Hk, pl.Wk, shift, (const unsigned short *)residual, relu,
                       (double *)nullptr C64_GEO_ARG(pl));
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// ------------------------------------------------------------------------------------------------------------ stem layer
// The network's first convolution: Cin <= 8 feature channels (7 for SALSA) -> 64, reading the extractor's OWN output layout
// (float32, planar [N][Cin][H][W]) -- no layout change, no cast pass.  One workgroup = one 8 x 32 pixel tile: the halo tile goes
// to LDS as [pixel][8 channels] bf16 (16 B per pixel, channels Cin..7 zero), which makes a B fragment (one pixel, one tap, 8
// channels = 8 consecutive K) a single ds_read_b128.  K = 10 taps x 8 (tap 9 has zero weights) = 5 MFMA steps; the filter
// (64 x 80 bf16, prepared by the caller) sits in registers.  The layer is bound by its 64-channel output stream, so the
// optional epilogue (folded BatchNorm shift + ReLU, inference) saves a whole read + write of that tensor.
namespace {

#ifdef STEM_NO_STORE // probe: every value computed, nothing written
#define STEM_STORE_COND &&v.x == 0x12345678u && v.w == 0x9abcdef0u
#else
#define STEM_STORE_COND
#endif
constexpr int STH = 8, SHALO_H = STH + 2; // stem tile: 8 rows x TW pixels, wave w owns rows 2w, 2w+1 and all 64 output channels

// STATS (training): the launch is persistent (a workgroup walks tiles blockIdx.x, + gridDim.x, ...) and also leaves the
// per-channel sum / sum of squares of its bf16-rounded outputs as one float64 row [2][64] per workgroup for the BatchNorm that
// follows, accumulated like the 64 -> 64 kernel's (conv64_stats: DPP quad sums, lane q of a quad keeps channel group q).
#ifndef STEM_FWD_WPS
#define STEM_FWD_WPS 2 // round 5: with (256, 1) the compiler parked the accumulators in 64 AGPRs next to 160 - 208 VGPRs: one (training) / two (inference) waves per SIMD; held to 256 / 2 registers they stay in VGPRs: two / three waves, 277 -> 210 us (training, with the statistics epilogue), 0.361 -> 0.313 ms (inference sub-batch)
#endif
template <bool STATS>
__global__ __launch_bounds__(256, STEM_FWD_WPS) void conv3x3_stem_fwd_kernel(const float *__restrict__ x, const unsigned short *__restrict__ wq,
                                                               unsigned short *__restrict__ y, int N, int Cin, int H, int W, long x_batch_stride,
                                                               long x_channel_stride, const float *__restrict__ shift, int relu,
                                                               double *__restrict__ stats_part)
{
    __shared__ __attribute__((aligned(16))) unsigned short xs2[2][SHALO_H * HALO_W * 8]; // double-buffered halo tile (round 4)
    __shared__ __attribute__((aligned(16))) unsigned short ys[4 * TW * ROW]; // per wave: one output row, [pixel][ROW]
    __shared__ __attribute__((aligned(16))) float shs[64]; // the folded-BatchNorm shift (inference), read from LDS in the epilogue: as
                                                           // global loads inside the tile loop they were 16 DEPENDENT round trips per
                                                           // tile (`global_load_dwordx4; s_waitcnt vmcnt(0)` per channel group and row)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int px = lane & 31, khalf = lane >> 5;
    if (tid < 64) shs[tid] = shift ? shift[tid] : 0.f; // (the first barrier of the tile loop publishes it)
    bf16x8 af[5][2]; // [K step][co half]: filter row co = 32*mt + (lane&31), tap 2*step + khalf, its 8 channels
#pragma unroll
    for (int ks = 0; ks < 5; ks++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++) af[ks][mt] = *(const bf16x8 *)(wq + ((mt * 32 + px) * 10 + 2 * ks + khalf) * 8);
    const int tiles_w = (W + TW - 1) / TW, tiles_h = (H + STH - 1) / STH;
    const long n_tiles = (long)N * tiles_h * tiles_w;
    float rs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, rq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // Round 4.  Counters on the round-3 kernel (tools/probes/pmc_probe.sh): waves spent 65 % of their cycles waiting with the
    // VALU 25 % and the matrix pipe 8 % busy, at ~2 resident waves per SIMD.  What they waited for: __syncthreads() is a fence + a
    // barrier, i.e. `s_waitcnt vmcnt(0)` -- twice per tile every wave sat out the acknowledgement of the 32 KB it had just
    // STORED (and, at the top of a tile, the whole latency of the halo loads it had just issued).  Now: the launch is persistent
    // in both variants; the halo tile is double-buffered in LDS; the barriers are raw `s_barrier`s behind `s_waitcnt lgkmcnt(0)`
    // (LDS visibility is all they have to order); and a tile's global loads are issued TWO tiles ahead and converted into the
    // other LDS buffer between the MFMAs and the epilogue of the tile before -- the one place where everything older in the
    // memory queue (the loads themselves, one tile old, and the stores of the tile before that) has long retired, so the
    // compiler's vmcnt(0) in front of the conversion costs nothing and no wait ever follows a store.
    constexpr int SPF = (SHALO_H * HALO_W + 255) / 256; // halo pixels per thread
    float pv[SPF][8];
    bool pin[SPF];
    // Round 5 (as in the weight gradient below, where the same three changes took 14 % off an issue-bound kernel): tile coordinates
    // from cursors stepped by the grid size with carries instead of 64-bit divisions (two decodes per tile), every address = a
    // wave-uniform 64-bit base + a 32-bit offset, validity by unsigned compares joined with `&` (no nested EXEC-mask regions)
    struct Cursor { int tw, th, n; };
    const int G_ = (int)gridDim.x;
    const int dgw = G_ % tiles_w, dgh = (G_ / tiles_w) % tiles_h, dgn = G_ / (tiles_w * tiles_h);
    auto cursor_at = [&](long t) {
        Cursor c;
        c.tw = (int)(t % tiles_w); c.th = (int)((t / tiles_w) % tiles_h); c.n = (int)(t / ((long)tiles_w * tiles_h));
        return c;
    };
    auto advance = [&](Cursor &c) {
        c.tw += dgw;
        if (c.tw >= tiles_w) { c.tw -= tiles_w; c.th += 1; }
        c.th += dgh;
        if (c.th >= tiles_h) { c.th -= tiles_h; c.n += 1; }
        c.n += dgn;
    };
    const int xcs32 = (int)x_channel_stride;
    auto prefetch = [&](const Cursor &t) {
        const int tw = t.tw, th = t.th;
        const float *xn = x + (long)t.n * x_batch_stride;
#pragma unroll
        for (int j = 0; j < SPF; j++) { // neighbouring lanes = neighbouring columns: every plane's loads coalesce
            const int p = tid + j * 256;
            const int hh = p / HALO_W, ww = p - hh * HALO_W;
            const int h = th * STH + hh - 1, wcol = tw * TW + ww - 1;
            pin[j] = (p < SHALO_H * HALO_W) & ((unsigned)h < (unsigned)H) & ((unsigned)wcol < (unsigned)W);
            const unsigned o = pin[j] ? (unsigned)(h * W + wcol) : 0u;
#pragma unroll
            for (int c = 0; c < 8; c++) { // unconditional (valid) loads.  NOT `c < Cin ? c * stride : 0`: the compiler then re-used the
                const int cc = c < Cin ? c : Cin - 1; // c = 0 load for the planes beyond Cin behind a `s_waitcnt vmcnt(0)` -- a full
                pv[j][c] = xn[o + (unsigned)(cc * xcs32)]; // memory round trip in the middle of every prefetch
            }
        }
    };
    auto convert = [&](int buf) { // registers -> bf16 [pixel][8 channels] in LDS: one 16-byte write packs a pixel
#pragma unroll
        for (int j = 0; j < SPF; j++) {
            const int p = tid + j * 256;
            uint4 pk;
            pk.x = pack_bf16((pin[j] && 0 < Cin) ? pv[j][0] : 0.f, (pin[j] && 1 < Cin) ? pv[j][1] : 0.f);
            pk.y = pack_bf16((pin[j] && 2 < Cin) ? pv[j][2] : 0.f, (pin[j] && 3 < Cin) ? pv[j][3] : 0.f);
            pk.z = pack_bf16((pin[j] && 4 < Cin) ? pv[j][4] : 0.f, (pin[j] && 5 < Cin) ? pv[j][5] : 0.f);
            pk.w = pack_bf16((pin[j] && 6 < Cin) ? pv[j][6] : 0.f, (pin[j] && 7 < Cin) ? pv[j][7] : 0.f);
            if (p < SHALO_H * HALO_W) *(uint4 *)(xs2[buf] + p * 8) = pk;
        }
    };
#define STEM_BARRIER()                                       \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_s_barrier();                        \
    } while (0)
    Cursor cur = cursor_at(blockIdx.x), pf = cur; // this tile / the next tile to request
    if ((long)blockIdx.x < n_tiles) {
        prefetch(pf);
        advance(pf);
        convert(0);
        if ((long)blockIdx.x + gridDim.x < n_tiles) {
            prefetch(pf);
            advance(pf);
        }
    }
    int it = 0;
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, it++, advance(cur)) {
    STEM_BARRIER(); // tile's LDS copy is complete (every wave's part), and the other buffer's readers (the tile before) are done
    const unsigned short *xs = xs2[it & 1];
    const int tw = cur.tw, th = cur.th;
    unsigned short *yb = y + (long)cur.n * H * W * CH; // this image (wave-uniform); the stores add 32-bit offsets
    f32x16 acc[2][2];
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++) acc[rr][mt] = f32x16{};
#pragma unroll
    for (int ks = 0; ks < 5; ks++) {
        const int tap = 2 * ks + khalf < 9 ? 2 * ks + khalf : 8; // tap 9: any address, its weights are zero
        const int r = tap / 3, sx = tap - 3 * r;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const bf16x8 b = *(const bf16x8 *)(xs + ((wv * 2 + rr + r) * HALO_W + px + sx) * 8);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) acc[rr][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][mt], b, acc[rr][mt], 0, 0, 0);
        }
    }
    if (tile + gridDim.x < n_tiles) { // the next tile: its loads are one tile old -> the other LDS buffer; then the loads of the tile after
        convert((it + 1) & 1);
        if (tile + 2 * (long)gridDim.x < n_tiles) {
            prefetch(pf);
            advance(pf);
        }
    }
    // Epilogue.  D gives a lane four groups of four consecutive output channels of ONE pixel: stored directly that is 8-byte
    // pieces, 16 store instructions per 128-byte pixel.  Each wave turns its row around through a private LDS strip
    // ([pixel][64 co], rows padded to 144 bytes) instead and writes 16 bytes per lane, 8 lanes per pixel: every store
    // instruction covers 8 whole pixels = 1 KB of contiguous memory (0.62 -> 0.52 ms on an 8 x 4800 x 200 sub-batch).
    unsigned short *strip = ys + wv * (TW * ROW);
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int g = 0; g < 4; g++) { // D rows (reg&3) + 8*(reg>>2) + 4*(lane>>5): four consecutive output channels
                float v4[4] = {acc[rr][mt][4 * g], acc[rr][mt][4 * g + 1], acc[rr][mt][4 * g + 2], acc[rr][mt][4 * g + 3]};
                if (shift) {
                    const float4 sh = *(const float4 *)(shs + 32 * mt + 4 * khalf + 8 * g);
                    v4[0] += sh.x; v4[1] += sh.y; v4[2] += sh.z; v4[3] += sh.w;
                    if (relu) {
#pragma unroll
                        for (int j = 0; j < 4; j++) v4[j] = fmaxf(v4[j], 0.f);
                    }
                }
                uint2 v;
                v.x = pack_bf16(v4[0], v4[1]);
                v.y = pack_bf16(v4[2], v4[3]);
                *(uint2 *)(strip + px * ROW + 32 * mt + 8 * g + 4 * khalf) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the strip is wave-private: no barrier, only LDS order
        const int h = th * STH + wv * 2 + rr;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int p = it * 8 + (lane >> 3), piece = lane & 7;
            const uint4 v = *(const uint4 *)(strip + p * ROW + piece * 8);
            const int wcol = tw * TW + p;
            if ((h < H) & (wcol < W) STEM_STORE_COND) {
#if STEM_OUT_NT // round-5 probe: the first layer's output stores non-temporal
                typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(u4v_t{v.x, v.y, v.z, v.w}, (u4v_t *)(yb + (unsigned)((h * W + wcol) * CH + piece * 8)));
#else
                *(uint4 *)(yb + (unsigned)((h * W + wcol) * CH + piece * 8)) = v;
#endif
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads done before the next row overwrites the strip
    }
    if (STATS) { // the wave's rows 8 th + 2 wv + {0, 1} = conv64_stats' rows (4 th' + 2 rg + rr) with th' = 2 th + wv / 2, rg = wv & 1
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const f32x16 a2[2] = {acc[0][mt], acc[1][mt]};
            conv64_stats(a2, rs[mt], rq[mt], 2 * th + (wv >> 1), tw, H, W, px, wv & 1);
        }
    }
    } // tile loop
    if (STATS) {
        float *lstats = (float *)ys; // [wave][sum | sum of squares][64]: the strips are free now
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int d = 4; d <= 16; d <<= 1) {
                    rs[mt][j] += __shfl_xor(rs[mt][j], d);
                    rq[mt][j] += __shfl_xor(rq[mt][j], d);
                }
        if (px < 4) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    lstats[wv * 128 + 32 * mt + 4 * khalf + 8 * px + j] = rs[mt][j];
                    lstats[wv * 128 + 64 + 32 * mt + 4 * khalf + 8 * px + j] = rq[mt][j];
                }
        }
        __syncthreads();
        if (tid < 128)
            stats_part[(long)blockIdx.x * 128 + tid] = (double)lstats[tid] + (double)lstats[128 + tid] + (double)lstats[256 + tid] + (double)lstats[384 + tid];
    }
}

} // namespace

#ifndef STEM_FWD_GRID
#define STEM_FWD_GRID 1536
#endif
#ifndef STEM_STATS_GRID
#define STEM_STATS_GRID 1024
#endif
/* number of partial rows salsa_nn_conv3x3_stem_stats writes (= its persistent workgroup count) */
extern "C" int salsa_nn_conv3x3_stem_stats_blocks(int64_t N, int H, int W)
{
    if (N <= 0 || H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH) return 0;
    const long tiles = (long)N * ((H + STH - 1) / STH) * ((W + TW - 1) / TW);
    return (int)(tiles >= STEM_STATS_GRID ? STEM_STATS_GRID : tiles);
}

// training: the plain first-layer convolution from a persistent launch that also leaves its output's per-channel partial sums
// (float64 rows stats_part[blocks][2][64]) for the BatchNorm that follows -- see salsa_nn_conv3x3_c64_stats
extern "C" int salsa_nn_conv3x3_stem_stats(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *wq, void *y,
                                           double *stats_part, int64_t N, int Cin, int H, int W, void *hip_stream)
{
    if (!x || !wq || !y || !stats_part || N <= 0 || Cin <= 0 || Cin > 8 || H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH ||
        x_channel_stride < (int64_t)H * W || x_batch_stride < x_channel_stride * Cin ||
        x_channel_stride * Cin >= INT32_MAX /* 32-bit element offsets inside an image */)
        return -1;
    const unsigned nb = (unsigned)salsa_nn_conv3x3_stem_stats_blocks(N, H, W);
    hipLaunchKernelGGL(conv3x3_stem_fwd_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, x, (const unsigned short *)wq,
                       (unsigned short *)y, (int)N, Cin, H, W, (long)x_batch_stride, (long)x_channel_stride, (const float *)nullptr, 0,
                       stats_part);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// x float32 planar [N][Cin][H][W] (Cin <= 8; rows contiguous, batch / channel strides in elements, so a time-cropped view of
// the extractor's output needs no copy), wq bf16 [64][10][8] = w[co][tap][ci] zero-padded (tap 9 and ci >= Cin zero),
// y bf16 channels-last [N][H][W][64]; shift NULL: plain convolution, else y = [relu](conv + shift[co]) (folded BatchNorm)
extern "C" int salsa_nn_conv3x3_stem(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *wq,
                                     const float *shift, void *y, int relu, int64_t N, int Cin, int H, int W, void *hip_stream)
{
    if (!x || !wq || !y || N <= 0 || Cin <= 0 || Cin > 8 || H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH ||
        x_channel_stride < (int64_t)H * W || x_batch_stride < x_channel_stride * Cin ||
        x_channel_stride * Cin >= INT32_MAX /* 32-bit element offsets inside an image */)
        return -1;
    const long tiles = (long)N * ((H + STH - 1) / STH) * ((W + TW - 1) / TW);
    if (tiles >= INT32_MAX) return -1;
    const unsigned nb = (unsigned)(tiles >= STEM_FWD_GRID ? STEM_FWD_GRID : tiles); // persistent: six workgroups per CU
    hipLaunchKernelGGL(conv3x3_stem_fwd_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, x,
                       (const unsigned short *)wq, (unsigned short *)y, (int)N, Cin, H, W, (long)x_batch_stride, (long)x_channel_stride,
                       shift, relu, (double *)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// ------------------------------------------------------------------------------------------------------------ weight gradient
// dW[co][tap][ci] = sum over pixels p of dy[p][co] * x[p + tap][ci]: a GEMM whose reduction runs over PIXELS, so both MFMA
// operands need 8 consecutive pixels of ONE channel per lane while memory (and the LDS tiles) are pixel-major.  gfx950's
// transposing LDS read does exactly that: ds_read_b64_tr_b16 with source lane i of a 16-lane group pointing at 4
// consecutive channels of pixel row i/4 (channel block i%4) returns to lane l channel l%16 of those 4 pixel rows
// (tools/probes/ds_tr_probe.hip prints the mapping), so two reads build one operand fragment: A = dy^T (32 co x 16 pixels),
// B = x (16 pixels x 32 ci).  A persistent workgroup keeps 64 x 576 float32 partial sums in registers (wave = one co half
// x one ci half x 9 taps = 9 accumulator tiles) over all its pixel tiles and adds them to the float32 result at the end.
namespace {

constexpr int WT_H = 4, WT_W = 32;                       // pixel tile
constexpr int WHALO_W = WT_W + 2, WHALO_H = WT_H + 2;

// one operand fragment = two transposing reads (pixel rows +0..3 and +4..7); issued without waiting: LDS_TR_WAIT below
struct tr_frag {
    unsigned long long lo, hi;
};
#define LDS_TR_ISSUE(f, addr, imm)                                                                                         \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                              \
                 : "=&v"((f).lo), "=&v"((f).hi)                                                                            \
                 : "v"(addr), "n"(imm), "n"((imm) + 4 * ROW * 2))
#define LDS_TR_WAIT(f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"((f).lo), "+v"((f).hi))
__device__ __forceinline__ bf16x8 tr_value(const tr_frag &f)
{
    union { unsigned long long q[2]; bf16x8 v; } u;
    u.q[0] = f.lo;
    u.q[1] = f.hi;
    return u.v;
}


#ifndef WRW_DEPTH
#define WRW_DEPTH 2 // operand fragments requested ahead of the one being multiplied: 1: 603, 2: 676, 3: 654, 4: 635 TFLOP/s at
                    // 32 x 640 x 200 (one fragment ahead left every step waiting out most of an LDS round trip)
#endif
// waits until all but the `n` newest LDS reads have returned (they return in order), naming fragment f as now valid
#define LDS_TR_WAIT_N(f, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"((f).lo), "+v"((f).hi))

template <int Q> __device__ __forceinline__ void wrw_issue(tr_frag &f, const unsigned ga, const unsigned xa)
{
    constexpr int ks_ = Q / 10, j_ = Q % 10, rr_ = ks_ >> 1, hw_ = ks_ & 1;
    if constexpr (j_ == 0) LDS_TR_ISSUE(f, ga, 2 * ((rr_ * WT_W + 16 * hw_) * ROW));
    else LDS_TR_ISSUE(f, xa, 2 * (((rr_ + (j_ - 1) / 3) * WHALO_W + 16 * hw_ + (j_ - 1) % 3) * ROW));
}

// (A variant in which a wave owns both co halves and a group of 4 - 5 taps -- 0.72 instead of 1.11 fragment reads per MFMA --
// measured no faster, 621 - 648 TFLOP/s: the LDS read rate is not what this kernel waits for.)
template <int Q>
__device__ __forceinline__ void wrw_steps(tr_frag (&fr)[WRW_DEPTH + 1], bf16x8 &a, f32x16 (&acc)[9], const unsigned ga, const unsigned xa)
{
    constexpr int R = WRW_DEPTH + 1;
    if constexpr (Q < 80) {
        if constexpr (Q + WRW_DEPTH < 80) wrw_issue<Q + WRW_DEPTH>(fr[(Q + WRW_DEPTH) % R], ga, xa);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (Q % 10 == 0) a = tr_value(fr[Q % R]);
        else acc[Q % 10 - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_value(fr[Q % R]), acc[Q % 10 - 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (Q + 1 < 80) { // fragment Q+1 must have landed; the younger ones (two reads each) may still be in flight
            constexpr int younger = (Q + WRW_DEPTH < 80 ? WRW_DEPTH : 79 - Q) - 1;
            if constexpr (younger <= 0) LDS_TR_WAIT_N(fr[(Q + 1) % R], 0);
            else if constexpr (younger == 1) LDS_TR_WAIT_N(fr[(Q + 1) % R], 2);
            else if constexpr (younger == 2) LDS_TR_WAIT_N(fr[(Q + 1) % R], 4);
            else LDS_TR_WAIT_N(fr[(Q + 1) % R], 6);
        }
        wrw_steps<Q + 1>(fr, a, acc, ga, xa);
    }
}

// XF (round 4): x is the RAW output x1 of the convolution before and the operand is a = dropout(relu(batch_norm(x1))), formed in
// registers between the global load and the LDS write of each piece (this kernel stages through registers): see XformArgs.
template <bool XF>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_wrw_kernel(const unsigned short *__restrict__ x,
                                                                 const unsigned short *__restrict__ dy,
                                                                 float *__restrict__ dw, int N, int H, int W, const Geo geo,
                                                                 float *__restrict__ part /* deterministic mode: [gridDim.x][64*9*64] */,
                                                                 const XformArgs xf)
{
    __shared__ __attribute__((aligned(16))) unsigned short xl[WHALO_H * WHALO_W * ROW];
    __shared__ __attribute__((aligned(16))) unsigned short gl[WT_H * WT_W * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mb = wv & 1, nb = wv >> 1;                 // this wave: co 32*mb.., ci 32*nb..
    const int i16 = lane & 15, cb = (lane >> 4) & 1, kh = lane >> 5;
    // per-lane source address inside a 16-pixel x 32-channel operand block (see the comment above)
    const unsigned a_lane = 2u * (unsigned)((8 * kh + (i16 >> 2)) * ROW + 32 * mb + 16 * cb + 4 * (i16 & 3));
    const unsigned b_lane = 2u * (unsigned)((8 * kh + (i16 >> 2)) * ROW + 32 * nb + 16 * cb + 4 * (i16 & 3));
    const unsigned gbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)gl;
    const unsigned xbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)xl;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = f32x16{};
    const int tiles_w = (W + WT_W - 1) / WT_W, tiles_h = (H + WT_H - 1) / WT_H;
    constexpr int XP = (WHALO_H * WHALO_W * 8 + 255) / 256, GP = WT_H * WT_W * 8 / 256; // 16-byte pieces per thread
    uint4 px_[XP], pg_[GP];
    // Round 3: the fetch used to decode the tile with 64-bit divisions and rebuild a 64-bit address with a division by 34 for
    // every one of its 11 loads -- ~45 instructions per load, as many issue cycles per tile as the 72 MFMAs (probe builds at
    // 32 x 640 x 200: 0.445 ms, 0.252 without the fetch, 0.242 without the multiply, 0.114 with neither: purely additive).
    // Now a cursor steps (tw, th, n) by the grid size with carries, the tile's origin is ONE wave-uniform 64-bit pointer, and a
    // load is that pointer + a 32-bit offset from the thread's halo position (q / 34 by multiply-shift, exact for q < 236).
    struct Cursor { int tw, th, n; };
    const TileWalk walk = tile_walk(N, tiles_h, tiles_w, (int)gridDim.x, (int)blockIdx.x); // XCD-aware order (see tile_walk)
    const int stride_ = walk.G;
    auto cursor_of = [&](const TilePos &p) {
        Cursor c;
        c.tw = p.tw;
        tile_coords(walk, p, c.n, c.th);
        return c;
    };
    static_assert(WHALO_W == 34 && WHALO_H * WHALO_W + 31 < 236, "q / 34 == (q * 241) >> 13 holds for q < 236");
    auto fetch = [&](const Cursor &c) { // one tile's x (with halo, zeros outside the image) and dy into registers
        const int h0 = c.th * WT_H, w0 = c.tw * WT_W;
        const unsigned short *xo = x + geo_off(geo, c.n, h0 - 1, w0 - 1);   // halo pixel (0, 0); wave-uniform
        const unsigned short *go = dy + geo_off(geo, c.n, h0, w0);
        const int p0 = tid >> 3, piece8 = (tid & 7) * 8;
#pragma unroll
        for (int j = 0; j < XP; j++) {
            const int q = p0 + 32 * j;                   // halo pixel of this piece, row-major in the 6 x 34 halo
            const int hh = (q * 241) >> 13, ww = q - hh * WHALO_W;
            const bool ok = q < WHALO_H * WHALO_W && (unsigned)(h0 - 1 + hh) < (unsigned)H && (unsigned)(w0 - 1 + ww) < (unsigned)W;
            px_[j] = make_uint4(0u, 0u, 0u, 0u);
#ifndef WRW64_NO_FETCH // (probe builds: WRW64_NO_FETCH / WRW64_NO_LDSWRITE / WRW64_NO_MULT drop one phase each)
            if (ok) px_[j] = *(const uint4 *)(xo + (hh * geo.sh + ww * geo.sw + piece8));
#endif
        }
#pragma unroll
        for (int j = 0; j < GP; j++) {
            const int q = p0 + 32 * j, hh = q >> 5, ww = q & 31;
            pg_[j] = make_uint4(0u, 0u, 0u, 0u);
#ifndef WRW64_NO_FETCH
            if (h0 + hh < H && w0 + ww < W) pg_[j] = *(const uint4 *)(go + (hh * geo.sh + ww * geo.sw + piece8));
#endif
        }
    };
    const int n_local = walk.n_local;                // tiles of this workgroup's XCD; `tile` = position in that sequence
    int tile = (int)blockIdx.x / walk.nx;
    TilePos pcur = tile_pos(walk, tile);
    __shared__ __attribute__((aligned(16))) float xtab[XF ? 2 * 64 : 4]; // XF: scale[64], shift[64]
    if (XF) {
        if (tid < 64) {
            const float sc = xf.invstd[tid] * xf.gamma[tid];
            xtab[tid] = sc;
            xtab[64 + tid] = xf.beta[tid] - xf.mean[tid] * sc;
        }
    } // (the loop's first __syncthreads() publishes the table)
    if (tile < n_local) fetch(cursor_of(pcur));
    for (; tile < n_local; tile += stride_) {
        __syncthreads(); // the previous tile's LDS reads are done
#ifndef WRW64_NO_LDSWRITE
        if (XF) { // a = dropout(relu(x1 * scale + shift)) on the pieces inside the image (outside: the zero padding of the ACTIVATION)
            const Cursor c = cursor_of(pcur);
            const int h0 = c.th * WT_H, w0 = c.tw * WT_W;
            const long origin = geo_off(geo, c.n, h0 - 1, w0 - 1);
            const int p0 = tid >> 3, piece8 = (tid & 7) * 8;
            const float4 s0 = *(const float4 *)(xtab + piece8), s1 = *(const float4 *)(xtab + piece8 + 4);
            const float4 t0 = *(const float4 *)(xtab + 64 + piece8), t1 = *(const float4 *)(xtab + 64 + piece8 + 4);
            const float xsc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, xsh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int j = 0; j < XP; j++) {
                const int q = p0 + 32 * j;
                const int hh = (q * 241) >> 13, ww = q - hh * WHALO_W;
                const bool ok = q < WHALO_H * WHALO_W && (unsigned)(h0 - 1 + hh) < (unsigned)H && (unsigned)(w0 - 1 + ww) < (unsigned)W;
                if (ok) {
                    const long e0 = origin + (long)hh * geo.sh + (long)ww * geo.sw + piece8;
                    const unsigned in[4] = {px_[j].x, px_[j].y, px_[j].z, px_[j].w};
                    unsigned out[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float a = fmaxf(__uint_as_float(in[k] << 16) * xsc[2 * k] + xsh[2 * k], 0.f);
                        float b = fmaxf(__uint_as_float(in[k] & 0xffff0000u) * xsc[2 * k + 1] + xsh[2 * k + 1], 0.f);
                        if (xf.drop.thresh) {
                            const unsigned hsh = drop_hash((unsigned)(e0 >> 1) + k, xf.drop.seed);
                            a = (hsh & 0xFFFFu) >= xf.drop.thresh ? a * xf.drop.scale : 0.f;
                            b = (hsh >> 16) >= xf.drop.thresh ? b * xf.drop.scale : 0.f;
                        }
                        out[k] = pack_bf16(a, b);
                    }
                    px_[j] = make_uint4(out[0], out[1], out[2], out[3]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < XP; j++) {
            const int i = tid + j * 256;
            if (i < WHALO_H * WHALO_W * 8) *(uint4 *)(xl + (long)(i >> 3) * ROW + (i & 7) * 8) = px_[j];
        }
#pragma unroll
        for (int j = 0; j < GP; j++) {
            const int i = tid + j * 256;
            *(uint4 *)(gl + (long)(i >> 3) * ROW + (i & 7) * 8) = pg_[j];
        }
#endif
        __syncthreads();
        tile_advance(walk, pcur);
        if (tile + stride_ < n_local) fetch(cursor_of(pcur)); // in flight during the multiply below
        // 8 K-steps (4 rows x 2 halves of 16 consecutive pixels) x 9 taps, software-pipelined by hand like the forward kernel:
        // the fragment of step q+1 is requested before step q's MFMA issues.  Step q = ks * 10 + j: j = 0 is the dy
        // fragment of K-step ks, j = 1..9 the x fragment of tap j-1.
        const unsigned ga = gbase + a_lane, xa = xbase + b_lane;
        tr_frag fr[WRW_DEPTH + 1];
        bf16x8 a;
        wrw_issue<0>(fr[0], ga, xa);
        if constexpr (WRW_DEPTH >= 2) wrw_issue<1>(fr[1], ga, xa);
        if constexpr (WRW_DEPTH >= 3) wrw_issue<2>(fr[2], ga, xa);
        if constexpr (WRW_DEPTH >= 4) wrw_issue<3>(fr[3], ga, xa);
        if constexpr (WRW_DEPTH == 1) LDS_TR_WAIT_N(fr[0], 0);
        else if constexpr (WRW_DEPTH == 2) LDS_TR_WAIT_N(fr[0], 2);
        else if constexpr (WRW_DEPTH == 3) LDS_TR_WAIT_N(fr[0], 4);
        else LDS_TR_WAIT_N(fr[0], 6);
#ifndef WRW64_NO_MULT
        wrw_steps<0>(fr, a, acc, ga, xa);
#else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    }
    // D[m = co][n = ci]: column = lane&31 = ci, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = co within the block
#ifndef WRW64_NO_ATOMIC // (probe)
    if (part) {
        // deterministic mode: the workgroup's 64 x 9 x 64 block goes to its slab.  Written as the accumulators lie that is 144
        // scattered 4-byte stores per lane (the wide kernel's probe builds: a quarter of the launch); instead each tap's 64 x 64
        // plane is turned through LDS (the x tile's buffer is free) into its memory layout and written as 16-byte stores,
        // whole 256-byte rows (round 4)
        float *blk = (float *)xl; // [64 co][64 ci]
        static_assert(sizeof(xl) >= 64 * 64 * 4, "one tap's plane");
        float *out = part + (long)blockIdx.x * (64L * 9 * 64);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            __syncthreads();
#pragma unroll
            for (int reg = 0; reg < 16; reg++)
                blk[(32 * mb + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * 64 + 32 * nb + (lane & 31)] = acc[tap][reg];
            __syncthreads();
            const int ft = geo.tap_t ? (tap % 3) * 3 + tap / 3 : tap; // (transposed geometry: tap (ky, kx) is the filter's (kx, ky))
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int co = 16 * r + (tid >> 4), seg = tid & 15;
                *(float4 *)(out + ((long)(co * 9 + ft)) * CH + seg * 4) = *(const float4 *)(blk + co * 64 + seg * 4);
            }
        }
        return;
    }
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int co = 32 * mb + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), ci = 32 * nb + (lane & 31);
            atomicAdd(dw + ((long)(co * 9 + (geo.tap_t ? (tap % 3) * 3 + tap / 3 : tap)) * CH + ci), acc[tap][reg]);
        }
#endif
}

} // namespace

#ifndef C64_WRW_MID
#define C64_WRW_MID 256 // persistent workgroups of the weight gradient at 4096 .. 16383 tiles, and (C64_WRW_BIG) beyond; with slabs (round 4) at
                        // 32 x 320 x 100: 256: 99 us, 384: 105, 512: 101 (the launch + its slab reduction); with atomics 384 was best
#endif
#ifndef C64_WRW_BIG
#define C64_WRW_BIG 512
#endif
// dw: float32 [64 co][3][3][64 ci], ADDED to (zero it first); x, dy: [N][H][W][64] bf16
extern "C" int salsa_nn_conv3x3_c64_wrw(const void *x, const void *dy, float *dw, int64_t N, int H, int W, void *hip_stream)
{
    if (!x || !dy || !dw || N <= 0 || H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH) return -1;
    const C64Plan pl = c64_plan(N, H, W, WT_H, WT_W);
    const long tiles = pl.tiles;
    // persistent workgroups, two per CU; fewer when there are few tiles (every workgroup ends with 36 864 float atomics)
    // (mid sizes, 32 x 320 x 100: 256 workgroups 0.158 ms, 384: 0.141, 512: 0.149)
    const unsigned nb = (unsigned)(tiles >= 16384 ? C64_WRW_BIG : tiles >= 4096 ? C64_WRW_MID : tiles >= 128 ? 128 : tiles);
    int rc = 0;
    float *part = salsa_nn_det_begin((int)nb, 64L * 9 * 64, (hipStream_t)hip_stream, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(conv3x3_c64_wrw_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, (const unsigned short *)x,
                       (const unsigned short *)dy, dw, (int)N, pl.Hk, pl.Wk, pl.geo, part, XformArgs{});
    if (part) return salsa_nn_det_finish(part, (int)nb, 64L * 9 * 64, dw, (hipStream_t)hip_stream);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// The same with x = the RAW output x1 of the convolution before and the operand a = dropout(relu(batch_norm(x1))) formed on the
// fly (mean / invstd: the BatchNorm's batch statistics, salsa_nn_bn_train_finalize; drop_p / drop_seed as salsa_nn_bn_train_fwd).
extern "C" int salsa_nn_conv3x3_c64_wrw_xform(const void *x1, const void *dy, float *dw, const float *mean, const float *invstd,
                                              const float *gamma, const float *beta, float drop_p, uint32_t drop_seed, int64_t N, int H,
                                              int W, void *hip_stream)
{
    if (!x1 || !dy || !dw || !mean || !invstd || !gamma || !beta || drop_p < 0.f || drop_p >= 1.f || N <= 0 || H <= 0 || W <= 0 ||
        N * H * W >= INT32_MAX / CH)
        return -1;
    const C64Plan pl = c64_plan(N, H, W, WT_H, WT_W);
    const long tiles = pl.tiles;
    const unsigned nb = (unsigned)(tiles >= 16384 ? C64_WRW_BIG : tiles >= 4096 ? C64_WRW_MID : tiles >= 128 ? 128 : tiles);
    int rc = 0;
    float *part = salsa_nn_det_begin((int)nb, 64L * 9 * 64, (hipStream_t)hip_stream, &rc);
    if (rc) return rc;
    const XformArgs xf = {mean, invstd, gamma, beta, drop_args(drop_p, drop_seed)};
    hipLaunchKernelGGL(conv3x3_c64_wrw_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, (const unsigned short *)x1,
                       (const unsigned short *)dy, dw, (int)N, pl.Hk, pl.Wk, pl.geo, part, xf);
    if (part) return salsa_nn_det_finish(part, (int)nb, 64L * 9 * 64, dw, (hipStream_t)hip_stream);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// ------------------------------------------------------------------------------------------ weight gradient, first layer
// dW[co][ci][tap] = sum_p dy[p][co] * x[ci][p + tap] for the (Cin <= 7) -> 64 first layer: the same pixel-reduction GEMM as
// above with N = Cin*9 <= 63 columns (one 64-column block, column c = ci*9 + tap; column 63 multiplies a zero plane).  The A
// operand (dy^T) comes from the pixel-major dy tile through transposing LDS reads exactly as in the 64 -> 64 kernel.  The B
// operand needs, per lane, 8 consecutive pixels of ONE (ci, tap) -- and x is float32 PLANAR (the extractor's layout: no bf16
// channels-last copy of the input is made for this kernel), so consecutive pixels ARE consecutive floats: the tile's halo is
// converted to bf16 once and stored in LDS as three column-shifted copies per (ci, halo row), which makes every fragment one
// aligned 16-byte read.  33.5 GFLOP against 0.64 GB of input: HBM-bound; a wave owns one 32 x 32 block of dW.
namespace {

constexpr int SW_XROW = 40;                               // bf16 per (ci, halo row, shift) row: 32 + pad (80-byte pitch)
constexpr int SW_XS = 8 * WHALO_H * 3 * SW_XROW;          // 8 channel planes (those >= Cin stay zero)
constexpr int SW_XPF = (7 * WHALO_H * WHALO_W + 255) / 256; // float32 halo elements per thread

template <int KS>
__device__ __forceinline__ void stem_wrw_steps(f32x16 &acc, const unsigned ga, const unsigned xa)
{
    if constexpr (KS < 8) {
        constexpr int rr = KS >> 1, hw = KS & 1;
        tr_frag fa;
        bf16x8 b;
        LDS_TR_ISSUE(fa, ga, 2 * ((rr * WT_W + 16 * hw) * ROW));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(xa), "n"(2 * (rr * 3 * SW_XROW + 16 * hw)));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa.lo), "+v"(fa.hi), "+v"(b));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_value(fa), b, acc, 0, 0, 0);
        stem_wrw_steps<KS + 1>(acc, ga, xa);
    }
}

// MODE 2 (see the kernel): the same input fragment against the masked-gradient tile and the normalised-activation tile; its eight
// values also go into this lane's share of S0 (v_dot2_f32_bf16 with a pair of ones: one register, where a third product against a
// fragment of ones would cost sixteen)
// S0 must leave out the OUTPUT pixels of a tile that hang over the image's right / bottom edge by itself (G and Xh see g = xh = 0
// there, but those pixels' input patches next to the edge are real): om[hw][q] = the pair of ones for pixel pair q of the lane's eight
// pixels in column half hw (zero where the pixel is outside), rows outside (rr >= h_left) are dropped per row.
template <int KS>
__device__ __forceinline__ void stem_wrw_steps2(f32x16 &acc_g, f32x16 &acc_x, float &s0, float &t, const unsigned ga, const unsigned ga2,
                                                const unsigned xa, const unsigned (&om)[2][4], const int h_left)
{
    if constexpr (KS < 8) {
        constexpr int rr = KS >> 1, hw = KS & 1;
        tr_frag fa, fx;
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t b;
        LDS_TR_ISSUE(fa, ga, 2 * ((rr * WT_W + 16 * hw) * ROW));
        LDS_TR_ISSUE(fx, ga2, 2 * ((rr * WT_W + 16 * hw) * ROW));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(xa), "n"(2 * (rr * 3 * SW_XROW + 16 * hw)));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa.lo), "+v"(fa.hi), "+v"(fx.lo), "+v"(fx.hi), "+v"(b));
        const bf16x8 bv = __builtin_bit_cast(bf16x8, b);
        acc_g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_value(fa), bv, acc_g, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_value(fx), bv, acc_x, 0, 0, 0);
        // (one asm statement with its own trailing wait states: a dot product's result needs them before an ordinary vector
        // instruction may read it, and the compiler's hazard recognizer cannot see into asm statements -- four separate ones let the
        // add below read t too early; the builtin form was worse: the compiler then read the fragment's registers after re-using them)
        if (hw == 0) t = 0.f;
        asm volatile("v_dot2_f32_bf16 %0, %1, %5, %0\n\tv_dot2_f32_bf16 %0, %2, %6, %0\n\tv_dot2_f32_bf16 %0, %3, %7, %0\n\t"
                     "v_dot2_f32_bf16 %0, %4, %8, %0\n\ts_nop 4"
                     : "+v"(t)
                     : "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(om[hw][0]), "v"(om[hw][1]), "v"(om[hw][2]), "v"(om[hw][3]));
        if (hw == 1) s0 += rr < h_left ? t : 0.f;
        stem_wrw_steps2<KS + 1>(acc_g, acc_x, s0, t, ga, ga2, xa, om, h_left);
    }
}
struct StemBnf { const float *mean, *invstd, *gamma, *beta; }; // MODE 2: the forward's statistics and the affine parameters [64]
constexpr int SW_SLAB = 2 * 64 * 64 + 3 * 64; // MODE 2: a workgroup's partial sums: G[64][64] (column 63 = dbeta), Xh[64][64], S0[2][64], dgamma[64]

// BN: dy is not the gradient of the convolution's output but of the BatchNorm (+ ReLU) output behind it, and x1 that
// BatchNorm's input (= this convolution's output): the tile's dx = a (g masked - b - (x1 - mean) k) -- the BatchNorm backward's
// apply pass, coefficients from salsa_nn_bn_bwd(dx = NULL) -- is formed while the tile goes into LDS, rounded to bf16 exactly
// as the separate pass would have stored it.  This layer's weight gradient is dx's ONLY reader (the network input needs no
// gradient), so the 524-MB dx is never written or read: 4 tensor passes become 2.
// MODE 2 (round 5): the BatchNorm backward's REDUCTION pass folded in as well.  dx = a (g - b - xh k') is linear in the two totals
// b = mean(g), k' = mean(g xh) that are only known after a pass over (g, x1) -- the pass this kernel makes anyway.  So it multiplies
// the input patches with the masked gradient g AND with the normalised activation xh = (x1 - mean) invstd,
//     G[co][c] = sum_p g[p][co] patch[p][c],   Xh[co][c] = sum_p xh[p][co] patch[p][c],   S0[c] = sum_p patch[p][c]  (v_dot2 on the side),
// gets dbeta = sum g as G's column 63 (the spare eighth input plane holds ones) and adds up dgamma = sum g xh on the way; a small
// launch afterwards combines the workgroups' partial sums:
//     dW[co][c] = a[co] (G[co][c] - b[co] S0[c] - k'[co] Xh[co][c]),   a = gamma invstd.
// The salsa_nn_bn_bwd(dx = NULL) launch (a second read of the 524-MB g and x1, ~145 us) disappears; the operands are g (exact in
// bf16) and bf16(xh) where MODE 1 rounds dx to bf16: the same order of rounding error.
template <int MODE>
#ifndef STEM_WRW_WPS
#define STEM_WRW_WPS 1
#endif
__global__ __launch_bounds__(256, MODE == 2 ? 2 : STEM_WRW_WPS) void conv3x3_stem_wrw_kernel(const float *__restrict__ x, long xbs, long xcs,
                                                               const unsigned short *__restrict__ dy, float *__restrict__ dw,
                                                               int N, int Cin, int H, int W,
                                                               const unsigned short *__restrict__ x1, const float *__restrict__ coef,
                                                               int relu, float *__restrict__ part /* deterministic mode: [gridDim.x][64*Cin*9]; MODE 2: [gridDim.x][SW_SLAB] */,
                                                               const StemBnf bnf)
{
    constexpr bool BN = MODE != 0;
    // (8 spare elements in front: the halo conversion stores every element into all three shifted copies UNCONDITIONALLY -- columns
    // -2, -1, 32, 33 land in the 8 padding columns of this row or of the one before, which nothing reads; behind `0 <= col < 32`
    // each of the 18 stores was an EXEC-mask region of its own)
    __shared__ __attribute__((aligned(16))) unsigned short xs_raw[SW_XS + 8];
    unsigned short *const xs = xs_raw + 8;
    __shared__ __attribute__((aligned(16))) unsigned short gl[WT_H * WT_W * ROW];
    __shared__ __attribute__((aligned(16))) unsigned short gl2[MODE == 2 ? WT_H * WT_W * ROW : 8]; // MODE 2: the xh tile

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mb = wv & 1, nb = wv >> 1; // this wave: co 32*mb.., columns 32*nb..
    const int i16 = lane & 15, cb = (lane >> 4) & 1, kh = lane >> 5;
    const unsigned a_lane = 2u * (unsigned)((8 * kh + (i16 >> 2)) * ROW + 32 * mb + 16 * cb + 4 * (i16 & 3));
    const int c = 32 * nb + (lane & 31), ci = c / 9, tap = c - 9 * ci; // column 63: ci = 7, a zero plane (Cin <= 7)
    const unsigned b_lane = 2u * (unsigned)(((ci * WHALO_H + tap / 3) * 3 + tap % 3) * SW_XROW + 8 * kh);
    const unsigned ga = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)gl + a_lane;
    const unsigned xa = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)xs + b_lane;
    const unsigned ga2 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)gl2 + a_lane;
    for (int i = tid; i < SW_XS / 8; i += 256) ((uint4 *)xs)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (MODE == 2) { // the spare eighth input plane (column 63 = its tap 0) holds ONES: G[co][63] = sum_p g[p][co] = dbeta
        __syncthreads();
        for (int i = tid; i < WHALO_H * 3 * SW_XROW; i += 256) xs[7 * WHALO_H * 3 * SW_XROW + i] = 0x3F80;
    }
    f32x16 acc = f32x16{}, acc_x = f32x16{};
    float s0 = 0.f;  // MODE 2: this lane's share of S0[c] (its half of the k-steps' pixels)
    float s_dg[8];   // MODE 2: this thread's share of dgamma (its pieces always cover the same 8 channels)
#pragma unroll
    for (int e = 0; e < 8; e++) s_dg[e] = 0.f;
    const int tiles_w = (W + WT_W - 1) / WT_W, tiles_h = (H + WT_H - 1) / WT_H;
    const long n_tiles = (long)N * tiles_h * tiles_w;
    constexpr int GP = WT_H * WT_W * 8 / 256;
    const int n_x = Cin * WHALO_H * WHALO_W;
    // Round 5: TWO register sets of tile loads in flight (the tile after next is fetched while this one is converted and
    // multiplied) and raw barriers.  Before: one set, fetched after the second __syncthreads() of a tile and needed right after the
    // first one of the next -- a fence + barrier = s_waitcnt vmcnt(0), so every tile exposed a whole memory latency minus eight
    // MFMAs, at two resident workgroups per CU (185 registers).  The second set costs 38 registers the kernel had spare at that
    // occupancy.  Every load is unconditional (a clamped in-bounds address; validity bits applied at conversion), so that nothing
    // but loads sits between the loads of a set.
    struct TileRegs {
        float px[SW_XPF];
        uint4 pg[GP], pq[BN ? GP : 1];
        unsigned okx, okg; // validity bits of the halo elements / the tile pieces
        int th, tw;
    };
    // Tile coordinates by a cursor that steps by the grid size with carries (decoding every tile index with 64-bit divisions --
    // tile % tiles_w, (tile / tiles_w) % tiles_h, tile / (tiles_w tiles_h) -- was ~230 scalar instructions per tile on one dependent
    // chain).  Past the end the cursor stays on the last tile (fetched, never used).
    struct Cursor { int tw, th, n; long t; };
    const long G = gridDim.x, t_last = n_tiles - 1;
    const int dgw = (int)(G % tiles_w), dgh = (int)((G / tiles_w) % tiles_h), dgn = (int)(G / ((long)tiles_w * tiles_h));
    auto cursor_at = [&](long t) {
        Cursor c;
        c.t = t; c.tw = (int)(t % tiles_w); c.th = (int)((t / tiles_w) % tiles_h); c.n = (int)(t / ((long)tiles_w * tiles_h));
        return c;
    };
    const Cursor c_last = cursor_at(t_last);
    auto advance = [&](Cursor &c) { // + G, clamped to the last tile
        c.t += G;
        c.tw += dgw;
        if (c.tw >= tiles_w) { c.tw -= tiles_w; c.th += 1; }
        c.th += dgh;
        if (c.th >= tiles_h) { c.th -= tiles_h; c.n += 1; }
        c.n += dgn;
        if (c.t >= t_last) c = c_last;
    };
    auto fetch = [&](const Cursor &tile, TileRegs &r) {
        const int tw = tile.tw, th = tile.th;
        const long n = tile.n;
        const int h0 = th * WT_H, w0 = tw * WT_W;
        r.th = th; r.tw = tw; r.okx = 0u; r.okg = 0u;
        // one wave-uniform 64-bit base per tensor and image + a 32-bit offset per load, validity by unsigned compares joined with `&`:
        // written with `&&` and 64-bit element offsets every load was three nested EXEC-mask regions around three 64-bit multiply-adds
        const float *xb = x + n * xbs;
        const long pix0 = n * H * W * CH;
        const unsigned short *gb = dy + pix0, *qb = BN ? x1 + pix0 : nullptr;
        const int xcs32 = (int)xcs;
#pragma unroll
        for (int j = 0; j < SW_XPF; j++) {
            const int i = tid + j * 256, cc = i / (WHALO_H * WHALO_W), rr = i - cc * (WHALO_H * WHALO_W);
            const int hh = rr / WHALO_W, ww = rr - hh * WHALO_W;
            const int h = h0 + hh - 1, wc = w0 + ww - 1;
            const bool ok = (i < n_x) & ((unsigned)h < (unsigned)H) & ((unsigned)wc < (unsigned)W);
            r.okx |= (ok ? 1u : 0u) << j;
            r.px[j] = xb[ok ? (unsigned)(cc * xcs32 + h * W + wc) : 0u];
        }
#pragma unroll
        for (int j = 0; j < GP; j++) {
            const int i = tid + j * 256, piece = i & 7, p = i >> 3;
            const int hh = p / WT_W, ww = p - hh * WT_W;
            const int h = h0 + hh, wc = w0 + ww;
            const bool ok = (h < H) & (wc < W);
            r.okg |= (ok ? 1u : 0u) << j;
            const unsigned off = (ok ? (unsigned)((h * W + wc) * CH) : 0u) + (unsigned)(piece * 8);
            r.pg[j] = *(const uint4 *)(gb + off);
            if (BN) r.pq[j] = *(const uint4 *)(qb + off);
        }
    };
    // one bf16 pair of the tile: BatchNorm-backward apply on (g, x1) -> dx, rounded to bf16 (a pixel outside the image has
    // g = x1 = 0 and would give -a (b - mean k): the caller zeroes those pieces instead).  A thread's pieces always cover the
    // same 8 channels (piece index = tid & 7), so their 56 coefficients live in registers (a table in LDS cost seven LDS reads
    // per element: 0.43 ms for this kernel); the arithmetic is bn_bwd_apply_kernel's, operation for operation.
    float c_a[8], c_b[8], c_mu[8], c_k[8], c_is[8], c_be[8], c_ga[8];
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int ch = (tid & 7) * 8 + e;
            c_a[e] = coef[ch]; c_b[e] = coef[CH + ch]; c_mu[e] = coef[2 * CH + ch]; c_k[e] = coef[3 * CH + ch];
            c_is[e] = coef[4 * CH + ch]; c_be[e] = coef[5 * CH + ch]; c_ga[e] = coef[6 * CH + ch];
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int ch = (tid & 7) * 8 + e;
            c_mu[e] = bnf.mean[ch]; c_is[e] = bnf.invstd[ch]; c_ga[e] = bnf.gamma[ch]; c_be[e] = bnf.beta[ch];
            c_a[e] = c_b[e] = c_k[e] = 0.f;
        }
    }
    // MODE 2: one bf16 pair of the tile -> the masked gradient pair (returned) and the normalised-activation pair (xh2), both as the
    // matrix operands; dgamma from the unrounded float32 values, as bn_bwd_reduce_kernel forms it
    auto bnf_pair = [&](const unsigned g2, const unsigned q2, const int e0, const bool inside, unsigned &xh2, float &dg0, float &dg1) -> unsigned {
        const float g0 = __uint_as_float(g2 << 16), g1 = __uint_as_float(g2 & 0xffff0000u);
        const float x0 = __uint_as_float(q2 << 16), x1v = __uint_as_float(q2 & 0xffff0000u);
        const float xh0 = (x0 - c_mu[e0]) * c_is[e0], xh1 = (x1v - c_mu[e0 + 1]) * c_is[e0 + 1];
        const bool live0 = inside & !(relu && !(xh0 * c_ga[e0] + c_be[e0] > 0.f));
        const bool live1 = inside & !(relu && !(xh1 * c_ga[e0 + 1] + c_be[e0 + 1] > 0.f));
        const float gk0 = live0 ? g0 : 0.f, gk1 = live1 ? g1 : 0.f;
        asm("v_fmac_f32 %0, %1, %2" : "+v"(dg0) : "v"(gk0), "v"(xh0)); // (written out: left to the compiler the eight accumulators
        asm("v_fmac_f32 %0, %1, %2" : "+v"(dg1) : "v"(gk1), "v"(xh1)); //  cost the kernel ~100 registers -- 353 against 245)
        xh2 = pack_bf16(inside ? xh0 : 0.f, inside ? xh1 : 0.f);
        return pack_bf16(gk0, gk1); // (exact: g is a bf16 value or zero)
    };
    auto bn_pair = [&](const unsigned g2, const unsigned q2, const int e0) -> unsigned {
        float r[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float g = __uint_as_float(e ? (g2 & 0xffff0000u) : (g2 << 16)), xv = __uint_as_float(e ? (q2 & 0xffff0000u) : (q2 << 16));
            const int ch = e0 + e;
            const float xc = xv - c_mu[ch];
            const bool zero = relu && !((xc * c_is[ch]) * c_ga[ch] + c_be[ch] > 0.f);
            r[e] = c_a[ch] * ((zero ? 0.f : g) - c_b[ch] - xc * c_k[ch]);
        }
        return pack_bf16(r[0], r[1]);
    };
    auto raw_barrier = [&]() { // LDS order only: the loads in flight stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // registers -> LDS: the halo as three column-shifted bf16 copies per (ci, row); the (g [, x1]) tile as dx rows
    auto convert = [&](const TileRegs &r) {
#pragma unroll
        for (int j = 0; j < SW_XPF; j++) {
            const int i = tid + j * 256, cc = i / (WHALO_H * WHALO_W), rr = i - cc * (WHALO_H * WHALO_W);
            const int hh = rr / WHALO_W, ww = rr - hh * WHALO_W;
            if (i < n_x) {
                const unsigned short bits = (unsigned short)(pack_bf16(((r.okx >> j) & 1u) ? r.px[j] : 0.f, 0.f) & 0xffffu); // round to nearest even
#pragma unroll
                for (int kx = 0; kx < 3; kx++) { // copy kx holds halo columns kx .. kx + 31 at positions 0 .. 31
                    xs[((cc * WHALO_H + hh) * 3 + kx) * SW_XROW + ww - kx] = bits;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GP; j++) {
            const int i = tid + j * 256;
            const bool inside = (r.okg >> j) & 1u;
            uint4 v = r.pg[j];
            if (MODE == 1) {
                v.x = bn_pair(r.pg[j].x, r.pq[j].x, 0);
                v.y = bn_pair(r.pg[j].y, r.pq[j].y, 2);
                v.z = bn_pair(r.pg[j].z, r.pq[j].z, 4);
                v.w = bn_pair(r.pg[j].w, r.pq[j].w, 6);
            }
            if (MODE == 2) {
                uint4 u;
                v.x = bnf_pair(r.pg[j].x, r.pq[j].x, 0, inside, u.x, s_dg[0], s_dg[1]);
                v.y = bnf_pair(r.pg[j].y, r.pq[j].y, 2, inside, u.y, s_dg[2], s_dg[3]);
                v.z = bnf_pair(r.pg[j].z, r.pq[j].z, 4, inside, u.z, s_dg[4], s_dg[5]);
                v.w = bnf_pair(r.pg[j].w, r.pq[j].w, 6, inside, u.w, s_dg[6], s_dg[7]);
                *(uint4 *)(gl2 + (long)(i >> 3) * ROW + (i & 7) * 8) = u;
            } else if (!inside) v = make_uint4(0u, 0u, 0u, 0u);
            *(uint4 *)(gl + (long)(i >> 3) * ROW + (i & 7) * 8) = v;
        }
    };
    auto multiply = [&](const int th, const int tw) __attribute__((always_inline)) {
        if constexpr (MODE == 2) {
            const int h_left = H - th * WT_H, w_left = W - tw * WT_W;
            unsigned om[2][4];
#pragma unroll
            for (int hw = 0; hw < 2; hw++) {
                const int nv = w_left - (16 * hw + 8 * kh); // valid pixels among this lane's eight (<= 0: none, >= 8: all)
#pragma unroll
                for (int q = 0; q < 4; q++) om[hw][q] = (2 * q < nv ? 0x3F80u : 0u) | (2 * q + 1 < nv ? 0x3F800000u : 0u);
            }
            float t = 0.f;
            stem_wrw_steps2<0>(acc, acc_x, s0, t, ga, ga2, xa, om, h_left);
        } else stem_wrw_steps<0>(acc, ga, xa);
    };
    TileRegs ra, rb;
    long tile = blockIdx.x; // (< n_tiles: the launch has at most one workgroup per tile)
    // Every fetch is issued unconditionally (past the end: the last tile again, never converted): with a fetch or a tile behind a
    // branch the compiler's wait insertion merges the paths and drains BOTH sets (vmcnt 13 .. 0) where the older set alone
    // is needed (vmcnt 27 .. 14)
    Cursor nf = cursor_at(tile); // the next tile to request
    fetch(nf, ra);
    advance(nf);
    fetch(nf, rb);
    advance(nf);
    __syncthreads(); // the zero fill of xs (and nothing else: the loads above land in registers)
    for (; tile + G < n_tiles; tile += 2 * G) {
        raw_barrier(); // the previous tile's LDS reads are done
        convert(ra);
        raw_barrier();
        const int tha = ra.th, twa = ra.tw;
        fetch(nf, ra); // (tile + 2 G) in flight during this tile's multiply and the whole next tile
        advance(nf);
        multiply(tha, twa);
        raw_barrier();
        convert(rb);
        raw_barrier();
        const int thb = rb.th, twb = rb.tw;
        fetch(nf, rb); // (tile + 3 G)
        advance(nf);
        multiply(thb, twb);
    }
    if (tile < n_tiles) {
        raw_barrier();
        convert(ra);
        raw_barrier();
        multiply(ra.th, ra.tw);
    }
    if (MODE == 2) { // this workgroup's slab: G (column 63: dbeta), Xh, S0 (the two lane halves), dgamma
        float *slab = part + (long)blockIdx.x * SW_SLAB;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int co = 32 * mb + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            slab[co * 64 + c] = acc[reg];
            slab[64 * 64 + co * 64 + c] = acc_x[reg];
        }
        if (mb == 0) slab[2 * 64 * 64 + 64 * kh + c] = s0;
        // dgamma: the 32 threads that share a channel group (tid & 7) through LDS, in a fixed order
        raw_barrier(); // (the last multiply's reads of gl are done)
        float *red = (float *)gl; // [8 values][256 threads]
#pragma unroll
        for (int e = 0; e < 8; e++) red[e * 256 + tid] = s_dg[e];
        __syncthreads();
        if (tid < 64) { // channel (tid & 7) * 8 + (tid >> 3)
            const int grp = tid & 7, v = tid >> 3;
            float t = 0.f;
            for (int q = 0; q < 32; q++) t += red[v * 256 + grp + 8 * q];
            slab[2 * 64 * 64 + 128 + grp * 8 + v] = t;
        }
        return;
    }
    // D[m = co][n = column]: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (c < Cin * 9 && (long)blockIdx.x < n_tiles) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int co = 32 * mb + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            salsa_nn_accumulate(dw, part, 64L * Cin * 9, (int)blockIdx.x, (long)co * (Cin * 9) + c, acc[reg]);
        }
    }
}

} // namespace

// dw: float32 [64 co][Cin][3][3] contiguous, ADDED to (zero it first); x float32 planar [N][Cin <= 7][H][W] (strides in elements,
// rows contiguous), dy bf16 channels-last [N][H][W][64]
extern "C" int salsa_nn_conv3x3_stem_wrw(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *dy, float *dw,
                                         int64_t N, int Cin, int H, int W, void *hip_stream)
{
    if (!x || !dy || !dw || N <= 0 || Cin <= 0 || Cin > 7 || H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH ||
        x_channel_stride < (int64_t)H * W || x_batch_stride < x_channel_stride * Cin ||
        x_channel_stride * Cin >= INT32_MAX /* 32-bit element offsets inside an image */)
        return -1;
    const long tiles = (long)N * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    const unsigned nb = (unsigned)(tiles >= 1280 ? 1280 : tiles); // persistent, five per CU (30 KB of LDS each)
    int rc = 0;
    float *part = salsa_nn_det_begin((int)nb, 64L * Cin * 9, (hipStream_t)hip_stream, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(conv3x3_stem_wrw_kernel<0>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, x, (long)x_batch_stride,
                       (long)x_channel_stride, (const unsigned short *)dy, dw, (int)N, Cin, H, W, (const unsigned short *)nullptr,
                       (const float *)nullptr, 0, part, StemBnf{});
    if (part) return salsa_nn_det_finish(part, (int)nb, 64L * Cin * 9, dw, (hipStream_t)hip_stream);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

#ifndef STEM_WRW_BN_GRID
#define STEM_WRW_BN_GRID 512
#endif
namespace {
// MODE 2's second launch: workgroup co adds the slabs' partial sums of its output channel in float64 -- thread (column k = tid & 63,
// slab lane tid >> 6) takes every 16th slab in ascending order, lane 0 then adds the 16 lanes' sums in lane order: a fixed order,
// bit-reproducible -- and combines them: dW[co][k] += a (G - b S0[k] - k' Xh), dbeta = sum g, dgamma = sum g xh.
__global__ __launch_bounds__(1024) void stem_wrw_bnf_finalize_kernel(const float *__restrict__ part, int nslab, double M, int ncol /* Cin * 9 */,
                                                                     const StemBnf bnf, float *__restrict__ dw, float *__restrict__ dgamma,
                                                                     float *__restrict__ dbeta)
{
    __shared__ double red[5][1024];
    const int co = blockIdx.x, k = threadIdx.x & 63, sl = threadIdx.x >> 6;
    double sg = 0.0, sx = 0.0, s0 = 0.0, sb = 0.0, sd = 0.0;
    for (int b0 = sl; b0 < nslab; b0 += 16 * 4) { // four slabs' loads in flight per thread; the additions keep their order
        float v[4][5];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = b0 + 16 * u;
            const float *q = part + (long)(b < nslab ? b : b0) * SW_SLAB;
            v[u][0] = q[co * 64 + k];
            v[u][1] = q[64 * 64 + co * 64 + k];
            v[u][2] = q[2 * 64 * 64 + k] + q[2 * 64 * 64 + 64 + k]; // S0: the two lane halves
            v[u][3] = q[co * 64 + 63];                              // dbeta: G's column of ones
            v[u][4] = q[2 * 64 * 64 + 128 + co];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (b0 + 16 * u < nslab) {
                sg += (double)v[u][0]; sx += (double)v[u][1]; s0 += (double)v[u][2]; sb += (double)v[u][3]; sd += (double)v[u][4];
            }
    }
    red[0][threadIdx.x] = sg; red[1][threadIdx.x] = sx; red[2][threadIdx.x] = s0; red[3][threadIdx.x] = sb; red[4][threadIdx.x] = sd;
    __syncthreads();
    if (sl != 0) return;
    double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < 16; q++)
#pragma unroll
        for (int a = 0; a < 5; a++) t[a] += red[a][q * 64 + k];
    const double a = (double)bnf.gamma[co] * (double)bnf.invstd[co], bb = t[3] / M, kk = t[4] / M;
    if (k < ncol) dw[(long)co * ncol + k] += (float)(a * (t[0] - bb * t[2] - kk * t[1]));
    if (k == 0) {
        dbeta[co] = (float)t[3];
        dgamma[co] = (float)t[4];
    }
}
} // namespace

/* workspace of salsa_nn_conv3x3_stem_wrw_bnf: one slab of partial sums per workgroup */
extern "C" size_t salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes(int64_t N, int H, int W)
{
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const long tiles = (long)N * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    return sizeof(float) * SW_SLAB * (size_t)(tiles >= STEM_WRW_BN_GRID ? STEM_WRW_BN_GRID : tiles);
}

/* The first layer's weight gradient with the WHOLE BatchNorm (+ ReLU) backward behind it folded in (conv3x3_stem_wrw_kernel<2>): g =
 * gradient of the BatchNorm's output, x1 = its input, both bf16 channels-last; mean / invstd = the forward's saved statistics.  Leaves
 * dw (ADDED to), dgamma, dbeta; two launches (the pass over g / x1 / x, and the combination of the workgroups' slabs in `ws`), always
 * bit-reproducible.  Replaces salsa_nn_bn_bwd(dx = NULL) + salsa_nn_conv3x3_stem_wrw_bn: one pass over g and x1 instead of two. */
extern "C" int salsa_nn_conv3x3_stem_wrw_bnf(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *g,
                                             const void *x1, const float *mean, const float *invstd, const float *gamma,
                                             const float *beta, int relu, float *dw, float *dgamma, float *dbeta, void *ws,
                                             size_t ws_bytes, int64_t N, int Cin, int H, int W, void *hip_stream)
{
    if (!x || !g || !x1 || !mean || !invstd || !gamma || !beta || !dw || !dgamma || !dbeta || !ws || N <= 0 || Cin <= 0 || Cin > 7 ||
        H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH || x_channel_stride < (int64_t)H * W || x_batch_stride < x_channel_stride * Cin ||
        x_channel_stride * Cin >= INT32_MAX)
        return -1;
    if (ws_bytes < salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes(N, H, W)) return -5;
    const long tiles = (long)N * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    const unsigned nb = (unsigned)(tiles >= STEM_WRW_BN_GRID ? STEM_WRW_BN_GRID : tiles);
    const StemBnf bnf = {mean, invstd, gamma, beta};
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(conv3x3_stem_wrw_kernel<2>, dim3(nb), dim3(256), 0, st, x, (long)x_batch_stride, (long)x_channel_stride,
                       (const unsigned short *)g, dw, (int)N, Cin, H, W, (const unsigned short *)x1, (const float *)nullptr, relu,
                       (float *)ws, bnf);
    hipLaunchKernelGGL(stem_wrw_bnf_finalize_kernel, dim3(64), dim3(1024), 0, st, (const float *)ws, (int)nb, (double)(N * H * W), Cin * 9,
                       bnf, dw, dgamma, dbeta);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

// The same with the BatchNorm (+ ReLU) that follows the first layer differentiated on the fly: g = gradient of the BatchNorm's
// OUTPUT, x1 = its input (the first layer's output), both bf16 channels-last; coef = the [7][64] table salsa_nn_bn_bwd leaves in
// coef_ws (call it with dx = NULL: it then skips its apply pass, whose only reader would have been this kernel).
extern "C" int salsa_nn_conv3x3_stem_wrw_bn(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *g,
                                            const void *x1, const float *coef, int relu, float *dw, int64_t N, int Cin, int H, int W,
                                            void *hip_stream)
{
    if (!x || !g || !x1 || !coef || !dw || N <= 0 || Cin <= 0 || Cin > 7 || H <= 0 || W <= 0 || N * H * W >= INT32_MAX / CH ||
        x_channel_stride < (int64_t)H * W || x_batch_stride < x_channel_stride * Cin ||
        x_channel_stride * Cin >= INT32_MAX /* 32-bit element offsets inside an image */)
        return -1;
    const long tiles = (long)N * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    // persistent; this instantiation holds 246 registers (two sets of tile loads in flight): TWO workgroups per CU are resident, so
    // the grid is a whole number of rounds of 512 (1280 left the last half round on half the CUs)
    const unsigned nb = (unsigned)(tiles >= STEM_WRW_BN_GRID ? STEM_WRW_BN_GRID : tiles);
    int rc = 0;
    float *part = salsa_nn_det_begin((int)nb, 64L * Cin * 9, (hipStream_t)hip_stream, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(conv3x3_stem_wrw_kernel<1>, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, x, (long)x_batch_stride,
                       (long)x_channel_stride, (const unsigned short *)g, dw, (int)N, Cin, H, W, (const unsigned short *)x1, coef, relu, part, StemBnf{});
    if (part) return salsa_nn_det_finish(part, (int)nb, 64L * Cin * 9, dw, (hipStream_t)hip_stream);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}
