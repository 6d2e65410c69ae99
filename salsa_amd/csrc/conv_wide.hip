// conv_wide.hip -- 3x3 / stride 1 / pad 1 convolution for the WIDE layers of the SELD CRNN (128 / 256 / 512 channels on
// 160x50 / 80x25 / 40x12 maps: 58 % of the network's convolution FLOPs), channels-last bf16, on the gfx950 matrix cores.
//
// The 64 -> 64 kernel (conv_mfma.hip) keeps its whole filter in registers and tiles the image in 4 x 32 pixel rectangles.
// Neither carries over: a 128..512-channel filter does not fit a register file, and 32-pixel-wide rectangles waste up to
// 62 % of the matrix tiles on maps 12..50 pixels wide.  So this kernel is an implicit GEMM over the FLATTENED pixel axis
//   C^T[co][p] = sum_{tap, ci} W[co][tap][ci] * X[p + off(tap)][ci],      p = (n*H + h)*W + w  in  [0, N*H*W)
// computed transposed like the 64 -> 64 kernel (A = 16 contiguous bytes of a filter row, B = 16 contiguous bytes of a pixel,
// D pixel-major), with both operands staged through LDS:
//   * a workgroup (8 waves) owns TM = 512 consecutive pixels x TN = 128 (or 64) output channels; a wave owns 64 pixels x all
//     TN channels = 2 x (TN/32) accumulator tiles, so one 1-KB filter fragment feeds 2 MFMAs in each of 8 waves and the
//     filter stream from L2 is 8 B/cycle/CU instead of the 32 a 128-pixel tile would need;
//   * the input chunk (32 channels of every pixel the tile's 9 taps touch) sits in LDS in a PADDED image layout: local row
//     lr = virtual row - first, virtual row v(n, h) = n*(H+1) + h + 1 (one shared zero row between images), slot =
//     lr*(W+2) + w + 1 (zero columns at both ends).  A tap is then a constant slot offset dy*(W+2) + dx for every pixel --
//     no per-tap masks, no halo logic in the multiply loop -- and the zero padding is written once per chunk by the
//     staging pass, which knows every slot's (n, h, w);
//   * both LDS tiles use 64-byte rows (32 bf16) with the 16-byte piece index XOR-ed by (row >> 2) & 3: the 16-lane groups
//     of a ds_read_b128 then touch all 64 banks exactly once (rows 4 apart would otherwise collide 4-way).
// K loop: for each 32-channel chunk: stage X once; for each filter row r: stage the 3 taps' weights; 3 taps x 2 k16 steps
// x 2*(TN/32) MFMAs per wave.  The data gradient is the same kernel on dy with the filter flipped and transposed.
#include "build_guard.h" // probe switches need -DSALSA_PROBE_BUILD; SALSA_BUILD_FLAGS (generated: tools/gen_build_guard.py)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "nn_det.h"

#ifndef WIDE_OUT_NT
#define WIDE_OUT_NT 0
#endif
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;                // input channels per chunk (2 MFMA k-steps)
constexpr int MAX_XL_BYTES = 53248;   // input chunk: up to 832 slots of 64 bytes (a multiple of 1 KiB: whole wave-loads)

#ifndef WIDE_STORE16
#define WIDE_STORE16 1 // 16-byte epilogue stores through v_permlane32_swap (0: 8-byte stores)
#endif
__device__ uint4 wide_zero16; // zero-initialised: the source of every padding piece
// Its address, pinned in a scalar register pair for the life of a kernel.  Referring to the variable at the point of use lets the
// compiler re-materialise the address there, and for a __device__ variable that is a GOT load: s_getpc + s_load_dwordx2 +
// s_waitcnt lgkmcnt(0) -- a scalar-memory round trip in FRONT of every staged wave-load, 9 per 128-pixel tile in the weight
// gradient (found in the ISA in round 4; the counters only showed "waiting").
__device__ __forceinline__ const unsigned short *wide_zero_ptr()
{
    const unsigned short *z = (const unsigned short *)&wide_zero16;
    asm volatile("" : "+s"(z));
    return z;
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// byte offset of 16-byte piece q of 64-byte row `row` (XOR swizzle, see the header)
__device__ __forceinline__ unsigned swz_off(int row, int q) { return (unsigned)row * 64u + (unsigned)((q ^ ((row >> 2) & 3)) * 16); }

template <int CTRL> __device__ __forceinline__ float wide_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Staging is LDS-direct (global_load_lds_dwordx4: a wave-instruction fills 64 consecutive 16-byte LDS pieces, no staging
// registers) and double-buffered: the loads of step i+1 -- the next filter row's weights, and at the first row of a channel
// chunk the NEXT chunk's input pixels -- are issued right after the single barrier of step i and land while its MFMAs run.
// Because the LDS side of such a load is fixed (piece i of the buffer <- lane i), the XOR swizzle is applied on the GLOBAL
// side: the lane that fills position (row, pos) fetches channel piece pos ^ ((row >> 2) & 3) of that row; padding positions
// fetch a 16-byte zero constant.
template <int TN, int WV>
__global__ __launch_bounds__(64 * WV) void conv3x3_wide_kernel(const unsigned short *__restrict__ x,
                                                               const unsigned short *__restrict__ w,
                                                               unsigned short *__restrict__ y, int N, int H, int W, int CIN,
                                                               int COUT, int n_rows /* local rows of the padded layout */,
                                                               int xl_bytes /* one input buffer, a multiple of 1024 */,
                                                               const float *__restrict__ shift /* [COUT] or NULL */,
                                                               const unsigned short *__restrict__ residual /* like y, or NULL */,
                                                               int relu, double *__restrict__ stats_part /* training: see the epilogue */)
{
    constexpr int CT = TN / 32;          // output-channel tiles per wave
    constexpr int NT = 64 * WV, TM = 64 * WV;
    constexpr int WL_BYTES = 3 * TN * 64; // one weight buffer: 3 taps x TN rows x 64 bytes
    constexpr int W_INSTR = (WL_BYTES / 16 + NT - 1) / NT;           // wave-loads per thread per weight chunk
    constexpr int X_INSTR = (MAX_XL_BYTES / 16 + NT - 1) / NT;       // ... per input chunk (upper bound)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *xl0 = lds, *wl0 = lds + 2 * xl_bytes;            // xl[2], then wl[2]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, khalf = lane >> 5;
    const unsigned short *const zero16 = wide_zero_ptr();
    const long P = (long)N * H * W;
    const long p0 = (long)blockIdx.x * TM;
    const int co0 = blockIdx.y * TN;
    const int W2 = W + 2, HW = H * W;
    // virtual (padded) row of the tile's first pixel, minus one: local row 0
    const int n_first = (int)(p0 / HW), h_first = (int)((p0 - (long)n_first * HW) / W);
    const int v_first = n_first * (H + 1) + h_first; // = v(n_first, h_first) - 1
    // ---- staging descriptors of this thread's input pieces (the same for every channel chunk): element offset in x of the
    // 8 channels this lane fetches (without the chunk offset), or -1: zeros
    int src_off[X_INSTR];
    const int n_pieces = n_rows * W2 * 4;
#pragma unroll
    for (int j = 0; j < X_INSTR; j++) {
        const int idx = tid + j * NT;
        const int slot = idx >> 2, pos = idx & 3;
        src_off[j] = -1;
        if (idx < n_pieces) {
            const int lr = slot / W2, ws = slot - lr * W2;
            const int v = v_first + lr;
            const int n = v / (H + 1), hv = v - n * (H + 1);
            const int q = pos ^ ((slot >> 2) & 3);
            if (hv >= 1 && n < N && ws >= 1 && ws <= W) src_off[j] = (((n * H + hv - 1) * W + ws - 1) * CIN) + q * 8;
        }
    }
    // weight pieces: position i of a buffer = row i/4 (= tap s * TN + co), pos i%4 <- channel piece pos ^ ((row >> 2) & 3)
    int wsrc[W_INSTR];
#pragma unroll
    for (int j = 0; j < W_INSTR; j++) {
        const int idx = tid + j * NT, row = idx >> 2, pos = idx & 3;
        const int sidx = row / TN, co = row - sidx * TN;
        wsrc[j] = ((co0 + co) * 9 + sidx) * CIN + (pos ^ ((row >> 2) & 3)) * 8; // + r*3*CIN + cc per step
    }
    auto stage_x = [&](int cc, int buf) {
#pragma unroll
        for (int j = 0; j < X_INSTR; j++) {
            if ((j * NT + wv * 64) * 16 < xl_bytes && (j * NT + wv * 64) < n_pieces) { // wave-uniform: this wave-load lies inside the chunk
                const unsigned short *src = src_off[j] >= 0 ? x + (long)src_off[j] + cc : zero16;
#ifdef WIDE_X_ZERO // (probe)
                src = zero16;
#endif
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(xl0 + buf * xl_bytes + (j * NT + wv * 64) * 16), 16, 0, 0);
            }
        }
    };
    auto stage_w = [&](int cc, int r, int buf) {
#pragma unroll
        for (int j = 0; j < W_INSTR; j++) {
            if ((j * NT + wv * 64) * 16 < WL_BYTES) { // wave-uniform
                const unsigned short *src = w + (long)wsrc[j] + r * 3 * CIN + cc;
#ifdef WIDE_W_ZERO // (probe: the filter loads answered from one cached line)
                src = zero16;
#endif
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(wl0 + buf * WL_BYTES + (j * NT + wv * 64) * 16), 16, 0, 0);
            }
        }
    };
    // ---- this lane's two output pixels (one per 32-pixel tile of the wave) and their LDS slots
    unsigned b_addr[9][2];
    long pix[2];
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        long p = p0 + wv * 64 + pt * 32 + l32;
        pix[pt] = p;
        if (p >= P) p = P - 1; // (results discarded)
        const int n = (int)(p / HW), rem = (int)(p - (long)n * HW), h = rem / W, wc = rem - h * W;
        const int slot0 = (n * (H + 1) + h + 1 - v_first) * W2 + wc + 1;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int slot = slot0 + (t / 3 - 1) * W2 + (t % 3 - 1);
            b_addr[t][pt] = swz_off(slot, khalf);
        }
    }
    unsigned a_addr[CT]; // filter fragment of co-tile ct, tap 0, k-step 0
#pragma unroll
    for (int ct = 0; ct < CT; ct++) a_addr[ct] = swz_off(ct * 32 + l32, khalf);
    f32x16 acc[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++) acc[ct][pt] = f32x16{};

    const int n_steps = (CIN / KC) * 3; // step = (channel chunk, filter row)
    stage_x(0, 0);
    stage_w(0, 0, 0);
    // One step = 3 taps x 2 k-steps = 6 sub-steps of (CT filter fragments + 2 pixel fragments -> 2*CT MFMAs).  The LDS reads are
    // software-pipelined by hand (the compiler otherwise waits for ALL outstanding reads before every MFMA group): sub-step i+1's
    // fragments are requested before sub-step i's MFMAs issue, and the counted wait lets exactly those younger reads stay in
    // flight.  Reads are asm with the tap's constant part as the instruction's immediate offset.
    const unsigned lds_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)lds;
    auto compute = [&](const int r, const unsigned xl_off, const unsigned wl_off) {
        unsigned a_lo[CT], a_hi[CT], b_ad[3][2];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            a_lo[ct] = lds_base + wl_off + a_addr[ct];
            a_hi[ct] = a_lo[ct] ^ 32u; // (the buffer offsets are multiples of 1024: the XOR commutes with the addition)
        }
#pragma unroll
        for (int sidx = 0; sidx < 3; sidx++)
#pragma unroll
            for (int pt = 0; pt < 2; pt++) b_ad[sidx][pt] = lds_base + xl_off + (r == 0 ? b_addr[sidx][pt] : r == 1 ? b_addr[3 + sidx][pt] : b_addr[6 + sidx][pt]);
        bf16x8 fa[2][CT], fb[2][2];
        auto issue = [&](const int i) {
            const int sidx = i >> 1, k16 = i & 1;
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                if (sidx == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i & 1][ct]) : "v"(k16 ? a_hi[ct] : a_lo[ct]), "n"(0));
                if (sidx == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i & 1][ct]) : "v"(k16 ? a_hi[ct] : a_lo[ct]), "n"(TN * 64));
                if (sidx == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i & 1][ct]) : "v"(k16 ? a_hi[ct] : a_lo[ct]), "n"(2 * TN * 64));
            }
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const unsigned ad = k16 ? b_ad[sidx][pt] ^ 32u : b_ad[sidx][pt];
                asm volatile("ds_read_b128 %0, %1" : "=v"(fb[i & 1][pt]) : "v"(ad));
            }
        };
        issue(0);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (i + 1 < 6) {
                issue(i + 1);
                __builtin_amdgcn_sched_barrier(0);
                // sub-step i's fragments have landed when only the CT + 2 younger reads are outstanding
                if (CT == 4) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int pt = 0; pt < 2; pt++) {
                    asm volatile("" : "+v"(fa[i & 1][ct]), "+v"(fb[i & 1][pt])); // (the MFMAs stay behind the wait)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][ct], fb[i & 1][pt], acc[ct][pt], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int step = 0, cc = 0; step < n_steps; cc += KC) {
        const int xb = (cc / KC) & 1;
#pragma unroll
        for (int r = 0; r < 3; r++, step++) {
            __syncthreads(); // (waits for this thread's loads, then the barrier) the step's data is complete; the other buffers are free
            // (Round 4: a counted wait at the second filter row that leaves the next chunk's input loads in flight for one more
            // step measured no gain, 87 - 90 us either way at 128 -> 128: what the staging costs -- 60 us with every tile load
            // answered from one cached line, WIDE_X_ZERO / WIDE_W_ZERO -- is the rate at which 64-byte pieces arrive from L2,
            // not the deadline they are given.)
            const int wb = step & 1;
            {   // the next step's loads, in flight during this step's MFMAs
                const int nr = r == 2 ? 0 : r + 1, ncc = r == 2 ? cc + KC : cc;
                if (step + 1 < n_steps) stage_w(ncc, nr, wb ^ 1);
                if (r == 0 && cc + KC < CIN) stage_x(cc + KC, xb ^ 1);
            }
            compute(r, (unsigned)(xb * xl_bytes), (unsigned)(2 * xl_bytes + wb * WL_BYTES));
        }
    }
    // Training (stats_part != NULL): the per-channel sum and sum of squares of this tile's bf16-ROUNDED outputs -- what the
    // BatchNorm that follows would otherwise read the whole tensor for (15 statistics passes, 0.17 ms per step) -- as ONE float64
    // row pair stats_part[blockIdx.x][sum | sum of squares][COUT] (this workgroup's TN channels of it), the layout of the
    // BatchNorm kernels' own partial table.  Per wave like conv64_stats (conv_mfma.hip): two DPP quad permutes sum neighbouring
    // pixel columns, lane q of a quad keeps channel group g = q, three shuffles combine the 8 quads of a half-wave; the
    // workgroup's waves meet in LDS (the tile buffers are free by then).
    if (stats_part) {
        const float m0 = pix[0] < P ? 1.f : 0.f, m1 = pix[1] < P ? 1.f : 0.f;
        const int quad = l32 & 3;
        float rs[CT][4], rq[CT][4];
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float t[4], q[4];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const unsigned u0 = pack_bf16(acc[ct][0][4 * g + 2 * k], acc[ct][0][4 * g + 2 * k + 1]);
                    const unsigned u1 = pack_bf16(acc[ct][1][4 * g + 2 * k], acc[ct][1][4 * g + 2 * k + 1]);
                    const float a0 = __uint_as_float(u0 << 16), a1 = __uint_as_float(u0 & 0xffff0000u);
                    const float b0 = __uint_as_float(u1 << 16), b1 = __uint_as_float(u1 & 0xffff0000u);
                    t[2 * k] = a0 * m0 + b0 * m1;
                    t[2 * k + 1] = a1 * m0 + b1 * m1;
                    q[2 * k] = fmaf(a0 * m0, a0, b0 * m1 * b0);
                    q[2 * k + 1] = fmaf(a1 * m0, a1, b1 * m1 * b1);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    t[j] += wide_dpp<0xB1>(t[j]); q[j] += wide_dpp<0xB1>(q[j]); // quad_perm [1,0,3,2]
                    t[j] += wide_dpp<0x4E>(t[j]); q[j] += wide_dpp<0x4E>(q[j]); // quad_perm [2,3,0,1]
                    if (g == 0) { rs[ct][j] = 0.f; rq[ct][j] = 0.f; }
                    rs[ct][j] = quad == g ? t[j] : rs[ct][j];
                    rq[ct][j] = quad == g ? q[j] : rq[ct][j];
                }
            }
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int d = 4; d <= 16; d <<= 1) {
                    rs[ct][j] += __shfl_xor(rs[ct][j], d);
                    rq[ct][j] += __shfl_xor(rq[ct][j], d);
                }
        float *lstats = (float *)lds; // [wave][sum | sum of squares][TN]
        __syncthreads();              // every wave is done with the tile buffers
        if (l32 < 4) {
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int c = ct * 32 + 8 * l32 + 4 * khalf + j; // (quad == l32 for these lanes)
                    lstats[(wv * 2 + 0) * TN + c] = rs[ct][j];
                    lstats[(wv * 2 + 1) * TN + c] = rq[ct][j];
                }
        }
        __syncthreads();
        for (int i = tid; i < 2 * TN; i += NT) {
            double sum = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < WV; w8++) sum += (double)lstats[w8 * 2 * TN + i];
            stats_part[((long)blockIdx.x * 2 + i / TN) * COUT + co0 + i % TN] = sum;
        }
    }
    // D: column = lane&31 = pixel, row (= co within the tile) = (reg&3) + 8*(reg>>2) + 4*(lane>>5): a lane holds runs of 4
    // consecutive channels and its partner lane (the same pixel, lane +- 32) the other half of each 8-channel run.
    // v_permlane32_swap trades runs between the two so that each lane owns whole 8-channel runs: 16-byte stores, half as many
    // (round 3; the 8-byte version was store-issue-bound in the 64 -> 64 kernel: conv_mfma.hip conv64_epilogue).
    // Inference epilogue (shift != NULL): folded BatchNorm shift (+ residual) (+ ReLU) on the float32 sums, one rounding.
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        const bool inside = pix[pt] < P;                   // (the same for both lanes of a pair)
        const long off = (inside ? pix[pt] : 0) * COUT + co0 + 4 * khalf;
        unsigned short *o = y + (inside ? pix[pt] : 0) * COUT + co0 + 8 * khalf;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            uint2 pk[4];
            // Round 4: the co-tile's four shift vectors and four residual pieces are requested TOGETHER, before any is used.  Fetched
            // next to their use (inside the g loop, behind the `if`s) the compiler waited `vmcnt(0)` after every single load:
            // 64 dependent memory round trips at the end of every workgroup, in the inference epilogue AND in the training
            // step's data gradients that add the skip branch's gradient (tools/asm_serial_loads.py).
            float4 shq[4];
            uint2 rvq[4];
            if (shift) {
#pragma unroll
                for (int g = 0; g < 4; g++) shq[g] = *(const float4 *)(shift + co0 + 4 * khalf + ct * 32 + 8 * g);
                if (residual) {
#pragma unroll
                    for (int g = 0; g < 4; g++) rvq[g] = *(const uint2 *)(residual + off + ct * 32 + 8 * g);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; g++) rvq[g] = make_uint2(0u, 0u);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float v4[4] = {acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]};
                if (shift) {
                    const float4 sh = shq[g];
                    v4[0] += sh.x; v4[1] += sh.y; v4[2] += sh.z; v4[3] += sh.w;
                    {
                        const uint2 rv = rvq[g]; // (+0.0f when there is no residual)
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
#if WIDE_STORE16
#pragma unroll
            for (int g = 0; g < 4; g += 2) { // swap(A, B): lanes 32-63 of A <-> lanes 0-31 of B
                const auto sx = __builtin_amdgcn_permlane32_swap(pk[g].x, pk[g + 1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(pk[g].y, pk[g + 1].y, false, false);
                if (inside) {
#if WIDE_OUT_NT // round-5 probe: the output's 16-byte stores non-temporal
                    typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(u4v_t{sx[0], sy[0], sx[1], sy[1]}, (u4v_t *)(o + ct * 32 + 8 * g));
#else
                    *(uint4 *)(o + ct * 32 + 8 * g) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
#endif
                }
            }
#else
#pragma unroll
            for (int g = 0; g < 4; g++)
                if (inside) *(uint2 *)(y + off + ct * 32 + 8 * g) = pk[g];
#endif
        }
    }
}

// local rows of the padded layout a TM-pixel tile can touch: its own rows, one above and below, one zero row per image start
int wide_rows(int TM, int H, int W)
{
    const int own = (TM + W - 2) / W + 1;                // rows a run of TM pixels can span
    const int images = (TM + H * W - 2) / (H * W) + 1;   // images it can span
    return own + 2 + images;
}
int wide_xl_bytes(int TM, int H, int W) { return ((wide_rows(TM, H, W) * (W + 2) * 64) + 1023) & ~1023; }

template <int TN, int WV>
int wide_launch(const void *x, const void *w, void *y, long N, int H, int W, int Cin, int Cout, hipStream_t st, const float *shift,
                const void *residual, int relu, double *stats_part)
{
    constexpr int TM = 64 * WV;
    const long P = N * H * W;
    const int xlb = wide_xl_bytes(TM, H, W);
    const size_t lds = 2 * (size_t)xlb + 2 * 3 * TN * 64;
    // more than 64 KiB of dynamic LDS needs the attribute; it is per device, so it is (cheaply) set on every launch rather than
    // cached in a process-wide flag that a second GPU driven by the same process would never see
    if (hipFuncSetAttribute((const void *)conv3x3_wide_kernel<TN, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return -6;
    hipLaunchKernelGGL((conv3x3_wide_kernel<TN, WV>), dim3((unsigned)((P + TM - 1) / TM), (unsigned)(Cout / TN)), dim3(64 * WV), lds, st,
                       (const unsigned short *)x, (const unsigned short *)w, (unsigned short *)y, (int)N, H, W, Cin, Cout,
                       wide_rows(TM, H, W), xlb, shift, (const unsigned short *)residual, relu, stats_part);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

} // namespace

extern "C" int salsa_nn_conv3x3_wide_supported(int64_t N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin < KC || Cin % KC || Cout < 64 || Cout % 64) return 0;
    if (N * H * W * (int64_t)(Cin > Cout ? Cin : Cout) >= INT32_MAX) return 0;          // 32-bit element offsets
    return wide_xl_bytes(256, H, W) <= MAX_XL_BYTES;
}

// tile choice: 512 pixels x 128 channels per workgroup when that still gives every CU a workgroup; smaller tiles for the small
// maps (40 x 12 x 32 clips = 15 360 pixels: 30 tiles of 512).  Returns the pixels per tile (512 | 256), *tn the channels (128 | 64).
static int wide_tile(int64_t N, int H, int W, int Cout, int *tn)
{
    const long P = (long)N * H * W;
    const bool big_ok = wide_xl_bytes(512, H, W) <= MAX_XL_BYTES;
    if (Cout % 128 == 0 && big_ok && (P + 511) / 512 * (Cout / 128) >= 192) { *tn = 128; return 512; }
#ifndef WIDE_SMALL_TN64
#define WIDE_SMALL_TN64 0 // (round-4 experiment) small maps as 256-pixel x 64-channel tiles, two workgroups per CU, instead of 256 x 128, one per CU:
                           // 512 -> 512 at 40 x 12: 68.0 -> 66.0 us (+3 %, for twice the input re-reads): not adopted
#endif
    if (Cout % 128 == 0 && (P + 255) / 256 * (Cout / 128) >= 192 && !(WIDE_SMALL_TN64 && 2 * wide_xl_bytes(256, H, W) + 2 * 3 * 64 * 64 <= 80 * 1024)) { *tn = 128; return 256; }
    *tn = 64;
    return (big_ok && (P + 511) / 512 * (Cout / 64) >= 192) ? 512 : 256;
}

static int wide_dispatch(const void *x, const void *w, void *y, int64_t N, int H, int W, int Cin, int Cout, void *hip_stream,
                         const float *shift, const void *residual, int relu, double *stats_part = nullptr)
{
    if (!x || !w || !y || x == y || !salsa_nn_conv3x3_wide_supported(N, H, W, Cin, Cout)) return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    int tn;
    const int tm = wide_tile(N, H, W, Cout, &tn);
    if (tn == 128 && tm == 512) return wide_launch<128, 8>(x, w, y, N, H, W, Cin, Cout, st, shift, residual, relu, stats_part);
    if (tn == 128) return wide_launch<128, 4>(x, w, y, N, H, W, Cin, Cout, st, shift, residual, relu, stats_part);
    if (tm == 512) return wide_launch<64, 8>(x, w, y, N, H, W, Cin, Cout, st, shift, residual, relu, stats_part);
    return wide_launch<64, 4>(x, w, y, N, H, W, Cin, Cout, st, shift, residual, relu, stats_part);
}

/* training: rows of partial statistics salsa_nn_conv3x3_wide_stats writes (= its pixel tiles), 0 if unsupported */
extern "C" int salsa_nn_conv3x3_wide_stats_blocks(int64_t N, int H, int W, int Cin, int Cout)
{
    if (!salsa_nn_conv3x3_wide_supported(N, H, W, Cin, Cout)) return 0;
    int tn;
    const int tm = wide_tile(N, H, W, Cout, &tn);
    return (int)(((long)N * H * W + tm - 1) / tm);
}

/* training: the plain convolution, which also leaves the per-channel sum / sum of squares of its (bf16-rounded) output as float64
 * partial rows stats_part[blocks][2][Cout] for the BatchNorm that follows (salsa_nn_bn_train_fwd's stats_part / stats_blocks) */
extern "C" int salsa_nn_conv3x3_wide_stats(const void *x, const void *w, void *y, double *stats_part, int64_t N, int H, int W, int Cin,
                                           int Cout, void *hip_stream)
{
    if (!stats_part) return -1;
    return wide_dispatch(x, w, y, N, H, W, Cin, Cout, hip_stream, nullptr, nullptr, 0, stats_part);
}

// x: [N][H][W][Cin] bf16, w: [Cout][3][3][Cin] bf16 (a channels-last torch.nn.Conv2d weight), y: [N][H][W][Cout] bf16
extern "C" int salsa_nn_conv3x3_wide(const void *x, const void *w, void *y, int64_t N, int H, int W, int Cin, int Cout,
                                     void *hip_stream)
{
    return wide_dispatch(x, w, y, N, H, W, Cin, Cout, hip_stream, nullptr, nullptr, 0);
}

// inference: y = [relu](conv(x, w) + shift[co] [+ residual]) -- w pre-scaled by the folded BatchNorm factor gamma / sigma
extern "C" int salsa_nn_conv3x3_wide_bias_act(const void *x, const void *w, const float *shift, const void *residual, void *y,
                                              int relu, int64_t N, int H, int W, int Cin, int Cout, void *hip_stream)
{
    if (!shift) return -1;
    return wide_dispatch(x, w, y, N, H, W, Cin, Cout, hip_stream, shift, residual, relu);
}

// ------------------------------------------------------------------------------------------------------------ weight gradient
// dW[co][tap][ci] = sum over pixels p of dy[p][co] * x[p + off(tap)][ci] for the wide layers: a GEMM whose reduction runs over
// PIXELS, so both MFMA operands need 8 consecutive pixels of ONE channel per lane while memory is pixel-major -- gfx950's
// transposing LDS read (ds_read_b64_tr_b16, mapping in conv_mfma.hip "weight gradient") builds them from pixel-major tiles.
// Same flattened pixel axis and padded slot layout as the forward kernel, with two differences: the tiles are NOT swizzled (a
// transposing read of 32 lanes covers four whole consecutive 64-byte slots = every bank once), and a workgroup walks MANY pixel
// tiles, so the (n, h, w) arithmetic of the forward kernel's set-up is replaced by two small index tables built once per map
// shape by the caller: vpos[p] = padded slot of pixel p, inv[slot] = pixel of a padded slot or -1.
// Workgroup = 4 waves = 128 output channels (32 per wave) x one 32-channel input chunk x 9 taps (9 accumulator tiles per wave)
// over a share of the pixel tiles; float32 partial sums are added to dW at the end.
namespace {

constexpr int WG_TM = 128;                          // pixels per tile
constexpr int WG_XL = 28672;                        // input tile: up to 448 slots of 64 bytes
constexpr int WG_DL = WG_TM * 4 * 64;               // dy tile: 4 channel groups x 128 pixels x 64 bytes

struct wtr_frag {
    unsigned long long lo, hi;
};
__device__ __forceinline__ bf16x8 wtr_value(const wtr_frag &f)
{
    union { unsigned long long q[2]; bf16x8 v; } u;
    u.q[0] = f.lo;
    u.q[1] = f.hi;
    return u.v;
}
#define WTR_ISSUE(f, a0, a1) asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"((f).lo), "=&v"((f).hi) : "v"(a0), "v"(a1))
#define WTR_WAIT(f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"((f).lo), "+v"((f).hi))
// the same with compile-time byte offsets in the instructions' offset fields (no address arithmetic per fragment)
#define WTR_ISSUE_OFF(f, a0, a1, o0, o1) asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%5" : "=&v"((f).lo), "=&v"((f).hi) : "v"(a0), "v"(a1), "n"(o0), "n"(o1))
template <class F, int... I> __device__ __forceinline__ void wg_static_for(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

// Every tile costs two dependent global round trips before its first MFMA (index tables, then the tiles themselves) against
// ~1 us of multiply, so the loop is pipelined two deep: while tile i is multiplied, tile i+1's LDS-direct loads are in flight
// into the other half of LDS and tile i+2's table look-ups into registers.  (First version, everything serial with two
// workgroups per CU to cover for each other: 324 - 488 TFLOP/s, on a par with MIOpen.)
#ifndef WG_KQ
#define WG_KQ 2                                      // wave groups that split a tile's k-steps (pixels): 2 = eight waves, two per SIMD, so that
#endif                                               // one wave's LDS round trips sit under the other's MFMAs; 1 = the four-wave kernel of round 2
constexpr int WG_NT = 256 * WG_KQ;                  // threads per workgroup
constexpr int WG_XP = (WG_XL / 16 + WG_NT - 1) / WG_NT; // input wave-loads per thread per tile (upper bound): 7 | 4
constexpr int WG_KS = WG_TM / 16 / WG_KQ;           // k-steps per tile and wave: 8 | 4
constexpr int WG_LDS = WG_KQ == 2 && 2 * (WG_XL + WG_DL) < 9 * 16 * 256 * 4 ? 9 * 16 * 256 * 4 : 2 * (WG_XL + WG_DL); // tiles, or the pair reduction (144 KiB)
#ifndef WG_DEPTH
#define WG_DEPTH 2                                   // fragments in flight ahead of the MFMA (<= 4: the counted waits below); measured
                                                     // 1: 433, 2: 583, 3: 558, 4: 543 TFLOP/s on 256 -> 256
#endif

// End of a weight-gradient workgroup, shared by the two kernels below: the two k-halves of a channel group meet in LDS (the tile
// buffers are free now), then the 128 x 9 x 32 block goes to dW (atomics, from the first half's waves) or to the workgroup's slab
// (deterministic mode).  Probe builds (128 -> 128 at 160 x 50): the multiply loop alone 62 us, + this epilogue 87 -- 144 scattered
// 4-byte stores per lane, the accumulator layout (lane = input channel, register = output channel) as it lies.  The slab path
// therefore turns the block through LDS into its memory layout [co][tap][ci] and all eight waves write it as 16-byte stores
// (18 per lane, whole 128-byte rows) -- round 4.
__device__ __forceinline__ void wide_wrw_finish(f32x16 (&acc)[9], unsigned char *wlds, float *__restrict__ dw, float *__restrict__ part,
                                                int CIN, int COUT, int co0, int ci0, int wv, int kq, int lane, int kh)
{
    static_assert(WG_KQ == 2, "the pair reduction below");
    __syncthreads();
    {   // 9 x 16 floats per lane, lane-major: 36 KiB per wave pair
        float *red = (float *)wlds + (wv * 64 + lane) * 4;
        if (kq == 1) {
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *(float4 *)(red + (t * 4 + g) * 1024) = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float4 v = *(const float4 *)(red + (t * 4 + g) * 1024);
                    acc[t][4 * g] += v.x; acc[t][4 * g + 1] += v.y; acc[t][4 * g + 2] += v.z; acc[t][4 * g + 3] += v.w;
                }
        }
    }
    // D[m = co][n = ci]: column = lane&31 = ci, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = co within the wave's 32
#ifndef WRW_NO_ATOMIC // (probe)
    if (part) {
        __syncthreads(); // (the pair buffer has been read)
        float *blk = (float *)wlds; // [128 co][9 taps][32 ci]
        if (kq == 0) {
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int reg = 0; reg < 16; reg++)
                    blk[((32 * wv + (reg & 3) + 8 * (reg >> 2) + 4 * kh) * 9 + t) * 32 + (lane & 31)] = acc[t][reg];
        }
        __syncthreads();
        float *out = part + (long)blockIdx.x * ((long)COUT * 9 * CIN) + (long)co0 * 9 * CIN + ci0;
        const int tid = threadIdx.x, seg = tid & 7;
#pragma unroll 6
        for (int r = 0; r < 128 * 9 * 8 / WG_NT; r++) {
            const int row = (r * WG_NT + tid) >> 3; // = co * 9 + tap of the block
            *(float4 *)(out + (long)row * CIN + seg * 4) = *(const float4 *)(blk + row * 32 + seg * 4);
        }
        return;
    }
    if (kq == 1) return;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int co = co0 + 32 * wv + (reg & 3) + 8 * (reg >> 2) + 4 * kh, ci = ci0 + (lane & 31);
            atomicAdd(dw + ((long)(co * 9 + t) * CIN + ci), acc[t][reg]);
        }
#endif
}

template <int W2C /* W + 2 known at compile time (the CRNN's maps), or 0 */>
__global__ __launch_bounds__(WG_NT, 1) void conv3x3_wide_wrw_kernel(const unsigned short *__restrict__ x,
                                                                  const unsigned short *__restrict__ dy, float *__restrict__ dw,
                                                                  const int *__restrict__ vpos, const int *__restrict__ inv,
                                                                  const int2 *__restrict__ tbounds, long P, int W, int CIN, int COUT,
                                                                  int n_shares, float *__restrict__ part /* deterministic mode: [n_shares][COUT*9*CIN] */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds[]; // xl[2][WG_XL], dl[2][WG_DL]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wv = wave & 3, kq = wave >> 2;
    const int i16 = lane & 15, cb = (lane >> 4) & 1, kh = lane >> 5;
    const unsigned short *const zero16 = wide_zero_ptr();
    const int co0 = blockIdx.y * 128, ci0 = blockIdx.z * 32;
    const int W2 = W + 2;
    const unsigned lane_chunk = (unsigned)(cb * 32 + (i16 & 3) * 8);   // this lane's 4 channels inside a 64-byte slot
    const unsigned lbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)wlds;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = f32x16{};
    const long n_tiles = (P + WG_TM - 1) / WG_TM;

    // table values of one tile, fetched a tile ahead: vb = first padded slot, ns = slots, pixel of each staged slot, slot of
    // each of this lane's 16 fragment pixel rows
    struct Look {
        int vb, ns;
        int pix[WG_XP];
        int vp[WG_KS][2];
    };
    auto lookup = [&](long tile, Look &L) {
        const long p0 = tile * WG_TM;
        const long p_last = p0 + WG_TM - 1 < P ? p0 + WG_TM - 1 : P - 1;
        // the tile's first padded slot and slot count from the per-tile table through a SCALAR load (uniform address): a vector
        // load here would make the inv[] loads below wait for it and -- vector loads retiring in order -- for the LDS-direct loads
        // issued just before (measured: the whole kernel 15 % slower; computing the two values with integer divisions: 30 %)
        const int2 tb = tbounds[__builtin_amdgcn_readfirstlane((int)tile)];
        L.vb = tb.x;
        L.ns = tb.y;
        (void)p_last;
#pragma unroll
        for (int j = 0; j < WG_XP; j++) {
            const int sl = (j * WG_NT + tid) >> 2;
            L.pix[j] = inv[L.vb + (sl < L.ns ? sl : 0)]; // (clamped address; the value is ignored beyond ns)
            if (sl >= L.ns) L.pix[j] = -1;
        }
#pragma unroll
        for (int ks = 0; ks < WG_KS; ks++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                long p = p0 + 16 * (kq * WG_KS + ks) + 8 * kh + (i16 >> 2) + 4 * hf;
                if (p >= P) p = P - 1; // (dy is zero there: no contribution)
                L.vp[ks][hf] = vpos[p];
            }
    };
    auto stage = [&](long tile, const Look &L, int buf) {
        const long p0 = tile * WG_TM;
        unsigned char *xl = wlds + buf * WG_XL, *dl = wlds + 2 * WG_XL + buf * WG_DL;
#pragma unroll
        for (int j = 0; j < WG_XP; j++) {
            if (j * WG_NT + wave * 64 < L.ns * 4) { // wave-uniform: this wave's 64 pieces (16 slots) hold slots of the tile
                const int piece = tid & 3;
                const unsigned short *src = L.pix[j] >= 0 ? x + ((long)L.pix[j] * CIN + ci0 + piece * 8) : zero16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(xl + (j * WG_NT + wave * 64) * 16), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < WG_DL / 16 / WG_NT; k++) {
            const int idx = k * WG_NT + tid, piece = idx & 3, j = (idx >> 2) & (WG_TM - 1), cg = idx >> 9;
            const unsigned short *src = p0 + j < P ? dy + ((p0 + j) * COUT + co0 + cg * 32 + piece * 8) : zero16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dl + (k * WG_NT + wave * 64) * 16), 16, 0, 0);
        }
    };
    auto multiply = [&](const Look &L, int buf) {
        const unsigned xbase = lbase + (unsigned)(buf * WG_XL), dbase = lbase + (unsigned)(2 * WG_XL + buf * WG_DL);
        // dy fragment of k-step ks: pixel rows 16 ks + 8 kh + (i16 >> 2) (+4), this wave's channel group
        const unsigned d_lane = dbase + (unsigned)((wv * WG_TM + 16 * WG_KS * kq + 8 * kh + (i16 >> 2)) * 64) + lane_chunk;
        unsigned xa[WG_KS][2];
#pragma unroll
        for (int ks = 0; ks < WG_KS; ks++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) xa[ks][hf] = xbase + (unsigned)((L.vp[ks][hf] - L.vb - (W2 + 1)) * 64) + lane_chunk;
        // 8 k-steps x (1 dy fragment + 9 x fragments -> 9 MFMAs).  With one wave per SIMD nothing else covers the LDS round trip
        // (~150 cycles against a 32-cycle MFMA), so the fragment stream runs WG_DEPTH fragments ahead of the MFMA that consumes it
        // (one fragment ahead: 13.5 k cycles per tile for 2.3 k cycles of matrix work); the counted wait lets exactly the younger
        // fragments' reads (two per fragment) stay outstanding.  Phase-drop probe builds of the finished kernel (128 -> 128,
        // 0.142 ms): without the final atomics 0.125, without the multiply 0.080, without the staging 0.114, staging alone 0.058.
        constexpr int NQ = WG_KS * 10, RING = WG_DEPTH + 1;
        wtr_frag fr[RING];
        bf16x8 a;
        // Fragment q = (k-step q / 10, j = q % 10): j = 0 the dy fragment, j = 1..9 the x fragment of tap j - 1.  Every address is a
        // per-k-step base register plus a COMPILE-TIME offset in the instruction (round 4; before: two v_add per fragment, 235
        // VALU instructions per tile and wave against 36 MFMAs): the dy fragments of all k-steps hang off one register; an x
        // fragment is its k-step's row base + (tap row * W2 + tap column) * 64 -- an immediate when the map width is a template
        // parameter (W2C), otherwise the tap row is added once per three fragments and only the column is immediate.
        unsigned cr0 = 0, cr1 = 0;
        const unsigned rowb = (unsigned)(W2 * 64);
        auto issue = [&fr, &xa, &cr0, &cr1, d_lane, rowb](auto q1c) {
            constexpr int q1 = decltype(q1c)::value, ks1 = q1 / 10, j1 = q1 % 10;
            if constexpr (j1 == 0) {
                WTR_ISSUE_OFF(fr[q1 % RING], d_lane, d_lane, ks1 * 16 * 64, ks1 * 16 * 64 + 4 * 64);
            } else {
                constexpr int t = j1 - 1, dh = t / 3, dwc = t % 3;
                if constexpr (W2C > 0) {
                    WTR_ISSUE_OFF(fr[q1 % RING], xa[ks1][0], xa[ks1][1], (dh * W2C + dwc) * 64, (dh * W2C + dwc) * 64);
                } else {
                    if constexpr (dwc == 0) {
                        cr0 = xa[ks1][0] + (unsigned)dh * rowb;
                        cr1 = xa[ks1][1] + (unsigned)dh * rowb;
                    }
                    WTR_ISSUE_OFF(fr[q1 % RING], cr0, cr1, dwc * 64, dwc * 64);
                }
            }
        };
        wg_static_for([&issue](auto qc) { issue(qc); }, std::make_integer_sequence<int, WG_DEPTH>{});
        wg_static_for([&issue, &fr, &a, &acc](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q + WG_DEPTH < NQ) issue(std::integral_constant<int, q + WG_DEPTH>{});
            __builtin_amdgcn_sched_barrier(0);
            {   // fragment q has landed when at most the 2 * min(WG_DEPTH, NQ - 1 - q) younger reads are outstanding
                constexpr int younger = 2 * (NQ - 1 - q < WG_DEPTH ? NQ - 1 - q : WG_DEPTH);
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fr[q % RING].lo), "+v"(fr[q % RING].hi) : "n"(younger));
            }
            constexpr int j = q % 10;
            if constexpr (j == 0) a = wtr_value(fr[q % RING]);
            else acc[j - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wtr_value(fr[q % RING]), acc[j - 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, NQ>{});
    };
    // software pipeline over this workgroup's tiles, tile(i) = blockIdx.x + i * n_shares: at iteration i the data of tile i is
    // in buffer i & 1 with its tables in Lcur, tile i+1's tables are in Lnext
    Look Lcur, Lnext;
    long tile = blockIdx.x;
    if (tile < n_tiles) {
        lookup(tile, Lcur);
        stage(tile, Lcur, 0);
        if (tile + n_shares < n_tiles) lookup(tile + n_shares, Lnext);
    }
    for (int buf = 0; tile < n_tiles; tile += n_shares, buf ^= 1) {
        __syncthreads(); // (drains this thread's loads) tile's data is complete and the other buffer is free
#ifndef WRW_NO_STAGE // (probe builds: WRW_NO_STAGE / WRW_NO_MULT / WRW_NO_ATOMIC drop one phase each)
        if (tile + n_shares < n_tiles) stage(tile + n_shares, Lnext, buf ^ 1);
#endif
        Look Lnn = Lnext;
        if (tile + 2 * n_shares < n_tiles) lookup(tile + 2 * n_shares, Lnn);
#ifndef WRW_NO_MULT
        multiply(Lcur, buf);
#endif
        Lcur = Lnext;
        Lnext = Lnn;
    }
    wide_wrw_finish(acc, wlds, dw, part, CIN, COUT, co0, ci0, wv, kq, lane, kh);
}


// ---- round 4: THREE tile buffers, index tables through LDS.
// Probe builds of the kernel above (128 -> 128 at 160 x 50, batch 32, inputs from HBM): whole 139 us; without the staging 97; without
// the multiply 88; without the final atomics 118.  Staging and multiplying each need ~70 us and together 118: the two-buffer
// pipeline hides one tile time (~1.1 us of MFMAs) of a load chain -- table look-up, then the tile itself, mostly L2 misses --
// that is longer than that.  (Deeper LDS-read pipelining inside the multiply, WG_DEPTH 3 / 4: no change.)  Here tile i + 2 is
// staged while tile i is multiplied, and the look-ups run four tiles ahead.  What makes that possible:
//  * every load is LDS-direct, the tables included (4-byte global_load_lds into small LDS rings: inv[] of a tile's slots, 3
//    entries; vpos[] of its 128 pixels, 5 entries), so no load returns into a register and the compiler inserts no vmcnt wait of
//    its own; the kernel waits by COUNT at the top of an iteration -- everything but the loads of the previous iteration --
//    and every wave issues a launch-constant number of loads per iteration (x: nx = 1..3 by wave, dy: 4, tables: 1);
//    The count errs on the safe side only: where the compiler turns a `cond ? tensor : zeros` source into two exec-masked loads
//    a wave has MORE loads in flight than counted and the wait is stricter than needed, never laxer; nothing in the loop lets it
//    issue fewer (the LDS-direct builtins have side effects: they are neither merged nor dropped).
//  * the x buffers are sized for the map (W = 50: 304 slots instead of 448): 3 x (19 + 32) KiB + rings = 159 KiB of LDS.
// Maps too wide for that keep the kernel above.
#ifndef WRW_AUX
#define WRW_AUX 0 // cache-policy bits of the tile loads (probe: 1 = sc0, 2 = nt, 16 = sc1)
#endif
constexpr int W3_VP_RING = 5, W3_INV_RING = 3;
constexpr int W3_LDS_MAX = 160 * 1024;

// upper bound of a tile's slot count (tile_bounds[].y of salsa_nn_conv3x3_wide_tables), rounded up to whole 16-slot wave-loads
static int wrw3_slots(int H, int W)
{
    const int W2 = W + 2, rc = (WG_TM - 1 + W - 1) / W, ic = (WG_TM - 1 + H * W - 1) / (H * W);
    return (WG_TM - 1 + 2 * rc + W2 * ic + 2 * W2 + 3 + 15) & ~15;
}
static int wrw3_lds_bytes(int H, int W)
{
    const int xs = wrw3_slots(H, W), n_inv_w = (xs + 63) / 64;
    return 3 * (xs * 64 + WG_DL) + W3_INV_RING * n_inv_w * 256 + W3_VP_RING * 512 + 256;
}
static bool wrw3_supported(int H, int W)
{
    return WG_KQ == 2 && (wrw3_slots(H, W) + 63) / 64 + 2 <= 8 && wrw3_slots(H, W) / 16 <= 24 && wrw3_lds_bytes(H, W) <= W3_LDS_MAX;
}

template <int W2C>
__global__ __launch_bounds__(WG_NT, 1) void conv3x3_wide_wrw3_kernel(const unsigned short *__restrict__ x,
                                                                   const unsigned short *__restrict__ dy, float *__restrict__ dw,
                                                                   const int *__restrict__ vpos, const int *__restrict__ inv,
                                                                   const int2 *__restrict__ tbounds, long P, int W, int CIN, int COUT,
                                                                   int n_shares, float *__restrict__ part, int xs /* slots per x buffer */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds[]; // xl[3][xs * 64], dl[3][WG_DL], inv ring, vpos ring, dump
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wv = wave & 3, kq = wave >> 2;
    const int i16 = lane & 15, cb = (lane >> 4) & 1, kh = lane >> 5;
    const unsigned short *const zero16 = wide_zero_ptr();
    const int co0 = blockIdx.y * 128, ci0 = blockIdx.z * 32;
    const int W2 = W2C > 0 ? W2C : W + 2;
    const unsigned lane_chunk = (unsigned)(cb * 32 + (i16 & 3) * 8);
    const unsigned lbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)wlds;
    const int XB = xs * 64, n_inv_w = (xs + 63) >> 6, n_xw = xs >> 4;       // bytes of an x buffer; wave-loads of a tile's inv[] / x slots
    const int o_dl = 3 * XB, o_inv = o_dl + 3 * WG_DL, o_vp = o_inv + W3_INV_RING * n_inv_w * 256, o_dump = o_vp + W3_VP_RING * 512;
    const int nx = (n_xw - wave + 7) >> 3;                                   // this wave's x wave-loads per tile (q = wave, wave + 8, ...)
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = f32x16{};
    const long n_tiles = (P + WG_TM - 1) / WG_TM;
    const int n_mine = (int)((n_tiles - (long)blockIdx.x + n_shares - 1) / n_shares);       // tiles blockIdx.x, + n_shares, ...
    auto tile_of = [&](int i) { return (long)blockIdx.x + (long)i * n_shares; };

    // tables of tile i: ONE 4-byte LDS-direct load per wave
    auto lookup = [&](int i) {
        const long tile = tile_of(i);
        const int2 tb = tbounds[__builtin_amdgcn_readfirstlane((int)tile)];
        const int *src;
        int dst;
        if (wave < n_inv_w) {           // inv[first slot + sl]: the pixel of slot sl (clamped address; the value is ignored beyond the tile's count)
            const int sl = wave * 64 + lane;
            src = inv + tb.x + (sl < tb.y ? sl : 0);
            dst = o_inv + ((i % W3_INV_RING) * n_inv_w + wave) * 256;
        } else if (wave < n_inv_w + 2) { // vpos[p]: the slot of each of the tile's 128 pixels
            const int h = wave - n_inv_w;
            long p = tile * WG_TM + h * 64 + lane;
            if (p >= P) p = P - 1;      // (dy is zero there: no contribution)
            src = vpos + p;
            dst = o_vp + (i % W3_VP_RING) * 512 + h * 256;
        } else {                        // (a wave without a share: the same number of loads in flight as everyone else)
            src = vpos; // (every lane the same valid word)
            dst = o_dump;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(wlds + dst), 4, 0, 0);
    };
    // tile i into buffer b; its inv[] table has landed in the ring
    auto stage = [&](int i, int b) {
        const long tile = tile_of(i), p0 = tile * WG_TM;
        const int ns = tbounds[__builtin_amdgcn_readfirstlane((int)tile)].y;
        const unsigned tinv = lbase + (unsigned)(o_inv + (i % W3_INV_RING) * n_inv_w * 256);
        int pix[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int sl = (wave + 8 * j) * 16 + (lane >> 2);
            pix[j] = -1;
            if (j < nx) asm volatile("ds_read_b32 %0, %1" : "=v"(pix[j]) : "v"(tinv + (unsigned)(sl * 4)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pix[0]), "+v"(pix[1]), "+v"(pix[2]));
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (j < nx) { // wave-uniform
                const int q = wave + 8 * j, sl = q * 16 + (lane >> 2), piece = lane & 3;
                const unsigned short *src = (sl < ns && pix[j] >= 0) ? x + ((long)pix[j] * CIN + ci0 + piece * 8) : zero16;
#if defined(WRW_STAGE_NOLOAD) || defined(WRW_STAGE_NOX) // (probe: the staging code with its scalar / LDS round trips, but zeros instead of the tensor)
                src = zero16;
#endif
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(wlds + b * XB + q * 1024), 16, 0, WRW_AUX);
            }
        }
#pragma unroll
        for (int k = 0; k < WG_DL / 16 / WG_NT; k++) {
            const int idx = k * WG_NT + tid, piece = idx & 3, j = (idx >> 2) & (WG_TM - 1), cg = idx >> 9;
            const unsigned short *src = p0 + j < P ? dy + ((p0 + j) * COUT + co0 + cg * 32 + piece * 8) : zero16;
#if defined(WRW_STAGE_NOLOAD) || defined(WRW_STAGE_NODY)
            src = zero16;
#endif
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(wlds + o_dl + b * WG_DL + (k * WG_NT + wave * 64) * 16), 16, 0, WRW_AUX);
        }
    };
    auto multiply = [&](int i, int b) {
        const int vb = tbounds[__builtin_amdgcn_readfirstlane((int)tile_of(i))].x;
        const unsigned xbase = lbase + (unsigned)(b * XB), dbase = lbase + (unsigned)(o_dl + b * WG_DL);
        const unsigned d_lane = dbase + (unsigned)((wv * WG_TM + 16 * WG_KS * kq + 8 * kh + (i16 >> 2)) * 64) + lane_chunk;
        // the padded slot of each of this lane's fragment pixel rows: vpos ring entry, pixel 16 (kq KS + ks) + 8 kh + (i16 >> 2) (+ 4)
        const unsigned tvp = lbase + (unsigned)(o_vp + (i % W3_VP_RING) * 512 + (16 * WG_KS * kq + 8 * kh + (i16 >> 2)) * 4);
        int2 vp[WG_KS];
        wg_static_for([&vp, tvp](auto kc) {
            constexpr int ks = decltype(kc)::value;
            asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(vp[ks]) : "v"(tvp), "n"(ks * 16), "n"(ks * 16 + 4));
        }, std::make_integer_sequence<int, WG_KS>{});
        static_assert(WG_KS == 4, "the wait below names four registers");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vp[0]), "+v"(vp[1]), "+v"(vp[2]), "+v"(vp[3]));
        unsigned xa[WG_KS][2];
#pragma unroll
        for (int ks = 0; ks < WG_KS; ks++) {
            xa[ks][0] = xbase + (unsigned)((vp[ks].x - vb - (W2 + 1)) * 64) + lane_chunk;
            xa[ks][1] = xbase + (unsigned)((vp[ks].y - vb - (W2 + 1)) * 64) + lane_chunk;
        }
        constexpr int NQ = WG_KS * 10, RING = WG_DEPTH + 1;
        wtr_frag fr[RING];
        bf16x8 a;
        unsigned cr0 = 0, cr1 = 0;
        const unsigned rowb = (unsigned)(W2 * 64);
        auto issue = [&fr, &xa, &cr0, &cr1, d_lane, rowb](auto q1c) { // (see the kernel above)
            constexpr int q1 = decltype(q1c)::value, ks1 = q1 / 10, j1 = q1 % 10;
            if constexpr (j1 == 0) {
                WTR_ISSUE_OFF(fr[q1 % RING], d_lane, d_lane, ks1 * 16 * 64, ks1 * 16 * 64 + 4 * 64);
            } else {
                constexpr int t = j1 - 1, dh = t / 3, dwc = t % 3;
                if constexpr (W2C > 0) {
                    WTR_ISSUE_OFF(fr[q1 % RING], xa[ks1][0], xa[ks1][1], (dh * W2C + dwc) * 64, (dh * W2C + dwc) * 64);
                } else {
                    if constexpr (dwc == 0) {
                        cr0 = xa[ks1][0] + (unsigned)dh * rowb;
                        cr1 = xa[ks1][1] + (unsigned)dh * rowb;
                    }
                    WTR_ISSUE_OFF(fr[q1 % RING], cr0, cr1, dwc * 64, dwc * 64);
                }
            }
        };
        wg_static_for([&issue](auto qc) { issue(qc); }, std::make_integer_sequence<int, WG_DEPTH>{});
        wg_static_for([&issue, &fr, &a, &acc](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q + WG_DEPTH < NQ) issue(std::integral_constant<int, q + WG_DEPTH>{});
            __builtin_amdgcn_sched_barrier(0);
            {
                constexpr int younger = 2 * (NQ - 1 - q < WG_DEPTH ? NQ - 1 - q : WG_DEPTH);
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fr[q % RING].lo), "+v"(fr[q % RING].hi) : "n"(younger));
            }
            constexpr int j = q % 10;
            if constexpr (j == 0) a = wtr_value(fr[q % RING]);
            else acc[j - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wtr_value(fr[q % RING]), acc[j - 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, NQ>{});
    };

    // prologue: the tables of tiles 0..2, then tiles 0 and 1, then the tables of tile 3 (whose ring entry is tile 0's)
#pragma unroll
    for (int i = 0; i < 3; i++)
        if (i < n_mine) lookup(i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (0 < n_mine) stage(0, 0);
    int prev = 0; // loads this wave issued after the ones iteration i needs
    if (1 < n_mine) { stage(1, 1); prev = nx + 4; }
    __builtin_amdgcn_s_barrier(); // (every wave has read tile 0's table)
    if (3 < n_mine) { lookup(3); prev += 1; }
    int b = 0;
    for (int i = 0; i < n_mine; i++) {
        // tile i and the tables of tile i + 2 have landed when only the previous iteration's loads are in flight
        switch (prev) {
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef WRW_NO_BARRIER // (probe)
        __builtin_amdgcn_s_barrier(); // ... for every wave; and buffer (i + 2) % 3 = (i - 1) % 3 is free
#endif
        // The loads of this iteration are a chain of scalar and LDS round trips (tile bounds, table values, addresses) before the
        // first of them issues: ~25 % of an iteration during which this wave feeds no MFMA.  The two waves of a SIMD (k-halves
        // kq = 0 / 1) therefore take turns: one issues its loads before its multiply, the other after it.
        prev = 0;
        auto loads = [&]() {
#ifndef WRW_NO_STAGE
            if (i + 2 < n_mine) { stage(i + 2, b == 0 ? 2 : b - 1); prev = nx + 4; }
#endif
#ifndef WRW_NO_LOOKUP // (probe)
            if (i + 4 < n_mine) { lookup(i + 4); prev += 1; }
#endif
        };
#ifndef WRW_NO_STAGGER
        if (kq == 0) loads();
#else
        loads();
#endif
#ifndef WRW_NO_MULT
        multiply(i, b);
#endif
#ifndef WRW_NO_STAGGER
        if (kq == 1) loads();
#endif
        b = b == 2 ? 0 : b + 1;
    }
    __syncthreads(); // (drains everything; the tile buffers are free for the pair reduction)
    wide_wrw_finish(acc, wlds, dw, part, CIN, COUT, co0, ci0, wv, kq, lane, kh);
}

} // namespace

extern "C" int salsa_nn_conv3x3_wide_wrw_supported(int64_t N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin < 32 || Cin % 32 || Cout < 128 || Cout % 128) return 0;
    if (N * H * W * (int64_t)(Cin > Cout ? Cin : Cout) >= INT32_MAX || Cout / 128 > 65535 || Cin / 32 > 65535) return 0;
    return (int64_t)wide_rows(WG_TM, H, W) * (W + 2) * 64 <= WG_XL;
}

// Index tables of a map shape (host arrays the caller uploads once): vpos[N*H*W], inv[salsa_nn_conv3x3_wide_table_len],
// tile_bounds[2 * salsa_nn_conv3x3_wide_tile_count]
extern "C" int64_t salsa_nn_conv3x3_wide_table_len(int64_t N, int H, int W) { return (N * (H + 1) + 3) * (int64_t)(W + 2) + 16; }
extern "C" int64_t salsa_nn_conv3x3_wide_tile_count(int64_t N, int H, int W) { return (N * H * W + WG_TM - 1) / WG_TM; }
extern "C" int salsa_nn_conv3x3_wide_tables(int64_t N, int H, int W, int *vpos, int *inv, int *tile_bounds)
{
    if (N <= 0 || H <= 0 || W <= 0 || !vpos || !inv || !tile_bounds || N * (H + 2) * (int64_t)(W + 2) >= INT32_MAX) return -1;
    const int64_t len = salsa_nn_conv3x3_wide_table_len(N, H, W);
    for (int64_t i = 0; i < len; i++) inv[i] = -1;
    int64_t p = 0;
    for (int64_t n = 0; n < N; n++)
        for (int h = 0; h < H; h++)
            for (int w = 0; w < W; w++, p++) {
                const int64_t v = (n * (H + 1) + h + 1) * (W + 2) + w + 1;
                vpos[p] = (int)v;
                inv[v] = (int)p;
            }
    const int64_t P = N * H * W, W2 = W + 2;
    for (int64_t t = 0; t < (P + WG_TM - 1) / WG_TM; t++) { // per 128-pixel tile: first padded slot its taps touch, number of slots
        const int64_t pf = t * WG_TM, pl = pf + WG_TM - 1 < P ? pf + WG_TM - 1 : P - 1;
        tile_bounds[2 * t] = (int)(vpos[pf] - (W2 + 1));
        tile_bounds[2 * t + 1] = (int)(vpos[pl] - tile_bounds[2 * t] + W2 + 2);
    }
    return 0;
}

// dw: float32 [Cout][3][3][Cin], ADDED to (zero it first); x [N][H][W][Cin], dy [N][H][W][Cout] bf16; d_vpos / d_inv / d_tbounds:
// the device copies of salsa_nn_conv3x3_wide_tables(N, H, W)
extern "C" int salsa_nn_conv3x3_wide_wrw(const void *x, const void *dy, float *dw, const int *d_vpos, const int *d_inv,
                                         const int *d_tbounds, int64_t N, int H, int W, int Cin, int Cout, void *hip_stream)
{
    if (!x || !dy || !dw || !d_vpos || !d_inv || !d_tbounds || !salsa_nn_conv3x3_wide_wrw_supported(N, H, W, Cin, Cout)) return -1;
    const long P = (long)N * H * W;
    const long tiles = (P + WG_TM - 1) / WG_TM;
    const long pairs = (long)(Cout / 128) * (Cin / 32);
    long shares = (256 + pairs - 1) / pairs; // one workgroup per CU (its 120 KiB of LDS fill it)
    if (shares > tiles) shares = tiles;
    if (shares < 1) shares = 1;
    int rc = 0;
    float *part = salsa_nn_det_begin((int)shares, (long)Cout * 9 * Cin, (hipStream_t)hip_stream, &rc);
    if (rc) return rc;
    const dim3 grid((unsigned)shares, (unsigned)(Cout / 128), (unsigned)(Cin / 32));
    const bool three = wrw3_supported(H, W); // three tile buffers when the map's x tiles are small enough for that (see the kernel)
    auto launch = [&](auto w2c) -> int {
        constexpr int W2C = decltype(w2c)::value;
        // 120 - 159 KiB of dynamic LDS needs the attribute (per device: set on every launch)
        if (three) {
            if (hipFuncSetAttribute((const void *)conv3x3_wide_wrw3_kernel<W2C>, hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_MAX) != hipSuccess)
                return -6;
            const int lds = wrw3_lds_bytes(H, W) > WG_LDS ? wrw3_lds_bytes(H, W) : WG_LDS; // (>= the pair reduction's 144 KiB)
            hipLaunchKernelGGL(conv3x3_wide_wrw3_kernel<W2C>, grid, dim3(WG_NT), lds, (hipStream_t)hip_stream, (const unsigned short *)x,
                               (const unsigned short *)dy, dw, d_vpos, d_inv, (const int2 *)d_tbounds, P, W, Cin, Cout, (int)shares, part,
                               wrw3_slots(H, W));
            return 0;
        }
        if (hipFuncSetAttribute((const void *)conv3x3_wide_wrw_kernel<W2C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return -6;
        hipLaunchKernelGGL(conv3x3_wide_wrw_kernel<W2C>, grid, dim3(WG_NT), WG_LDS, (hipStream_t)hip_stream, (const unsigned short *)x,
                           (const unsigned short *)dy, dw, d_vpos, d_inv, (const int2 *)d_tbounds, P, W, Cin, Cout, (int)shares, part);
        return 0;
    };
    // the CRNN's three wide map widths get instantiations with every tap offset an immediate; any other width the generic one
    rc = W == 50 ? launch(std::integral_constant<int, 52>{}) : W == 25 ? launch(std::integral_constant<int, 27>{})
       : W == 12 ? launch(std::integral_constant<int, 14>{}) : launch(std::integral_constant<int, 0>{});
    if (rc) return rc;
    if (part) return salsa_nn_det_finish(part, (int)shares, (long)Cout * 9 * Cin, dw, (hipStream_t)hip_stream);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}
