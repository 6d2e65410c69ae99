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
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WV = 8;                 // waves per workgroup
constexpr int TM = 64 * WV;           // pixels per workgroup
constexpr int KC = 32;                // input channels per chunk (2 MFMA k-steps)
constexpr int XL_BYTES = 53248;       // input chunk: up to 832 slots of 64 bytes
constexpr int MAX_PIECES = (XL_BYTES / 16 + 64 * WV - 1) / (64 * WV); // 16-byte pieces a thread stages per chunk

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// byte offset of 16-byte piece q of 64-byte row `row` (XOR swizzle, see the header)
__device__ __forceinline__ unsigned swz_off(int row, int q) { return (unsigned)row * 64u + (unsigned)((q ^ ((row >> 2) & 3)) * 16); }

template <int TN>
__global__ __launch_bounds__(64 * WV) void conv3x3_wide_kernel(const unsigned short *__restrict__ x,
                                                               const unsigned short *__restrict__ w,
                                                               unsigned short *__restrict__ y, int N, int H, int W, int CIN,
                                                               int COUT, int n_rows /* local rows of the padded layout */)
{
    constexpr int CT = TN / 32; // output-channel tiles per wave
    __shared__ __attribute__((aligned(16))) unsigned char xl[XL_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char wl[3 * TN * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l32 = lane & 31, khalf = lane >> 5;
    const long P = (long)N * H * W;
    const long p0 = (long)blockIdx.x * TM;
    const int co0 = blockIdx.y * TN;
    const int W2 = W + 2, HW = H * W;
    // virtual (padded) row of the tile's first pixel, minus one: local row 0
    const int n_first = (int)(p0 / HW), h_first = (int)((p0 - (long)n_first * HW) / W);
    const int v_first = n_first * (H + 1) + h_first; // = v(n_first, h_first) - 1
    // ---- staging descriptors of this thread's input pieces (the same for every channel chunk)
    int src_off[MAX_PIECES];      // element offset of the piece's first channel in x (without the chunk offset), or -1: zeros
    unsigned dst_off[MAX_PIECES]; // byte offset in xl, or 0xffffffff: nothing to write
    const int n_slots = n_rows * W2;
#pragma unroll
    for (int j = 0; j < MAX_PIECES; j++) {
        const int idx = tid + j * 64 * WV;
        const int slot = idx >> 2, q = idx & 3;
        src_off[j] = -1;
        dst_off[j] = 0xffffffffu;
        if (slot < n_slots) {
            const int lr = slot / W2, ws = slot - lr * W2;
            const int v = v_first + lr;
            const int n = v / (H + 1), hv = v - n * (H + 1);
            dst_off[j] = swz_off(slot, q);
            if (hv >= 1 && n < N && ws >= 1 && ws <= W) src_off[j] = (((n * H + hv - 1) * W + ws - 1) * CIN) + q * 8;
        }
    }
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

    for (int cc = 0; cc < CIN; cc += KC) {
        __syncthreads(); // the previous chunk's reads of xl are done
#pragma unroll
        for (int j = 0; j < MAX_PIECES; j++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (src_off[j] >= 0) v = *(const uint4 *)(x + (long)src_off[j] + cc);
            if (dst_off[j] != 0xffffffffu) *(uint4 *)(xl + dst_off[j]) = v;
        }
        for (int r = 0; r < 3; r++) {
            __syncthreads(); // the previous filter row's reads of wl are done (and, for r = 0, xl is complete after the next barrier)
#pragma unroll
            for (int j = 0; j < 3 * TN * 4 / (64 * WV); j++) { // 3 taps x TN rows x 4 pieces
                const int idx = tid + j * 64 * WV;
                const int q = idx & 3, row = idx >> 2; // row = s * TN + co
                const int s = row / TN, co = row - s * TN;
                const uint4 v = *(const uint4 *)(w + ((long)(co0 + co) * 9 + r * 3 + s) * CIN + cc + q * 8);
                *(uint4 *)(wl + swz_off(row, q)) = v;
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 3; s++)
#pragma unroll
                for (int k16 = 0; k16 < 2; k16++) {
                    bf16x8 a[CT], b[2];
#pragma unroll
                    for (int ct = 0; ct < CT; ct++) a[ct] = *(const bf16x8 *)(wl + ((a_addr[ct] + (unsigned)(s * TN * 64)) ^ (unsigned)(k16 * 32)));
#pragma unroll
                    for (int pt = 0; pt < 2; pt++) b[pt] = *(const bf16x8 *)(xl + (b_addr[r * 3 + s][pt] ^ (unsigned)(k16 * 32)));
#pragma unroll
                    for (int ct = 0; ct < CT; ct++)
#pragma unroll
                        for (int pt = 0; pt < 2; pt++)
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct], b[pt], acc[ct][pt], 0, 0, 0);
                }
        }
    }
    // D: column = lane&31 = pixel, row (= co within the tile) = (reg&3) + 8*(reg>>2) + 4*(lane>>5): 8-byte bf16 stores
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        if (pix[pt] < P) {
            unsigned short *o = y + pix[pt] * COUT + co0 + 4 * khalf;
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    uint2 v;
                    v.x = pack_bf16(acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1]);
                    v.y = pack_bf16(acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]);
                    *(uint2 *)(o + ct * 32 + 8 * g) = v;
                }
        }
    }
}

// local rows of the padded layout a 512-pixel tile can touch: its own rows, one above and below, one zero row per image start
int wide_rows(int H, int W)
{
    const int own = (TM + W - 2) / W + 1;                // rows a run of TM pixels can span
    const int images = (TM + H * W - 2) / (H * W) + 1;   // images it can span
    return own + 2 + images;
}

} // namespace

extern "C" int salsa_nn_conv3x3_wide_supported(int64_t N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin < KC || Cin % KC || Cout < 64 || Cout % 64) return 0;
    if (N * H * W * (int64_t)(Cin > Cout ? Cin : Cout) >= INT32_MAX) return 0;          // 32-bit element offsets
    return (int64_t)wide_rows(H, W) * (W + 2) * 64 <= XL_BYTES;
}

// x: [N][H][W][Cin] bf16, w: [Cout][3][3][Cin] bf16 (a channels-last torch.nn.Conv2d weight), y: [N][H][W][Cout] bf16
extern "C" int salsa_nn_conv3x3_wide(const void *x, const void *w, void *y, int64_t N, int H, int W, int Cin, int Cout,
                                     void *hip_stream)
{
    if (!x || !w || !y || x == y || !salsa_nn_conv3x3_wide_supported(N, H, W, Cin, Cout)) return -1;
    const long P = (long)N * H * W;
    const int rows = wide_rows(H, W);
    const unsigned nb = (unsigned)((P + TM - 1) / TM);
    if (Cout % 128 == 0)
        hipLaunchKernelGGL(conv3x3_wide_kernel<128>, dim3(nb, (unsigned)(Cout / 128)), dim3(64 * WV), 0, (hipStream_t)hip_stream,
                           (const unsigned short *)x, (const unsigned short *)w, (unsigned short *)y, (int)N, H, W, Cin, Cout, rows);
    else
        hipLaunchKernelGGL(conv3x3_wide_kernel<64>, dim3(nb, (unsigned)(Cout / 64)), dim3(64 * WV), 0, (hipStream_t)hip_stream,
                           (const unsigned short *)x, (const unsigned short *)w, (unsigned short *)y, (int)N, H, W, Cin, Cout, rows);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}
