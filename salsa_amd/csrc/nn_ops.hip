// nn_ops.hip -- gfx950 kernels for the memory-bound layers of the SELD CRNN (C ABI: include/salsa_nn.h).
// Everything here is a streaming pass: 16-byte accesses, channels-last so that a thread's vector is contiguous.
#include "build_guard.h" // probe switches need -DSALSA_PROBE_BUILD; SALSA_BUILD_FLAGS (generated: tools/gen_build_guard.py)
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include "../../include/salsa_nn.h"
#include "nn_det.h"
#include "nn_common.h"

namespace {

struct f32x4 {
    float v[4];
};
struct bf16x8 {
    unsigned short v[8];
};

__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
// two float32 -> packed bf16 (low half = a), round to nearest even: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// Round-5 probes: the 16-byte vector loads / stores of the streaming passes (BatchNorm, pooling) as non-temporal accesses
// (-DNN_NT_LD=1 / -DNN_NT_ST=1).  Every tensor these passes touch is read or written exactly once per launch.
#ifndef NN_NT_LD
#define NN_NT_LD 0
#endif
#ifndef NN_NT_ST
#define NN_NT_ST 0
#endif
typedef unsigned nn_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nn_ld16(const void *p)
{
    if (NN_NT_LD) {
        const nn_u4v t = __builtin_nontemporal_load((const nn_u4v *)p);
        return make_uint4(t.x, t.y, t.z, t.w);
    }
    return *(const uint4 *)p;
}
__device__ __forceinline__ void nn_st16(void *p, const uint4 v)
{
    if (NN_NT_ST) {
        const nn_u4v t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, (nn_u4v *)p);
    } else
        *(uint4 *)p = v;
}

template <typename V, int L> struct vec_io;
template <> struct vec_io<f32x4, 4> {
    typedef float4 raw_t;
    static __device__ __forceinline__ raw_t load_raw(const void *p) { return *(const float4 *)p; }
    static __device__ __forceinline__ void convert(const raw_t &t, float *f) { f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w; }
    static __device__ __forceinline__ void load(const void *p, float *f)
    {
        const float4 t = *(const float4 *)p;
        f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
    }
    static __device__ __forceinline__ void store(void *p, const float *f) { *(float4 *)p = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct vec_io<bf16x8, 8> {
    typedef uint4 raw_t;
    static __device__ __forceinline__ raw_t load_raw(const void *p) { return nn_ld16(p); }
    static __device__ __forceinline__ void convert(const raw_t &t, float *f)
    {
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f[2 * i] = bf2f((unsigned short)(w[i] & 0xffffu));
            f[2 * i + 1] = bf2f((unsigned short)(w[i] >> 16));
        }
    }
    static __device__ __forceinline__ void load(const void *p, float *f) { convert(load_raw(p), f); }
    static __device__ __forceinline__ void store(void *p, const float *f)
    {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
        nn_st16(p, make_uint4(w[0], w[1], w[2], w[3]));
    }
};

// one thread = one output pixel x L channels (16 bytes); x, y channels-last
template <typename V, int L>
__global__ __launch_bounds__(256) void avgpool2x2_fwd_kernel(const char *__restrict__ x, char *__restrict__ y, long n_vec,
                                                             int H, int W, int C)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_vec) return;
    const int cv = C / L, Ho = H / 2, Wo = W / 2;
    const int c = (int)(i % cv);
    long r = i / cv;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long n = r / Ho;
    const long esz = 16 / L; // bytes per element
    const char *p = x + (((n * H + 2 * ho) * W + 2 * wo) * C + (long)c * L) * esz;
    float a[L], b[L], cc[L], d[L], o[L];
    vec_io<V, L>::load(p, a);
    vec_io<V, L>::load(p + (long)C * esz, b);
    vec_io<V, L>::load(p + (long)W * C * esz, cc);
    vec_io<V, L>::load(p + ((long)W * C + C) * esz, d);
#pragma unroll
    for (int k = 0; k < L; k++) o[k] = (((a[k] + b[k]) + cc[k]) + d[k]) / 4.0f; // the reference implementation's order
    vec_io<V, L>::store(y + i * 16, o);
}

// one thread = one INPUT pixel x L channels: grad_x = grad_y[h/2][w/2] / 4, zero in a dropped odd row / column
template <typename V, int L>
__global__ __launch_bounds__(256) void avgpool2x2_bwd_kernel(const char *__restrict__ gy, char *__restrict__ gx, long n_vec,
                                                             int H, int W, int C)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_vec) return;
    const int cv = C / L, Ho = H / 2, Wo = W / 2;
    const int c = (int)(i % cv);
    long r = i / cv;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const long n = r / H;
    float g[L];
    if (h < 2 * Ho && w < 2 * Wo) {
        vec_io<V, L>::load(gy + ((((n * Ho + h / 2) * Wo + w / 2) * cv + c) * 16), g);
#pragma unroll
        for (int k = 0; k < L; k++) g[k] = g[k] / 4.0f;
    } else {
#pragma unroll
        for (int k = 0; k < L; k++) g[k] = 0.f;
    }
    vec_io<V, L>::store(gx + i * 16, g);
}


// ------------------------------------------------------------------------------------------------ BatchNorm2d (+ add) (+ ReLU)
// x: [M][C] channels-last (M = N*H*W).  A thread owns one column group of L channels (16 bytes); a 256-thread block is
// (256/cv) rows x cv column groups, cv = C / L, and walks BN_ROWS_PER_THREAD rows per thread.  Reductions: float32
// per-thread partials -> LDS tree over the block's rows -> one partial row per block (no atomics: thousands of blocks would
// serialise on 2*C addresses), summed in float64 by the finalize kernels.
#ifndef BN_RPT
#define BN_RPT 32
#endif
constexpr int BN_ROWS_PER_THREAD = BN_RPT; // rows per thread of the reduction kernels on LARGE maps (bn_rpt below picks fewer on small ones)
#ifndef BN_BATCH_N
#define BN_BATCH_N 4
#endif
constexpr int BN_BATCH = BN_BATCH_N; // rows whose loads are in flight together in the reduction kernels
static_assert(BN_RPT % BN_BATCH_N == 0, "row batches");
// Rows per thread, by map size (round 4 experiment, OFF).  With the fixed 32 (reductions) / 16 (element-wise passes) rows per
// thread the 33-MB and 16-MB maps of stages 3 / 4 are cut into 120 - 500 workgroups, and r3_bn_network_probe.txt shows 35 us forward /
// 63 us backward per BatchNorm there "whatever the size".  Fewer rows per thread on small maps (~1000 / ~2000 workgroups) did NOT
// help: forward unchanged, backward 0.063 -> 0.142 ms at 40 x 12 x 512 and 0.068 -> 0.183 at 80 x 25 x 256 (eight times the partial
// rows for the finalize kernels and a block-reduction tree per 4 rows instead of per 32), step 10.78 -> 10.75 ms = noise
// (profiles/r4_bn_adaptive_rows.txt).  The fixed cost of those passes is not their parallelism.
#ifndef BN_ADAPTIVE_ROWS
#define BN_ADAPTIVE_ROWS 0
#endif
__host__ __device__ inline int bn_rpt(long M, int rpi)
{
    int r = BN_ROWS_PER_THREAD;
    while (BN_ADAPTIVE_ROWS && r > BN_BATCH && (M + (long)rpi * r - 1) / ((long)rpi * r) < 1024) r >>= 1;
    return r;
}
constexpr int BN_APPLY_ROWS_MAX = 16;
__host__ __device__ inline int bn_apply_rows(long M, int rpi)
{
    int r = BN_APPLY_ROWS_MAX;
    while (BN_ADAPTIVE_ROWS && r > 2 && (M + (long)rpi * r - 1) / ((long)rpi * r) < 2048) r >>= 1;
    return r;
}

template <int L, typename A = float>
__device__ __forceinline__ void bn_block_reduce(A (&a)[L], A (&b)[L], A *red /*[2L][256]*/, int cv, A *part_a, A *part_b, int c0)
{
    const int tid = threadIdx.x; // red[value][thread]: neighbouring lanes touch neighbouring banks
#pragma unroll
    for (int k = 0; k < L; k++) {
        red[k * 256 + tid] = a[k];
        red[(L + k) * 256 + tid] = b[k];
    }
    __syncthreads();
    for (int s = 128; s >= cv; s >>= 1) { // threads tid and tid + s own the same column group (s is a multiple of cv)
        if (tid < s) {
#pragma unroll
            for (int k = 0; k < 2 * L; k++) red[k * 256 + tid] += red[k * 256 + tid + s];
        }
        __syncthreads();
    }
    if (tid < cv) { // this block's partial sums (no atomics: thousands of blocks would serialise on 2*C addresses)
#pragma unroll
        for (int k = 0; k < L; k++) {
            part_a[c0 + k] = red[k * 256 + tid];
            part_b[c0 + k] = red[(L + k) * 256 + tid];
        }
    }
}

// Sum the blocks' partials [nblk][2][C] for FOUR channels (both sums) in float64: a 256-thread block = 8 outputs
// (o = which * 4 + channel) x 32 threads that each add every 32nd partial, then an LDS tree over the 32.  On return
// red[o] holds the totals; the statistics kernels below finish their per-channel arithmetic in the same launch.
template <typename P>
__device__ __forceinline__ void bn_sum_partials4(const P *__restrict__ part, int nblk, int C, int c0, double *red /*[256]*/)
{
    const int o = threadIdx.x & 7, seg = threadIdx.x >> 3;
    const int c = c0 + (o & 3), which = o >> 2;
    double acc = 0.0;
    if (c < C) // eight independent loads in flight per thread (one at a time, the loop was a chain of ~32 memory round trips:
               // 7.9 us per launch, 42 launches per training step); the additions keep their order
        for (int b = seg; b < nblk; b += 32 * 8) {
            P v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int bb = b + 32 * u;
                v[u] = part[((long)(bb < nblk ? bb : b) * 2 + which) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (b + 32 * u < nblk) acc += (double)v[u];
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 8; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
}

// part[block][0][C] = sum x, part[block][1][C] = sum x^2 over the block's rows
template <typename V, int L>
__global__ __launch_bounds__(256) void bn_stats_kernel(const char *__restrict__ x, long M, int C, double *__restrict__ part)
{
    // float64 sums (the same instruction rate as float32 on this chip, and the pass is HBM-bound): var = E[x^2] - mean^2 keeps
    // its digits when |mean| >> std, where float32 sums of x^2 lose them (float32 PARTIALS alone cost 2e-3 of the variance at
    // |mean| = 300 sigma)
    __shared__ double red[256 * 2 * L];
    const int cv = C / L, rpi = 256 / cv;
    const int cx = threadIdx.x % cv, ry = threadIdx.x / cv;
    const int rpt = bn_rpt(M, rpi);
    const long chunk = (long)rpi * rpt;
    double s[L], ss[L];
#pragma unroll
    for (int k = 0; k < L; k++) s[k] = ss[k] = 0.0;
    for (long row0 = (long)blockIdx.x * chunk; row0 < M; row0 += (long)gridDim.x * chunk) {
        // BN_BATCH rows' loads are issued together, with no branch between them (a row past the end re-reads the last row
        // and is weighted 0): row-at-a-time code with its bounds test waited for every load before issuing the next
#pragma unroll 1
        for (int it0 = 0; it0 < rpt; it0 += BN_BATCH) {
            float v[BN_BATCH][L], wgt[BN_BATCH];
#pragma unroll
            for (int u = 0; u < BN_BATCH; u++) {
                const long r = row0 + (long)(it0 + u) * rpi + ry;
                wgt[u] = r < M ? 1.f : 0.f;
                vec_io<V, L>::load(x + ((r < M ? r : M - 1) * cv + cx) * 16, v[u]);
            }
            // float32 over the batch's BN_BATCH rows, float64 across batches: a quarter of the float64 instructions (all-float64
            // made this HBM-bound pass 30 % slower) and still 2e-5 of the variance at |mean| = 300 sigma
#pragma unroll
            for (int k = 0; k < L; k++) {
                float ps = 0.f, pq = 0.f;
#pragma unroll
                for (int u = 0; u < BN_BATCH; u++) {
                    const float vk = v[u][k] * wgt[u];
                    ps += vk;
                    pq = fmaf(vk, vk, pq);
                }
                s[k] += (double)ps;
                ss[k] += (double)pq;
            }
        }
    }
    double *mine = part + (long)blockIdx.x * 2 * C;
    bn_block_reduce<L, double>(s, ss, red, cv, mine, mine + C, cx * L);
}

// training: mean / invstd from the blocks' partial sums, running statistics (momentum, unbiased variance); 4 channels per block
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double *__restrict__ part, int nblk, long M, int C, float eps,
                                                          float momentum, float *__restrict__ save_mean,
                                                          float *__restrict__ save_invstd, float *__restrict__ running_mean,
                                                          float *__restrict__ running_var, long long *__restrict__ batches_tracked)
{
    if (batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1; // nn.BatchNorm2d.num_batches_tracked
    __shared__ double red[256];
    bn_sum_partials4(part, nblk, C, blockIdx.x * 4, red);
    const int c = blockIdx.x * 4 + threadIdx.x;
    if (threadIdx.x >= 4 || c >= C) return;
    const double mean = red[threadIdx.x] / (double)M;
    double var = red[4 + threadIdx.x] / (double)M - mean * mean;
    if (var < 0) var = 0;
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// y = [relu]( ((x - mean) * invstd) * gamma + beta [+ residual] ).  Same thread geometry as the reductions: a thread keeps
// its column group's constants in registers and walks BN_APPLY_ROWS rows (one thread per vector would re-load 4 x L
// constants for every 16 bytes of payload and become instruction-bound).
// Dropout fused behind the ReLU: a counter-based hash of the element index decides what is kept, so the backward regenerates
// the mask instead of reading one.  One 32-bit hash serves two neighbouring elements (16 bits each).
template <int L> __device__ inline void drop_keep(long first_elem, DropArgs d, bool keep[L])
{
#pragma unroll
    for (int k = 0; k < L; k += 2) {
        const unsigned h = drop_hash((unsigned)(first_elem >> 1) + k / 2, d.seed);
        keep[k] = (h & 0xFFFFu) >= d.thresh;
        keep[k + 1] = (h >> 16) >= d.thresh;
    }
}

template <typename V, int L>
__global__ __launch_bounds__(256) void bn_apply_kernel(const char *__restrict__ x, char *__restrict__ y,
                                                       const char *__restrict__ residual, long M, int C,
                                                       const float *__restrict__ mean, const float *__restrict__ invstd,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                                                       DropArgs drop, unsigned char *__restrict__ live_bits = nullptr)
{
    const int cv = C / L, rpi = 256 / cv;
    const int cx = threadIdx.x % cv, ry = threadIdx.x / cv;
    float mu[L], is[L], ga[L], be[L];
#pragma unroll
    for (int k = 0; k < L; k++) {
        mu[k] = mean[cx * L + k];
        is[k] = invstd[cx * L + k];
        ga[k] = gamma[cx * L + k];
        be[k] = beta[cx * L + k];
    }
    const int nrows = bn_apply_rows(M, rpi);
    const long row0 = (long)blockIdx.x * rpi * nrows;
    // Rows in batches of four with ALL of a batch's loads issued before anything is converted (a row past the end re-reads the last
    // row and is not stored).  Written as load / compute / store per row with a `break` at the end of the map, every row was its own
    // memory round trip (tools/asm_serial_loads.py: four serialised loads in the loop): invisible on the large maps, where enough
    // waves are resident to hide it, and most of the time of the small ones (round 4).
    static_assert(BN_APPLY_ROWS_MAX % 4 == 0, "row batches");
    for (int it0 = 0; it0 < nrows; it0 += 4) {
        if (row0 + (long)it0 * rpi >= M) break; // (uniform per workgroup: the whole batch is past the end)
        typename vec_io<V, L>::raw_t rx[4], rr[4];
        long vecs[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long r = row0 + (long)(it0 + u) * rpi + ry;
            ok[u] = r < M;
            vecs[u] = (ok[u] ? r : M - 1) * cv + cx;
            rx[u] = vec_io<V, L>::load_raw(x + vecs[u] * 16);
            if (residual) rr[u] = vec_io<V, L>::load_raw(residual + vecs[u] * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float v[L], rs[L];
            vec_io<V, L>::convert(rx[u], v);
            if (residual) vec_io<V, L>::convert(rr[u], rs);
#pragma unroll
            for (int k = 0; k < L; k++) {
                float o = ((v[k] - mu[k]) * is[k]) * ga[k] + be[k];
                if (residual) o += rs[k];
                v[k] = relu ? fmaxf(o, 0.f) : o;
            }
            if (drop.thresh) {
                bool keep[L];
                drop_keep<L>(vecs[u] * L, drop, keep);
#pragma unroll
                for (int k = 0; k < L; k++) v[k] = keep[k] ? v[k] * drop.scale : 0.f;
            }
            if (ok[u]) {
                vec_io<V, L>::store(y + vecs[u] * 16, v);
                if (live_bits) { // training: one byte per vector, bit k = "output k is positive" -- what the backward passes otherwise read y for
                    unsigned bb = 0;
#pragma unroll
                    for (int k = 0; k < L; k++) bb |= (v[k] > 0.f ? 1u : 0u) << k;
                    live_bits[vecs[u]] = (unsigned char)bb;
                }
            }
        }
    }
}

// The stem's BatchNorm + ReLU is followed by a 2x2 average pool (models/model_utils.py:187-228): fused, the full-resolution
// activation (524 MB per step at 32 x 640 x 200 x 64) is neither written nor read back -- one thread = one POOLED pixel x L
// channels: four input vectors, normalise + ReLU each, average in float32, one rounding.  x: [N][H][W][C], y: [N][H/2][W/2][C].
template <typename V, int L>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const char *__restrict__ x, char *__restrict__ y, long n_vec, int H, int W,
                                                            int C, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const char *__restrict__ res /* like x, added before the ReLU; or NULL */,
                                                            unsigned char *__restrict__ live_bits = nullptr /* full resolution */)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_vec) return;
    const int cv = C / L, Ho = H / 2, Wo = W / 2;
    const int c = (int)(i % cv);
    long r = i / cv;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long n = r / Ho;
    const char *p = x + ((((n * H + 2 * ho) * W + 2 * wo) * cv + c) * 16);
    float v[4][L], o[L];
    vec_io<V, L>::load(p, v[0]);
    vec_io<V, L>::load(p + (long)cv * 16, v[1]);
    vec_io<V, L>::load(p + (long)W * cv * 16, v[2]);
    vec_io<V, L>::load(p + ((long)W * cv + cv) * 16, v[3]);
    float rv[4][L];
    if (res) {
        const char *q = res + (p - x);
        vec_io<V, L>::load(q, rv[0]);
        vec_io<V, L>::load(q + (long)cv * 16, rv[1]);
        vec_io<V, L>::load(q + (long)W * cv * 16, rv[2]);
        vec_io<V, L>::load(q + ((long)W * cv + cv) * 16, rv[3]);
    }
    unsigned bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < L; k++) {
        const float mu = mean[c * L + k], is = invstd[c * L + k], ga = gamma[c * L + k], be = beta[c * L + k];
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float pre = ((v[q][k] - mu) * is) * ga + be + (res ? rv[q][k] : 0.f); // (same order as the backward's mask)
            bits[q] |= (pre > 0.f ? 1u : 0u) << k;
            acc += fmaxf(pre, 0.f);
        }
        o[k] = acc * 0.25f;
    }
    vec_io<V, L>::store(y + i * 16, o);
    if (live_bits) { // one byte per full-resolution vector (see bn_apply_kernel)
        const long v0 = (p - x) / 16;
        live_bits[v0] = (unsigned char)bits[0];
        live_bits[v0 + cv] = (unsigned char)bits[1];
        live_bits[v0 + (long)W * cv] = (unsigned char)bits[2];
        live_bits[v0 + (long)W * cv + cv] = (unsigned char)bits[3];
    }
}

// gradient vector of input row r when dy is the POOLED gradient [N][H/2][W/2][C]: dy[pooled row] / 4, zero in a dropped odd
// row / column (F.avg_pool2d floor mode)
template <typename V, int L>
__device__ __forceinline__ void pooled_grad(const char *__restrict__ gp, long r, int cv, int cx, int H, int W, float (&g)[L])
{
    const int w = (int)(r % W);
    const long t = r / W;
    const int h = (int)(t % H);
    const long n = t / H;
    const int Ho = H / 2, Wo = W / 2;
    const bool inside = h < 2 * Ho && w < 2 * Wo;
    vec_io<V, L>::load(gp + ((((n * Ho + (inside ? h / 2 : 0)) * Wo + (inside ? w / 2 : 0)) * cv + cx) * 16), g);
#pragma unroll
    for (int k = 0; k < L; k++) g[k] = inside ? g[k] * 0.25f : 0.f;
}

// g = dy * (y > 0) ; sums[0][C] += sum g (= dbeta) ; sums[1][C] += sum g * xhat (= dgamma)
// MASK: 0 no ReLU, 1 ReLU mask from the stored output y, 2 recomputed from x, 3 recomputed from x and the residual that was
// added (passed in y's place: the fused-pool layers store no full-resolution output), 4 (round 4) from the bit plane the forward
// left (passed in y's place: one byte per vector instead of a 16-byte read of y / the residual); DROP: fused dropout (compile-time, so
// the element loop has no branches)
template <typename V, int L, int MASK, bool DROP, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const char *__restrict__ dy, const char *__restrict__ y,
                                                            const char *__restrict__ x, long M, int C,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            DropArgs drop, float *__restrict__ part, int PH = 0, int PW = 0)
{
    __shared__ float red[256 * 2 * L];
    const int cv = C / L, rpi = 256 / cv;
    const int cx = threadIdx.x % cv, ry = threadIdx.x / cv;
    const int rpt = bn_rpt(M, rpi);
    const long chunk = (long)rpi * rpt;
    float db[L], dg[L], mu[L], is[L], ga[L], be[L];
#pragma unroll
    for (int k = 0; k < L; k++) {
        db[k] = dg[k] = 0.f;
        mu[k] = mean[cx * L + k];
        is[k] = invstd[cx * L + k];
        ga[k] = gamma[cx * L + k];
        be[k] = beta[cx * L + k];
    }
    for (long row0 = (long)blockIdx.x * chunk; row0 < M; row0 += (long)gridDim.x * chunk) {
#pragma unroll 1
        for (int it0 = 0; it0 < rpt; it0 += BN_BATCH) {
            float g[BN_BATCH][L], xv[BN_BATCH][L], yv[BN_BATCH][L];
            long vec[BN_BATCH];
            bool ok[BN_BATCH];
            unsigned mb[BN_BATCH];
            typename vec_io<V, L>::raw_t rg[BN_BATCH], rx[BN_BATCH], ry_[BN_BATCH];
#pragma unroll
            for (int u = 0; u < BN_BATCH; u++) { // all of the batch's loads first (see bn_stats_kernel)
                const long r = row0 + (long)(it0 + u) * rpi + ry;
                ok[u] = r < M;
                vec[u] = (ok[u] ? r : M - 1) * cv + cx;
                // (raw loads here, conversions in a second loop: written as load + convert per vector, the MASK 4 instantiation
                // came out as load / wait / convert / load / wait ... -- four serialised round trips per batch)
                if (POOL) pooled_grad<V, L>(dy, ok[u] ? r : M - 1, cv, cx, PH, PW, g[u]); // dy = the pooled gradient
                else rg[u] = vec_io<V, L>::load_raw(dy + vec[u] * 16);
                rx[u] = vec_io<V, L>::load_raw(x + vec[u] * 16);
#ifndef BN_PROBE_NO_Y // (timing probe: what the backward passes would cost if the ReLU mask came from a bit plane instead of y)
                if (MASK == 1 || MASK == 3) ry_[u] = vec_io<V, L>::load_raw(y + vec[u] * 16); // (MASK 3: y is the RESIDUAL that was added)
#endif
                if (MASK == 4) mb[u] = ((const unsigned char *)y)[vec[u]]; // the forward's bit plane: 1 byte instead of 16
            }
#pragma unroll
            for (int u = 0; u < BN_BATCH; u++) {
                if (!POOL) vec_io<V, L>::convert(rg[u], g[u]);
                vec_io<V, L>::convert(rx[u], xv[u]);
#ifndef BN_PROBE_NO_Y
                if (MASK == 1 || MASK == 3) vec_io<V, L>::convert(ry_[u], yv[u]);
#else
                if (MASK == 1 || MASK == 3) for (int k = 0; k < L; k++) yv[u][k] = 1.f;
#endif
            }
#pragma unroll
            for (int u = 0; u < BN_BATCH; u++) {
                bool keep[L];
                if (DROP && MASK != 1 && MASK != 4) drop_keep<L>(vec[u] * L, drop, keep); // a stored y (or its bit plane) already holds the dropped zeros
#pragma unroll
                for (int k = 0; k < L; k++) {
                    const float xh = (xv[u][k] - mu[k]) * is[k];
                    // ReLU mask: from the stored output, or -- when nothing was added before the ReLU -- recomputed from x
                    // (the sign of the float32 pre-activation survives its rounding to the stored dtype), saving y's read
                    bool live = ok[u];
                    if (MASK == 1) live = live && yv[u][k] > 0.f;
                    if (MASK == 2) live = live && xh * ga[k] + be[k] > 0.f;
                    if (MASK == 3) live = live && xh * ga[k] + be[k] + yv[u][k] > 0.f;
                    if (MASK == 4) live = live && ((mb[u] >> k) & 1u);
                    if (DROP && MASK != 1 && MASK != 4) live = live && keep[k];
                    const float gk = live ? (DROP ? g[u][k] * drop.scale : g[u][k]) : 0.f;
                    db[k] += gk;
                    dg[k] = fmaf(gk, xh, dg[k]); // (explicit: every MASK instantiation must round alike -- bit planes on == off)
                }
            }
        }
    }
    float *mine = part + (long)blockIdx.x * 2 * C;
    bn_block_reduce<L>(db, dg, red, cv, mine, mine + C, cx * L);
}

// coefficients of dx = a * (g - b - (x - mean) * k): coef[0..3][C] = a = gamma*invstd, b = dbeta/M, mean, k = invstd*dgamma/M ;
// coef[4..5][C] = invstd, beta (for the mask recomputation)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float *__restrict__ part, int nblk, long M, int C,
                                                              const float *__restrict__ gamma, const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, const float *__restrict__ beta,
                                                              float *__restrict__ coef, float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta)
{
    __shared__ double red[256];
    bn_sum_partials4(part, nblk, C, blockIdx.x * 4, red);
    const int c = blockIdx.x * 4 + threadIdx.x;
    if (threadIdx.x >= 4 || c >= C) return;
    const double db = red[threadIdx.x], dg = red[4 + threadIdx.x];
    dbeta[c] = (float)db;
    dgamma[c] = (float)dg;
    coef[c] = gamma[c] * invstd[c];
    coef[C + c] = (float)(db / (double)M);
    coef[2 * C + c] = mean[c];
    coef[3 * C + c] = (float)((double)invstd[c] * dg / (double)M);
    coef[4 * C + c] = invstd[c];
    coef[5 * C + c] = beta[c];
    coef[6 * C + c] = gamma[c];
}

// One workgroup's rows of the backward element-wise pass.  YM (compile-time, so that nothing but loads sits between the loads of a
// batch): 0 = no second operand (mask from x, or none), 1 = y is a tensor like x (the stored output, or -- mask_from_x 2 -- the
// residual that was added), 3 = y is the forward's bit plane.
template <typename V, int L, bool POOL, int YM>
__device__ __forceinline__ void bn_bwd_apply_rows(const char *__restrict__ dy, const char *__restrict__ y, const char *__restrict__ x,
                                                  char *__restrict__ dx, char *__restrict__ dres, long M, int cv, int rpi, int cx, int ry,
                                                  const float (&a)[L], const float (&b)[L], const float (&mu)[L], const float (&kk)[L],
                                                  const float (&is)[L], const float (&be)[L], const float (&ga)[L], int mask_from_x,
                                                  const DropArgs &drop, int PH, int PW)
{
    const int nrows = bn_apply_rows(M, rpi);
    const long row0 = (long)blockIdx.x * rpi * nrows;
    // (rows in batches of four, loads first: see bn_apply_kernel)
    for (int it0 = 0; it0 < nrows; it0 += 4) {
        if (row0 + (long)it0 * rpi >= M) break;
        typename vec_io<V, L>::raw_t rg[4], rx[4], ryv[4];
        float gp[4][L];
        unsigned mbv[4] = {0u, 0u, 0u, 0u};
        long vecs[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long r = row0 + (long)(it0 + u) * rpi + ry;
            ok[u] = r < M;
            const long rc = ok[u] ? r : M - 1;
            vecs[u] = rc * cv + cx;
            if (POOL) pooled_grad<V, L>(dy, rc, cv, cx, PH, PW, gp[u]);
            else rg[u] = vec_io<V, L>::load_raw(dy + vecs[u] * 16);
            rx[u] = vec_io<V, L>::load_raw(x + vecs[u] * 16);
            if (YM == 3) mbv[u] = ((const unsigned char *)y)[vecs[u]];
#ifndef BN_PROBE_NO_Y
            if (YM == 1) ryv[u] = vec_io<V, L>::load_raw(y + vecs[u] * 16);
#endif
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float g[L], xv[L], yv[L];
            if (POOL) {
#pragma unroll
                for (int k = 0; k < L; k++) g[k] = gp[u][k];
            } else vec_io<V, L>::convert(rg[u], g);
            vec_io<V, L>::convert(rx[u], xv);
#ifndef BN_PROBE_NO_Y
            if (YM == 1) vec_io<V, L>::convert(ryv[u], yv);
#else
            if (YM == 1) for (int k = 0; k < L; k++) yv[k] = 1.f;
#endif
            if (drop.thresh) {
                bool keep[L];
                if (YM == 0 || mask_from_x == 2) drop_keep<L>(vecs[u] * L, drop, keep);
#pragma unroll
                for (int k = 0; k < L; k++) g[k] = ((YM != 0 && mask_from_x != 2) || keep[k]) ? g[k] * drop.scale : 0.f;
            }
#pragma unroll
            for (int k = 0; k < L; k++) {
                const float xc = xv[k] - mu[k];
                bool zero;
                if (YM == 3) zero = !((mbv[u] >> k) & 1u);
                else if (YM == 1) zero = mask_from_x == 2 ? !((xc * is[k]) * ga[k] + be[k] + yv[k] > 0.f) // y = the residual that was added
                                                          : !(yv[k] > 0.f);
                else zero = mask_from_x && !((xc * is[k]) * ga[k] + be[k] > 0.f);
                if (zero) g[k] = 0.f;
                xv[k] = a[k] * (g[k] - b[k] - xc * kk[k]);
            }
            if (ok[u]) {
                vec_io<V, L>::store(dx + vecs[u] * 16, xv);
                if (dres) vec_io<V, L>::store(dres + vecs[u] * 16, g);
            }
        }
    }
}

template <typename V, int L, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const char *__restrict__ dy, const char *__restrict__ y,
                                                           const char *__restrict__ x, char *__restrict__ dx,
                                                           char *__restrict__ dres, long M, int C,
                                                           const float *__restrict__ coef, int mask_from_x, DropArgs drop,
                                                           int PH = 0, int PW = 0)
{
    const int cv = C / L, rpi = 256 / cv;
    const int cx = threadIdx.x % cv, ry = threadIdx.x / cv;
    float a[L], b[L], mu[L], kk[L], is[L], be[L], ga[L];
#pragma unroll
    for (int k = 0; k < L; k++) {
        const int c = cx * L + k;
        a[k] = coef[c];
        b[k] = coef[C + c];
        mu[k] = coef[2 * C + c];
        kk[k] = coef[3 * C + c];
        is[k] = coef[4 * C + c];
        be[k] = coef[5 * C + c];
        ga[k] = coef[6 * C + c];
    }
    if (mask_from_x == 3) bn_bwd_apply_rows<V, L, POOL, 3>(dy, y, x, dx, dres, M, cv, rpi, cx, ry, a, b, mu, kk, is, be, ga, mask_from_x, drop, PH, PW);
    else if (y) bn_bwd_apply_rows<V, L, POOL, 1>(dy, y, x, dx, dres, M, cv, rpi, cx, ry, a, b, mu, kk, is, be, ga, mask_from_x, drop, PH, PW);
    else bn_bwd_apply_rows<V, L, POOL, 0>(dy, y, x, dx, dres, M, cv, rpi, cx, ry, a, b, mu, kk, is, be, ga, mask_from_x, drop, PH, PW);
}

// ------------------------------------------------------------------------------------------------ convolution filter bank
// Every step the hand-written convolutions need each float32 master filter (Cout, Cin, 3, 3) as bf16 in two layouts: the
// forward kernel's [Cout][3][3][Cin] and the data-gradient kernel's flipped / transposed [Cin][3][3][Cout].  Through torch
// that was four small kernels per layer per step (cast, permute copy, flip, permute copy: 68 launches, ~0.5 ms); here ONE
// launch does all layers.  A block = one layer's 32 x 32 (co, ci) tile with its nine taps: the 32 x 288 floats go through LDS
// (row pitch 289: both read patterns below are bank-conflict-free) and leave as 64-byte runs in either layout.
// desc: 11 x int64 per layer = src, fwd, bwd pointers, Cout, Cin, element strides of src (co, ci, ky, kx), first block, taps
// (9, or 1 for the 1x1 shortcut filters: forward [Cout][Cin], data-gradient [Cin][Cout]).
__global__ __launch_bounds__(256) void conv_filter_bank_kernel(const long long *__restrict__ desc, int n_layers)
{
    __shared__ float tile[32][289];
    int l = 0;
    while (l + 1 < n_layers && (long long)blockIdx.x >= desc[(l + 1) * 11 + 9]) l++;
    const long long *d = desc + l * 11;
    const float *src = (const float *)d[0];
    __hip_bfloat16 *fwd = (__hip_bfloat16 *)d[1], *bwd = (__hip_bfloat16 *)d[2];
    const int Cout = (int)d[3], Cin = (int)d[4], taps = (int)d[10], rw = 32 * taps; // taps = 9 or 1
    const long s_co = d[5], s_ci = d[6], s_ky = d[7], s_kx = d[8];
    const int t = (int)(blockIdx.x - d[9]), tiles_ci = Cin / 32, co0 = (t / tiles_ci) * 32, ci0 = (t % tiles_ci) * 32;
    if (s_ci == 1 && !(s_co & 3) && !(s_ky & 3) && !(s_kx & 3)) { // channels-last master filter (the trainer's): 16-byte loads along ci
        for (int e = threadIdx.x; e < 8 * rw; e += 256) {
            const int co = e / (8 * taps), q = e % (8 * taps), ci = (q % 8) * 4, tap = q / 8;
            const float4 v = *(const float4 *)(src + (co0 + co) * s_co + (ci0 + ci) + (tap / 3) * s_ky + (tap % 3) * s_kx);
            tile[co][ci * taps + tap] = v.x;
            tile[co][(ci + 1) * taps + tap] = v.y;
            tile[co][(ci + 2) * taps + tap] = v.z;
            tile[co][(ci + 3) * taps + tap] = v.w;
        }
    } else {
        for (int e = threadIdx.x; e < 32 * rw; e += 256) { // consecutive threads walk the source's contiguous axis: ci when the
            // master filter is channels-last (s_ci == 1), the taps when it is torch's default (co, ci, ky, kx)
            const int co = e / rw, q = e % rw;
            const int ci = s_ci == 1 ? q % 32 : q / taps, tap = s_ci == 1 ? q / 32 : q % taps;
            tile[co][ci * taps + tap] = src[(co0 + co) * s_co + (ci0 + ci) * s_ci + (tap / 3) * s_ky + (tap % 3) * s_kx];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * rw; e += 256) { // two neighbouring channels per thread: 4-byte stores, 64-byte runs per 16 lanes
        const int a = (e % 16) * 2, tap = (e / 16) % taps, b = e / (16 * taps);
        *(unsigned *)(fwd + ((long)(co0 + b) * taps + tap) * Cin + ci0 + a) = pack_bf16(tile[b][a * taps + tap], tile[b][(a + 1) * taps + tap]); // (co = b, ci = a)
        if (bwd)
            *(unsigned *)(bwd + ((long)(ci0 + b) * taps + (taps - 1 - tap)) * Cout + co0 + a) = pack_bf16(tile[a][b * taps + tap], tile[a + 1][b * taps + tap]); // (ci = b, co = a)
    }
}

// ------------------------------------------------------------------------------------------------------------ frequency mean
// The decoder's first step (models/decoders.py: x.mean(dim=3), then (B, C, T) -> (B, T, C)): x bf16 channels-last [N][H][W][C]
// -> float32 [H][N][C] (time-major, the GRU's order) or [N][H][C]; torch's reduction over the strided axis ran at 0.4 TB/s.
__global__ __launch_bounds__(256) void freq_mean_fwd_kernel(const unsigned short *__restrict__ x, float *__restrict__ y, long rows,
                                                            int N, int H, int W, int C, int time_major)
{
    const int cv = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cv) return;
    const long r = i / cv;
    const int c8 = (int)(i - r * cv);
    const uint4 *p = (const uint4 *)(x + (r * W) * C + c8 * 8);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < W; w++) {
        const uint4 v = p[(long)w * cv];
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            acc[2 * k] += __uint_as_float(u[k] << 16);
            acc[2 * k + 1] += __uint_as_float(u[k] & 0xFFFF0000u);
        }
    }
    const float inv = 1.f / (float)W;
    const long n = r / H, h = r - n * H;
    float *o = y + ((time_major ? h * N + n : r) * C + c8 * 8);
    *(float4 *)o = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    *(float4 *)(o + 4) = make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv);
}

// dx[n][h][w][c] = g[row(n, h)][c] / W, bf16 channels-last
__global__ __launch_bounds__(256) void freq_mean_bwd_kernel(const float *__restrict__ g, unsigned short *__restrict__ dx, long rows,
                                                            int N, int H, int W, int C, int time_major)
{
    const int cv = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * W * cv) return;
    const long rw = i / cv;
    const int c8 = (int)(i - rw * cv);
    const long r = rw / W;
    const long n = r / H, h = r - n * H;
    const float *src = g + ((time_major ? h * N + n : r) * C + c8 * 8);
    const float4 a = *(const float4 *)src, b = *(const float4 *)(src + 4);
    const float inv = 1.f / (float)W;
    uint4 o;
    o.x = pack_bf16(a.x * inv, a.y * inv);
    o.y = pack_bf16(a.z * inv, a.w * inv);
    o.z = pack_bf16(b.x * inv, b.y * inv);
    o.w = pack_bf16(b.z * inv, b.w * inv);
    *(uint4 *)(dx + rw * C + c8 * 8) = o;
}

// ------------------------------------------------------------------------------------------------ deterministic reductions
// dw[i] += sum over the slabs of ws[slab][i], in a FIXED order (what makes the result reproducible; it need not be slab order): a
// workgroup owns 4 * 256 / SL consecutive elements and its SL slab lanes each add every SL-th slab in ascending order, then the SL
// partial sums are added in lane order.  (First version: one thread per four elements walking all slabs -- 4 to 36
// workgroups for the first layer's 1280 slabs and the 64 -> 64 layers' 384: 138 / 33 us per call.)
template <int SL> // slab lanes per workgroup (16: many slabs -- the 64 -> 64 and first-layer gradients; 4: the wide layers' 4 - 64 slabs)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ ws, int slabs, long n, float *__restrict__ dw)
{
    constexpr int EX = 256 / SL;
    __shared__ float4 red[SL][EX];
    const int ex = threadIdx.x % EX, sl = threadIdx.x / EX;
    const long i = ((long)blockIdx.x * EX + ex) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i + 3 < n) {
        int s = sl;
        for (; s + 3 * SL < slabs; s += 4 * SL) { // four loads in flight; the additions keep their order
            const float4 a = *(const float4 *)(ws + (long)s * n + i), b = *(const float4 *)(ws + (long)(s + SL) * n + i);
            const float4 c = *(const float4 *)(ws + (long)(s + 2 * SL) * n + i), d = *(const float4 *)(ws + (long)(s + 3 * SL) * n + i);
            acc.x = (((acc.x + a.x) + b.x) + c.x) + d.x; acc.y = (((acc.y + a.y) + b.y) + c.y) + d.y;
            acc.z = (((acc.z + a.z) + b.z) + c.z) + d.z; acc.w = (((acc.w + a.w) + b.w) + c.w) + d.w;
        }
        for (; s < slabs; s += SL) {
            const float4 a = *(const float4 *)(ws + (long)s * n + i);
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
    } else if (i < n) { // (n not a multiple of 4: the last elements one by one)
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = sl; s < slabs; s += SL)
            for (int k = 0; k < 4 && i + k < n; k++) t[k] += ws[(long)s * n + i + k];
        acc = make_float4(t[0], t[1], t[2], t[3]);
    }
    red[sl][ex] = acc;
    __syncthreads();
    if (sl == 0 && i < n) {
        float4 t = red[0][ex];
#pragma unroll
        for (int k = 1; k < SL; k++) {
            const float4 v = red[k][ex];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        const float o[4] = {t.x, t.y, t.z, t.w};
        for (int k = 0; k < 4 && i + k < n; k++) dw[i + k] += o[k];
    }
}

__global__ __launch_bounds__(256) void slab_reduce_strided_kernel(const float *__restrict__ ws, int slabs, int C, float *__restrict__ oa,
                                                                  float *__restrict__ ob)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float acc = 0.f;
    for (int s = 0; s < slabs; s++) acc += ws[((long)s * 2 + blockIdx.y) * C + c];
    (blockIdx.y ? ob : oa)[c] += acc;
}

// out[c] += sum over rows of x[row][c] for up to two float32 matrices of one shape (the GRU's two bias gradients: rows = T * B,
// C = D * 3H); out must hold zeros.  Workgroup = 64 columns x 4 row lanes over a slice of the rows; one float atomic per column.
__global__ __launch_bounds__(256) void colsum2_kernel(const float *__restrict__ xa, const float *__restrict__ xb, float *__restrict__ oa,
                                                      float *__restrict__ ob, long M, int C, int rows_per_block,
                                                      float *__restrict__ part /* deterministic mode: [gridDim.y][2][C] */)
{
    __shared__ float red[4][64];
    const float *x = blockIdx.z ? xb : xa;
    float *o = blockIdx.z ? ob : oa;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float acc = 0.f;
    if (col < C) {
        long r = r0 + rl;
        for (; r + 12 < r1; r += 16) { // four rows in flight
            const float v0 = x[r * C + col], v1 = x[(r + 4) * C + col], v2 = x[(r + 8) * C + col], v3 = x[(r + 12) * C + col];
            acc += (v0 + v1) + (v2 + v3);
        }
        for (; r < r1; r += 4) acc += x[r * C + col];
    }
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && col < C) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (part) part[((long)blockIdx.y * 2 + blockIdx.z) * C + col] = v;
        else atomicAdd(o + col, v);
    }
}

// ------------------------------------------------------------------------------------------------------------ SELD loss
// models/interfaces.py:304-355: sed = mean BCE-with-logits(logit, sed_gt); doa = sum over the x / y / z blocks of
// sum(|p - t| m) / sum(m) with m = sed_gt (the three blocks share sum(m)); loss = w_sed sed + w_doa doa -- and the gradients of
// sed w.r.t. logit and of doa w.r.t. the predictions (unweighted: the backward launch scales them by what flows in).
// Two launches of SELD_BLOCKS workgroups (a single 1024-thread workgroup took 64 us for the 123 k elements of a training step):
// partial sums per workgroup in float64, fixed order = deterministic; the second launch adds the partials (every workgroup for
// itself), writes the three values and the gradients of its slice.
constexpr int SELD_BLOCKS = 64;
__global__ __launch_bounds__(256) void seld_loss_partial_kernel(const float *__restrict__ logit, const float *__restrict__ doa,
                                                                const float *__restrict__ sed_gt, const float *__restrict__ doa_gt,
                                                                long rows, int nc, double *__restrict__ partial /* [SELD_BLOCKS][3] */)
{
    __shared__ double red[3][256];
    const long n = rows * nc;
    double s_bce = 0.0, s_mask = 0.0, s_abs = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)SELD_BLOCKS * 256) {
        const long r = i / nc;
        const int c = (int)(i - r * nc);
        const float x = logit[i], z = sed_gt[i];
        // max(x, 0) - x z + log1p(exp(-|x|))  (torch's binary_cross_entropy_with_logits)
        s_bce += (double)(fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))));
        s_mask += (double)z;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) a += fabsf(doa[r * 3 * nc + k * nc + c] - doa_gt[r * 3 * nc + k * nc + c]) * z;
        s_abs += (double)a;
    }
    red[0][threadIdx.x] = s_bce;
    red[1][threadIdx.x] = s_mask;
    red[2][threadIdx.x] = s_abs;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int k = 0; k < 3; k++) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void seld_loss_finish_kernel(const float *__restrict__ logit, const float *__restrict__ doa,
                                                               const float *__restrict__ sed_gt, const float *__restrict__ doa_gt,
                                                               long rows, int nc, float w_sed, float w_doa,
                                                               const double *__restrict__ partial, float *__restrict__ out3,
                                                               float *__restrict__ g_logit, float *__restrict__ g_doa)
{
    __shared__ double tot[3];
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int b = 0; b < SELD_BLOCKS; b++) t += partial[b * 3 + threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    const long n = rows * nc;
    const float sed = (float)(tot[0] / (double)n);
    const float msum = (float)tot[1];
    const float d = (float)tot[2] / msum; // 0 / 0 = nan when no class is active anywhere, as in the reference
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out3[0] = w_sed * sed + w_doa * d;
        out3[1] = sed;
        out3[2] = d;
    }
    const float inv_n = 1.f / (float)n, inv_m = 1.f / msum;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)SELD_BLOCKS * 256) {
        const long r = i / nc;
        const int c = (int)(i - r * nc);
        const float x = logit[i], z = sed_gt[i];
        g_logit[i] = (1.f / (1.f + expf(-x)) - z) * inv_n;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const long j = r * 3 * nc + k * nc + c;
            const float e = doa[j] - doa_gt[j];
            g_doa[j] = (e > 0.f ? 1.f : e < 0.f ? -1.f : 0.f) * z * inv_m;
        }
    }
}

// backward: g_logit *= g_loss w_sed + g_sed ; g_doa *= g_loss w_doa + g_doa_loss   (the incoming gradients are device scalars
// or NULL = 0), out of place
__global__ __launch_bounds__(256) void seld_loss_scale_kernel(const float *__restrict__ a, long na, const float *__restrict__ b,
                                                              long nb, const float *__restrict__ g_loss,
                                                              const float *__restrict__ g_sed, const float *__restrict__ g_d,
                                                              float w_sed, float w_doa, float *__restrict__ oa,
                                                              float *__restrict__ ob)
{
    const float gl = g_loss ? *g_loss : 0.f;
    const float fa = gl * w_sed + (g_sed ? *g_sed : 0.f), fb = gl * w_doa + (g_d ? *g_d : 0.f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (long)gridDim.x * 256) {
        if (i < na) oa[i] = a[i] * fa;
        else ob[i - na] = b[i - na] * fb;
    }
}

} // namespace

// One workspace PER DEVICE (round-4 advice: a single process-global pointer, re-pointed by whichever device ran a forward last,
// made the backward kernels of device A write their slabs through device B's pointer in any single-process multi-device pattern).
// salsa_nn_set_deterministic registers the workspace for the device that is current at the call; the kernels' launchers look up
// the device current at THEIR call, so a device without a workspace keeps the atomics instead of borrowing a neighbour's.
constexpr int DET_MAX_DEV = 64;
static float *g_det_ws[DET_MAX_DEV] = {};
static size_t g_det_bytes[DET_MAX_DEV] = {};
static int det_device()
{
    int d = -1;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < DET_MAX_DEV) ? d : -1;
}
extern "C" int salsa_nn_set_deterministic(void *ws, size_t bytes)
{
    if (!ws) { // off, everywhere
        for (int i = 0; i < DET_MAX_DEV; i++) g_det_ws[i] = nullptr, g_det_bytes[i] = 0;
        return 0;
    }
    const int d = det_device();
    if (d < 0) return -6;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, ws) == hipSuccess && at.type == hipMemoryTypeDevice && at.device != d) return -1; // not this device's memory
    g_det_ws[d] = (float *)ws;
    g_det_bytes[d] = bytes;
    return 0;
}
extern "C" int salsa_nn_get_deterministic(void)
{
    const int d = det_device();
    return d >= 0 && g_det_ws[d] != nullptr;
}
float *salsa_nn_det_begin(int slabs, long n, hipStream_t st, int *rc)
{
    *rc = 0;
    const int d = det_device();
    if (d < 0 || !g_det_ws[d]) return nullptr;
    const size_t need = (size_t)slabs * (size_t)n * sizeof(float);
    if (need > g_det_bytes[d]) {
        *rc = -5;
        return nullptr;
    }
    // (no clearing pass: every kernel's workgroups cover every element of their slabs -- grids are sized so that no workgroup is
    // idle -- and the GPU test runs on a NaN-poisoned workspace; a 75-MB memset per call was most of this mode's cost)
    (void)st;
    return g_det_ws[d];
}
int salsa_nn_det_finish(const float *ws, int slabs, long n, float *dw, hipStream_t st)
{
    if (slabs > 16) hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, ws, slabs, n, dw);
    else hipLaunchKernelGGL(slab_reduce_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, slabs, n, dw);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

extern "C" {

int salsa_nn_avgpool2x2_fwd(const void *x, void *y, int dtype, int64_t N, int H, int W, int C, void *hip_stream)
{
    const int L = dtype == 1 ? 8 : 4;
    if (!x || !y || N <= 0 || H < 2 || W < 2 || C <= 0 || C % L || (dtype != 0 && dtype != 1)) return -1;
    const long n_vec = (long)N * (H / 2) * (W / 2) * (C / L);
    const dim3 grid((unsigned)((n_vec + 255) / 256));
    if (dtype == 1)
        hipLaunchKernelGGL((avgpool2x2_fwd_kernel<bf16x8, 8>), grid, dim3(256), 0, (hipStream_t)hip_stream, (const char *)x, (char *)y, n_vec, H, W, C);
    else
        hipLaunchKernelGGL((avgpool2x2_fwd_kernel<f32x4, 4>), grid, dim3(256), 0, (hipStream_t)hip_stream, (const char *)x, (char *)y, n_vec, H, W, C);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_nn_avgpool2x2_bwd(const void *grad_y, void *grad_x, int dtype, int64_t N, int H, int W, int C, void *hip_stream)
{
    const int L = dtype == 1 ? 8 : 4;
    if (!grad_y || !grad_x || N <= 0 || H < 2 || W < 2 || C <= 0 || C % L || (dtype != 0 && dtype != 1)) return -1;
    const long n_vec = (long)N * H * W * (C / L);
    const dim3 grid((unsigned)((n_vec + 255) / 256));
    if (dtype == 1)
        hipLaunchKernelGGL((avgpool2x2_bwd_kernel<bf16x8, 8>), grid, dim3(256), 0, (hipStream_t)hip_stream, (const char *)grad_y, (char *)grad_x, n_vec, H, W, C);
    else
        hipLaunchKernelGGL((avgpool2x2_bwd_kernel<f32x4, 4>), grid, dim3(256), 0, (hipStream_t)hip_stream, (const char *)grad_y, (char *)grad_x, n_vec, H, W, C);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

static int bn_geometry_ok(int dtype, int64_t M, int C)
{
    const int L = dtype == 1 ? 8 : 4;
    if ((dtype != 0 && dtype != 1) || M <= 0 || C <= 0 || C % L) return 0;
    const int cv = C / L;
    if ((int64_t)M * C >= ((int64_t)1 << 32)) return 0; // element offsets are 32-bit unsigned inside the kernels
    return cv <= 256 && (cv & (cv - 1)) == 0; // a block is (256/cv) rows x cv column groups
}
#ifndef BN_MAX_BLOCKS
#define BN_MAX_BLOCKS 1024 // (2048: the reductions run no faster -- 0.203 vs 0.202 ms on 32 x 640 x 200 x 64 -- and the finalize
#endif                     // kernels that sum the partial rows take twice as long there, 0.026 / 0.030 vs 0.015 / 0.014 ms; round 3)
constexpr unsigned BN_MAX_REDUCE_BLOCKS = BN_MAX_BLOCKS; // 4 per CU; a block takes several row chunks when M is large
static unsigned bn_reduce_blocks(int dtype, int64_t M, int C)
{
    const int rpi = 256 / (C / (dtype == 1 ? 8 : 4));
    const long per = (long)rpi * bn_rpt(M, rpi);
    const long n = (M + per - 1) / per;
    return (unsigned)(n < BN_MAX_REDUCE_BLOCKS ? n : BN_MAX_REDUCE_BLOCKS);
}
static unsigned bn_apply_blocks(int dtype, int64_t M, int C)
{
    const int rpi = 256 / (C / (dtype == 1 ? 8 : 4));
    const long per = (long)rpi * bn_apply_rows(M, rpi);
    return (unsigned)((M + per - 1) / per);
}
#define NN_LAUNCH(KERNEL, grid, block, ...)                                                               \
    do {                                                                                                  \
        if (dtype == 1) hipLaunchKernelGGL((KERNEL<bf16x8, 8>), grid, block, 0, st, __VA_ARGS__);          \
        else hipLaunchKernelGGL((KERNEL<f32x4, 4>), grid, block, 0, st, __VA_ARGS__);                      \
    } while (0)

int salsa_nn_bn_supported(int dtype, int64_t M, int C) { return bn_geometry_ok(dtype, M, C); }
/* bytes of the sums_ws scratch: 2*C float64 sums + one partial pair per reduction block (float64 forward, float32 backward) */
size_t salsa_nn_bn_workspace_bytes(int dtype, int64_t M, int C)
{
    if (!bn_geometry_ok(dtype, M, C)) return 0;
    return sizeof(double) * 2 * C + sizeof(double) * 2 * (size_t)C * bn_reduce_blocks(dtype, M, C);
}

static int bn_train_fwd_impl(const void *x, void *y, const void *residual, int dtype, int64_t M, int C, const float *gamma,
                             const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                             float *save_mean, float *save_invstd, double *sums_ws, int relu, float drop_p, uint32_t drop_seed,
                             int64_t *batches_tracked, const double *stats_part, int stats_blocks, void *live_bits, void *hip_stream)
{
    if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || !sums_ws || !bn_geometry_ok(dtype, M, C) || drop_p < 0.f ||
        drop_p >= 1.f || (int64_t)M * C >= ((int64_t)1 << 32))
        return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    const unsigned nblk = bn_reduce_blocks(dtype, M, C);
    const double *part = sums_ws + 2 * C; // [nblk][2][C] float64 partial sums
    int nparts = (int)nblk;
    if (stats_part && stats_blocks > 0) { // the producing convolution already left them (salsa_nn_conv3x3_c64_stats)
        part = stats_part;
        nparts = stats_blocks;
    } else {
        NN_LAUNCH(bn_stats_kernel, dim3(nblk), dim3(256), (const char *)x, (long)M, C, sums_ws + 2 * C);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nparts, (long)M, C, eps, momentum, save_mean,
                       save_invstd, running_mean, running_var, (long long *)batches_tracked);
    NN_LAUNCH(bn_apply_kernel, dim3(bn_apply_blocks(dtype, M, C)), dim3(256), (const char *)x, (char *)y,
              (const char *)residual, (long)M, C, save_mean, save_invstd, gamma, beta, relu, drop_args(drop_p, drop_seed),
              (unsigned char *)live_bits);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_nn_bn_train_fwd(const void *x, void *y, const void *residual, int dtype, int64_t M, int C, const float *gamma,
                          const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                          float *save_mean, float *save_invstd, double *sums_ws, int relu, float drop_p, uint32_t drop_seed,
                          int64_t *batches_tracked, const double *stats_part, int stats_blocks, void *hip_stream)
{
    return bn_train_fwd_impl(x, y, residual, dtype, M, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd,
                             sums_ws, relu, drop_p, drop_seed, batches_tracked, stats_part, stats_blocks, nullptr, hip_stream);
}

/* The same, also leaving live_bits[M * C / L] (L = 8 bf16 | 4 float32 elements per byte: bit k of byte i = "output element L i + k is
 * positive"): the ReLU mask for salsa_nn_bn_bwd(relu = 2), which then reads one byte where it read 16 of y (round 4). */
int salsa_nn_bn_train_fwd_bits(const void *x, void *y, const void *residual, int dtype, int64_t M, int C, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                               float *save_mean, float *save_invstd, double *sums_ws, int relu, float drop_p, uint32_t drop_seed,
                               int64_t *batches_tracked, const double *stats_part, int stats_blocks, void *live_bits, void *hip_stream)
{
    if (!live_bits || !relu) return -1;
    return bn_train_fwd_impl(x, y, residual, dtype, M, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd,
                             sums_ws, relu, drop_p, drop_seed, batches_tracked, stats_part, stats_blocks, live_bits, hip_stream);
}

/* Training statistics ONLY (round 4): mean / invstd, running statistics and the batch count from the per-workgroup partial rows
 * the producing convolution left (stats_part[stats_blocks][2][C] float64) -- no normalisation pass: the CONSUMER of the activation
 * applies relu(bn(x)) (+ dropout) while it stages its operand (salsa_nn_conv3x3_c64_xform, salsa_nn_conv3x3_c64_wrw_xform). */
int salsa_nn_bn_train_finalize(const double *stats_part, int stats_blocks, int64_t M, int C, float eps, float momentum,
                               float *running_mean, float *running_var, float *save_mean, float *save_invstd,
                               int64_t *batches_tracked, void *hip_stream)
{
    if (!stats_part || stats_blocks <= 0 || M <= 0 || C <= 0 || !save_mean || !save_invstd) return -1;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)hip_stream, stats_part, stats_blocks, (long)M, C, eps,
                       momentum, save_mean, save_invstd, running_mean, running_var, (long long *)batches_tracked);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_nn_bn_eval_fwd(const void *x, void *y, const void *residual, int dtype, int64_t M, int C, const float *gamma,
                         const float *beta, const float *mean, const float *invstd, int relu, void *hip_stream)
{
    if (!x || !y || !gamma || !beta || !mean || !invstd || !bn_geometry_ok(dtype, M, C)) return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    NN_LAUNCH(bn_apply_kernel, dim3(bn_apply_blocks(dtype, M, C)), dim3(256), (const char *)x, (char *)y,
              (const char *)residual, (long)M, C, mean, invstd, gamma, beta, relu, drop_args(0.f, 0u));
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_nn_bn_bwd(const void *dy, const void *y_or_null, const void *x, void *dx, void *dres_or_null, int dtype, int64_t M,
                    int C, const float *gamma, const float *beta, const float *save_mean, const float *save_invstd, int relu,
                    float *dgamma, float *dbeta, double *sums_ws, float *coef_ws, float drop_p, uint32_t drop_seed, void *hip_stream)
{
    const bool bits = relu == 2; // y_or_null is the forward's bit plane (salsa_nn_bn_train_fwd_bits)
    if (bits && !y_or_null) return -1;
    const int mask_from_x = bits ? 3 : (relu && !y_or_null);
    const DropArgs drop = drop_args(drop_p, drop_seed);
    if (drop_p < 0.f || drop_p >= 1.f || (int64_t)M * C >= ((int64_t)1 << 32)) return -1;
    // dx == NULL: only dgamma, dbeta and the coefficient table coef_ws[7][C] are produced, for a consumer that forms dx on the fly
    // (salsa_nn_conv3x3_stem_wrw_bn: the first layer's weight gradient is dx's only reader)
    if (!dy || !x || (!dx && dres_or_null) || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !sums_ws || !coef_ws ||
        !bn_geometry_ok(dtype, M, C))
        return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    const unsigned nblk = bn_reduce_blocks(dtype, M, C);
    float *part = (float *)(sums_ws + 2 * C);
#define BN_REDUCE(MASK, DROP)                                                                                                   \
    do {                                                                                                                        \
        if (dtype == 1)                                                                                                         \
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16x8, 8, MASK, DROP>), dim3(nblk), dim3(256), 0, st, (const char *)dy,   \
                               (const char *)y_or_null, (const char *)x, (long)M, C, save_mean, save_invstd, gamma, beta, drop, \
                               part);                                                                                           \
        else                                                                                                                    \
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<f32x4, 4, MASK, DROP>), dim3(nblk), dim3(256), 0, st, (const char *)dy,    \
                               (const char *)y_or_null, (const char *)x, (long)M, C, save_mean, save_invstd, gamma, beta, drop, \
                               part);                                                                                           \
    } while (0)
    const int mask_mode = !relu ? 0 : bits ? 4 : (y_or_null ? 1 : 2);
    if (drop.thresh) {
        if (mask_mode == 0) BN_REDUCE(0, true);
        else if (mask_mode == 1) BN_REDUCE(1, true);
        else if (mask_mode == 4) BN_REDUCE(4, true);
        else BN_REDUCE(2, true);
    } else {
        if (mask_mode == 0) BN_REDUCE(0, false);
        else if (mask_mode == 1) BN_REDUCE(1, false);
        else if (mask_mode == 4) BN_REDUCE(4, false);
        else BN_REDUCE(2, false);
    }
#undef BN_REDUCE
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, (int)nblk, (long)M, C, gamma, save_mean,
                       save_invstd, beta, coef_ws, dgamma, dbeta);
    if (dx)
        NN_LAUNCH(bn_bwd_apply_kernel, dim3(bn_apply_blocks(dtype, M, C)), dim3(256), (const char *)dy, (const char *)y_or_null,
                  (const char *)x, (char *)dx, (char *)dres_or_null, (long)M, C, coef_ws, mask_from_x, drop);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

/* BatchNorm (training) + ReLU + 2x2 average pool in one pass over x (the stem's tail, models/model_utils.py:187-228):
 * x [N][H][W][C] -> y [N][H/2][W/2][C]; statistics over all N*H*W rows like salsa_nn_bn_train_fwd. */
static int bn_train_fwd_pool_impl(const void *x, void *y, const void *residual, int dtype, int64_t N, int H, int W, int C,
                                  const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                  float *running_var, float *save_mean, float *save_invstd, double *sums_ws, int64_t *batches_tracked,
                                  const double *stats_part, int stats_blocks, void *live_bits, void *hip_stream)
{
    const int64_t M = N * H * W;
    if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || !sums_ws || N <= 0 || H < 2 || W < 2 || !bn_geometry_ok(dtype, M, C))
        return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    const unsigned nblk = bn_reduce_blocks(dtype, M, C);
    const double *part = sums_ws + 2 * C; // [nblk][2][C] float64 partial sums
    int nparts = (int)nblk;
    if (stats_part && stats_blocks > 0) { // the producing convolution already left them (salsa_nn_conv3x3_c64_stats)
        part = stats_part;
        nparts = stats_blocks;
    } else {
        NN_LAUNCH(bn_stats_kernel, dim3(nblk), dim3(256), (const char *)x, (long)M, C, sums_ws + 2 * C);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nparts, (long)M, C, eps, momentum, save_mean,
                       save_invstd, running_mean, running_var, (long long *)batches_tracked);
    const long n_vec = (long)N * (H / 2) * (W / 2) * (C / (dtype == 1 ? 8 : 4));
    NN_LAUNCH(bn_apply_pool_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), (const char *)x, (char *)y, n_vec, H, W, C,
              save_mean, save_invstd, gamma, beta, (const char *)residual, (unsigned char *)live_bits);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_nn_bn_train_fwd_pool(const void *x, void *y, const void *residual, int dtype, int64_t N, int H, int W, int C,
                               const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                               float *running_var, float *save_mean, float *save_invstd, double *sums_ws, int64_t *batches_tracked,
                               const double *stats_part, int stats_blocks, void *hip_stream)
{
    return bn_train_fwd_pool_impl(x, y, residual, dtype, N, H, W, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean,
                                  save_invstd, sums_ws, batches_tracked, stats_part, stats_blocks, nullptr, hip_stream);
}

/* ... also leaving the full-resolution ReLU bit plane live_bits[N * H * W * C / L] (bytes of pixels the pool drops -- an odd last
 * row / column -- are not written and not used) for salsa_nn_bn_bwd_pool_bits */
int salsa_nn_bn_train_fwd_pool_bits(const void *x, void *y, const void *residual, int dtype, int64_t N, int H, int W, int C,
                                    const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                    float *running_var, float *save_mean, float *save_invstd, double *sums_ws, int64_t *batches_tracked,
                                    const double *stats_part, int stats_blocks, void *live_bits, void *hip_stream)
{
    if (!live_bits) return -1;
    return bn_train_fwd_pool_impl(x, y, residual, dtype, N, H, W, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean,
                                  save_invstd, sums_ws, batches_tracked, stats_part, stats_blocks, live_bits, hip_stream);
}

/* its backward: dy_pooled [N][H/2][W/2][C] -> dx [N][H][W][C] (+ dres, the residual's gradient, when one was added; + dgamma,
 * dbeta); the ReLU mask is recomputed from x (and the residual) */
static int bn_bwd_pool_impl(const void *dy_pooled, const void *x, const void *residual /* or the bit plane */, bool bits, void *dx,
                            void *dres, int dtype, int64_t N, int H, int W, int C, const float *gamma, const float *beta,
                            const float *save_mean, const float *save_invstd, float *dgamma, float *dbeta, double *sums_ws,
                            float *coef_ws, void *hip_stream)
{
    const int64_t M = N * H * W;
    if (!dy_pooled || !x || !dx || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !sums_ws || !coef_ws ||
        (residual && !bits && !dres) || (bits && !residual) || N <= 0 || H < 2 || W < 2 || !bn_geometry_ok(dtype, M, C))
        return -1;
    hipStream_t st = (hipStream_t)hip_stream;
    const unsigned nblk = bn_reduce_blocks(dtype, M, C);
    float *part = (float *)(sums_ws + 2 * C);
    const DropArgs drop = drop_args(0.f, 0u);
#define BN_POOL_REDUCE(V, L, MASK)                                                                                            \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<V, L, MASK, false, true>), dim3(nblk), dim3(256), 0, st, (const char *)dy_pooled, \
                       (const char *)residual, (const char *)x, (long)M, C, save_mean, save_invstd, gamma, beta, drop, part, H, W)
    if (dtype == 1) {
        if (bits) BN_POOL_REDUCE(bf16x8, 8, 4);
        else if (residual) BN_POOL_REDUCE(bf16x8, 8, 3);
        else BN_POOL_REDUCE(bf16x8, 8, 2);
    } else {
        if (bits) BN_POOL_REDUCE(f32x4, 4, 4);
        else if (residual) BN_POOL_REDUCE(f32x4, 4, 3);
        else BN_POOL_REDUCE(f32x4, 4, 2);
    }
#undef BN_POOL_REDUCE
    const int apply_mode = bits ? 3 : residual ? 2 : 1;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, (int)nblk, (long)M, C, gamma, save_mean,
                       save_invstd, beta, coef_ws, dgamma, dbeta);
    if (dtype == 1)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16x8, 8, true>), dim3(bn_apply_blocks(dtype, M, C)), dim3(256), 0, st,
                           (const char *)dy_pooled, (const char *)residual, (const char *)x, (char *)dx, (char *)dres, (long)M, C,
                           coef_ws, apply_mode, drop, H, W);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<f32x4, 4, true>), dim3(bn_apply_blocks(dtype, M, C)), dim3(256), 0, st,
                           (const char *)dy_pooled, (const char *)residual, (const char *)x, (char *)dx, (char *)dres, (long)M, C,
                           coef_ws, apply_mode, drop, H, W);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_nn_bn_bwd_pool(const void *dy_pooled, const void *x, const void *residual, void *dx, void *dres, int dtype, int64_t N, int H,
                         int W, int C, const float *gamma, const float *beta, const float *save_mean, const float *save_invstd,
                         float *dgamma, float *dbeta, double *sums_ws, float *coef_ws, void *hip_stream)
{
    return bn_bwd_pool_impl(dy_pooled, x, residual, false, dx, dres, dtype, N, H, W, C, gamma, beta, save_mean, save_invstd, dgamma, dbeta,
                            sums_ws, coef_ws, hip_stream);
}

/* ... with the ReLU mask from the forward's bit plane (salsa_nn_bn_train_fwd_pool_bits) instead of recomputing it from x and the
 * residual: dres (or NULL) = the gradient of the residual that was added */
int salsa_nn_bn_bwd_pool_bits(const void *dy_pooled, const void *x, const void *live_bits, void *dx, void *dres, int dtype, int64_t N,
                              int H, int W, int C, const float *gamma, const float *beta, const float *save_mean,
                              const float *save_invstd, float *dgamma, float *dbeta, double *sums_ws, float *coef_ws, void *hip_stream)
{
    return bn_bwd_pool_impl(dy_pooled, x, live_bits, true, dx, dres, dtype, N, H, W, C, gamma, beta, save_mean, save_invstd, dgamma, dbeta,
                            sums_ws, coef_ws, hip_stream);
}

/* bf16 copies of all hand-written convolution layers' float32 filters in ONE launch: the forward layout [Cout][3][3][Cin] and
 * (when bwd != 0) the data-gradient layout [Cin][3][3][Cout] with the taps flipped.  desc (device): n_layers x 11 int64 =
 * {src, fwd, bwd, Cout, Cin, src element strides co / ci / ky / kx, first block, taps (9 | 1)}; layer l owns (Cout/32)*(Cin/32) blocks from
 * its first block on; n_blocks = their total.  Cout and Cin must be multiples of 32. */
int salsa_nn_conv_filter_bank(const void *desc, int n_layers, int n_blocks, void *hip_stream)
{
    if (!desc || n_layers <= 0 || n_blocks <= 0) return -1;
    hipLaunchKernelGGL(conv_filter_bank_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)hip_stream, (const long long *)desc,
                       n_layers);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

/* SELD training loss and its gradients (salsa_amd/crnn/loss.py; reference models/interfaces.py:304-355): logit, sed_gt
 * [rows][nc]; doa, doa_gt [rows][3 nc] float32 contiguous.  out3 = {loss, sed, doa}; g_logit / g_doa = d sed / d logit and
 * d doa / d prediction (unweighted); partial_ws: SALSA_SELD_LOSS_WS float64 values of scratch. */
int salsa_nn_seld_loss(const float *logit, const float *doa, const float *sed_gt, const float *doa_gt, int64_t rows, int nc,
                       float w_sed, float w_doa, float *out3, float *g_logit, float *g_doa, double *partial_ws, void *hip_stream)
{
    if (!logit || !doa || !sed_gt || !doa_gt || !out3 || !g_logit || !g_doa || !partial_ws || rows <= 0 || nc <= 0) return -1;
    hipLaunchKernelGGL(seld_loss_partial_kernel, dim3(SELD_BLOCKS), dim3(256), 0, (hipStream_t)hip_stream, logit, doa, sed_gt, doa_gt,
                       (long)rows, nc, partial_ws);
    hipLaunchKernelGGL(seld_loss_finish_kernel, dim3(SELD_BLOCKS), dim3(256), 0, (hipStream_t)hip_stream, logit, doa, sed_gt, doa_gt,
                       (long)rows, nc, w_sed, w_doa, partial_ws, out3, g_logit, g_doa);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

/* out_a = a (g_loss w_sed + g_sed), out_b = b (g_loss w_doa + g_doa): the loss's backward; g_* are device scalars or NULL (= 0) */
int salsa_nn_seld_loss_bwd(const float *a, int64_t na, const float *b, int64_t nb, const float *g_loss, const float *g_sed,
                           const float *g_doa, float w_sed, float w_doa, float *out_a, float *out_b, void *hip_stream)
{
    if (!a || !b || !out_a || !out_b || na <= 0 || nb <= 0) return -1;
    const long blocks = (na + nb + 255) / 256;
    hipLaunchKernelGGL(seld_loss_scale_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, (hipStream_t)hip_stream, a,
                       (long)na, b, (long)nb, g_loss, g_sed, g_doa, w_sed, w_doa, out_a, out_b);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

/* mean over the frequency axis of a bf16 channels-last map x [N][H][W][C] -> float32 y, [H][N][C] when time_major else [N][H][C];
 * C % 8 == 0.  _bwd: dx = g / W broadcast over W, bf16 channels-last. */
int salsa_nn_freq_mean_fwd(const void *x, float *y, int64_t N, int H, int W, int C, int time_major, void *hip_stream)
{
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return -1;
    const long rows = (long)N * H, n = rows * (C / 8);
    hipLaunchKernelGGL(freq_mean_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                       (const unsigned short *)x, y, rows, (int)N, H, W, C, time_major);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

int salsa_nn_freq_mean_bwd(const float *g, void *dx, int64_t N, int H, int W, int C, int time_major, void *hip_stream)
{
    if (!g || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return -1;
    const long rows = (long)N * H, n = rows * W * (C / 8);
    hipLaunchKernelGGL(freq_mean_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, g,
                       (unsigned short *)dx, rows, (int)N, H, W, C, time_major);
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

/* out_a[c] += sum_rows a[row][c] (and out_b from b when b != NULL): float32 [M][C] row-major, the outputs must hold zeros. */
int salsa_nn_colsum2(const float *a, const float *b, float *out_a, float *out_b, int64_t M, int C, void *hip_stream)
{
    if (!a || !out_a || (b && !out_b) || M <= 0 || C <= 0) return -1;
    const int rpb = 64; // rows per workgroup: 16 per row lane
    const unsigned gy = (unsigned)((M + rpb - 1) / rpb);
    int rc = 0;
    float *part = salsa_nn_det_begin((int)gy, 2L * C, (hipStream_t)hip_stream, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum2_kernel, dim3((unsigned)((C + 63) / 64), gy, b ? 2u : 1u), dim3(256), 0,
                       (hipStream_t)hip_stream, a, b, out_a, out_b, (long)M, C, rpb, part);
    if (part) { // slab row = [out_a | out_b]: two strided reductions (the outputs are separate arrays)
        hipLaunchKernelGGL(slab_reduce_strided_kernel, dim3((unsigned)((C + 255) / 256), b ? 2u : 1u), dim3(256), 0, (hipStream_t)hip_stream,
                           part, (int)gy, C, out_a, out_b);
    }
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------------------ Adam, one launch
namespace {
struct AdamGrads {
    const float *g[SALSA_NN_ADAM_MAX_TENSORS];
};
__global__ __launch_bounds__(256) void adam_step_kernel(const salsa_nn_adam_entry *__restrict__ table, const AdamGrads grads,
                                                        const int2 *__restrict__ chunks, float w1 /* 1 - beta1 */, float beta2,
                                                        float w2 /* 1 - beta2 */, float step_size, float bc2_sqrt, float eps, float wd)
{
    const int2 ch = chunks[blockIdx.x];
    const salsa_nn_adam_entry e = table[ch.x];
    const float *__restrict__ g = grads.g[ch.x];
    const long first = (long)ch.y * SALSA_NN_ADAM_CHUNK;
    const long last = first + SALSA_NN_ADAM_CHUNK < e.n ? first + SALSA_NN_ADAM_CHUNK : e.n;
    auto update = [&](float &p, float &m, float &v, float gr) {
        if (wd != 0.f) gr += p * wd;
        m = m + w1 * (gr - m);                       // lerp(m, g, 1 - beta1), weight < 0.5
        v = beta2 * v + w2 * gr * gr;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        p -= step_size * m / denom;
    };
    const bool vec = ((((uintptr_t)e.p | (uintptr_t)e.m | (uintptr_t)e.v | (uintptr_t)g) & 15) == 0) && (e.n & 3) == 0;
    if (vec) {
        for (long i = first + 4 * threadIdx.x; i < last; i += 4 * 256) {
            float4 p = *(float4 *)(e.p + i), m = *(float4 *)(e.m + i), v = *(float4 *)(e.v + i);
            const float4 gr = *(const float4 *)(g + i);
            update(p.x, m.x, v.x, gr.x); update(p.y, m.y, v.y, gr.y); update(p.z, m.z, v.z, gr.z); update(p.w, m.w, v.w, gr.w);
            *(float4 *)(e.p + i) = p; *(float4 *)(e.m + i) = m; *(float4 *)(e.v + i) = v;
        }
    } else {
        for (long i = first + threadIdx.x; i < last; i += 256) update(e.p[i], e.m[i], e.v[i], g[i]);
    }
}
} // namespace

extern "C" int salsa_nn_adam_step(const salsa_nn_adam_entry *table, const float *const *grads, int n_tensors, const int *chunks, int n_chunks,
                                  double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, void *hip_stream)
{
    if (!table || !grads || !chunks || n_tensors <= 0 || n_tensors > SALSA_NN_ADAM_MAX_TENSORS || n_chunks <= 0 || step < 1) return -1;
    AdamGrads ga;
    for (int i = 0; i < n_tensors; i++) {
        if (!grads[i]) return -1;
        ga.g[i] = grads[i];
    }
    for (int i = n_tensors; i < SALSA_NN_ADAM_MAX_TENSORS; i++) ga.g[i] = nullptr;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)hip_stream, table, ga, (const int2 *)chunks,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1), (float)sqrt(bc2), (float)eps,
                       (float)weight_decay); // (the hyper-parameters combine in double, as torch's kernel does per element)
    return hipGetLastError() == hipSuccess ? 0 : -6;
}

