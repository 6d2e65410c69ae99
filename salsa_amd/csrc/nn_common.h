// nn_common.h -- internal: what the BatchNorm kernels (nn_ops.hip) and the convolutions that apply a BatchNorm on the fly while
// staging their input (conv_mfma.hip, round 4) must agree on bit for bit: the counter-based dropout mask.
#pragma once
#include <hip/hip_runtime.h>
namespace {
struct DropArgs {
    unsigned thresh; // drop when the element's 16 bits < thresh ; 0 = no dropout
    unsigned seed;
    float scale;     // 65536 / (65536 - thresh)
};
__device__ inline unsigned drop_hash(unsigned pair, unsigned seed)
{
    unsigned h = pair * 0x9E3779B1u + seed;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
static inline DropArgs drop_args(float p, unsigned seed)
{
    DropArgs d = {0u, seed, 1.f};
    if (p > 0.f) {
        d.thresh = (unsigned)(p * 65536.f + 0.5f);
        if (d.thresh > 65535u) d.thresh = 65535u;
        d.scale = 65536.f / (float)(65536u - d.thresh);
    }
    return d;
}
} // namespace
