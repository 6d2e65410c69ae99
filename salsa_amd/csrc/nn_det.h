// nn_det.h -- internal: the deterministic-reduction hooks shared by the weight-gradient translation units (include/salsa_nn.h:
// salsa_nn_set_deterministic).  Implemented in nn_ops.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
// -> slabs x n floats of workspace (zeroed on the stream) when the deterministic mode is on, else NULL; *rc = -5 when it is on
// and the workspace is too small
float *salsa_nn_det_begin(int slabs, long n, hipStream_t st, int *rc);
// dw[i] += sum over slabs (in slab order) of ws[slab * n + i]
int salsa_nn_det_finish(const float *ws, int slabs, long n, float *dw, hipStream_t st);
// one partial sum: into the workgroup's slab (deterministic mode) or atomically into the result
__device__ __forceinline__ void salsa_nn_accumulate(float *__restrict__ dw, float *__restrict__ part, long n, int slab, long off, float v)
{
    if (part) part[(long)slab * n + off] = v;
    else atomicAdd(dw + off, v);
}
