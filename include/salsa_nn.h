/*
 * salsa_nn.h -- C ABI of the memory-bound layers of the SELD CRNN consumer that are hand-written for MI355X (the
 * convolutions stay with MIOpen).  Tensors are channels-last ([N][H][W][C], C fastest), device pointers, caller-owned;
 * dtype: 0 = float32, 1 = bfloat16; asynchronous on the given HIP stream.
 *
 *   salsa_nn_avgpool2x2_{fwd,bwd}: F.avg_pool2d(x, 2) of the upstream model (models/model_utils.py:187-228 ConvBlock,
 *   the strided residual stages of the PANN ResNet22): y[n][h][w][c] = (x[2h][2w] + x[2h][2w+1] + x[2h+1][2w] +
 *   x[2h+1][2w+1]) / 4 accumulated in float32 in that order; odd trailing rows / columns are dropped (floor mode) and
 *   get zero gradient.  C must be a multiple of 8 (bf16) or 4 (float32).
 */
#ifndef SALSA_NN_H
#define SALSA_NN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int salsa_nn_avgpool2x2_fwd(const void *x, void *y, int dtype, int64_t N, int H, int W, int C, void *hip_stream);
int salsa_nn_avgpool2x2_bwd(const void *grad_y, void *grad_x, int dtype, int64_t N, int H, int W, int C, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
