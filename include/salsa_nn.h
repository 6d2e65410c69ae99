/*
 * salsa_nn.h -- C ABI of the layers of the SELD CRNN consumer that are hand-written for MI355X: average pools, fused
 * BatchNorm, and EVERY convolution of the training step on the matrix cores -- forward, data gradient and weight gradient of
 * the first layer (salsa_nn_conv3x3_stem*), the 64 -> 64 layers (salsa_nn_conv3x3_c64*), the 128 / 256 / 512-channel layers
 * (salsa_nn_conv3x3_wide*) and the 1x1 shortcuts (salsa_nn_conv1x1*); nothing of the step is left on MIOpen.  Tensors are channels-last ([N][H][W][C], C fastest), device pointers, caller-owned;
 * dtype: 0 = float32, 1 = bfloat16; asynchronous on the given HIP stream.
 *
 *   salsa_nn_avgpool2x2_{fwd,bwd}: F.avg_pool2d(x, 2) of the upstream model (models/model_utils.py:187-228 ConvBlock,
 *   the strided residual stages of the PANN ResNet22): y[n][h][w][c] = (x[2h][2w] + x[2h][2w+1] + x[2h+1][2w] +
 *   x[2h+1][2w+1]) / 4 accumulated in float32 in that order; odd trailing rows / columns are dropped (floor mode) and
 *   get zero gradient.  C must be a multiple of 8 (bf16) or 4 (float32).
 *
 *   salsa_nn_bn_*: nn.BatchNorm2d of the upstream blocks fused with what follows it there -- the residual add and the
 *   ReLU (models/model_utils.py:187-228, :312-367): y = [relu]( ((x - mean) * invstd) * gamma + beta [+ residual] ).
 *   x, y, residual, dy, dx: [M][C] (M = N*H*W) in `dtype`; gamma, beta, statistics: float32 [C].  Training statistics are
 *   accumulated per block in float32 and across blocks in float64 (sums_ws: salsa_nn_bn_workspace_bytes of scratch) and the running statistics are updated like torch does
 *   (momentum, unbiased variance).  Backward: with g = dy * (y > 0) (relu = 0: g = dy),
 *   dbeta = sum g, dgamma = sum g * xhat, dx = gamma * invstd * (g - dbeta/M - xhat * dgamma/M), and dres = g when the
 *   forward had a residual.  With relu != 0 and y_or_null = NULL (allowed when the forward had NO residual) the mask is
 *   recomputed from x, which saves reading y.  coef_ws: 7*C floats of scratch.  salsa_nn_bn_supported: C/L must be a power of two <= 256
 *   (L = 8 for bf16, 4 for float32).
 *
 *   salsa_nn_conv3x3_c64: the 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (the stem's second
 *   convolution and the four of the first residual stage: 42 % of the network's convolution FLOPs) on the matrix cores,
 *   bf16 in / float32 accumulate / bf16 out.  x, y: [N][H][W][64]; w: [64 co][3][3][64 ci] (a channels-last
 *   torch.nn.Conv2d weight).  The data gradient is the same call on dy with the filter flipped and transposed
 *   (w'[ci][r][s][co] = w[co][2-r][2-s][ci]).  salsa_amd/csrc/conv_mfma.hip describes the kernel.
 */
#ifndef SALSA_NN_H
#define SALSA_NN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int salsa_nn_avgpool2x2_fwd(const void *x, void *y, int dtype, int64_t N, int H, int W, int C, void *hip_stream);
int salsa_nn_avgpool2x2_bwd(const void *grad_y, void *grad_x, int dtype, int64_t N, int H, int W, int C, void *hip_stream);

int salsa_nn_conv3x3_c64(const void *x, const void *w, void *y, int64_t N, int H, int W, void *hip_stream);
/* inference: y = [relu](conv(x, w) + shift[co] [+ residual]) with the BatchNorm that follows folded in: w pre-scaled by
 * gamma / sqrt(var + eps) per output channel, shift = beta - mean * gamma / sqrt(var + eps) (float32 [64]); residual bf16 or NULL */
int salsa_nn_conv3x3_c64_bias_act(const void *x, const void *w, const float *shift, const void *residual, void *y, int relu,
                                  int64_t N, int H, int W, void *hip_stream);
/* the same followed by the 2x2 average pool of the upstream stem (taken in float32 before the single rounding):
 * y bf16 [N][H/2][W/2][64]; H and W even */
int salsa_nn_conv3x3_c64_bias_act_pool(const void *x, const void *w, const float *shift, const void *residual, void *y, int relu,
                                       int64_t N, int H, int W, void *hip_stream);
/* the network's first layer, Cin <= 8 -> 64 channels, on the extractor's output layout: x float32 planar [N][Cin][H][W] with
 * contiguous rows and the given batch / channel strides (elements; a time-cropped view needs no copy);
 * wq bf16 [64 co][10 taps][8 ci] (taps row-major, tap 9 and ci >= Cin zero); y bf16 channels-last [N][H][W][64];
 * shift NULL: plain convolution, else y = [relu](conv + shift[co]) with the BatchNorm folded as above */
int salsa_nn_conv3x3_stem(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *wq, const float *shift,
                          void *y, int relu, int64_t N, int Cin, int H, int W, void *hip_stream);
/* the WIDE 3x3 / stride 1 / pad 1 convolutions (Cin a multiple of 32, Cout a multiple of 64: the 128 / 256 / 512-channel
 * residual stages, models/model_utils.py:312-367, :429-500): implicit GEMM over the flattened pixel axis, both operands staged
 * through LDS (salsa_amd/csrc/conv_wide.hip).  x [N][H][W][Cin], w [Cout][3][3][Cin], y [N][H][W][Cout], all bf16; float32
 * accumulation.  The data gradient is the same call on dy with w'[ci][r][s][co] = w[co][2-r][2-s][ci].
 * salsa_nn_conv3x3_wide_supported: 1 if the shape is taken (the padded input chunk of a 512-pixel tile must fit its LDS buffer). */
int salsa_nn_conv3x3_wide_supported(int64_t N, int H, int W, int Cin, int Cout);
int salsa_nn_conv3x3_wide(const void *x, const void *w, void *y, int64_t N, int H, int W, int Cin, int Cout, void *hip_stream);
/* inference: y = [relu](conv(x, w) + shift[co] [+ residual]) with the BatchNorm that follows folded in (w pre-scaled per output
 * channel, shift float32 [Cout], residual bf16 like y or NULL), applied to the float32 sums before the single rounding */
int salsa_nn_conv3x3_wide_bias_act(const void *x, const void *w, const float *shift, const void *residual, void *y, int relu,
                                   int64_t N, int H, int W, int Cin, int Cout, void *hip_stream);
/* weight gradient of the wide layers (Cin a multiple of 32, Cout a multiple of 128): dw float32 [Cout][3][3][Cin] += sum_pixels
 * dy[p][co] * x[p+tap][ci] (zero it first), transposing LDS reads on the same flattened pixel axis.  The kernel walks many pixel
 * tiles per workgroup and takes its (n, h, w) arithmetic from two index tables of the map shape, built on the HOST by
 * salsa_nn_conv3x3_wide_tables (vpos: N*H*W ints, inv: salsa_nn_conv3x3_wide_table_len ints, tile_bounds: 2 *
 * salsa_nn_conv3x3_wide_tile_count ints) and uploaded once by the caller. */
int salsa_nn_conv3x3_wide_wrw_supported(int64_t N, int H, int W, int Cin, int Cout);
int64_t salsa_nn_conv3x3_wide_table_len(int64_t N, int H, int W);
int64_t salsa_nn_conv3x3_wide_tile_count(int64_t N, int H, int W);
int salsa_nn_conv3x3_wide_tables(int64_t N, int H, int W, int *vpos_host, int *inv_host, int *tile_bounds_host);
/* Training forward of the wide layers that also leaves the per-channel sum / sum of squares of its (bf16-rounded) output as
 * float64 partial rows stats_part[salsa_nn_conv3x3_wide_stats_blocks(...)][2][Cout] in its epilogue, like
 * salsa_nn_conv3x3_c64_stats: the BatchNorm that follows (salsa_nn_bn_train_fwd: stats_part, stats_blocks) makes no statistics
 * pass over y. */
int salsa_nn_conv3x3_wide_stats_blocks(int64_t N, int H, int W, int Cin, int Cout);
int salsa_nn_conv3x3_wide_stats(const void *x, const void *w, void *y, double *stats_part, int64_t N, int H, int W, int Cin, int Cout,
                                void *hip_stream);
int salsa_nn_conv3x3_wide_wrw(const void *x, const void *dy, float *dw, const int *d_vpos, const int *d_inv, const int *d_tile_bounds,
                              int64_t N, int H, int W, int Cin, int Cout, void *hip_stream);
/* Training forward of the 64 -> 64 layer that ALSO leaves the per-channel sum and sum of squares of its (bf16-rounded) output as
 * per-workgroup float64 rows stats_part[salsa_nn_conv3x3_c64_stats_blocks(N, H, W)][2][64], accumulated in its epilogue: the
 * BatchNorm that follows (salsa_nn_bn_train_fwd / _pool: stats_part, stats_blocks) then makes no statistics pass over y. */
int salsa_nn_conv3x3_c64_stats_blocks(int64_t N, int H, int W);
int salsa_nn_conv3x3_c64_stats(const void *x, const void *w, void *y, double *stats_part, int64_t N, int H, int W, void *hip_stream);
/* Training forward of the first layer from a persistent launch that also leaves the per-channel partial sums of its output
 * (see salsa_nn_conv3x3_c64_stats): stats_part[salsa_nn_conv3x3_stem_stats_blocks(N, H, W)][2][64] float64. */
int salsa_nn_conv3x3_stem_stats_blocks(int64_t N, int H, int W);
int salsa_nn_conv3x3_stem_stats(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *wq, void *y,
                                double *stats_part, int64_t N, int Cin, int H, int W, void *hip_stream);
/* weight gradient of the (Cin <= 7) -> 64 first layer: dw float32 [64 co][Cin][3][3] contiguous += sum_pixels dy[p][co] *
 * x[ci][p + tap] (zero it first); x float32 planar as in salsa_nn_conv3x3_stem, dy bf16 channels-last [N][H][W][64] */
int salsa_nn_conv3x3_stem_wrw(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *dy, float *dw, int64_t N,
                              int Cin, int H, int W, void *hip_stream);
/* The same with the BatchNorm (+ ReLU) behind the first layer differentiated on the fly: g = gradient of the BatchNorm's OUTPUT,
 * x1 = the BatchNorm's input (= the first layer's output), coef = the [7][64] table salsa_nn_bn_bwd leaves in coef_ws.  Call
 * salsa_nn_bn_bwd with dx = NULL (it then only produces dgamma, dbeta and coef_ws): dx's only reader is this weight gradient,
 * so the full-resolution dx is neither written nor read. */
int salsa_nn_conv3x3_stem_wrw_bn(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *g, const void *x1,
                                 const float *coef, int relu, float *dw, int64_t N, int Cin, int H, int W, void *hip_stream);

/* ... and with the BatchNorm backward's REDUCTION folded in as well (round 5): dx = a (g - b - xh k') is linear in the two totals
 * b = mean(g), k' = mean(g xh), so the pass over (g, x1, x) accumulates G = sum g (x) patch, Xh = sum xh (x) patch, S0 = sum patch,
 * dbeta = sum g, dgamma = sum g xh per workgroup, and a second small launch combines the slabs: dW = a (G - b S0 - k' Xh).  Replaces
 * salsa_nn_bn_bwd(dx = NULL) + salsa_nn_conv3x3_stem_wrw_bn (one pass over the 524-MB g and x1 instead of two).  mean / invstd: the
 * forward's saved statistics; ws: salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes(N, H, W) bytes of device scratch (-5 when smaller); dw is
 * ADDED to, dgamma / dbeta are written.  Always bit-reproducible (slabs added in a fixed order, float64). */
size_t salsa_nn_conv3x3_stem_wrw_bnf_ws_bytes(int64_t N, int H, int W);
int salsa_nn_conv3x3_stem_wrw_bnf(const float *x, int64_t x_batch_stride, int64_t x_channel_stride, const void *g, const void *x1,
                                  const float *mean, const float *invstd, const float *gamma, const float *beta, int relu, float *dw,
                                  float *dgamma, float *dbeta, void *ws, size_t ws_bytes, int64_t N, int Cin, int H, int W,
                                  void *hip_stream);
/* weight gradient of the same layer: dw float32 [64 co][3][3][64 ci] += sum_pixels dy[p][co] * x[p+tap][ci] (zero it first) */
int salsa_nn_conv3x3_c64_wrw(const void *x, const void *dy, float *dw, int64_t N, int H, int W, void *hip_stream);

/* Round 4: a conv -> BatchNorm -> ReLU (-> dropout) -> conv chain of the reference blocks (models/model_utils.py:213-228, :345-367)
 * without the normalised activation ever being stored.  The first convolution leaves its RAW output x1 and the per-workgroup
 * statistics (salsa_nn_conv3x3_*_stats); salsa_nn_bn_train_finalize turns those into mean / invstd (and updates the running
 * statistics) with no pass over x1; the SECOND convolution's forward (salsa_nn_conv3x3_c64_xform_stats: y = conv(a, w) + the
 * statistics epilogue for the BatchNorm that follows) and weight gradient (salsa_nn_conv3x3_c64_wrw_xform) form
 * a = dropout(relu(((x1 - mean) * invstd) * gamma + beta)) while they stage their operand -- as x1 * scale + shift with
 * scale = invstd * gamma, shift = beta - mean * scale in float32, rounded once to bf16; the dropout mask is the one
 * salsa_nn_bn_train_fwd would draw for (drop_p, drop_seed), so salsa_nn_bn_bwd regenerates the same.  The convolution's zero padding
 * applies to a, not to x1.  All four vectors: float32 [64]. */
int salsa_nn_bn_train_finalize(const double *stats_part, int stats_blocks, int64_t M, int C, float eps, float momentum,
                               float *running_mean, float *running_var, float *save_mean, float *save_invstd,
                               int64_t *batches_tracked, void *hip_stream);
int salsa_nn_conv3x3_c64_xform_stats(const void *x1, const void *w, void *y, double *stats_part, const float *mean,
                                     const float *invstd, const float *gamma, const float *beta, float drop_p, uint32_t drop_seed,
                                     int64_t N, int H, int W, void *hip_stream);
int salsa_nn_conv3x3_c64_wrw_xform(const void *x1, const void *dy, float *dw, const float *mean, const float *invstd,
                                   const float *gamma, const float *beta, float drop_p, uint32_t drop_seed, int64_t N, int H, int W,
                                   void *hip_stream);

/* drop_p > 0 fuses the dropout that follows the ReLU in the upstream residual block (models/resnet.py:78): an element is kept
 * when a counter-based hash of (its index, drop_seed) says so and scaled by 1/(1-p), p quantised to 1/65536; the backward takes
 * the same (drop_p, drop_seed) and regenerates the mask, so none is stored.  M*C must be below 2^32. */
int salsa_nn_bn_supported(int dtype, int64_t M, int C);
size_t salsa_nn_bn_workspace_bytes(int dtype, int64_t M, int C); /* size of sums_ws (8-byte aligned) */
int salsa_nn_bn_train_fwd(const void *x, void *y, const void *residual, int dtype, int64_t M, int C, const float *gamma,
                          const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                          float *save_mean, float *save_invstd, double *sums_ws, int relu, float drop_p, uint32_t drop_seed,
                          int64_t *batches_tracked /* nn.BatchNorm2d.num_batches_tracked (device), += 1; may be NULL */,
                          const double *stats_part, int stats_blocks /* per-workgroup [2][C] float64 sum / sum-of-squares rows left
                          by the convolution that produced x (salsa_nn_conv3x3_c64_stats): the statistics pass over x is
                          skipped; NULL / 0: the call computes them itself */,
                          void *hip_stream);
int salsa_nn_bn_eval_fwd(const void *x, void *y, const void *residual, int dtype, int64_t M, int C, const float *gamma,
                         const float *beta, const float *mean, const float *invstd, int relu, void *hip_stream);
int salsa_nn_bn_bwd(const void *dy, const void *y_or_null, const void *x, void *dx, void *dres_or_null, int dtype, int64_t M,
                    int C, const float *gamma, const float *beta, const float *save_mean, const float *save_invstd, int relu,
                    float *dgamma, float *dbeta, double *sums_ws, float *coef_ws, float drop_p, uint32_t drop_seed,
                    void *hip_stream);

/* BatchNorm (training statistics) [+ residual] + ReLU + the 2x2 average pool that follows them -- in the upstream stem
 * (models/model_utils.py:187-228) and, with the block's shortcut as `residual`, at the end of every residual block whose
 * successor starts with the stride-2 pool (:312-367) -- one pass over x: x [N][H][W][C] -> y [N][H/2][W/2][C] (normalise, add,
 * ReLU each of the four pixels, average in float32, one rounding); the full-resolution activation is never written.  Backward
 * from the POOLED gradient: dx, and dres (the residual's gradient) when there was one; the ReLU mask is recomputed from x (and
 * the residual), the same float32 operations in the same order as the forward. */
int salsa_nn_bn_train_fwd_pool(const void *x, void *y, const void *residual, int dtype, int64_t N, int H, int W, int C,
                               const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                               float *running_var, float *save_mean, float *save_invstd, double *sums_ws, int64_t *batches_tracked,
                               const double *stats_part, int stats_blocks, void *hip_stream);
int salsa_nn_bn_bwd_pool(const void *dy_pooled, const void *x, const void *residual, void *dx, void *dres, int dtype, int64_t N, int H,
                         int W, int C, const float *gamma, const float *beta, const float *save_mean, const float *save_invstd,
                         float *dgamma, float *dbeta, double *sums_ws, float *coef_ws, void *hip_stream);

/* The ReLU mask as a BIT PLANE (round 4).  Where something is added before the ReLU (the residual blocks' second BatchNorm) the
 * backward cannot recompute the mask from x alone: it read the stored output y (or, for the pooled variant, the residual) -- 16 bytes
 * per vector in each of its two passes -- only to test `> 0`.  The *_bits forward variants also leave live_bits: one BYTE per
 * 16-byte vector (M * C / 8 bytes for bf16, M * C / 4 for float32; bit k = "output element k of the vector is positive", after the
 * dropout if any), and the backward reads that instead: salsa_nn_bn_bwd with relu = 2 and live_bits in y's place,
 * salsa_nn_bn_bwd_pool_bits with live_bits in the residual's place (dres, or NULL, still receives the residual's gradient).
 * Same results bit for bit (the bits ARE the tests the backward made).  Measured: 0.18 ms of a 10.2-ms training step. */
int salsa_nn_bn_train_fwd_bits(const void *x, void *y, const void *residual, int dtype, int64_t M, int C, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                               float *save_mean, float *save_invstd, double *sums_ws, int relu, float drop_p, uint32_t drop_seed,
                               int64_t *batches_tracked, const double *stats_part, int stats_blocks, void *live_bits, void *hip_stream);
int salsa_nn_bn_train_fwd_pool_bits(const void *x, void *y, const void *residual, int dtype, int64_t N, int H, int W, int C,
                                    const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                    float *running_var, float *save_mean, float *save_invstd, double *sums_ws, int64_t *batches_tracked,
                                    const double *stats_part, int stats_blocks, void *live_bits, void *hip_stream);
int salsa_nn_bn_bwd_pool_bits(const void *dy_pooled, const void *x, const void *live_bits, void *dx, void *dres, int dtype, int64_t N,
                              int H, int W, int C, const float *gamma, const float *beta, const float *save_mean,
                              const float *save_invstd, float *dgamma, float *dbeta, double *sums_ws, float *coef_ws, void *hip_stream);

/* The 1x1 / stride 1 convolutions of the residual shortcuts (models/model_utils.py:340-349) over the flattened pixel axis
 * (salsa_amd/csrc/conv_1x1.hip): x bf16 [M][Cin] (channels-last pixels, M = N*H*W), w bf16 [Cout][Cin], y bf16 [M][Cout];
 * the data gradient is the same call on dy with the transposed filter [Cin][Cout]; the weight gradient is float32 [Cout][Cin],
 * ADDED to dw (zero it first).  *_supported: Cin % 16 == 0 and Cout % 64 == 0 (forward), Cin % 64 == 0 and Cout % 128 == 0 (wrw). */
int salsa_nn_conv1x1_supported(int64_t M, int Cin, int Cout);
int salsa_nn_conv1x1(const void *x, const void *w, void *y, int64_t M, int Cin, int Cout, void *hip_stream);
int salsa_nn_conv1x1_wrw_supported(int64_t M, int Cin, int Cout);
int salsa_nn_conv1x1_wrw(const void *x, const void *dy, float *dw, int64_t M, int Cin, int Cout, void *hip_stream);

/* The bf16 working copies of the float32 master filters (what autocast's per-layer casts + permutes + flips produce in the
 * reference's training step, models/seld_models.py:68-76 under torch.cuda.amp) for ALL layers in one launch: forward layout
 * [Cout][3][3][Cin] and, when the descriptor's bwd pointer is not 0, the data-gradient layout [Cin][3][3][Cout] with flipped taps.
 * desc (device memory): n_layers x 11 int64 = {src float32*, fwd bf16*, bwd bf16*, Cout, Cin, src element strides for
 * (co, ci, ky, kx), first block, taps (9: 3x3 filter, 1: 1x1 filter)}; layer l owns (Cout/32)*(Cin/32) consecutive blocks;
 * n_blocks = the total.  Cout, Cin % 32 == 0. */
int salsa_nn_conv_filter_bank(const void *desc, int n_layers, int n_blocks, void *hip_stream);

/* The SELD training loss and its gradients in one launch (reference models/interfaces.py:304-355, compute_loss: 0.3 x
 * BCE-with-logits of the event logits + 0.7 x the sum over x / y / z of the activity-masked mean absolute error): logit, sed_gt
 * [rows][nc], doa, doa_gt [rows][3 nc] (blocks x | y | z), float32 contiguous.  out3 = {loss, sed loss, doa loss};
 * g_logit = d sed / d logit, g_doa = d doa / d prediction, unweighted.  salsa_nn_seld_loss_bwd scales them by the incoming
 * gradients (device scalars, NULL = 0): out_a = a (g_loss w_sed + g_sed), out_b = b (g_loss w_doa + g_doa). */
#define SALSA_SELD_LOSS_WS 192 /* float64 values of scratch (partial_ws) salsa_nn_seld_loss needs: 64 workgroups x 3 partial sums */
int salsa_nn_seld_loss(const float *logit, const float *doa, const float *sed_gt, const float *doa_gt, int64_t rows, int nc,
                       float w_sed, float w_doa, float *out3, float *g_logit, float *g_doa, double *partial_ws, void *hip_stream);
int salsa_nn_seld_loss_bwd(const float *a, int64_t na, const float *b, int64_t nb, const float *g_loss, const float *g_sed,
                           const float *g_doa, float w_sed, float w_doa, float *out_a, float *out_b, void *hip_stream);

/* The decoder's frequency mean (reference models/decoders.py: x.mean(dim=3) then (B, C, T) -> (B, T, C)) in one pass:
 * x bf16 channels-last [N][H][W][C] -> float32 y [H][N][C] (time_major != 0: the GRU scans' order) or [N][H][C]; C % 8 == 0.
 * _bwd: dx[n][h][w][c] = g[row(n, h)][c] / W, bf16 channels-last. */
int salsa_nn_freq_mean_fwd(const void *x, float *y, int64_t N, int H, int W, int C, int time_major, void *hip_stream);
int salsa_nn_freq_mean_bwd(const float *g, void *dx, int64_t N, int H, int W, int C, int time_major, void *hip_stream);

/* Column sums of one or two float32 row-major [M][C] matrices, ADDED to out_a / out_b (zero them first; b may be NULL): the GRU's
 * bias gradients db_ih = sum_(t,b) dgi, db_hh = sum_(t,b) dgh in one launch (torch's reduction / a ones-vector GEMV: ~17 us each). */
int salsa_nn_colsum2(const float *a, const float *b, float *out_a, float *out_b, int64_t M, int C, void *hip_stream);

/* Deterministic weight gradients (round 4).  Every weight-gradient kernel (salsa_nn_conv3x3_c64_wrw, _stem_wrw, _stem_wrw_bn,
 * _wide_wrw, salsa_nn_conv1x1_wrw) and salsa_nn_colsum2 is a reduction over pixels split across workgroups.  By default the
 * workgroups combine their float32 partial sums with atomic adds: the order of arrival decides the rounding, so two identical
 * training steps differ in the last bits.  salsa_nn_set_deterministic(ws, bytes) with a device workspace switches all of them to
 * partial SLABS (every workgroup stores its partial sums into its own slab of `ws`) followed by one reduction launch that adds the
 * slabs in slab order -- bit-identical results run to run, at the price of the slab traffic (<= 76 MB per call) and one more
 * launch per call.  The workspace is registered FOR THE DEVICE THAT IS CURRENT at the call (one per device; memory of another
 * device is refused with -1) and every launcher uses the workspace of the device current at ITS call -- a device without one keeps
 * the atomics.  ws = NULL switches every device back to atomics.  The calls return -5 when `bytes` is too small for a shape
 * (SALSA_NN_DET_WS_BYTES covers every layer of the SELD CRNN at the bench's sizes).  On one device the workspace is shared by all
 * calls: use one stream at a time, as the trainer does (two backward passes racing on two streams of one device need the atomics).
 * salsa_nn_get_deterministic: 1 when the current device has a workspace. */
#define SALSA_NN_DET_WS_BYTES ((size_t)160 << 20)
int salsa_nn_set_deterministic(void *ws, size_t bytes);
int salsa_nn_get_deterministic(void);

/* One Adam step over MANY parameter tensors in one launch (round 4) -- torch.optim.Adam's update rule (its fused CUDA kernel,
 * ADAM_MODE ORIGINAL, amsgrad off; the reference's optimiser, experiments/configs/seld.yml:37-52 through Lightning):
 *     g += weight_decay p;  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g g;
 *     p -= (lr / (1 - beta1^step)) m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * all float32.  torch's own fused path takes three multi_tensor_apply launches for this network's 143 tensors, two of them ~40
 * workgroups of tiny tensors (0.165 ms per step for 0.39 GB of traffic).
 *   table  device array [n_tensors] of {float *p, *m, *v; int64 n} (salsa_nn_adam_entry): stable from step to step;
 *   grads  HOST array [n_tensors] of device pointers to the gradients (they move from step to step: passed in the kernel arguments);
 *   chunks device array [n_chunks] of {int tensor, int first element / SALSA_NN_ADAM_CHUNK}: one workgroup each;
 * at most SALSA_NN_ADAM_MAX_TENSORS tensors per call.  Returns 0, -1 on bad arguments. */
#define SALSA_NN_ADAM_MAX_TENSORS 192
#define SALSA_NN_ADAM_CHUNK 8192
typedef struct { float *p, *m, *v; int64_t n; } salsa_nn_adam_entry;
int salsa_nn_adam_step(const salsa_nn_adam_entry *table, const float *const *grads, int n_tensors, const int *chunks, int n_chunks,
                       double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
