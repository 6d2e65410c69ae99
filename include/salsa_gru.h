/*
 * salsa_gru.h -- C ABI of the fused GRU scan used by the SELD CRNN consumer (models/decoders.py:44-46 in the upstream
 * repo: nn.GRU(512, 256, num_layers=2, bidirectional=True)).  One launch runs the whole time recurrence of one layer
 * for every (sample, direction) pair; the input projections W_ih x + b_ih are one dense GEMM done by the caller.
 * All tensors are float32, device pointers, caller-owned; asynchronous on the given HIP stream.
 *
 *   gi    [T][B][D][3H]  input projections, gate order r,z,n (PyTorch's)
 *   whh   [D][3H][H]     weight_hh (PyTorch layout, row = gate*H + unit); whh_t [D][H][3H] is its transpose per direction
 *   bhh   [D][3H]
 *   hs    [T][B][D][H]   hidden state AFTER step t (direction 1 scans t = T-1..0 and still stores at index t)
 *   saved [T][B][D][4H]  r, z, n, (W_hn h + b_hn) per step, for the backward scan (may be NULL for inference)
 */
#ifndef SALSA_GRU_H
#define SALSA_GRU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int salsa_gru_scan_fwd(const float *gi, const float *whh_t, const float *bhh, float *hs, float *saved, int T, int B, int D,
                       int H, void *hip_stream);

/* Register-resident variants for H = 256: the recurrent weights (PyTorch layout whh, rounded to float16) stay in the
 * registers of a 1024-thread workgroup for the whole scan instead of being streamed from L2 every step; float32 state,
 * gates and accumulation.  Used without gradients (saved = NULL) and for training under bf16 autocast, where the backward
 * variant differentiates the same rounded weights (same arguments as salsa_gru_scan_bwd). */
int salsa_gru_scan_fwd_regw(const float *gi, const float *whh, const float *bhh, float *hs, float *saved, int T, int B, int D,
                            int H, void *hip_stream);
int salsa_gru_scan_bwd_regw(const float *dhs, const float *whh, const float *hs, const float *saved, float *dgi, float *dgh,
                            int T, int B, int D, int H, void *hip_stream);

/* dhs [T][B][D][H] = gradient wrt every hs[t].  Outputs: dgi [T][B][D][3H] (gradient wrt gi) and dgh [T][B][D][3H]
 * (gradient wrt W_hh h_prev + b_hh; the caller forms dW_hh = sum_t,b dgh^T h_prev and db_hh with one GEMM). */
int salsa_gru_scan_bwd(const float *dhs, const float *whh, const float *hs, const float *saved, float *dgi, float *dgh,
                       int T, int B, int D, int H, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
