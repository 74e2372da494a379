// elementwise.h -- LayerNorm and small HBM-bound helpers.
#pragma once
#include "ccx_common.h"

// y[m][:] = LN(x[m][:]) * gamma + beta, fp32 statistics; out_bf16 and/or out_f32 may be null.
int ccx_launch_layernorm(ccx_ctx* ctx, const float* x, long ldx, const float* gamma, const float* beta,
                         bf16_t* out_bf16, float* out_f32, long ldo, int M, int D, float eps, hipStream_t stream);
// dst_bf16[i] = bf16(src_f32[i])
int ccx_launch_f32_to_bf16(ccx_ctx* ctx, const float* src, bf16_t* dst, long n, hipStream_t stream);
int ccx_launch_fill_u16(ccx_ctx* ctx, bf16_t* dst, bf16_t v, long n, hipStream_t stream);
int ccx_launch_peak_normalize(ccx_ctx* ctx, const float* x, float* y, long stride, const int* n_samples_dev, int B, float eps,
                              hipStream_t stream);
// dst[i][0 .. lens[i]) = ((const float*)src_ptrs[i])[0 .. lens[i]); tables in device memory
int ccx_launch_gather_rows(ccx_ctx* ctx, const long* src_ptrs_dev, const int* lens_dev, int n_rows, int max_len, float* dst,
                           long stride, hipStream_t stream);
// out[b] = unbiased variance of x[b][0 .. n_b) (fp64 accumulation, fixed order)
int ccx_launch_row_variance(ccx_ctx* ctx, const float* x, long stride, const int* n_samples_dev, int B, float* out, hipStream_t stream);
// out[r] = cosine similarity of a[r][:] and b[r % b_rows][:] (ATen's formula, eps 1e-8)
int ccx_launch_cosine_rows(ccx_ctx* ctx, const float* a, const float* b, int R, int D, int b_rows, float* out, hipStream_t stream);
// out[c][s][:] = sum over the turns t of speaker s of emb[c][t][:] * w[c][t] / (sum of that speaker's w[c][.])
int ccx_launch_speaker_profiles(ccx_ctx* ctx, const float* emb, const float* w, const int* spk_dev, int C, int T, int D, int S, float* out,
                                hipStream_t stream);
