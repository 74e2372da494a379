// gemm_bf16.h -- bf16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues.
// This one kernel family carries every linear / im2col-conv of the hot path
// (Whisper QKV/out/FFN/logits, conv stem, SepFormer/TDNN projections).
#pragma once
#include "ccx_common.h"

enum GemmEpi {
  EPI_BF16 = 0,        // out bf16 = acc + bias
  EPI_BF16_GELU = 1,   // out bf16 = gelu(acc + bias)
  EPI_F32_RESID = 2,   // out f32  = acc + bias + resid[row][n]      (resid may alias out)
  EPI_F32 = 3,         // out f32  = acc + bias
  EPI_HEADS = 4,       // q/k (/v) scattered head-major for the attention kernels
  EPI_BF16_RELU = 5,   // out bf16 = relu(acc + bias)
  EPI_F32_GELU_POS = 6, // out f32 = gelu(acc + bias) + resid[row % resid_mod][n]   (conv2 + pos-emb)
  EPI_BF16_LRELU_AFFINE = 7, // out bf16 = leaky_relu(acc + bias (+ resid)) * scale[n] + shift[n]  (TDNN conv + BatchNorm)
  EPI_BF16_ADD_RELU = 8  // out bf16 = relu(acc + bias + resid_bf16[out_row][n])   (ResNet basic block; resid may be null)
};

struct GemmParams {
  const bf16_t* A;   // [M, K] row-major, leading dimension lda (elements)
  const bf16_t* W;   // [N, K] row-major (torch Linear layout), leading dimension ldw
  long lda, ldw;
  int M, N, K;       // K % 64 == 0; N % 16 == 0 or the destination has ceil(N/16)*16 writable columns per row
  // Accumulated taps (convolutions as shifted GEMMs): C = sum_t A[row + t*a_tap_stride elements][0..K) * W[n][t*K .. (t+1)*K).
  // ntaps <= 1 is a plain GEMM.  W rows then hold ntaps*K elements (ldw >= ntaps*K).
  int ntaps; long a_tap_stride;
  const float* bias; // [N] fp32 or nullptr
  void* out;         // bf16 or f32 depending on epilogue
  long ldo;
  const float* resid; // f32
  long ldr;
  int resid_mod;     // if > 0: resid row = out_row % resid_mod
  const float* scale; const float* shift;  // EPI_BF16_LRELU_AFFINE: per-column affine after the activation (may be null)
  float slope;       // LeakyReLU negative slope
  // output row remap: rows arrive in groups of rpb_in; group g row i -> g*rpb_out + i + roff,
  // rows with i >= rpb_valid are dropped.  rpb_in == 0 -> identity.
  int rpb_in, rpb_out, roff, rpb_valid;
  // second level (images of padded rows): if img_rows_in > 0, group g = img*img_rows_in + r; rows with r >= img_rows_valid
  // are dropped and the output group is img*img_rows_out + r.
  int img_rows_in, img_rows_valid, img_rows_out;
  const bf16_t* resid_bf16; long ldrb;   // EPI_BF16_ADD_RELU
  // EPI_HEADS
  bf16_t* hq; bf16_t* hk; bf16_t* hv;  // destinations for column blocks 0,1,2 (each d_model wide)
  int d_model, n_head;   // head_dim fixed to 64
  int S, Spad;           // rows per sequence (M = B*S), padded sequence length of the destinations
  int v_transposed;      // 1: third block stored as V^T [B,H,64,Spad]; 0: [B,H,Spad,64]
  int first_block;       // which destination the first d_model columns go to (0=q,1=k)
};

// Launch on `stream`.  Returns CCX_OK or an error (message in ctx).
int ccx_launch_gemm(ccx_ctx* ctx, int epi, const GemmParams& p, hipStream_t stream);
