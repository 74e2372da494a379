// cross_x.h -- decode cross attention against the ENCODER OUTPUT (cross_x.hip).
#pragma once
#include "ccx_common.h"

// Whisper's decoder cross attention (openai-whisper model.py MultiHeadAttention with xa; called once per layer and decode step from
// back/api.py:1286-1292 / 1432-1438 / 1474-1480 through transcribe()) reads, per layer, K = xa Wk^T and V = xa Wv^T + bv of every
// sequence: 2 x 1500 x 768 bf16 per layer and sequence, 12 different caches for the 12 layers -- 94 % of a decode step's HBM bytes.
// Both are linear images of the SAME xa, so
//     scores_h[j] = q_h . K_j,h           = (q_h Wk_h) . xa_j            -- "expanded query" q'_h = q_h Wk_h, one 768-vector per head
//     out_h       = sum_j p_hj V_j,h      = (sum_j p_hj xa_j) Wv_h^T + bv -- (sum_j p_hj = 1)
// and one pass over xa (1500 x 768 bf16, shared by all layers) serves all heads: half the bytes per layer, and 1/24 of the cache.
// The price is 12 x the matrix work (every head against all 768 features instead of its 64), which goes to the matrix cores:
// heads are the N index of the MFMAs.
struct XsParams {
  const float* q;        // [rows][D] f32 queries (bias included, not scaled) -- used when x is null
  // query projection inside the expansion (dec_xq_fused_kernel): q = LN(x + pending slabs) Wq^T + bq computed per (head, 16 rows) block
  const float* x; const float* pend; int pend_n; long pend_stride;   // residual rows [rows][D] and split-K partial slabs (decoder.h)
  float* x_out;          // if non-null: the resolved residual rows are written here (must differ from x)
  const float* ln_g; const float* ln_b; float eps;
  const bf16_t* Wq;      // [D][D] row-major cross_attn.query.weight
  const float* bq;       // [D]
  // LayerNorm-free chain (dec_xq_lnfree_kernel): the raw residual rows as bf16 + their per-tile statistics; Wq then holds gamma o Wq,
  // bq the folded constant c, and q = rstd (xb Wq^T - mean ln_s) + c  (decoder.h, ACT_BF16_LN)
  const bf16_t* xb; const float2* ln_stats; const float* ln_s;
  const bf16_t* WkT;     // [H][D][64]: WkT[h][f][d] = Wk[h*64 + d][f]   (cross_attn.key.weight re-laid per head)
  bf16_t* xq;            // [rows][H][D]: expanded queries
  float* part_o;         // [rows][XS_SPLIT][H][D]: unnormalised contexts of each key half (ccx_xs_part_o_elems)
  float* part_ml;        // [rows][XS_SPLIT][16][2]: reference maximum and denominator of each key half (ccx_xs_part_ml_elems)
  const bf16_t* X;       // [sequences][x_seq_stride]: encoder output rows [S][D] per sequence
  long x_seq_stride;     // elements between sequences
  const int* row_seq;    // row -> sequence (prompt prefill: several rows per sequence); null: identity
  int rows_per_seq;      // > 1: rows s * rows_per_seq .. + rows_per_seq - 1 all belong to sequence row_seq[s * rows_per_seq] (prefill)
  const bf16_t* Wv;      // [D][D] row-major cross_attn.value.weight
  const float* bv;       // [D]
  bf16_t* out;           // [rows][D] attention output (input of cross_attn.out)
  int rows, H, S, D;
  float scale_log2e;     // (d_head ^ -0.25)^2 * log2(e)
  int lds_pad;           // LDS the streaming blocks claim without using it (occupancy cap while decode lanes overlap)
};

// key ranges per row, each streamed by a block of its own (fixed: a row's arithmetic must not depend on the launch)
#define XS_SPLIT 2
static inline size_t ccx_xs_part_o_elems(size_t rows, int H, int D) { return rows * XS_SPLIT * (size_t)H * D; }
static inline size_t ccx_xs_part_ml_elems(size_t rows) { return rows * XS_SPLIT * 16 * 2; }
// the three launches of one layer's cross attention: q' = expand(q); partial contexts of the key halves; out = merge(partials) Wv^T + bv
int ccx_launch_xs_cross_attention(ccx_ctx* ctx, const XsParams& p, hipStream_t stream);
// widths the kernels are instantiated for
bool ccx_xs_supported(int D, int H);
