// decoder.h -- parameter blocks and launchers for the decoder step kernels (decoder.hip).
#pragma once
#include "ccx_common.h"

// ACT_BF16_LN: bf16 rows of the RAW (resolved, not normalised) residual stream + per-row LayerNorm statistics; the LayerNorm is applied
// algebraically in the epilogue: LN(x) W^T + b = rstd (x (gamma o W)^T - mean s) + c with s_n = sum_k (gamma o W)_nk, c_n = sum_k beta_k W_nk + b_n
// (the "LayerNorm-free" decode chain of the X-stream path, whisper.hip)
enum { ACT_LN = 0, ACT_BF16 = 1, ACT_COMBINE = 2, ACT_BF16_LN = 3 };
// DEPI_RESOLVE: the residual add done by the PRODUCER (no split-K slabs): x[m][n] += acc + bias in place (fp32), a bf16 copy of the new
// row and, per 16-column tile of every row, (sum, sum of squares) -- the statistics the next ACT_BF16_LN consumer normalises with
enum { DEPI_BF16_GELU = 1, DEPI_PARTIAL = 2, DEPI_F32 = 3, DEPI_SELF_QKV = 4, DEPI_RESOLVE = 5 };

struct DecLinearParams {
  int M, N, K;
  const bf16_t* W; long ldw;      // [N][K]
  const float* bias;              // [N] or null
  // activation sources
  const float* x;                 // ACT_LN: [M][K] f32 residual stream (before the pending partials)
  const float* pend; int pend_n; long pend_stride;  // pending split-K partials [pend_n][M][K] folded into x
  float* x_out;                   // ACT_LN: if non-null, block (0,*,0) writes x + sum(pend) here (must differ from x)
  const float* ln_g; const float* ln_b; float eps;
  const bf16_t* act; long lda;    // ACT_BF16: [M][K]
  const float* part_o; const float* part_ml; int nsplit;  // ACT_COMBINE: [M][H][nsplit][64], [M][H][nsplit][2]
  // ACT_BF16_LN: statistics of the input rows [M][K / 16] (sum, sum of squares per 16-column tile), s [N]; `bias` holds c [N]
  const float2* ln_stats; const float* ln_s;
  // DEPI_RESOLVE: xres [M][N] f32 (read and written in place), xb [M][N] bf16, st_out [M][N / 16]
  float* xres; bf16_t* xb; float2* st_out;
  // outputs (DEPI_PARTIAL: out[z][m][n] with stride pend_stride between the grid.z slices)
  void* out; long ldo;
  // DEPI_SELF_QKV: q -> out (f32 [M][K]), k/v -> caches [sequence][H][cache_T][64] at pos[m]; the sequence of row m is
  // row_seq[m] (prompt prefill: several rows per sequence) or m itself when row_seq is null
  bf16_t* cache_k; bf16_t* cache_v; int cache_T; const int* pos; const int* row_seq;
};
int ccx_launch_dec_linear(ccx_ctx* ctx, int act, int epi, const DecLinearParams& p, hipStream_t stream);
// number of grid.z K-slices ccx_launch_dec_linear will use (= number of partial slabs written)
int ccx_dec_linear_ksplit(int K, int epi);
// out = bf16 LayerNorm(x + sum pend); if x_out != null also writes the resolved x there (must not alias x)
int ccx_launch_dec_resolve_ln(ccx_ctx* ctx, const float* x, const float* pend, int pend_n, long pend_stride, const float* g,
                              const float* b, bf16_t* out, float* x_out, int M, int K, float eps, hipStream_t stream);

// LayerNorm-free chain: x += sum pend (in place), xb = bf16(x), st = (sum, sum of squares) per 16-column tile of every row
int ccx_launch_dec_resolve_stats(ccx_ctx* ctx, float* x, const float* pend, int pend_n, long pend_stride, bf16_t* xb, float2* st, int M,
                                 int K, hipStream_t stream);

struct DecAttnParams {
  const float* q;       // [B][H][64] f32
  const bf16_t* k;      // [B][H][kv_T][64]
  const bf16_t* v;
  int H, kv_T;
  const int* pos;       // if non-null: keys = pos[b] + 1 (self attention), else T
  int T;
  float scale_log2e;
  bf16_t* out_bf16;     // FINAL: [B][H*64]
  float* part_o;        // partials [B][H][nsplit][64]
  float* part_ml;       // [B][H][nsplit][2]
  int lds_pad;          // dynamic LDS the blocks claim without using it: caps the blocks per CU (see ccx_whisper_decode)
  int stream_mode;      // cross attention only: 1 = dec_cross_stream_kernel (few waves, few bytes in flight per CU)
  // prompt prefill: rows (q / out index) != sequences (K/V index).  row_seq [rows] maps them (null: identity); rows_per_seq > 1
  // tells the cross attention that the rows of a sequence are consecutive, so that it can co-schedule them on one XCD
  const int* row_seq; int rows_per_seq;
  // fused query projection (small batches, ccx_launch_dec_cross_fused_q): q = LN(x + pending slabs) * Wq^T + bq computed by the
  // attention block itself instead of by a launch of its own.  Wq: the fragment-packed image dec_linear streams.
  const float* qx; const float* q_pend; int q_pend_n; long q_pend_stride; float* q_x_out;
  const float* q_ln_g; const float* q_ln_b; float q_eps;
  const bf16_t* q_W; const float* q_bias; int q_K;
};
int ccx_launch_dec_attention(ccx_ctx* ctx, const DecAttnParams& p, int B, int nsplit, bool final_out, hipStream_t stream);
// Cross attention of a small batch (B <= 16 rows) WITH its query projection: replaces ln_linear(Wcq) + ccx_launch_dec_attention(split
// partials).  Same arithmetic, operation for operation, as the two launches it replaces (q is bit-identical).
int ccx_launch_dec_cross_fused_q(ccx_ctx* ctx, const DecAttnParams& p, int B, int nsplit, hipStream_t stream);

struct DecSeqState {
  int pos, prompt_len, n_gen, done;
  int last_tok, pen_tok, last_ts_tok, n_tokens;
  float sum_logprob, no_speech_prob;
};

struct DecSelectParams {
  const float* logits; long ld_logits; int n_vocab;
  DecSeqState* state;
  const int* prompt; int max_prompt;
  int* cur_tok; int* pos;
  int* gen; int sample_len;
  int* n_done;
  const unsigned char* suppress_mask;  // [n_vocab rounded up to 4]
  int eot, blank, no_speech, timestamp_begin, max_initial_ts;
  // next-step embedding written by the select kernel: x[b] = tok_emb[next] + pos_emb[pos]
  const float* tok_emb; const float* pos_emb; float* x; int D;
  // sampling (GreedyDecoder.update with temperature > 0): sample_cfg = {float temperature, u32 seed_lo, u32 seed_hi} in
  // device memory (so that a captured step graph serves every temperature / seed); row0 = batch row of sequence 0 of
  // this launch (lanes), so that the noise of a sequence does not depend on how the batch is split
  const unsigned* sample_cfg; int row0;
  int sample;   // 0: greedy kernel (sample_cfg ignored); 1: kernel with the temperature > 0 branch
};
int ccx_launch_dec_select(ccx_ctx* ctx, const DecSelectParams& p, int B, hipStream_t stream);
int ccx_launch_dec_embed(ccx_ctx* ctx, const float* tok_emb, const float* pos_emb, const int* cur_tok, const int* pos,
                         float* x, int B, int D, hipStream_t stream);
// dst[i][:] = src[idx[i]][:] where idx[i] >= 0 (bf16 rows of D elements): the last prompt row of every sequence after a prefill pass
int ccx_launch_dec_gather_rows(ccx_ctx* ctx, const bf16_t* src, const int* idx, bf16_t* dst, int n, int D, hipStream_t stream);
int ccx_launch_dec_combine(ccx_ctx* ctx, const float* part_o, const float* part_ml, int nsplit, bf16_t* out, int M, int H,
                           hipStream_t stream);
