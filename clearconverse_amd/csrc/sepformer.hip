// sepformer.hip -- RE-SepFormer separator of libccx (K12-K17 in SURVEY.md section 2a).
//
// Replaces `self.separator.separate_batch(mix[1,T]) -> [1,T,2]` (reference back/api.py:1077; model
// speechbrain/resepformer-wsj02mix loaded at back/api.py:713).  Semantics follow SpeechBrain
// [UPSTREAM-RECALL]; the CPU restatement is oracle/sepformer_ref.py.
//
// Data layout: all utterances of a call are concatenated into ONE token buffer [n_tok, 128] (every
// utterance padded to whole 150-token chunks, as the reference's _padfeature does), so the segment
// transformer's GEMMs / LayerNorms run flat over all chunks of all utterances (ragged batches cost
// nothing) on the shared bf16 MFMA GEMM (gemm_bf16.hip).  Sequences (chunks for the segment model,
// one chunk-mean sequence per utterance for the memory model) are described by a (start, len)
// table consumed by the attention / global-norm kernels.
#include <map>
#include <math.h>
#include "../../include/ccx.h"
#include "ccx_common.h"
#include "elementwise.h"
#include <type_traits>
#include "gemm_bf16.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short bf16x4v;

// ---------------------------------------------------------------------------------------------
// Encoder: Conv1d(1 -> N=128, k=16, stride 8, no bias) + ReLU, written token-major [tok, 128] f32.
// Rows that are chunk padding are zero-filled.  One wave per token (2 filters per lane).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sep_encoder_kernel(const float* __restrict__ mix, long mix_stride,
                                                          const int* __restrict__ tok_utt, const int* __restrict__ tok_pos,
                                                          const int* __restrict__ utt_L, const float* __restrict__ w,  // [128][16]
                                                          float* __restrict__ feats, int n_tok) {
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (tok >= n_tok) return;
  const int u = tok_utt[tok], l = tok_pos[tok];
  float2 o = make_float2(0.f, 0.f);
  if (l < utt_L[u]) {
    const float* x = mix + (long)u * mix_stride + 8 * l;
    float xs[16];
#pragma unroll
    for (int k = 0; k < 16; k++) xs[k] = x[k];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      a = fmaf(w[(2 * lane) * 16 + k], xs[k], a);
      b = fmaf(w[(2 * lane + 1) * 16 + k], xs[k], b);
    }
    o = make_float2(fmaxf(a, 0.f), fmaxf(b, 0.f));
  }
  ((float2*)(feats + (long)tok * 128))[lane] = o;
}

// x' = x (+ hc[seq]) ; h = x' + pe[pos]   (block input and its positionally encoded stream)
__global__ void sep_block_input_kernel(const float* __restrict__ x, const float* __restrict__ hc, const int* __restrict__ tok_seq,
                                       const int* __restrict__ tok_pos, const float* __restrict__ pe, float* __restrict__ xin,
                                       float* __restrict__ h, int n_tok) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
  if (i >= (long)n_tok * 32) return;
  const int tok = (int)(i >> 5), c4 = (int)(i & 31);
  float4 v = ((const float4*)x)[i];
  if (hc) {
    const float4 a = ((const float4*)(hc + (long)tok_seq[tok] * 128))[c4];
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
  }
  ((float4*)xin)[i] = v;
  const float4 p = ((const float4*)(pe + (long)tok_pos[tok] * 128))[c4];
  ((float4*)h)[i] = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
}

// ---------------------------------------------------------------------------------------------
// Attention, head_dim 16, 8 heads, sequences given by (start, len).  One wave per (sequence, head,
// 16-query tile group); S^T = K Q^T with v_mfma_f32_16x16x16_bf16, so each lane owns one query
// column; the S^T accumulator registers are, unchanged, the A operand of O = P V (row = query,
// k = key), so P never leaves registers.  V is read as B operand (k = key, col = d).
// qkv: [n_tok, 384] bf16 (q | k | v, heads contiguous inside each 128).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void sep_attention_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ seq_start,
                                                            const int* __restrict__ seq_len, bf16_t* __restrict__ out,
                                                            float scale_log2e) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seq = blockIdx.x, head = blockIdx.y * 4 + wave;
  const int s0 = seq_start[seq], len = seq_len[seq];
  const int l15 = lane & 15, h4 = lane >> 4;
  const int n_qt = (len + 15) >> 4, n_kt = n_qt;
  const bf16_t* base = qkv + (long)s0 * 384 + head * 16;

  // K and V fragments of this (sequence, head).  Sequences of up to 160 tokens (every intra-chunk call: 150) keep them in
  // registers for all query tiles; the 2-byte V gathers then run once per sequence instead of once per query tile.
  auto load_k = [&](int kt) -> bf16x4v {
    int krow = kt * 16 + l15;
    krow = krow < len ? krow : len - 1;
    return *(const bf16x4v*)(base + 128 + (long)krow * 384 + 4 * h4);   // A operand: K[key l15][d = 4*h4 + j]
  };
  auto load_v = [&](int kt) -> bf16x4v {
    bf16x4v v;                                                           // B operand of O = P V: V[key = 4*h4 + j][d = l15]
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int vr = kt * 16 + 4 * h4 + j;
      vr = vr < len ? vr : len - 1;
      v[j] = (short)base[256 + (long)vr * 384 + l15];
    }
    return v;
  };
  const bool resident = n_kt <= 10;
  bf16x4v k_all[10], v_all[10];
  if (resident) {
#pragma unroll
    for (int t = 0; t < 10; t++) {
      const int kt = t < n_kt ? t : n_kt - 1;
      k_all[t] = load_k(kt);
      v_all[t] = load_v(kt);
    }
  }

  for (int qt = 0; qt < n_qt; qt++) {
    // B operand of S^T = K * Q^T:  B[k = d = 4*h4 + j][col = query l15]
    int qrow = qt * 16 + l15;
    qrow = qrow < len ? qrow : len - 1;
    const bf16x4v qf = *(const bf16x4v*)(base + (long)qrow * 384 + 4 * h4);
    float m_run = -1e30f, l_run = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};  // O tile: col = d (l15), rows = query 4*h4 + r
    // one block of up to 10 key tiles; FULL (exactly 10: every intra-chunk call) drops the per-tile branches, which
    // otherwise make up most of the instruction stream of this VALU-bound kernel
    auto key_block = [&](int kt0, int nk, auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      f32x4 s[10];
      bf16x4v vf[10];
#pragma unroll
      for (int t = 0; t < 10; t++) {
        if (FULL || t < nk) {
          const bf16x4v kf = resident ? k_all[t] : load_k(kt0 + t);
          s[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          vf[t] = resident ? v_all[t] : load_v(kt0 + t);
        }
      }
      // s[t][r] = score(query l15, key kt*16 + 4*h4 + r).  The kernel is VALU-bound (16-wide heads): keys past `len` can
      // only sit in the last key tile, and the scale is folded into the one fma in front of exp2 (m_run is kept scaled).
      float mx = -1e30f;
#pragma unroll
      for (int t = 0; t < 10; t++)
        if (FULL || t < nk) {
          if ((kt0 + t) * 16 + 16 > len) {
#pragma unroll
            for (int r = 0; r < 4; r++)
              if ((kt0 + t) * 16 + 4 * h4 + r >= len) s[t][r] = -INFINITY;
          }
#pragma unroll
          for (int r = 0; r < 4; r++) mx = fmaxf(mx, s[t][r]);
        }
      mx = fmaxf(mx, lane_xor16(mx));
      mx = fmaxf(mx, lane_xor32(mx));
      mx = fmaxf(m_run, mx * scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(m_run - mx);
      m_run = mx;
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 10; t++)
        if (FULL || t < nk) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(s[t][r], scale_log2e, -mx));
            s[t][r] = pv;
            ps += pv;
          }
        }
      l_run = l_run * alpha + ps;
      // alpha belongs to query l15 but the O tile holds queries 4*h4 + r on its rows: fetch per row
      float arow[4];
#pragma unroll
      for (int r = 0; r < 4; r++) arow[r] = __shfl(alpha, 4 * h4 + r, 64);
#pragma unroll
      for (int r = 0; r < 4; r++) o[r] *= arow[r];
#pragma unroll
      for (int t = 0; t < 10; t++)
        if (FULL || t < nk) {
          union { bf16x4v v; uint32_t u[2]; } pa;  // A operand: P[query l15][key 4*h4 + j]
          pa.u[0] = pack_bf16x2(s[t][0], s[t][1]);
          pa.u[1] = pack_bf16x2(s[t][2], s[t][3]);
          o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa.v, vf[t], o, 0, 0, 0);
        }
    };
    for (int kt0 = 0; kt0 < n_kt; kt0 += 10) {
      const int nk = (n_kt - kt0) < 10 ? (n_kt - kt0) : 10;
      if (nk == 10) key_block(kt0, 10, std::true_type{});
      else key_block(kt0, nk, std::false_type{});
    }
    float lt = l_run + lane_xor16(l_run);
    lt += lane_xor32(lt);
    // rows of the O tile are queries 4*h4 + r: fetch their normalisers
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int q = qt * 16 + 4 * h4 + r;
      const float inv = 1.0f / __shfl(lt, 4 * h4 + r, 64);
      if (q < len) out[(long)(s0 + q) * 128 + head * 16 + l15] = f32_to_bf16(o[r] * inv);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Final LayerNorm (eps 1e-6) of the encoder stack + GlobalLayerNorm over (time, channel) of each
// sequence + skip:  y = g_c * (LN(h) - mean) / sqrt(var + 1e-8) + b_c + xin.   One block/sequence,
// two passes (LN recomputed, h stays in L2).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 ln_row128(const float* __restrict__ row, const float* __restrict__ g,
                                            const float* __restrict__ b, int lane) {
  const float2 v = ((const float2*)row)[lane];
  const float mean = wave_reduce_sum(v.x + v.y) * (1.0f / 128.0f);
  const float dx = v.x - mean, dy = v.y - mean;
  const float rstd = rsqrtf(wave_reduce_sum(dx * dx + dy * dy) * (1.0f / 128.0f) + 1e-6f);
  const float2 gg = ((const float2*)g)[lane], bb = ((const float2*)b)[lane];
  return make_float2(dx * rstd * gg.x + bb.x, dy * rstd * gg.y + bb.y);
}

__global__ __launch_bounds__(256) void sep_final_norm_kernel(const float* __restrict__ h, const float* __restrict__ xin,
                                                             const int* __restrict__ seq_start, const int* __restrict__ seq_len,
                                                             const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                             const float* __restrict__ gln_g, const float* __restrict__ gln_b,
                                                             float* __restrict__ y) {
  __shared__ float red[8];
  const int seq = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s0 = seq_start[seq], len = seq_len[seq];
  float sum = 0.f;
  for (int r = wave; r < len; r += 4) {
    const float2 v = ln_row128(h + (long)(s0 + r) * 128, ln_g, ln_b, lane);
    sum += v.x + v.y;
  }
  sum = wave_reduce_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float n = (float)len * 128.0f;
  const float mean = (red[0] + red[1] + red[2] + red[3]) / n;
  float sq = 0.f;
  for (int r = wave; r < len; r += 4) {
    const float2 v = ln_row128(h + (long)(s0 + r) * 128, ln_g, ln_b, lane);
    sq += (v.x - mean) * (v.x - mean) + (v.y - mean) * (v.y - mean);
  }
  sq = wave_reduce_sum(sq);
  if (lane == 0) red[4 + wave] = sq;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[4] + red[5] + red[6] + red[7]) / n + 1e-8f);
  const float2 gg = ((const float2*)gln_g)[lane], bb = ((const float2*)gln_b)[lane];
  for (int r = wave; r < len; r += 4) {
    const float2 v = ln_row128(h + (long)(s0 + r) * 128, ln_g, ln_b, lane);
    const float2 xi = ((const float2*)(xin + (long)(s0 + r) * 128))[lane];
    ((float2*)(y + (long)(s0 + r) * 128))[lane] =
        make_float2(gg.x * (v.x - mean) * rstd + bb.x + xi.x, gg.y * (v.y - mean) * rstd + bb.y + xi.y);
  }
}

// ---------------------------------------------------------------------------------------------
// Fused position-wise FFN of a layer (d_model = 128):   h += W2 relu(W1 LN(h) + b1) + b2   in place.
// The two GEMMs of the unfused form move the d_ffn-wide hidden activation through HBM twice (4 KB of the
// 8.7 KB a token costs per layer); here a token's row is read once and written once.
//   * block = 8 waves x 32 tokens; a wave normalises its tokens in registers (a row sits in the four lanes
//     l15, l15+16, +32, +48) and keeps them as bf16 B-operand fragments for the whole tile;
//   * W1 / W2 stream through LDS by LDS-DMA in stages of 64 hidden units (16 KB + 16 KB) over a ring of four stages: two
//     stages stay in flight across raw s_barriers (counted s_waitcnt vmcnt), XOR-swizzled on the DMA source (ff_key1 /
//     ff_key2: conflict-free ds_read_b128 per tools/lds_bank_sim.py, tests/test_lds_layouts_cpu.py);
//   * MFMA operands are swapped as in gemm_bf16.hip (weights as "A"), so a lane owns 16 consecutive hidden
//     units of its token after GEMM 1 -- exactly a k-group of GEMM 2's B operand: relu + bf16 in registers,
//     no LDS round trip for the hidden activation.  The k order inside a 64-block is permuted to match
//     (hidden 16 hq + 8 s + e at MFMA k position 8 hq + e of step s); the W2 fragments follow it;
//   * the same permutation on GEMM 1's k axis makes the lane's input features its output features
//     (64 ob + 16 hq + 0..15), so the epilogue adds bias and the residual row 64 B at a time.
// ---------------------------------------------------------------------------------------------
#define FF_TOK 256
#define FF_HB 64                        // hidden units per stage
#define FF_P1 (FF_HB * 256)             // W1 part: HB rows x 128 features (256-byte rows, 16 chunks)   16 KB
#define FF_P2 (128 * FF_HB * 2)         // W2 part: 128 rows x HB hidden (128-byte rows, 8 chunks)     16 KB
#define FF_STAGE (FF_P1 + FF_P2)
#define FF_NSTAGE 4                     // LDS ring: stages st+1 .. st+2 stay in flight while stage st is consumed
#define FF_LDS (FF_NSTAGE * FF_STAGE + 4096)
typedef const __attribute__((address_space(1))) void* ff_gptr_t;
typedef __attribute__((address_space(3))) void* ff_lptr_t;

__device__ __forceinline__ int ff_key1(int r) { return (r & 3) | (((r >> 4) & 3) << 2); }          // 256-byte rows
__device__ __forceinline__ int ff_key2(int r) { return ((r >> 1) & 1) | (((r >> 5) & 1) << 2); }   // 128-byte rows (chunk = 2 hq + s)

// ---------------------------------------------------------------------------------------------
// Attention half of a transformer layer in ONE kernel for sequences of at most 160 tokens (every intra-chunk call: 150):
//   h += out_proj(attention(LayerNorm1(h) Wqkv^T + bqkv)) + bo
// Through separate kernels this half moves 650 MB per call over HBM (LayerNorm output, q|k|v, attention output, the residual
// twice); here a block (one sequence, 8 waves = 8 heads) reads its 150 x 128 residual rows once and writes them once.
//   1. LayerNorm of the rows -> bf16 in LDS (40 KB, 16-byte chunks XOR-swizzled by the row so that the fragment reads of 16
//      consecutive rows are conflict-free).
//   2. wave w = head w: q^T, k^T = W[16 w ..] x^T and v = x W^T with v_mfma_f32_16x16x32_bf16, weight fragments straight from
//      global memory into registers (48 rows x 128: 12 fragments).  The operand ORDER is chosen so that the accumulators already
//      are the operand layouts of the attention MFMAs: q^T / k^T tiles have the token on the lane (B / A operand of S^T = K Q^T),
//      the v tile has the head dimension on the lane (B operand of O = P V) -- q, k and v never leave registers.
//   3. attention as in sep_attention_kernel (K and V of all 160 keys resident), O tiles -> bf16 into the same LDS rows.
//   4. wave w = output columns 16 w ..: out-proj with the Wo fragments from global memory, C^T orientation (lane = token, four
//      consecutive columns), residual add, float4 store.
// ---------------------------------------------------------------------------------------------
constexpr int AB_MAX_TOK = 160;
__device__ __forceinline__ int ab_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }   // 16 chunks of 16 B per row

__global__ __launch_bounds__(512) void sep_attn_block_kernel(float* __restrict__ h, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                             const bf16_t* __restrict__ Wqkv, const float* __restrict__ bqkv,
                                                             const bf16_t* __restrict__ Wo, const float* __restrict__ bo,
                                                             const int* __restrict__ seq_start, const int* __restrict__ seq_len,
                                                             float scale_log2e, float eps) {
  __shared__ __attribute__((aligned(16))) char xs[AB_MAX_TOK * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, h4 = lane >> 4;
  const int seq = blockIdx.x;
  const int s0 = seq_start[seq], len = seq_len[seq];
  if (len <= 0) return;                                  // (block-uniform; the host never passes an empty sequence)
  const int n_t = (len + 15) >> 4;                       // token tiles (<= 10)

  // ---- 1. LayerNorm: wave w normalises rows w, w + 8, ...; a lane holds features 2 lane, 2 lane + 1 (ten rows at a time:
  //         keeps the live registers of this phase low) ----
  {
    const float2 g = ((const float2*)ln_g)[lane], b = ((const float2*)ln_b)[lane];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      float2 v[10];
#pragma unroll
      for (int i = 0; i < 10; i++) {
        const int r = wave + 8 * (10 * half + i);
        v[i] = r < len ? ((const float2*)(h + (long)(s0 + r) * 128))[lane] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 10; i++) {
        const int r = wave + 8 * (10 * half + i);
        const float mean = wave_reduce_sum(v[i].x + v[i].y) * (1.f / 128.f);
        const float dx = v[i].x - mean, dy = v[i].y - mean;
        const float rstd = rsqrtf(wave_reduce_sum(dx * dx + dy * dy) * (1.f / 128.f) + eps);
        uint32_t o = pack_bf16x2(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y);
        if (r >= len) o = 0;                             // rows past the sequence: zeros (their keys are masked, their queries dropped)
        *(uint32_t*)(xs + ab_off(r, lane >> 2) + (lane & 3) * 4) = o;
      }
    }
  }
  __syncthreads();

  // ---- 2. q^T, k^T, then v of head `wave` for every token tile (two passes over the rows: 32 + 16 weight registers at a time) ----
  bf16x4v q_all[10], k_all[10], v_all[10];
  {
    bf16x8 wq[4], wk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const long o = (long)(16 * wave + l15) * 128 + 32 * ks + 8 * h4;
      wq[ks] = *(const bf16x8*)(Wqkv + o);
      wk[ks] = *(const bf16x8*)(Wqkv + 128 * 128 + o);
    }
    const float4 bq = *(const float4*)(bqkv + 16 * wave + 4 * h4);
    const float4 bk = *(const float4*)(bqkv + 128 + 16 * wave + 4 * h4);
#pragma unroll
    for (int t = 0; t < 10; t++) {
      const int tt = t < n_t ? t : n_t - 1;              // tiles past the sequence repeat the last one (never used)
      f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq;
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const bf16x8 xf = *(const bf16x8*)(xs + ab_off(16 * tt + l15, 4 * ks + h4));
        aq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ks], xf, aq, 0, 0, 0);   // D[d][token]: lane = token, reg r = d 4 h4 + r
        ak = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk[ks], xf, ak, 0, 0, 0);
      }
      union { bf16x4v v; uint32_t u[2]; } c;
      c.u[0] = pack_bf16x2(aq[0] + bq.x, aq[1] + bq.y); c.u[1] = pack_bf16x2(aq[2] + bq.z, aq[3] + bq.w);
      q_all[t] = c.v;
      c.u[0] = pack_bf16x2(ak[0] + bk.x, ak[1] + bk.y); c.u[1] = pack_bf16x2(ak[2] + bk.z, ak[3] + bk.w);
      k_all[t] = c.v;
    }
  }
  {
    bf16x8 wv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) wv[ks] = *(const bf16x8*)(Wqkv + 256 * 128 + (long)(16 * wave + l15) * 128 + 32 * ks + 8 * h4);
    const float bv = bqkv[256 + 16 * wave + l15];
#pragma unroll
    for (int t = 0; t < 10; t++) {
      const int tt = t < n_t ? t : n_t - 1;
      f32x4 av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const bf16x8 xf = *(const bf16x8*)(xs + ab_off(16 * tt + l15, 4 * ks + h4));
        av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wv[ks], av, 0, 0, 0);   // D[token][d]: lane = d, reg r = token 4 h4 + r
      }
      union { bf16x4v v; uint32_t u[2]; } c;
      c.u[0] = pack_bf16x2(av[0] + bv, av[1] + bv); c.u[1] = pack_bf16x2(av[2] + bv, av[3] + bv);
      v_all[t] = c.v;
    }
  }
  __syncthreads();                                        // every wave is done with the normalised rows: the LDS rows take the O tiles now

  // ---- 3. attention of head `wave` (all keys resident; keys past `len` can only sit in the last key tile) ----
  for (int qt = 0; qt < n_t; qt++) {
    bf16x4v qf = q_all[0];
#pragma unroll
    for (int t = 1; t < 10; t++) qf = (t == qt) ? q_all[t] : qf;    // uniform select: q_all stays in registers
    f32x4 sc[10];
#pragma unroll
    for (int t = 0; t < 10; t++) sc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(k_all[t], qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    // sc[t][r] = score(query l15, key 16 t + 4 h4 + r)
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < 10; t++) {
      if (t * 16 + 16 > len) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (t * 16 + 4 * h4 + r >= len) sc[t][r] = -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 4; r++) mx = fmaxf(mx, sc[t][r]);
    }
    mx = fmaxf(mx, lane_xor16(mx));
    mx = fmaxf(mx, lane_xor32(mx));
    mx *= scale_log2e;
    float ps = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};                       // O tile: col = d (l15), rows = query 4 h4 + r
#pragma unroll
    for (int t = 0; t < 10; t++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(sc[t][r], scale_log2e, -mx));
        sc[t][r] = pv;
        ps += pv;
      }
      union { bf16x4v v; uint32_t u[2]; } pa;              // A operand: P[query l15][key 4 h4 + j]
      pa.u[0] = pack_bf16x2(sc[t][0], sc[t][1]);
      pa.u[1] = pack_bf16x2(sc[t][2], sc[t][3]);
      o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa.v, v_all[t], o, 0, 0, 0);
    }
    float lt = ps + lane_xor16(ps);
    lt += lane_xor32(lt);
    // rows of the O tile are queries 4 h4 + r: fetch their normalisers; column 16 wave + l15 of the attention output
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = qt * 16 + 4 * h4 + r;
      const float inv = 1.0f / __shfl(lt, 4 * h4 + r, 64);
      *(bf16_t*)(xs + ab_off(row, 2 * wave + (l15 >> 3)) + (l15 & 7) * 2) = f32_to_bf16(o[r] * inv);
    }
  }
  __syncthreads();

  // ---- 4. out-proj of columns 16 wave .. + 15, residual add ----
  {
    bf16x8 wo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) wo[ks] = *(const bf16x8*)(Wo + (long)(16 * wave + l15) * 128 + 32 * ks + 8 * h4);
    const float4 b4 = *(const float4*)(bo + 16 * wave + 4 * h4);
    for (int t = 0; t < n_t; t++) {
      const int tok = 16 * t + l15;
      float4* hp = (float4*)(h + (long)(s0 + (tok < len ? tok : len - 1)) * 128 + 16 * wave + 4 * h4);
      const float4 res = *hp;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const bf16x8 af = *(const bf16x8*)(xs + ab_off(tok, 4 * ks + h4));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[ks], af, acc, 0, 0, 0);    // D[col][token]: lane = token, reg r = col 4 h4 + r
      }
      if (tok < len) *hp = make_float4(res.x + acc[0] + b4.x, res.y + acc[1] + b4.y, res.z + acc[2] + b4.z, res.w + acc[3] + b4.w);
    }
  }
}

__global__ __launch_bounds__(512) void sep_ffn_kernel(float* __restrict__ h, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                      const bf16_t* __restrict__ W1, const float* __restrict__ b1,
                                                      const bf16_t* __restrict__ W2, const float* __restrict__ b2, int n_tok, int d_ffn,
                                                      float eps) {
  extern __shared__ __attribute__((aligned(16))) char ff_smem[];   // ring of FF_NSTAGE stages, then b1 (<= 1024 floats)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, hq = lane >> 4;
  const int tok0 = blockIdx.x * FF_TOK + wave * 32;
  const int n_stage = d_ffn / FF_HB;
  float* b1s = (float*)(ff_smem + FF_NSTAGE * FF_STAGE);

  // ---- weight staging: per stage and wave 2 DMA instructions of the W1 part (4 rows of 256 B each) and 2 of the
  //      W2 part (8 rows of 128 B each); the XOR swizzle is applied to the SOURCE chunk (the LDS image is lane-linear)
  const bf16_t* src1[2];
  const bf16_t* src2[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int r1 = (wave * 2 + i) * 4 + (lane >> 4);
    src1[i] = W1 + (long)r1 * 128 + (((lane & 15) ^ ff_key1(r1)) << 3);      // + stage * FF_HB rows
    const int r2 = (wave * 2 + i) * 8 + (lane >> 3);
    src2[i] = W2 + (long)r2 * d_ffn + (((lane & 7) ^ ff_key2(r2)) << 3);     // + stage * FF_HB columns
  }
  auto stage = [&](int st) {
    char* s1 = ff_smem + (st & (FF_NSTAGE - 1)) * FF_STAGE;
    char* s2 = s1 + FF_P1;
#pragma unroll
    for (int i = 0; i < 2; i++)
      __builtin_amdgcn_global_load_lds((ff_gptr_t)(src1[i] + (long)st * FF_HB * 128), (ff_lptr_t)(s1 + (wave * 2 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; i++)
      __builtin_amdgcn_global_load_lds((ff_gptr_t)(src2[i] + (long)st * FF_HB), (ff_lptr_t)(s2 + (wave * 2 + i) * 1024), 16, 0, 0);
  };

  // ---- LayerNorm of the wave's 32 tokens -> bf16 fragments xb[mt][ks] (features 64 (ks>>1) + 16 hq + 8 (ks&1) + 0..7) ----
  // (ordinary global loads: all of them retire before the first LDS-DMA is issued, see the vmcnt accounting below)
  bf16x8 xb[2][4];
  {
    float g[32], bt[32];
#pragma unroll
    for (int ob = 0; ob < 2; ob++)
#pragma unroll
      for (int i4 = 0; i4 < 4; i4++) {
        const float4 gv = ((const float4*)(ln_g + 64 * ob + 16 * hq))[i4], bv = ((const float4*)(ln_b + 64 * ob + 16 * hq))[i4];
        g[16 * ob + 4 * i4] = gv.x; g[16 * ob + 4 * i4 + 1] = gv.y; g[16 * ob + 4 * i4 + 2] = gv.z; g[16 * ob + 4 * i4 + 3] = gv.w;
        bt[16 * ob + 4 * i4] = bv.x; bt[16 * ob + 4 * i4 + 1] = bv.y; bt[16 * ob + 4 * i4 + 2] = bv.z; bt[16 * ob + 4 * i4 + 3] = bv.w;
      }
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      int tok = tok0 + mt * 16 + l15;
      tok = tok < n_tok ? tok : n_tok - 1;
      const float* xr = h + (long)tok * 128 + 16 * hq;
      float v[32];
#pragma unroll
      for (int ob = 0; ob < 2; ob++)
#pragma unroll
        for (int i4 = 0; i4 < 4; i4++) {
          const float4 t = ((const float4*)(xr + 64 * ob))[i4];
          v[16 * ob + 4 * i4] = t.x; v[16 * ob + 4 * i4 + 1] = t.y; v[16 * ob + 4 * i4 + 2] = t.z; v[16 * ob + 4 * i4 + 3] = t.w;
        }
      float sm = 0.f;
#pragma unroll
      for (int c = 0; c < 32; c++) sm += v[c];
      sm += lane_xor16(sm); sm += lane_xor32(sm);
      const float mean = sm * (1.f / 128.f);
      float sq = 0.f;
#pragma unroll
      for (int c = 0; c < 32; c++) { const float d = v[c] - mean; sq = fmaf(d, d, sq); }
      sq += lane_xor16(sq); sq += lane_xor32(sq);
      const float rstd = rsqrtf(sq * (1.f / 128.f) + eps);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        union { bf16x8 vv; uint32_t u[4]; } cv;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int c = 8 * ks + 2 * e;
          cv.u[e] = pack_bf16x2((v[c] - mean) * rstd * g[c] + bt[c], (v[c + 1] - mean) * rstd * g[c + 1] + bt[c + 1]);
        }
        xb[mt][ks] = cv.vv;
      }
    }
  }
  for (int i = tid; i < d_ffn; i += 512) b1s[i] = b1[i];     // b1 through LDS: a VGPR load inside the loop would drain the DMA ring

  f32x4 acc2[2][8];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int jo = 0; jo < 8; jo++) acc2[mt][jo] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int rq = 16 * (l15 >> 2) + (l15 & 3);            // permuted operand row inside a 64-row block (+ 4 j)
  const int k1 = l15;                                    // ff_key1 of that row (u | q << 2), independent of j
  const int k2 = ((l15 >> 1) & 1) | (((l15 >> 3) & 1) << 2);   // ff_key2 of that row
  // every wave issues the same 4 DMA instructions per stage, in stage order: with stages st+1, st+2 already issued,
  // vmcnt(8) means "my part of stage st has landed"; the barrier then makes all eight parts visible, and it also
  // tells that every wave is done reading stage st-1, whose buffer stage st+3 overwrites next.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stage(0);
  if (n_stage > 1) stage(1);
  if (n_stage > 2) stage(2);
  for (int st = 0; st < n_stage; st++) {
    const int ahead = n_stage - 1 - st;                  // stages issued after st (at most 2 here)
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st + 3 < n_stage) stage(st + 3);
    const char* s1 = ff_smem + (st & (FF_NSTAGE - 1)) * FF_STAGE;
    const char* s2 = s1 + FF_P1;
    f32x4 acc1[2][4];
    // bias slice of this stage: inline-asm reads (hipcc would wait vmcnt(0) -- the whole DMA ring -- in front of an
    // ordinary read of this part of the array); they are waited for after GEMM 1, where their latency is long gone
    f32x4 bv[4];
    {
      const uint32_t baddr = (uint32_t)(uintptr_t)(ff_lptr_t)(b1s + st * FF_HB + 16 * hq);
#pragma unroll
      for (int j = 0; j < 4; j++) asm volatile("ds_read_b128 %0, %1" : "=v"(bv[j]) : "v"(baddr + 16 * j) : "memory");
    }
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc1[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int chunk = 8 * (ks >> 1) + 2 * hq + (ks & 1);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bf16x8 w = *(const bf16x8*)(s1 + (rq + 4 * j) * 256 + ((chunk ^ k1) << 4));
        acc1[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xb[0][ks], acc1[0][j], 0, 0, 0);
        acc1[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xb[1][ks], acc1[1][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 sf[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int sx = 0; sx < 2; sx++) {
        union { bf16x8 vv; uint32_t u[4]; } cv;
        const f32x4 a = acc1[mt][2 * sx] + bv[2 * sx], b = acc1[mt][2 * sx + 1] + bv[2 * sx + 1];
        cv.u[0] = pack_bf16x2(fmaxf(a[0], 0.f), fmaxf(a[1], 0.f)); cv.u[1] = pack_bf16x2(fmaxf(a[2], 0.f), fmaxf(a[3], 0.f));
        cv.u[2] = pack_bf16x2(fmaxf(b[0], 0.f), fmaxf(b[1], 0.f)); cv.u[3] = pack_bf16x2(fmaxf(b[2], 0.f), fmaxf(b[3], 0.f));
        sf[mt][sx] = cv.vv;
      }
#pragma unroll
    for (int sx = 0; sx < 2; sx++) {
      const int chunk = 2 * hq + sx;
#pragma unroll
      for (int jo = 0; jo < 8; jo++) {
        const bf16x8 w = *(const bf16x8*)(s2 + (64 * (jo >> 2) + rq + 4 * (jo & 3)) * 128 + ((chunk ^ k2) << 4));
        acc2[0][jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, sf[0][sx], acc2[0][jo], 0, 0, 0);
        acc2[1][jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, sf[1][sx], acc2[1][jo], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: h[tok][64 ob + 16 hq + 4 jp + reg] += acc2 + b2 ----
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
    const int tok = tok0 + mt * 16 + l15;
    if (tok >= n_tok) continue;
#pragma unroll
    for (int ob = 0; ob < 2; ob++) {
      float4* hp = (float4*)(h + (long)tok * 128 + 64 * ob + 16 * hq);
      const float4* bp = (const float4*)(b2 + 64 * ob + 16 * hq);
#pragma unroll
      for (int jp = 0; jp < 4; jp++) {
        const float4 r = hp[jp], bv = bp[jp];
        const f32x4 a = acc2[mt][4 * ob + jp];
        hp[jp] = make_float4(a[0] + bv.x + r.x, a[1] + bv.y + r.y, a[2] + bv.z + r.z, a[3] + bv.w + r.w);
      }
    }
  }
}

// chunk means: mem[chunk][:] = mean over the chunk's 150 tokens
__global__ __launch_bounds__(128) void sep_chunk_mean_kernel(const float* __restrict__ x, float* __restrict__ mem, int seg) {
  const int chunk = blockIdx.x, c = threadIdx.x;
  const float* p = x + (long)chunk * seg * 128 + c;
  float s = 0.f;
  for (int t = 0; t < seg; t++) s += p[(long)t * 128];
  mem[(long)chunk * 128 + c] = s / (float)seg;
}

// PReLU (single slope) -> bf16 (input of the output 1x1 conv GEMM)
__global__ void sep_prelu_bf16_kernel(const float* __restrict__ x, const float* __restrict__ slope, bf16_t* __restrict__ out, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float a = slope[0];
  const float4 v = ((const float4*)x)[i];
  uint2 o;
  o.x = pack_bf16x2(v.x >= 0 ? v.x : a * v.x, v.y >= 0 ? v.y : a * v.y);
  o.y = pack_bf16x2(v.z >= 0 ? v.z : a * v.z, v.w >= 0 ? v.w : a * v.w);
  ((uint2*)out)[i] = o;
}

// ---------------------------------------------------------------------------------------------
// Mask (ReLU) * encoder output, ConvTranspose1d(128 -> 1, k16, s8) overlap-add, pad/trim to T:
//   est[u][t][spk] = sum_{l in {t/8, t/8-1}} sum_c feats[l][c] * relu(fc[l][2c+spk]) * wdec[c][t-8l]
// One wave per group of 8 output samples t = 8 l .. 8 l + 7 (both speakers): they all need exactly the token rows l (taps 0..7)
// and l - 1 (taps 8..15), which are read ONCE per wave -- with a wave per sample every 1.5 KB token row came out of L2
// sixteen times (18 GB per pipeline step, the whole 5.2 ms of the kernel).  Lanes split the 128 channels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sep_decoder_kernel(const float* __restrict__ feats, const float* __restrict__ fc,
                                                          const int* __restrict__ utt_tok0, const int* __restrict__ utt_L,
                                                          const int* __restrict__ utt_T, const float* __restrict__ wdec,  // [128][16]
                                                          float* __restrict__ out, long out_stride, int n_utt) {
  const int u = blockIdx.y;
  const int l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;     // sample group: t = 8 l + k
  const int T = utt_T[u], L = utt_L[u];
  const long t0 = 8L * l;
  if (t0 >= out_stride) return;
  float2* dst = (float2*)(out + ((long)u * out_stride + t0) * 2);
  if (t0 >= T || l > L) {  // rows past the utterance are zero (separate_batch pads / trims to T)
    if (lane < 8 && t0 + lane < out_stride) dst[lane] = make_float2(0.f, 0.f);
    return;
  }
  // g[token][spk][channel of the lane]: feats * relu(mask); token 0 = l (taps k), token 1 = l - 1 (taps k + 8)
  float g[2][2][2];
#pragma unroll
  for (int dl = 0; dl < 2; dl++) {
    const int ll = l - dl;
    const bool ok = ll >= 0 && ll < L;
    const long tok = utt_tok0[u] + (ok ? ll : 0);
    const float2 f = ((const float2*)(feats + tok * 128))[lane];          // channels 2*lane, 2*lane+1
    const float4 m = ((const float4*)(fc + tok * 256))[lane];             // (c0,s0) (c0,s1) (c1,s0) (c1,s1)
    const float z = ok ? 1.f : 0.f;
    g[dl][0][0] = z * f.x * fmaxf(m.x, 0.f); g[dl][0][1] = z * f.y * fmaxf(m.z, 0.f);
    g[dl][1][0] = z * f.x * fmaxf(m.y, 0.f); g[dl][1][1] = z * f.y * fmaxf(m.w, 0.f);
  }
  float4 w0[4], w1[4];                                                    // wdec rows of channels 2 lane, 2 lane + 1: 16 taps each
#pragma unroll
  for (int i = 0; i < 4; i++) {
    w0[i] = ((const float4*)(wdec + (2 * lane) * 16))[i];
    w1[i] = ((const float4*)(wdec + (2 * lane + 1) * 16))[i];
  }
  const float* w0f = (const float*)w0;
  const float* w1f = (const float*)w1;
  float2 res = make_float2(0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    float a0 = g[0][0][0] * w0f[k] + g[0][0][1] * w1f[k];
    float a1 = g[0][1][0] * w0f[k] + g[0][1][1] * w1f[k];
    a0 += g[1][0][0] * w0f[k + 8] + g[1][0][1] * w1f[k + 8];
    a1 += g[1][1][0] * w0f[k + 8] + g[1][1][1] * w1f[k + 8];
    a0 = wave_reduce_sum(a0);
    a1 = wave_reduce_sum(a1);
    if (lane == k) res = make_float2(a0, a1);
  }
  if (lane < 8 && t0 + lane < out_stride) dst[lane] = (t0 + lane < T) ? res : make_float2(0.f, 0.f);
}

struct HostT {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct SepLayer {
  float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *bqkv, *bo, *b1, *b2;
  bf16_t *Wqkv, *Wo, *W1, *W2;
};
struct SepBlock {
  std::vector<SepLayer> layers;
  float *lnf_g, *lnf_b, *gln_g, *gln_b;
};

inline bf16_t h2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

}  // namespace

struct ccx_sepformer {
  ccx_ctx* ctx = nullptr;
  ccx_sepformer_dims d{};
  int max_tokens = 0, max_utts = 0;
  bool finalized = false;
  std::map<std::string, HostT> staged;
  std::vector<void*> allocs;
  float *w_enc = nullptr, *w_dec = nullptr, *prelu = nullptr, *b_fc = nullptr, *pe = nullptr;
  bf16_t* W_fc = nullptr;
  std::vector<SepBlock> seg, mem;
  // workspaces
  float *feats = nullptr, *x = nullptr, *xin = nullptr, *h = nullptr, *fc = nullptr, *memx = nullptr, *memxin = nullptr,
        *memh = nullptr, *hc = nullptr;
  bf16_t *xn = nullptr, *qkv = nullptr, *att = nullptr;
  int *tok_utt = nullptr, *tok_pos = nullptr, *tok_chunk = nullptr, *tok_cpos = nullptr, *chunk_start = nullptr,
      *chunk_len = nullptr, *utt_L = nullptr, *utt_T = nullptr, *utt_tok0 = nullptr, *utt_chunk0 = nullptr, *utt_nchunk = nullptr,
      *mem_utt = nullptr, *mem_pos = nullptr;
  int pe_len = 0;
};

namespace {

#define STRY(expr)        \
  do {                    \
    int _rc = (expr);     \
    if (_rc) return _rc;  \
  } while (0)

template <typename T>
int salloc(ccx_sepformer* s, T** out, size_t count) {
  void* p = nullptr;
  const size_t bytes = ccx_align(count * sizeof(T), 256);
  CCX_HIP(s->ctx, hipMalloc(&p, bytes));
  CCX_HIP(s->ctx, hipMemset(p, 0, bytes));
  s->allocs.push_back(p);
  *out = (T*)p;
  return CCX_OK;
}
int sup_f32(ccx_sepformer* s, float** out, const float* src, size_t n) {
  STRY(salloc(s, out, n));
  CCX_HIP(s->ctx, hipMemcpy(*out, src, n * 4, hipMemcpyHostToDevice));
  return CCX_OK;
}
int sup_bf16(ccx_sepformer* s, bf16_t** out, const float* src, size_t n) {
  std::vector<bf16_t> tmp(n);
  for (size_t i = 0; i < n; i++) tmp[i] = h2bf(src[i]);
  STRY(salloc(s, out, n));
  CCX_HIP(s->ctx, hipMemcpy(*out, tmp.data(), n * 2, hipMemcpyHostToDevice));
  return CCX_OK;
}
int sneed(ccx_sepformer* s, const std::string& name, std::vector<int64_t> shape, const HostT** out) {
  auto it = s->staged.find(name);
  if (it == s->staged.end()) return ccx_fail(s->ctx, CCX_ERR_MISSING, "sepformer: tensor '%s' was never set", name.c_str());
  size_t want = 1, got = it->second.data.size();
  for (auto v : shape) want *= (size_t)v;
  if (want != got) return ccx_fail(s->ctx, CCX_ERR_ARG, "sepformer: tensor '%s' has %zu elements, expected %zu", name.c_str(), got, want);
  *out = &it->second;
  return CCX_OK;
}
#define SNEED(var, name, ...)                                              \
  const HostT* var = nullptr;                                              \
  STRY(sneed(s, (name), std::vector<int64_t>{__VA_ARGS__}, &var));

int load_block(ccx_sepformer* s, const std::string& prefix, SepBlock& B) {
  const int D = s->d.d_model, F = s->d.d_ffn;
  B.layers.resize(s->d.n_layers);
  for (int l = 0; l < s->d.n_layers; l++) {
    const std::string p = prefix + ".mdl.layers." + std::to_string(l);
    SepLayer& L = B.layers[l];
    SNEED(wi, p + ".self_att.att.in_proj_weight", 3 * D, D); SNEED(bi, p + ".self_att.att.in_proj_bias", 3 * D);
    SNEED(wo, p + ".self_att.att.out_proj.weight", D, D); SNEED(bo, p + ".self_att.att.out_proj.bias", D);
    SNEED(w1, p + ".pos_ffn.ffn.0.weight", F, D); SNEED(b1, p + ".pos_ffn.ffn.0.bias", F);
    SNEED(w2, p + ".pos_ffn.ffn.3.weight", D, F); SNEED(b2, p + ".pos_ffn.ffn.3.bias", D);
    SNEED(g1, p + ".norm1.norm.weight", D); SNEED(be1, p + ".norm1.norm.bias", D);
    SNEED(g2, p + ".norm2.norm.weight", D); SNEED(be2, p + ".norm2.norm.bias", D);
    STRY(sup_bf16(s, &L.Wqkv, wi->data.data(), wi->data.size())); STRY(sup_f32(s, &L.bqkv, bi->data.data(), 3 * D));
    STRY(sup_bf16(s, &L.Wo, wo->data.data(), wo->data.size())); STRY(sup_f32(s, &L.bo, bo->data.data(), D));
    STRY(sup_bf16(s, &L.W1, w1->data.data(), w1->data.size())); STRY(sup_f32(s, &L.b1, b1->data.data(), F));
    STRY(sup_bf16(s, &L.W2, w2->data.data(), w2->data.size())); STRY(sup_f32(s, &L.b2, b2->data.data(), D));
    STRY(sup_f32(s, &L.ln1_g, g1->data.data(), D)); STRY(sup_f32(s, &L.ln1_b, be1->data.data(), D));
    STRY(sup_f32(s, &L.ln2_g, g2->data.data(), D)); STRY(sup_f32(s, &L.ln2_b, be2->data.data(), D));
  }
  SNEED(fg, prefix + ".mdl.norm.norm.weight", D); SNEED(fb, prefix + ".mdl.norm.norm.bias", D);
  SNEED(gg, prefix + ".norm.weight", D); SNEED(gb, prefix + ".norm.bias", D);
  STRY(sup_f32(s, &B.lnf_g, fg->data.data(), D)); STRY(sup_f32(s, &B.lnf_b, fb->data.data(), D));
  STRY(sup_f32(s, &B.gln_g, gg->data.data(), D)); STRY(sup_f32(s, &B.gln_b, gb->data.data(), D));
  return CCX_OK;
}

// One SBTransformerBlock_wnormandskip over `n_tok` tokens organised in `n_seq` sequences.
// `max_len`: the longest sequence of the call (host side); up to 160 tokens the attention half of a layer is one kernel.
int run_block(ccx_sepformer* s, const SepBlock& B, const float* x, const float* hc, const int* tok_seq, const int* tok_pos,
              const int* seq_start, const int* seq_len, int n_tok, int n_seq, int max_len, float* xin, float* h, float* y, hipStream_t st) {
  ccx_ctx* ctx = s->ctx;
  const int D = s->d.d_model, F = s->d.d_ffn;
  hipLaunchKernelGGL(sep_block_input_kernel, dim3(ccx_cdiv(n_tok * 32, 256)), dim3(256), 0, st, x, hc, tok_seq, tok_pos, s->pe,
                     xin, h, n_tok);
  CCX_CHECK_LAUNCH(ctx);
  const float scale_log2e = 0.25f * 1.4426950408889634f;  // 1/sqrt(16)
  const char* fenv = getenv("CCX_SEP_FUSED_ATTN");          // read per call: tests compare the two paths in one process
  const bool fused = (fenv ? atoi(fenv) != 0 : true) && max_len <= AB_MAX_TOK;
  for (const SepLayer& L : B.layers) {
    GemmParams p;
    if (fused) {
      {
        // algorithmic work: QKV + out-proj GEMMs and the two attention products; bytes: the residual rows read and written once
        ccx_prof_scope ps(ctx, st, "sep_attn_block_kernel", 2.0 * n_tok * (double)D * 4 * D + 4.0 * n_tok * (double)max_len * D,
                          2.0 * n_tok * D * 4.0 + 2.0 * 4 * D * D);
        hipLaunchKernelGGL(sep_attn_block_kernel, dim3(n_seq), dim3(512), 0, st, h, L.ln1_g, L.ln1_b, L.Wqkv, L.bqkv, L.Wo, L.bo, seq_start,
                           seq_len, scale_log2e, 1e-6f);
      }
      CCX_CHECK_LAUNCH(ctx);
    } else {
    STRY(ccx_launch_layernorm(ctx, h, D, L.ln1_g, L.ln1_b, s->xn, nullptr, D, n_tok, D, 1e-6f, st));
    memset(&p, 0, sizeof(p));
    p.A = s->xn; p.lda = D; p.W = L.Wqkv; p.ldw = D; p.M = n_tok; p.N = 3 * D; p.K = D; p.bias = L.bqkv; p.out = s->qkv; p.ldo = 3 * D;
    STRY(ccx_launch_gemm(ctx, EPI_BF16, p, st));
    {
      ccx_prof_scope ps(ctx, st, "sep_attention_kernel", 0.0, 0.0);
      hipLaunchKernelGGL(sep_attention_kernel, dim3(n_seq, s->d.n_head / 4), dim3(256), 0, st, s->qkv, seq_start, seq_len, s->att,
                         scale_log2e);
    }
    CCX_CHECK_LAUNCH(ctx);
    memset(&p, 0, sizeof(p));
    p.A = s->att; p.lda = D; p.W = L.Wo; p.ldw = D; p.M = n_tok; p.N = D; p.K = D; p.bias = L.bo; p.out = h; p.ldo = D; p.resid = h; p.ldr = D;
    STRY(ccx_launch_gemm(ctx, EPI_F32_RESID, p, st));
    }
    {
      // LayerNorm 2 + Linear-ReLU-Linear + residual in one kernel (see sep_ffn_kernel)
      static ccx_lds_optin optin;
      CCX_HIP(ctx, optin.ensure(ctx->device, (const void*)sep_ffn_kernel, FF_LDS));
      ccx_prof_scope ps(ctx, st, "sep_ffn_kernel", 4.0 * n_tok * (double)D * F, 2.0 * n_tok * D * 4.0 + 4.0 * D * F);
      hipLaunchKernelGGL(sep_ffn_kernel, dim3(ccx_cdiv(n_tok, FF_TOK)), dim3(512), FF_LDS, st, h, L.ln2_g, L.ln2_b, L.W1, L.b1, L.W2, L.b2,
                         n_tok, F, 1e-6f);
    }
    CCX_CHECK_LAUNCH(ctx);
  }
  hipLaunchKernelGGL(sep_final_norm_kernel, dim3(n_seq), dim3(256), 0, st, h, xin, seq_start, seq_len, B.lnf_g, B.lnf_b, B.gln_g,
                     B.gln_b, y);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

}  // namespace

extern "C" {

int ccx_sepformer_create(ccx_ctx* ctx, const ccx_sepformer_dims* dims, int max_tokens, int max_utts, ccx_sepformer** out) {
  if (!ctx) return CCX_ERR_ARG;
  CCX_REQUIRE(ctx, dims && out, "ccx_sepformer_create: null argument");
  const ccx_sepformer_dims& d = *dims;
  CCX_REQUIRE(ctx, d.n_filters == 128 && d.d_model == 128 && d.kernel == 16 && d.stride == 8 && d.n_head == 8 && d.n_spk == 2,
              "sepformer: only the resepformer-wsj02mix geometry (128 filters, k16 s8, d_model 128, 8 heads, 2 speakers) is built");
  CCX_REQUIRE(ctx, d.d_ffn % 128 == 0 && d.d_ffn <= 1024 && d.segment >= 16 && d.n_layers >= 1 && d.n_blocks >= 1, "sepformer: bad dims");
  CCX_REQUIRE(ctx, max_tokens >= d.segment && max_utts >= 1, "sepformer: capacity too small");
  ccx_sepformer* s = new ccx_sepformer();
  s->ctx = ctx; s->d = d;
  s->max_tokens = ccx_cdiv(max_tokens, d.segment) * d.segment;
  s->max_utts = max_utts;
  *out = s;
  return CCX_OK;
}

void ccx_sepformer_destroy(ccx_sepformer* s) {
  if (!s) return;
  for (void* p : s->allocs) hipFree(p);
  delete s;
}

int ccx_sepformer_set_tensor(ccx_sepformer* s, const char* name, const float* data, int64_t numel) {
  if (!s) return CCX_ERR_ARG;
  CCX_REQUIRE(s->ctx, !s->finalized && name && data && numel > 0, "sepformer: set_tensor bad arguments");
  const std::string nm(name);
  CCX_REQUIRE(s->ctx, nm.rfind("encoder.", 0) == 0 || nm.rfind("decoder.", 0) == 0 || nm.rfind("masknet.", 0) == 0,
              "sepformer: unknown tensor name '%s'", name);
  HostT t;
  t.data.resize((size_t)numel);
  CCX_HIP(s->ctx, hipMemcpy(t.data.data(), data, (size_t)numel * 4, hipMemcpyDefault));
  s->staged[nm] = std::move(t);
  return CCX_OK;
}

int ccx_sepformer_finalize(ccx_sepformer* s) {
  if (!s) return CCX_ERR_ARG;
  CCX_REQUIRE(s->ctx, !s->finalized, "sepformer: finalize called twice");
  const ccx_sepformer_dims& d = s->d;
  const int D = d.d_model, F = d.d_ffn, N = d.n_filters;
  {
    SNEED(we, "encoder.conv1d.weight", N, 1, d.kernel);
    SNEED(wd, "decoder.weight", N, 1, d.kernel);
    SNEED(pr, "masknet.model.output_fc.0.weight", 1);
    SNEED(wf, "masknet.model.output_fc.1.weight", N * d.n_spk, D, 1);
    SNEED(bf, "masknet.model.output_fc.1.bias", N * d.n_spk);
    STRY(sup_f32(s, &s->w_enc, we->data.data(), we->data.size()));
    STRY(sup_f32(s, &s->w_dec, wd->data.data(), wd->data.size()));
    STRY(sup_f32(s, &s->prelu, pr->data.data(), 1));
    STRY(sup_bf16(s, &s->W_fc, wf->data.data(), wf->data.size()));
    STRY(sup_f32(s, &s->b_fc, bf->data.data(), bf->data.size()));
  }
  s->seg.resize(d.n_blocks);
  s->mem.resize(d.n_blocks > 0 ? d.n_blocks - 1 : 0);
  for (int i = 0; i < d.n_blocks; i++) {
    STRY(load_block(s, "masknet.model.seg_model." + std::to_string(i), s->seg[i]));
    if (i < d.n_blocks - 1) STRY(load_block(s, "masknet.model.mem_model." + std::to_string(i), s->mem[i]));
  }
  s->staged.clear();
  // positional encoding table (interleaved sin/cos), long enough for a chunk and for the longest memory sequence
  const int max_chunks = s->max_tokens / d.segment;
  s->pe_len = d.segment > max_chunks ? d.segment : max_chunks;
  {
    std::vector<float> pe((size_t)s->pe_len * D);
    for (int p = 0; p < s->pe_len; p++)
      for (int i = 0; i < D; i += 2) {
        const float den = expf((float)i * -(logf(10000.0f) / (float)D));
        pe[(size_t)p * D + i] = sinf((float)p * den);
        pe[(size_t)p * D + i + 1] = cosf((float)p * den);
      }
    STRY(sup_f32(s, &s->pe, pe.data(), pe.size()));
  }
  const size_t T = (size_t)s->max_tokens;
  STRY(salloc(s, &s->feats, T * D)); STRY(salloc(s, &s->x, T * D)); STRY(salloc(s, &s->xin, T * D)); STRY(salloc(s, &s->h, T * D));
  STRY(salloc(s, &s->fc, T * 2 * D)); STRY(salloc(s, &s->xn, T * D)); STRY(salloc(s, &s->qkv, T * 3 * D));
  STRY(salloc(s, &s->att, T * D));
  STRY(salloc(s, &s->memx, (size_t)max_chunks * D)); STRY(salloc(s, &s->memxin, (size_t)max_chunks * D));
  STRY(salloc(s, &s->memh, (size_t)max_chunks * D)); STRY(salloc(s, &s->hc, (size_t)max_chunks * D));
  STRY(salloc(s, &s->tok_utt, T)); STRY(salloc(s, &s->tok_pos, T)); STRY(salloc(s, &s->tok_chunk, T)); STRY(salloc(s, &s->tok_cpos, T));
  STRY(salloc(s, &s->chunk_start, (size_t)max_chunks)); STRY(salloc(s, &s->chunk_len, (size_t)max_chunks));
  STRY(salloc(s, &s->mem_utt, (size_t)max_chunks)); STRY(salloc(s, &s->mem_pos, (size_t)max_chunks));
  const size_t U = (size_t)s->max_utts;
  STRY(salloc(s, &s->utt_L, U)); STRY(salloc(s, &s->utt_T, U)); STRY(salloc(s, &s->utt_tok0, U)); STRY(salloc(s, &s->utt_chunk0, U));
  STRY(salloc(s, &s->utt_nchunk, U));
  s->finalized = true;
  return CCX_OK;
}

int ccx_sepformer_separate(ccx_sepformer* s, const float* mix, int64_t stride, const int* n_samples, int B, float* out,
                           void* stream_) {
  if (!s) return CCX_ERR_ARG;
  ccx_ctx* ctx = s->ctx;
  hipStream_t st = (hipStream_t)stream_;
  CCX_REQUIRE(ctx, s->finalized && mix && n_samples && out && B >= 1 && B <= s->max_utts, "sepformer_separate: bad arguments (B=%d, max %d)", B, s->max_utts);
  const ccx_sepformer_dims& d = s->d;
  const int D = d.d_model, seg = d.segment;
  // ---- host-side ragged bookkeeping (tiny) ----
  std::vector<int> uL(B), uT(B), utok0(B), uchunk0(B), unchunk(B);
  int n_tok = 0, n_chunk = 0, max_nchunk = 0;
  for (int b = 0; b < B; b++) {
    CCX_REQUIRE(ctx, n_samples[b] >= d.kernel && n_samples[b] <= stride, "sepformer_separate: utterance %d has %d samples (need >= %d, <= stride)", b, n_samples[b], d.kernel);
    uT[b] = n_samples[b];
    uL[b] = (n_samples[b] - d.kernel) / d.stride + 1;
    const int rest = seg - uL[b] % seg;  // a full extra chunk when L % seg == 0, as _padfeature does
    unchunk[b] = (uL[b] + rest) / seg;
    utok0[b] = n_tok; uchunk0[b] = n_chunk;
    n_tok += unchunk[b] * seg; n_chunk += unchunk[b];
    max_nchunk = unchunk[b] > max_nchunk ? unchunk[b] : max_nchunk;
  }
  CCX_REQUIRE(ctx, n_tok <= s->max_tokens, "sepformer_separate: %d tokens exceed capacity %d", n_tok, s->max_tokens);
  CCX_REQUIRE(ctx, n_chunk <= s->pe_len, "sepformer_separate: memory sequence too long");
  std::vector<int> t_utt(n_tok), t_pos(n_tok), t_chunk(n_tok), t_cpos(n_tok), c_start(n_chunk), c_len(n_chunk), m_utt(n_chunk), m_pos(n_chunk);
  for (int b = 0; b < B; b++) {
    for (int c = 0; c < unchunk[b]; c++) {
      const int cg = uchunk0[b] + c;
      c_start[cg] = utok0[b] + c * seg; c_len[cg] = seg; m_utt[cg] = b; m_pos[cg] = c;
      for (int i = 0; i < seg; i++) {
        const int t = c_start[cg] + i;
        t_utt[t] = b; t_pos[t] = c * seg + i; t_chunk[t] = cg; t_cpos[t] = i;
      }
    }
  }
#define UP(dst, vec) CCX_HIP(ctx, hipMemcpyAsync(dst, vec.data(), vec.size() * 4, hipMemcpyHostToDevice, st))
  UP(s->tok_utt, t_utt); UP(s->tok_pos, t_pos); UP(s->tok_chunk, t_chunk); UP(s->tok_cpos, t_cpos);
  UP(s->chunk_start, c_start); UP(s->chunk_len, c_len); UP(s->mem_utt, m_utt); UP(s->mem_pos, m_pos);
  UP(s->utt_L, uL); UP(s->utt_T, uT); UP(s->utt_tok0, utok0); UP(s->utt_chunk0, uchunk0); UP(s->utt_nchunk, unchunk);
#undef UP
  CCX_HIP(ctx, hipStreamSynchronize(st));  // host staging vectors go out of scope

  hipLaunchKernelGGL(sep_encoder_kernel, dim3(ccx_cdiv(n_tok, 4)), dim3(256), 0, st, mix, (long)stride, s->tok_utt, s->tok_pos,
                     s->utt_L, s->w_enc, s->feats, n_tok);
  CCX_CHECK_LAUNCH(ctx);
  const float* cur = s->feats;
  const float* hc = nullptr;
  for (int i = 0; i < d.n_blocks; i++) {
    STRY(run_block(s, s->seg[i], cur, hc, s->tok_chunk, s->tok_cpos, s->chunk_start, s->chunk_len, n_tok, n_chunk, seg, s->xin, s->h,
                   s->x, st));
    cur = s->x;
    if (i < d.n_blocks - 1) {
      hipLaunchKernelGGL(sep_chunk_mean_kernel, dim3(n_chunk), dim3(128), 0, st, s->x, s->memx, seg);
      CCX_CHECK_LAUNCH(ctx);
      // memory transformer: one sequence per utterance over its chunk means
      STRY(run_block(s, s->mem[i], s->memx, nullptr, s->mem_utt, s->mem_pos, s->utt_chunk0, s->utt_nchunk, n_chunk, B, max_nchunk, s->memxin,
                     s->memh, s->hc, st));
      hc = s->hc;
    }
  }
  hipLaunchKernelGGL(sep_prelu_bf16_kernel, dim3(ccx_cdiv(n_tok * 32, 256)), dim3(256), 0, st, s->x, s->prelu, s->xn, (long)n_tok * 32);
  CCX_CHECK_LAUNCH(ctx);
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = s->xn; p.lda = D; p.W = s->W_fc; p.ldw = D; p.M = n_tok; p.N = 2 * D; p.K = D; p.bias = s->b_fc; p.out = s->fc; p.ldo = 2 * D;
  STRY(ccx_launch_gemm(ctx, EPI_F32, p, st));
  hipLaunchKernelGGL(sep_decoder_kernel, dim3(ccx_cdiv(ccx_cdiv((int)stride, 8), 4), B), dim3(256), 0, st, s->feats, s->fc, s->utt_tok0, s->utt_L,
                     s->utt_T, s->w_dec, out, (long)stride, B);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

}  // extern "C"
