// cross_x.hip -- decode cross attention against the ENCODER OUTPUT xa instead of per-layer K/V caches (see cross_x.h for the algebra;
// reference call sites: back/api.py:1286-1292, 1432-1438, 1474-1480 -> openai-whisper MultiHeadAttention(x, xa) per decoder layer).
//
// Three launches per layer:
//   dec_xq_fused_kernel    q = LN(x + pending slabs) Wq^T + bq, then q'[row][h][:] = q[row][h*64 .. +64] Wk_h  (64 -> D per head)
//                          (dec_xq_expand_kernel: the expansion alone, from a given q -- the C-ABI operator, CCX_XS_FUSE_Q=0)
//   dec_xs_stream_kernel   partial contexts sum_j exp2(q'[row][h] . xa_j * scale - m) xa_j of a key half: ONE pass over the
//                          sequence's xa for all heads -- the HBM-bound part
//   dec_xv_project_kernel  merges the key halves, out[row][h*64 .. +64] = ctx[row][h] Wv_h^T + bv         (D -> 64 per head)
// (the kernels carry their own descriptions)
#include "cross_x.h"
#include "dec_ln.h"

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ bf16x8 pack8(const float4 a, const float4 b) {
  const u32x4 v = {pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w)};
  return __builtin_bit_cast(bf16x8, v);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// q' = per-head expansion of the query.  Block = (head, 16 rows); wave w owns features [w D/4, (w+1) D/4).
// MFMA roles: A = WkT_h [feature][d] (M = features), B = q^T [d][row] (N = rows), K = 64 (two k-steps).
// ---------------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void dec_xq_expand_kernel(XsParams p) {
  constexpr int FW = D / 4, NMT = FW / 16;
  const int h = blockIdx.x, r0 = blockIdx.y * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  int row = r0 + n;
  const bool live = row < p.rows;
  if (!live) row = p.rows - 1;
  bf16x8 qb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const float4* src = (const float4*)(p.q + (long)row * D + h * 64 + ks * 32 + g * 8);
    qb[ks] = pack8(src[0], src[1]);
  }
  const bf16_t* wbase = p.WkT + ((long)h * D + wave * FW + n) * 64 + g * 8;
  bf16_t* dst = p.xq + ((long)row * p.H + h) * D + wave * FW + g * 4;
#pragma unroll
  for (int mt = 0; mt < NMT; mt++) {
    const bf16x8 a0 = *(const bf16x8*)(wbase + (long)mt * 16 * 64);
    const bf16x8 a1 = *(const bf16x8*)(wbase + (long)mt * 16 * 64 + 32);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qb[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qb[1], c, 0, 0, 0);
    if (live) *(u32x2*)(dst + mt * 16) = (u32x2){pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3])};
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same expansion WITH the query projection and its LayerNorm inside: one launch instead of three (resolve + LayerNorm, the
// skinny linear Wq, the expansion) on a chain that is latency-bound launch by launch.  Block = (head, 16 rows):
//   A. every wave resolves and normalises four of the 16 residual rows (x + pending split-K slabs; the shared ln_* pieces of
//      decoder.hip) into LDS as bf16 -- the 12 heads' blocks redo this for the same rows (L2 traffic, no launch); the blocks of
//      head 0 also write the resolved rows to x_out (the residual stream's ping-pong buffer).  The head's 64 x D slice of Wq is
//      requested before, so it flies under the LayerNorm;
//   B. q^T [64 d x 16 rows] = Wq_h LN(x)^T + bq: wave w owns d = 16 w .. 16 w + 15 (D/32 MFMAs), result to LDS as bf16;
//   C. the expansion as in dec_xq_expand_kernel, its B operand out of LDS.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float xs_wave_sum(float v) { return wave_reduce_sum(v); }

template <int D>
__global__ __launch_bounds__(256) void dec_xq_fused_kernel(XsParams p) {
  constexpr int NKS = D / 32, FW = D / 4, NMT = FW / 16, RSX = 2 * D + 16, RSQ = 2 * 64 + 16, NV = D / 4, NI = (NV + 63) / 64;
  __shared__ __attribute__((aligned(16))) char xn_s[16 * RSX];
  __shared__ __attribute__((aligned(16))) char q_s[16 * RSQ];
  const int h = blockIdx.x, r0 = blockIdx.y * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  // the head's slice of Wq: wave w -> rows h*64 + 16 w + n
  bf16x8 wq[NKS];
  {
    const bf16_t* wsrc = p.Wq + (long)(h * 64 + wave * 16 + n) * D + g * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) wq[ks] = *(const bf16x8*)(wsrc + ks * 32);
  }
  // A. resolve + LayerNorm of rows 4 wave .. 4 wave + 3
  {
    float4 gg[NI], bb[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int idx = lane + 64 * i, ic = idx < NV ? idx : 0;
      gg[i] = ((const float4*)p.ln_g)[ic];
      bb[i] = ((const float4*)p.ln_b)[ic];
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int lr = wave * 4 + rr;
      int m = r0 + lr;
      const bool live = m < p.rows;
      if (!live) m = p.rows - 1;
      float4 v[NI];
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const int idx = lane + 64 * i, ic = idx < NV ? idx : 0;
        float4 a = ((const float4*)(p.x + (long)m * D))[ic];
        float4 q[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) q[s2] = ((const float4*)(p.pend + (long)(s2 < p.pend_n ? s2 : 0) * p.pend_stride + (long)m * D))[ic];
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) a = ln_add_pend(a, s2 < p.pend_n ? 1.f : 0.f, q[s2]);
        if (idx >= NV) a = make_float4(0.f, 0.f, 0.f, 0.f);
        v[i] = a;
        if (p.x_out && h == 0 && live && idx < NV) ((float4*)(p.x_out + (long)m * D))[idx] = a;
        sm += ln_sum4(a);
      }
      const float mean = xs_wave_sum(sm) / (float)D;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const float t = ln_sq4(v[i], mean);
        sq += (lane + 64 * i < NV) ? t : 0.f;
      }
      const float rstd = rsqrtf(xs_wave_sum(sq) / (float)D + p.eps);
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const int idx = lane + 64 * i;
        if (idx < NV) *(uint2*)(xn_s + lr * RSX + idx * 8) = ln_pack4(v[i], mean, rstd, gg[i], bb[i]);
      }
    }
  }
  __syncthreads();
  // B. q^T tile of this wave
  f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
  {
    const char* xsrc = xn_s + n * RSX + g * 16;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      const bf16x8 b = *(const bf16x8*)(xsrc + ks * 64);
      if (ks & 1) cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ks], b, cb, 0, 0, 0);
      else ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ks], b, ca, 0, 0, 0);
    }
  }
  // the expansion's weights fly under the LDS round trip of q
  const bf16_t* wbase = p.WkT + ((long)h * D + wave * FW + n) * 64 + g * 8;
  bf16x8 wk[NMT][2];
#pragma unroll
  for (int mt = 0; mt < NMT; mt++) {
    wk[mt][0] = *(const bf16x8*)(wbase + (long)mt * 16 * 64);
    wk[mt][1] = *(const bf16x8*)(wbase + (long)mt * 16 * 64 + 32);
  }
  {
    const float4 bias = *(const float4*)(p.bq + h * 64 + wave * 16 + g * 4);
    const float v0 = ca[0] + cb[0] + bias.x, v1 = ca[1] + cb[1] + bias.y, v2 = ca[2] + cb[2] + bias.z, v3 = ca[3] + cb[3] + bias.w;
    *(u32x2*)(q_s + n * RSQ + (wave * 16 + g * 4) * 2) = (u32x2){pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
  }
  __syncthreads();
  // C. expansion
  const bf16x8 qb0 = *(const bf16x8*)(q_s + n * RSQ + g * 16), qb1 = *(const bf16x8*)(q_s + n * RSQ + 64 + g * 16);
  const int row = r0 + n;
  bf16_t* dst = p.xq + ((long)(row < p.rows ? row : p.rows - 1) * p.H + h) * D + wave * FW + g * 4;
#pragma unroll
  for (int mt = 0; mt < NMT; mt++) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk[mt][0], qb0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk[mt][1], qb1, c, 0, 0, 0);
    if (row < p.rows) *(u32x2*)(dst + mt * 16) = (u32x2){pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3])};
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The expansion with the query projection of the LayerNorm-FREE chain: the producer of the residual stream (the self-attention output
// projection, DEPI_RESOLVE in decoder.hip) left the resolved rows as bf16 and (sum, sum of squares) per 16-column tile; the LayerNorm
// is applied algebraically, q = rstd (xb (gamma o Wq)^T - mean s) + c.  Block = (head, 16 rows): no resolve, no normalisation pass, the
// rows go from global memory straight into the MFMA B operand.
// ---------------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void dec_xq_lnfree_kernel(XsParams p) {
  constexpr int NKS = D / 32, FW = D / 4, NMT = FW / 16, RSQ = 2 * 64 + 16;
  __shared__ __attribute__((aligned(16))) char q_s[16 * RSQ];
  __shared__ float2 st_s[16];
  const int h = blockIdx.x, r0 = blockIdx.y * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  int row = r0 + n;
  const bool live = row < p.rows;
  if (!live) row = p.rows - 1;
  // the head's slice of gamma o Wq: wave w -> rows h*64 + 16 w + n; and the 16 raw rows (B operand)
  bf16x8 wq[NKS], xb[NKS];
  {
    const bf16_t* wsrc = p.Wq + (long)(h * 64 + wave * 16 + n) * D + g * 8;
    const bf16_t* xsrc = p.xb + (long)row * D + g * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) wq[ks] = *(const bf16x8*)(wsrc + ks * 32);
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) xb[ks] = *(const bf16x8*)(xsrc + ks * 32);
  }
  // (mean, rstd) of the 16 rows: four threads per row, a quarter of the D / 16 tiles each, (q0 + q1) + (q2 + q3) -- the order of
  // dec_linear_kernel<ACT_BF16_LN>
  if (threadIdx.x < 64) {
    const int r = threadIdx.x >> 2, qt = threadIdx.x & 3;
    int m = r0 + r;
    m = m < p.rows ? m : p.rows - 1;
    constexpr int T4 = (D / 16) / 4;
    const float2* sp = p.ln_stats + (long)m * (D / 16) + qt * T4;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < T4; i++) {
      const float2 v = sp[i];
      s1 += v.x; s2 += v.y;
    }
    s1 += dpp_mov<CCX_DPP_QUAD_XOR1>(s1); s2 += dpp_mov<CCX_DPP_QUAD_XOR1>(s2);
    s1 += dpp_mov<CCX_DPP_QUAD_XOR2>(s1); s2 += dpp_mov<CCX_DPP_QUAD_XOR2>(s2);
    const float mean = s1 / (float)D;
    const float var = fmaf(-mean, mean, s2 / (float)D);
    if (qt == 0) st_s[r] = make_float2(mean, rsqrtf(fmaxf(var, 0.f) + p.eps));
  }
  // the expansion's weights and the folded constants fly under the MFMAs
  const bf16_t* wbase = p.WkT + ((long)h * D + wave * FW + n) * 64 + g * 8;
  bf16x8 wk[NMT][2];
#pragma unroll
  for (int mt = 0; mt < NMT; mt++) {
    wk[mt][0] = *(const bf16x8*)(wbase + (long)mt * 16 * 64);
    wk[mt][1] = *(const bf16x8*)(wbase + (long)mt * 16 * 64 + 32);
  }
  const float4 cc = *(const float4*)(p.bq + h * 64 + wave * 16 + g * 4);
  const float4 ss = *(const float4*)(p.ln_s + h * 64 + wave * 16 + g * 4);
  f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < NKS; ks++) {
    if (ks & 1) cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ks], xb[ks], cb, 0, 0, 0);
    else ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ks], xb[ks], ca, 0, 0, 0);
  }
  __syncthreads();
  {
    const float2 ms = st_s[n];
    const float v0 = fmaf(ms.y, fmaf(-ms.x, ss.x, ca[0] + cb[0]), cc.x), v1 = fmaf(ms.y, fmaf(-ms.x, ss.y, ca[1] + cb[1]), cc.y);
    const float v2 = fmaf(ms.y, fmaf(-ms.x, ss.z, ca[2] + cb[2]), cc.z), v3 = fmaf(ms.y, fmaf(-ms.x, ss.w, ca[3] + cb[3]), cc.w);
    *(u32x2*)(q_s + n * RSQ + (wave * 16 + g * 4) * 2) = (u32x2){pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
  }
  __syncthreads();
  const bf16x8 qb0 = *(const bf16x8*)(q_s + n * RSQ + g * 16), qb1 = *(const bf16x8*)(q_s + n * RSQ + 64 + g * 16);
  bf16_t* dst = p.xq + ((long)row * p.H + h) * D + wave * FW + g * 4;
#pragma unroll
  for (int mt = 0; mt < NMT; mt++) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk[mt][0], qb0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk[mt][1], qb1, c, 0, 0, 0);
    if (live) *(u32x2*)(dst + mt * 16) = (u32x2){pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3])};
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// out = ctx Wv^T + bv per head, ctx = the merged key-half partials of dec_xs_stream_kernel.  Block = (head, 16 rows): all 256
// threads merge the head's contexts of the 16 rows into LDS as bf16 (ctx = (o0 w0 + o1 w1) / (l0 w0 + l1 w1), w_s = exp2(m_s - max m):
// the flash-decoding merge, in a fixed order), then wave w owns the head's output features [16 w, 16 w + 16).
// MFMA roles: A = Wv rows [out feature][f] (M), B = ctx^T [f][row] (N = rows, from LDS), K = D.
// ---------------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void dec_xv_project_kernel(XsParams p) {
  constexpr int NKS = D / 32, RSB = 2 * D + 16;
  __shared__ __attribute__((aligned(16))) char ctx_s[16 * RSB];
  const int h = blockIdx.x, r0 = blockIdx.y * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const int o0 = h * 64 + wave * 16;
  // weights first: they do not depend on the merge
  const bf16_t* wsrc = p.Wv + (long)(o0 + n) * D + g * 8;
  constexpr int CH = NKS % 6 == 0 ? 6 : 4;      // k-steps requested together
  static_assert(NKS % CH == 0, "D must be a multiple of 128");
  bf16x8 a[CH];
#pragma unroll
  for (int k = 0; k < CH; k++) a[k] = *(const bf16x8*)(wsrc + k * 32);
  {
    const int rr = threadIdx.x >> 4, c = threadIdx.x & 15;       // row of the tile, float4 column phase
    int row = r0 + rr;
    if (row >= p.rows) row = p.rows - 1;
    float m[XS_SPLIT], l[XS_SPLIT], wgt[XS_SPLIT];
    float mx = -INFINITY;
#pragma unroll
    for (int s2 = 0; s2 < XS_SPLIT; s2++) {
      const float2 ml = *(const float2*)(p.part_ml + (((long)row * XS_SPLIT + s2) * 16 + h) * 2);
      m[s2] = ml.x; l[s2] = ml.y;
      mx = fmaxf(mx, ml.x);
    }
    float den = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < XS_SPLIT; s2++) {
      wgt[s2] = __builtin_amdgcn_exp2f(m[s2] - mx);               // an empty half has m = -inf, l = 0: weight 0
      den = fmaf(l[s2], wgt[s2], den);
    }
    const float inv = 1.f / den;
#pragma unroll
    for (int s2 = 0; s2 < XS_SPLIT; s2++) wgt[s2] *= inv;
    const float* src = p.part_o + (((long)row * XS_SPLIT) * p.H + h) * D;
    const long sstride = (long)p.H * D;
#pragma unroll
    for (int k = 0; k < D / 64; k++) {
      const int f = (c + 16 * k) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int s2 = 0; s2 < XS_SPLIT; s2++) {
        const float4 o = *(const float4*)(src + s2 * sstride + f);
        v.x = fmaf(o.x, wgt[s2], v.x); v.y = fmaf(o.y, wgt[s2], v.y); v.z = fmaf(o.z, wgt[s2], v.z); v.w = fmaf(o.w, wgt[s2], v.w);
      }
      *(u32x2*)(ctx_s + rr * RSB + f * 2) = (u32x2){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    }
  }
  __syncthreads();
  const char* csrc = ctx_s + n * RSB + g * 16;
  f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c0 = 0; c0 < NKS; c0 += CH) {
    bf16x8 an[CH];
    if (c0 + CH < NKS) {
#pragma unroll
      for (int k = 0; k < CH; k++) an[k] = *(const bf16x8*)(wsrc + (c0 + CH + k) * 32);
    }
#pragma unroll
    for (int k = 0; k < CH; k++) {
      const bf16x8 b = *(const bf16x8*)(csrc + (c0 + k) * 64);
      if (k & 1) cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b, cb, 0, 0, 0);
      else ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b, ca, 0, 0, 0);
    }
    if (c0 + CH < NKS) {
#pragma unroll
      for (int k = 0; k < CH; k++) a[k] = an[k];
    }
  }
  const int row = r0 + n;
  if (row < p.rows) {
    const float4 bias = *(const float4*)(p.bv + o0 + g * 4);
    const float v0 = ca[0] + cb[0] + bias.x, v1 = ca[1] + cb[1] + bias.y, v2 = ca[2] + cb[2] + bias.z, v3 = ca[3] + cb[3] + bias.w;
    *(u32x2*)(p.out + (long)row * D + o0 + g * 4) = (u32x2){pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// the streaming kernel
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
template <int D, int RB = 1>
struct XsGeom {
  static constexpr int FW = D / 4;             // features per wave
  static constexpr int NKS = FW / 32;          // k-steps of a wave's partial scores
  static constexpr int NMT = FW / 16;          // M tiles of a wave's context slice
  static constexpr int PF = 4;                 // key tiles a wave keeps in flight (in situ, 768 sequences in 3 lanes: 2 -> 5.92, 3 -> 5.70, 4 -> 5.65, 6 -> 6.0 ms per step)
  static constexpr int RS = 2 * FW + 32;       // bytes per staged key row: +32 makes the transposed reads of 8 rows hit 64 different banks
  static constexpr int STRIP = 16 * RS;        // one wave's staging strip (private: no barrier)
  static constexpr int S_OFF = 4 * STRIP;      // partial-score exchange: [2 buffers][RB rows][4 waves][64 lanes] f32x4
  static constexpr int LDS = S_OFF + 2 * RB * 4 * 1024;
};
}  // namespace

// One block per (row, key half): XS_SPLIT = 2 blocks per sequence, ALWAYS -- a CU pulls ~25 GB/s whatever its occupancy, so the kernel
// takes as long as the CU with the most bytes: 384 rows as 384 blocks leave half the CUs with two rows and half with one (217 us), as
// 768 half-rows every CU gets three (the split is fixed so that a row's arithmetic never depends on the launch).  Two blocks per CU.
// The four waves work on the SAME 16-key tile of xa, each on its quarter of the
// D features (wave w: features [w D/4, (w+1) D/4)):
//   1. the wave's quarter of the tile is requested PF = 4 tiles ahead into named register images (D/128 16-byte loads per lane and
//      tile: lane = key l%16, feature chunk l/16 -- the MFMA A-operand image of S^T = xa_tile q'^T; the block's loads of one tile
//      cover 24 KB of contiguous memory; cacheable loads: with the non-temporal hint the two 64-byte halves of a line, which two
//      consecutive instructions ask for, cost 15 % of the rate),
//   2. partial scores over the wave's features: D/128 v_mfma_f32_16x16x32_bf16 against its slice of q' (registers, loaded once),
//   3. the four partial S^T [16 keys x 16 heads] meet in LDS (one barrier per tile, two buffers) and every wave adds them in the
//      same order, so all four hold the same scores and run the same softmax: base 2, against a FIXED reference (the maximum of
//      the row's first tile, see below) -- nothing is ever rescaled and the waves never exchange anything else,
//   4. the wave's quarter tile goes to its private LDS strip as it came ([key][feature], 32 B of row padding) and comes back
//      TRANSPOSED through ds_read_b64_tr_b16 as the A operand of ctx^T [16 features x 16 heads] += xa_tile^T p^T -- D/64
//      v_mfma_f32_16x16x16_bf16 whose B operand (4 keys x 1 head per lane) is exactly what the S^T accumulator holds after exp2.
// ~200 registers, 35 KB of LDS: two blocks per CU, 8 waves x 4 tiles x 6 KB = 192 KB of loads in flight per CU.
// The halves' partials (contexts relative to their own reference, reference, denominator) are merged by dec_xv_project_kernel.
// Deterministic: a row's numbers depend on nothing but its own q' and xa.
// RB > 1 (prompt prefill: p.rows_per_seq consecutive rows attend to ONE sequence's xa): a block takes RB rows of a sequence through
// the same pass -- RB sets of q', scores, references and accumulators against one stream of tiles and one set of transposed reads;
// per row the arithmetic is that of RB = 1, operation for operation.  One block per CU (RB = 4: ~430 registers).
template <int D, int RB>
__global__ __launch_bounds__(256, RB == 1 ? 2 : 1) void dec_xs_stream_kernel(XsParams p) {
  using G = XsGeom<D, RB>;
  constexpr int FW = G::FW, NKS = G::NKS, NMT = G::NMT, PF = G::PF, RS = G::RS;
  extern __shared__ __attribute__((aligned(16))) char xs_smem[];
  const int unit = blockIdx.x / XS_SPLIT, sp = blockIdx.x % XS_SPLIT;
  // rows of this block: RB == 1: row = unit; RB > 1: sequence s = unit / groups, rows s * rows_per_seq + gi * RB + j
  int row0, nrow;
  if (RB == 1) { row0 = unit; nrow = 1; }
  else {
    const int groups = (p.rows_per_seq + RB - 1) / RB;
    const int s_idx = unit / groups, gi = unit - s_idx * groups;
    row0 = s_idx * p.rows_per_seq + gi * RB;
    nrow = p.rows_per_seq - gi * RB < RB ? p.rows_per_seq - gi * RB : RB;
  }
  const int seq = p.row_seq ? p.row_seq[row0] : row0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int H = p.H, S = p.S;
  char* strip = xs_smem + wave * G::STRIP;
  char* sbuf = xs_smem + G::S_OFF;

  const bf16_t* Xp = p.X + (long)seq * p.x_seq_stride + wave * FW + g * 8;
  const int NTall = (S + 15) >> 4;
  const int T0 = NTall * sp / XS_SPLIT, NT = NTall * (sp + 1) / XS_SPLIT;      // this block's key tiles [T0, NT)

  bf16x8 img[PF][NKS];
  auto load_tile = [&](bf16x8 (&X)[NKS], int t) {
    int key = t * 16 + r;
    key = key < S ? key : S - 1;                 // the last tile's missing keys re-read the last row; their p is 0
    const bf16_t* src = Xp + (long)key * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) X[ks] = *(const bf16x8*)(src + ks * 32);
  };
#pragma unroll
  for (int j = 0; j < PF; j++)
    if (T0 + j < NT) load_tile(img[j], T0 + j);

  // the wave's slice of q' (B operand of the scores: head r, feature chunk g); heads >= H: zero; rows past the group: the last row's
  bf16x8 qf[RB][NKS];
#pragma unroll
  for (int jr = 0; jr < RB; jr++) {
    const int rw = row0 + (jr < nrow ? jr : nrow - 1);
    const bf16_t* qsrc = p.xq + ((long)rw * H + (r < H ? r : 0)) * D + wave * FW + g * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      qf[jr][ks] = *(const bf16x8*)(qsrc + ks * 32);
      if (r >= H) qf[jr][ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  f32x4 acc[RB][NMT];
  float m_ref[RB], m_seen[RB], l_part[RB];
#pragma unroll
  for (int jr = 0; jr < RB; jr++) { m_ref[jr] = -INFINITY; m_seen[jr] = -INFINITY; l_part[jr] = 0.f; }
  char* st_wr = strip + r * RS + g * 16;                                       // staging write: key r, feature chunk g
  const char* tr_rd = strip + (g * 4 + (r >> 2)) * RS + (r & 3) * 8;           // transposed read: lane 4q+p -> key 4g+q, features 4p..4p+3
  char* s_wr = sbuf + wave * 1024 + lane * 16;
  const char* s_rd = sbuf + lane * 16;
  const float scale = p.scale_log2e;

  auto tile = [&](const bf16x8 (&X)[NKS], int t, bool first) {
    const int boff = (t & 1) * (RB * 4096);
#pragma unroll
    for (int jr = 0; jr < RB; jr++) {
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) {
        if (ks & 1) sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X[ks], qf[jr][ks], sb, 0, 0, 0);
        else sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X[ks], qf[jr][ks], sa, 0, 0, 0);
      }
      *(f32x4*)(s_wr + boff + jr * 4096) = NKS > 1 ? sa + sb : sa;
    }
    // stage the quarter tile as it is (read back transposed below): private strip, the wave's own LDS operations stay in order
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) *(bf16x8*)(st_wr + ks * 64) = X[ks];
    __syncthreads();
    bf16x4 pb[RB];
    const int key0 = t * 16 + g * 4;             // lane: head r, keys 16 t + 4 g + j
#pragma unroll
    for (int jr = 0; jr < RB; jr++) {
      const char* sr = s_rd + boff + jr * 4096;
      const f32x4 s0 = *(const f32x4*)(sr), s1 = *(const f32x4*)(sr + 1024), s2 = *(const f32x4*)(sr + 2048), s3 = *(const f32x4*)(sr + 3072);
      const f32x4 sc = (s0 + s1) + (s2 + s3);
      float tv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) tv[j] = key0 + j < S ? sc[j] * scale : -INFINITY;
      float tmax = fmaxf(fmaxf(tv[0], tv[1]), fmaxf(tv[2], tv[3]));
      tmax = fmaxf(tmax, lane_xor16(tmax));
      tmax = fmaxf(tmax, lane_xor32(tmax));
      m_seen[jr] = fmaxf(m_seen[jr], tmax);
      if (first) m_ref[jr] = tmax;               // the block's first tile always holds live keys
      float pv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) pv[j] = __builtin_amdgcn_exp2f(tv[j] - m_ref[jr]);
      l_part[jr] += (pv[0] + pv[1]) + (pv[2] + pv[3]);
      const u32x2 pp = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
      pb[jr] = __builtin_bit_cast(bf16x4, pp);
    }
#pragma unroll
    for (int mt = 0; mt < NMT; mt++) {
      const bf16x4 xt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tr_rd + mt * 32));
#pragma unroll
      for (int jr = 0; jr < RB; jr++) acc[jr][mt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xt, pb[jr], acc[jr][mt], 0, 0, 0);
    }
  };

  // Softmax reference: the maximum of the block's FIRST tile (per row and head), never moved -- the accumulators are never rescaled.
  // p = exp2(t - m_ref) may then exceed 1; fp32 / bf16 carry it up to 2^127, and numerator and denominator share the reference, so the
  // result is exact.  Should a later tile exceed the reference by more than 2^100 (an attention peak of e^69 over the first 16 keys:
  // unseen), the block repeats its key range once with the maximum it then knows (every wave holds the same scores: a uniform decision).
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      bool over = false;
#pragma unroll
      for (int jr = 0; jr < RB; jr++) over |= m_seen[jr] > m_ref[jr] + 100.f;
      if (__builtin_amdgcn_ballot_w64(over) == 0ull) break;
#pragma unroll
      for (int jr = 0; jr < RB; jr++) m_ref[jr] = m_seen[jr];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < PF; j++)
        if (T0 + j < NT) load_tile(img[j], T0 + j);
    }
#pragma unroll
    for (int jr = 0; jr < RB; jr++) {
      l_part[jr] = 0.f;
#pragma unroll
      for (int mt = 0; mt < NMT; mt++) acc[jr][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int t0 = T0; t0 < NT; t0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; j++) {
        const int t = t0 + j;
        if (t < NT) {
          tile(img[j], t, pass == 0 && t == T0);
          if (t + PF < NT) load_tile(img[j], t + PF);
        }
      }
    }
  }

  // the block's partials: unnormalised contexts of its key range (relative to m_ref), its reference maximum and denominator
#pragma unroll
  for (int jr = 0; jr < RB; jr++) {
    if (jr < nrow) {
      float l_head = l_part[jr];
      l_head += lane_xor16(l_head);
      l_head += lane_xor32(l_head);
      const long u = (long)(row0 + jr) * XS_SPLIT + sp;
      if (r < H) {
        float* dst = p.part_o + (u * H + r) * D + wave * FW + g * 4;
#pragma unroll
        for (int mt = 0; mt < NMT; mt++) *(f32x4*)(dst + mt * 16) = acc[jr][mt];
      }
      if (wave == 0 && g == 0) *(float2*)(p.part_ml + (u * 16 + r) * 2) = make_float2(m_ref[jr], l_head);
    }
  }
}

namespace {

template <int D>
int launch_xs(ccx_ctx* ctx, const XsParams& p, hipStream_t stream) {
  using G = XsGeom<D>;
  // p.lds_pad: dynamic LDS the streaming blocks claim without using it -- 64 KB leave ONE block per CU (which pulls the same ~25 GB/s
  // as two), so that the chain kernels of the other decode lanes find wave slots and a shorter memory queue on every CU: 768-sequence
  // decode step 6.25 -> 5.87 ms with two lanes.  CCX_XS_LDS_PAD overrides (experiments).
  static const int forced_pad = [] { const char* e = getenv("CCX_XS_LDS_PAD"); return e ? atoi(e) : -1; }();
  const int lds_pad = forced_pad >= 0 ? forced_pad : p.lds_pad;
  // prompt prefill (rows_per_seq consecutive rows per sequence): four rows of a sequence share one pass over its xa
  // (CCX_XS_PREFILL_ROWS=1: one row per block, as the decode steps)
  static const int pf_rows = [] { const char* e = getenv("CCX_XS_PREFILL_ROWS"); return e ? atoi(e) : 4; }();
  CCX_REQUIRE(ctx, pf_rows == 1 || pf_rows == 4, "xs cross attention: CCX_XS_PREFILL_ROWS=%d unsupported (1 or 4 rows per block)", pf_rows);
  // rows of one sequence per block need the row -> sequence map (without it the kernel would take its row index as the sequence
  // and read xa beyond the encoded windows)
  CCX_REQUIRE(ctx, p.rows_per_seq <= 1 || p.row_seq != nullptr, "xs cross attention: rows_per_seq = %d needs row_seq", p.rows_per_seq);
  const bool multi = p.rows_per_seq > 1 && pf_rows == 4;
  static ccx_lds_optin optin1, optin4;       // per device, race-free (ccx_common.h)
  CCX_HIP(ctx, optin1.ensure(ctx->device, (const void*)dec_xs_stream_kernel<D, 1>));
  CCX_HIP(ctx, optin4.ensure(ctx->device, (const void*)dec_xs_stream_kernel<D, 4>));
  CCX_REQUIRE(ctx, lds_pad >= 0 && G::LDS + lds_pad <= 160 * 1024, "xs cross attention: LDS claim %d too large", lds_pad);
  CCX_REQUIRE(ctx, p.rows_per_seq <= 1 || p.rows % p.rows_per_seq == 0, "xs cross attention: %d rows are not a multiple of %d rows per sequence", p.rows, p.rows_per_seq);
  const dim3 small_grid(p.H, ccx_cdiv(p.rows, 16));
  const double wbytes = (double)p.H * 64 * D * 2;
  if (p.xb) {
    ccx_prof_scope ps(ctx, stream, D == 768 ? "dec_xq_lnfree_kernel<768>" : "dec_xq_lnfree_kernel", 2.0 * p.rows * D * (double)D * 2,
                      2.0 * wbytes + (double)p.rows * D * 2 + (double)p.rows * p.H * D * 2);
    hipLaunchKernelGGL(dec_xq_lnfree_kernel<D>, small_grid, dim3(256), 0, stream, p);
  } else if (p.x) {
    ccx_prof_scope ps(ctx, stream, D == 768 ? "dec_xq_fused_kernel<768>" : "dec_xq_fused_kernel", 2.0 * p.rows * D * (double)D * 2,
                      2.0 * wbytes + (double)p.rows * D * 4 * (1 + p.pend_n) + (double)p.rows * p.H * D * 2);
    hipLaunchKernelGGL(dec_xq_fused_kernel<D>, small_grid, dim3(256), 0, stream, p);
  } else {
    ccx_prof_scope ps(ctx, stream, D == 768 ? "dec_xq_expand_kernel<768>" : "dec_xq_expand_kernel", 2.0 * p.rows * p.H * 64 * D, wbytes + (double)p.rows * D * 4 + (double)p.rows * p.H * D * 2);
    hipLaunchKernelGGL(dec_xq_expand_kernel<D>, small_grid, dim3(256), 0, stream, p);
  }
  CCX_CHECK_LAUNCH(ctx);
  {
    ccx_prof_scope ps(ctx, stream, D == 768 ? (multi ? "dec_xs_stream_kernel<768,4>" : "dec_xs_stream_kernel<768,1>") : "dec_xs_stream_kernel", 4.0 * p.rows * 16 * (double)p.S * D,
                      (double)p.rows * ((double)p.S * D * 2 + p.H * D * 2.0 + XS_SPLIT * p.H * D * 4.0));
    if (multi) {
      const int units = p.rows / p.rows_per_seq * ccx_cdiv(p.rows_per_seq, 4);
      constexpr int lds4 = XsGeom<D, 4>::LDS;
      hipLaunchKernelGGL((dec_xs_stream_kernel<D, 4>), dim3(units * XS_SPLIT), dim3(256), lds4, stream, p);
    } else {
      hipLaunchKernelGGL((dec_xs_stream_kernel<D, 1>), dim3(p.rows * XS_SPLIT), dim3(256), G::LDS + lds_pad, stream, p);
    }
  }
  CCX_CHECK_LAUNCH(ctx);
  {
    ccx_prof_scope ps(ctx, stream, D == 768 ? "dec_xv_project_kernel<768>" : "dec_xv_project_kernel", 2.0 * p.rows * p.H * 64 * D, wbytes + (double)p.rows * p.H * D * 2 + (double)p.rows * D * 2);
    hipLaunchKernelGGL(dec_xv_project_kernel<D>, small_grid, dim3(256), 0, stream, p);
  }
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

}  // namespace

bool ccx_xs_supported(int D, int H) { return H >= 1 && H <= 16 && H * 64 == D && (D == 128 || D == 256 || D == 384 || D == 512 || D == 768); }

int ccx_launch_xs_cross_attention(ccx_ctx* ctx, const XsParams& p, hipStream_t stream) {
  CCX_REQUIRE(ctx, ccx_xs_supported(p.D, p.H), "xs cross attention: width %d with %d heads is not instantiated", p.D, p.H);
  CCX_REQUIRE(ctx, p.rows >= 1 && p.S >= 16 && (p.q || p.x || p.xb) && p.WkT && p.xq && p.X && p.Wv && p.bv && p.out, "xs cross attention: bad arguments");
  CCX_REQUIRE(ctx, !p.xb || (p.ln_stats && p.ln_s && p.Wq && p.bq), "xs cross attention: bad LayerNorm-free query arguments");
  CCX_REQUIRE(ctx, !p.x || (p.pend && p.ln_g && p.ln_b && p.Wq && p.bq && p.pend_n >= 0 && p.pend_n <= 4 && p.x_out != p.x), "xs cross attention: bad fused-query arguments");
  CCX_REQUIRE(ctx, p.part_o && p.part_ml, "xs cross attention: partial buffers missing");
  switch (p.D) {
    case 128: return launch_xs<128>(ctx, p, stream);
    case 256: return launch_xs<256>(ctx, p, stream);
    case 384: return launch_xs<384>(ctx, p, stream);
    case 512: return launch_xs<512>(ctx, p, stream);
    default: return launch_xs<768>(ctx, p, stream);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// C ABI: one layer's cross attention as a stand-alone operator (kernel-level parity tests, include/ccx.h)
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
inline bf16_t xs_host_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__global__ void xs_bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = bf16_to_f32(in[i]);
}
}  // namespace

extern "C" int ccx_cross_attention_xa(ccx_ctx* ctx, const float* q_dev, const float* wk_host, const float* wv_host, const float* bv_host,
                                      const uint16_t* xa_dev, const int* row_seq_host, int rows_per_seq, int rows, int n_seq, int n_head,
                                      int n_ctx, float* out_dev, void* stream_) {
  if (!ctx) return CCX_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int H = n_head, D = 64 * n_head, S = n_ctx;
  CCX_REQUIRE(ctx, ccx_xs_supported(D, H), "cross_attention_xa: %d heads (width %d) not instantiated", H, D);
  CCX_REQUIRE(ctx, q_dev && wk_host && wv_host && bv_host && xa_dev && out_dev && rows >= 1 && n_seq >= 1 && S >= 16, "cross_attention_xa: bad arguments");
  if (row_seq_host)
    for (int i = 0; i < rows; i++) CCX_REQUIRE(ctx, row_seq_host[i] >= 0 && row_seq_host[i] < n_seq, "cross_attention_xa: row_seq[%d] out of range", i);
  else CCX_REQUIRE(ctx, rows <= n_seq, "cross_attention_xa: more rows than sequences without a row map");
  if (rows_per_seq > 1) {
    CCX_REQUIRE(ctx, row_seq_host && rows % rows_per_seq == 0, "cross_attention_xa: rows_per_seq needs a row map and a multiple of it in rows");
    for (int i = 0; i < rows; i++)
      CCX_REQUIRE(ctx, row_seq_host[i] == row_seq_host[i - i % rows_per_seq], "cross_attention_xa: row %d is not in its group's sequence", i);
  }
  std::vector<bf16_t> wkt((size_t)D * D), wv((size_t)D * D);
  for (int hh = 0; hh < H; hh++)
    for (int f = 0; f < D; f++)
      for (int dd = 0; dd < 64; dd++) wkt[((size_t)hh * D + f) * 64 + dd] = xs_host_bf16(wk_host[(size_t)(hh * 64 + dd) * D + f]);
  for (size_t i = 0; i < wv.size(); i++) wv[i] = xs_host_bf16(wv_host[i]);
  bf16_t *d_wkt = nullptr, *d_wv = nullptr, *d_xq = nullptr, *d_out = nullptr;
  float *d_bv = nullptr, *d_po = nullptr, *d_pml = nullptr;
  int* d_rs = nullptr;
  auto cleanup = [&]() { hipFree(d_wkt); hipFree(d_wv); hipFree(d_xq); hipFree(d_out); hipFree(d_bv); hipFree(d_rs); hipFree(d_po); hipFree(d_pml); };
#define XS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return ccx_fail(ctx, CCX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
  XS_TRY(hipMalloc(&d_wkt, wkt.size() * 2)); XS_TRY(hipMalloc(&d_wv, wv.size() * 2)); XS_TRY(hipMalloc(&d_bv, (size_t)D * 4));
  XS_TRY(hipMalloc(&d_xq, (size_t)rows * H * D * 2)); XS_TRY(hipMalloc(&d_out, (size_t)rows * D * 2));
  XS_TRY(hipMalloc(&d_po, ccx_xs_part_o_elems(rows, H, D) * 4)); XS_TRY(hipMalloc(&d_pml, ccx_xs_part_ml_elems(rows) * 4));
  XS_TRY(hipMemcpy(d_wkt, wkt.data(), wkt.size() * 2, hipMemcpyHostToDevice));
  XS_TRY(hipMemcpy(d_wv, wv.data(), wv.size() * 2, hipMemcpyHostToDevice));
  XS_TRY(hipMemcpy(d_bv, bv_host, (size_t)D * 4, hipMemcpyHostToDevice));
  if (row_seq_host) {
    XS_TRY(hipMalloc(&d_rs, (size_t)rows * 4));
    XS_TRY(hipMemcpy(d_rs, row_seq_host, (size_t)rows * 4, hipMemcpyHostToDevice));
  }
  XsParams p;
  memset(&p, 0, sizeof(p));
  p.q = q_dev; p.WkT = d_wkt; p.xq = d_xq; p.X = xa_dev; p.x_seq_stride = (long)S * D; p.row_seq = d_rs; p.Wv = d_wv; p.bv = d_bv; p.out = d_out; p.part_o = d_po; p.part_ml = d_pml;
  p.rows = rows; p.H = H; p.S = S; p.D = D; p.scale_log2e = 0.125f * 1.4426950408889634f;
  p.rows_per_seq = rows_per_seq > 1 ? rows_per_seq : 0;
  int rc = ccx_launch_xs_cross_attention(ctx, p, stream);
  if (rc == CCX_OK) {
    const long n = (long)rows * D;
    hipLaunchKernelGGL(xs_bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_out, out_dev, n);
    XS_TRY(hipStreamSynchronize(stream));
  }
  cleanup();
#undef XS_TRY
  return rc;
}
