// cross_x.hip -- decode cross attention against the ENCODER OUTPUT xa instead of per-layer K/V caches (see cross_x.h for the algebra;
// reference call sites: back/api.py:1286-1292, 1432-1438, 1474-1480 -> openai-whisper MultiHeadAttention(x, xa) per decoder layer).
//
// Three launches per layer:
//   dec_xq_expand_kernel   q'[row][h][:] = q[row][h*64 .. +64] Wk_h                        (64 -> D per head; 1.2 MB of weights)
//   dec_xs_stream_kernel   ctx[row][h][:] = sum_j softmax_j(q'[row][h] . xa_j * scale) xa_j (ONE pass over the sequence's xa: the HBM-bound part)
//   dec_xv_project_kernel  out[row][h*64 .. +64] = ctx[row][h] Wv_h^T + bv                 (D -> 64 per head)
// dec_xs_stream_kernel: one block per row (sequence), four waves, one wave per SIMD with the whole register file (accumulators for all
// heads x all D features: D/4 registers).  A wave owns every fourth 16-key tile of xa and is a self-contained pipeline:
//   1. tile t + 1 is requested (D/32 16-byte loads per lane: lane = key l%16, feature chunk l/16 -- the MFMA A-operand image of
//      S^T = xa_tile q'^T, so the scores need no staging at all),
//   2. S^T [16 keys x 16 heads] = D/32 v_mfma_f32_16x16x32_bf16 against q' (LDS, B operand, heads on the N index),
//   3. base-2 softmax against a FIXED reference (the maximum of the wave's first tile; see the loop): no accumulator rescale, ever,
//   4. the tile goes to the wave's private LDS strip as it came ([key][feature], 32 B of row padding), and comes back TRANSPOSED
//      through ds_read_b64_tr_b16 as the A operand of ctx^T [16 features x 16 heads] += xa_tile^T p^T -- D/16 v_mfma_f32_16x16x16_bf16,
//      whose B operand (4 keys x 1 head per lane) is exactly what the S^T accumulator holds after exp2: no transpose of p either.
// No block-level barrier inside the loop; the four waves' partial (max, sum, ctx) are merged through LDS at the end in a fixed
// order (deterministic: a row's numbers depend on nothing but its own q' and xa).
#include "cross_x.h"

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
// out = ctx Wv^T + bv per head.  Block = (head, 16 rows); wave w owns the head's output features [16 w, 16 w + 16).
// MFMA roles: A = Wv rows [out feature][f] (M), B = ctx^T [f][row] (N = rows), K = D.
// ---------------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void dec_xv_project_kernel(XsParams p) {
  constexpr int NKS = D / 32;
  const int h = blockIdx.x, r0 = blockIdx.y * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  int row = r0 + n;
  const bool live = row < p.rows;
  if (!live) row = p.rows - 1;
  const int o0 = h * 64 + wave * 16;
  const bf16_t* wsrc = p.Wv + (long)(o0 + n) * D + g * 8;
  const bf16_t* csrc = p.xq + ((long)row * p.H + h) * D + g * 8;
  f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
  constexpr int CH = NKS % 6 == 0 ? 6 : 4;      // k-steps requested together
  static_assert(NKS % CH == 0, "D must be a multiple of 128");
#pragma unroll
  for (int c0 = 0; c0 < NKS; c0 += CH) {
    bf16x8 a[CH], b[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) {
      a[k] = *(const bf16x8*)(wsrc + (c0 + k) * 32);
      b[k] = *(const bf16x8*)(csrc + (c0 + k) * 32);
    }
#pragma unroll
    for (int k = 0; k < CH; k++) {
      if (k & 1) cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b[k], cb, 0, 0, 0);
      else ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b[k], ca, 0, 0, 0);
    }
  }
  if (live) {
    const float4 bias = *(const float4*)(p.bv + o0 + g * 4);
    const float v0 = ca[0] + cb[0] + bias.x, v1 = ca[1] + cb[1] + bias.y, v2 = ca[2] + cb[2] + bias.z, v3 = ca[3] + cb[3] + bias.w;
    *(u32x2*)(p.out + (long)row * D + o0 + g * 4) = (u32x2){pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// the streaming kernel
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
template <int D>
struct XsGeom {
  static constexpr int NKS = D / 32;           // k-steps of the score MFMAs (features)
  static constexpr int NMT = D / 16;           // M tiles of the context MFMAs (features)
  static constexpr int RS = 2 * D + 32;        // bytes per staged key row: +32 makes the transposed reads of 8 rows hit 64 different banks
  static constexpr int RSQ = 2 * D + 16;       // bytes per q' row
  static constexpr int STRIP = 16 * RS;        // one wave's staging strip
  static constexpr int Q_OFF = 4 * STRIP;
  static constexpr int ML_OFF = Q_OFF + 16 * RSQ;
  static constexpr int LDS = ML_OFF + 2 * 4 * 16 * 4;
  static constexpr int CROW = D + 4;           // floats per head row of the merge buffer (aliases the strips)
  static_assert(16 * CROW * 4 <= 4 * STRIP, "merge buffer must fit the staging strips");
};
}  // namespace

template <int D>
__global__ __launch_bounds__(256, 1) void dec_xs_stream_kernel(XsParams p) {
  using G = XsGeom<D>;
  constexpr int NKS = G::NKS, NMT = G::NMT, RS = G::RS, RSQ = G::RSQ;
  extern __shared__ __attribute__((aligned(16))) char xs_smem[];
  const int row = blockIdx.x;
  const int seq = p.row_seq ? p.row_seq[row] : row;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int H = p.H, S = p.S;
  char* strip = xs_smem + wave * G::STRIP;
  char* qlds = xs_smem + G::Q_OFF;
  float* mw = (float*)(xs_smem + G::ML_OFF);
  float* lw = mw + 64;

  const bf16_t* Xp = p.X + (long)seq * p.x_seq_stride;
  const int NT = (S + 15) >> 4;
  const int n_w = (NT - wave + 3) >> 2;          // tiles wave, wave + 4, ...

  bf16x8 XA[NKS];
  auto load_tile = [&](int t) {
    int key = t * 16 + r;
    key = key < S ? key : S - 1;                 // the last tile's missing keys re-read the last row; their p is 0
    const bf16_t* src = Xp + (long)key * D + g * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) XA[ks] = __builtin_nontemporal_load((const bf16x8*)(src + ks * 32));
  };
  if (n_w > 0) load_tile(wave);

  // q' of this row -> LDS (heads >= H: zero rows)
  {
    bf16_t* xq_row = p.xq + (long)row * H * D;
    constexpr int CPR = D / 8;                   // 16-byte chunks per head row
    for (int i = threadIdx.x; i < 16 * CPR; i += 256) {
      const int hh = i / CPR, c = i - hh * CPR;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (hh < H) v = *(const u32x4*)(xq_row + (long)hh * D + c * 8);
      *(u32x4*)(qlds + hh * RSQ + c * 16) = v;
    }
  }
  __syncthreads();

  f32x4 acc[NMT];
  float m_ref = -INFINITY, m_seen = -INFINITY, l_part = 0.f;
  const char* q_rd = qlds + r * RSQ + g * 16;                                  // B operand of the scores: head r, feature chunk g
  char* st_wr = strip + r * RS + g * 16;                                       // staging write: key r, feature chunk g
  const char* tr_rd = strip + (g * 4 + (r >> 2)) * RS + (r & 3) * 8;           // transposed read: lane 4q+p -> key 4g+q, features 4p..4p+3
  const float scale = p.scale_log2e;

  // Softmax reference: the maximum of the wave's FIRST tile (per head), never moved -- the accumulators (D/4 AccVGPRs, which the
  // VALU cannot touch) are never rescaled.  p = exp2(t - m_ref) may then exceed 1; fp32 / bf16 carry it up to 2^127, and numerator
  // and denominator share the reference, so the result is exact.  Should a later tile exceed the reference by more than 2^100
  // (an attention peak of e^69 over the first 16 keys: unseen), the wave repeats its tiles once with the maximum it then knows.
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      if (__builtin_amdgcn_ballot_w64(m_seen > m_ref + 100.f) == 0ull) break;
      m_ref = m_seen;
      if (n_w > 0) load_tile(wave);
    }
    l_part = 0.f;
#pragma unroll
    for (int mt = 0; mt < NMT; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ONE register image of a tile: scores straight out of it, then it goes to the strip and its registers take the next tile's
    // loads, which fly under the context MFMAs of this one (24 KB per wave in flight almost all the time: 24 MB over the chip).
    for (int i = 0; i < n_w; i++) {
      const int t = wave + 4 * i;
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) {
        const bf16x8 qf = *(const bf16x8*)(q_rd + ks * 64);
        if (ks & 1) sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(XA[ks], qf, sb, 0, 0, 0);
        else sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(XA[ks], qf, sa, 0, 0, 0);
      }
      // stage the tile as it is (read back transposed below); its registers are free for the next tile
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) *(bf16x8*)(st_wr + ks * 64) = XA[ks];
      if (i + 1 < n_w) load_tile(t + 4);
      // lane: head r, keys 16 t + 4 g + j
      float tv[4];
      const int key0 = t * 16 + g * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) tv[j] = key0 + j < S ? (sa[j] + sb[j]) * scale : -INFINITY;
      float tmax = fmaxf(fmaxf(tv[0], tv[1]), fmaxf(tv[2], tv[3]));
      tmax = fmaxf(tmax, lane_xor16(tmax));
      tmax = fmaxf(tmax, lane_xor32(tmax));
      m_seen = fmaxf(m_seen, tmax);
      if (i == 0 && pass == 0) m_ref = tmax;     // a wave's first tile always holds live keys
      float pv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) pv[j] = __builtin_amdgcn_exp2f(tv[j] - m_ref);
      l_part += (pv[0] + pv[1]) + (pv[2] + pv[3]);
      const u32x2 pp = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
      const bf16x4 pb = __builtin_bit_cast(bf16x4, pp);
#pragma unroll
      for (int mt = 0; mt < NMT; mt++) {
        const bf16x4 xt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tr_rd + mt * 32));
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xt, pb, acc[mt], 0, 0, 0);
      }
    }
  }

  // ---- merge the four waves (fixed order) ----
  float l_head = l_part;
  l_head += lane_xor16(l_head);
  l_head += lane_xor32(l_head);
  if (g == 0) { mw[wave * 16 + r] = m_ref; lw[wave * 16 + r] = l_head; }
  __syncthreads();                                // also: every wave is done with its staging strip
  float m_all = fmaxf(fmaxf(mw[r], mw[16 + r]), fmaxf(mw[32 + r], mw[48 + r]));
  float l_all = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < 4; w2++) l_all += lw[w2 * 16 + r] * __builtin_amdgcn_exp2f(mw[w2 * 16 + r] - m_all);
  const float mine = __builtin_amdgcn_exp2f(m_ref - m_all) / l_all;
  float* comb = (float*)xs_smem + r * G::CROW + g * 4;                        // [head][feature]
  for (int w2 = 0; w2 < 4; w2++) {
    if (wave == w2) {
#pragma unroll
      for (int mt = 0; mt < NMT; mt++) {
        f32x4 v = acc[mt] * mine;
        if (w2 > 0) v += *(const f32x4*)(comb + mt * 16);
        *(f32x4*)(comb + mt * 16) = v;
      }
    }
    __syncthreads();
  }
  // normalised contexts -> xq (in place of this row's q'), 16-byte stores
  {
    bf16_t* out_row = p.xq + (long)row * H * D;
    constexpr int CPR = D / 8;
    const float* cb = (const float*)xs_smem;
    for (int i = threadIdx.x; i < H * CPR; i += 256) {
      const int hh = i / CPR, c = i - hh * CPR;
      const float4 a = *(const float4*)(cb + hh * G::CROW + c * 8), b = *(const float4*)(cb + hh * G::CROW + c * 8 + 4);
      *(bf16x8*)(out_row + (long)hh * D + c * 8) = pack8(a, b);
    }
  }
}

namespace {

template <int D>
int launch_xs(ccx_ctx* ctx, const XsParams& p, hipStream_t stream) {
  using G = XsGeom<D>;
  static bool attr_set = false;
  if (!attr_set) {
    CCX_HIP(ctx, hipFuncSetAttribute((const void*)dec_xs_stream_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    attr_set = true;
  }
  const dim3 small_grid(p.H, ccx_cdiv(p.rows, 16));
  const double wbytes = (double)p.H * 64 * D * 2;
  {
    ccx_prof_scope ps(ctx, stream, D == 768 ? "dec_xq_expand_kernel<768>" : "dec_xq_expand_kernel", 2.0 * p.rows * p.H * 64 * D, wbytes + (double)p.rows * D * 4 + (double)p.rows * p.H * D * 2);
    hipLaunchKernelGGL(dec_xq_expand_kernel<D>, small_grid, dim3(256), 0, stream, p);
  }
  CCX_CHECK_LAUNCH(ctx);
  {
    ccx_prof_scope ps(ctx, stream, D == 768 ? "dec_xs_stream_kernel<768>" : "dec_xs_stream_kernel", 4.0 * p.rows * 16 * (double)p.S * D,
                      (double)p.rows * ((double)p.S * D * 2 + 2.0 * p.H * D * 2));
    hipLaunchKernelGGL(dec_xs_stream_kernel<D>, dim3(p.rows), dim3(256), G::LDS, stream, p);
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
  CCX_REQUIRE(ctx, p.rows >= 1 && p.S >= 16 && p.q && p.WkT && p.xq && p.X && p.Wv && p.bv && p.out, "xs cross attention: bad arguments");
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
                                      const uint16_t* xa_dev, const int* row_seq_host, int rows, int n_seq, int n_head, int n_ctx,
                                      float* out_dev, void* stream_) {
  if (!ctx) return CCX_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int H = n_head, D = 64 * n_head, S = n_ctx;
  CCX_REQUIRE(ctx, ccx_xs_supported(D, H), "cross_attention_xa: %d heads (width %d) not instantiated", H, D);
  CCX_REQUIRE(ctx, q_dev && wk_host && wv_host && bv_host && xa_dev && out_dev && rows >= 1 && n_seq >= 1 && S >= 16, "cross_attention_xa: bad arguments");
  if (row_seq_host)
    for (int i = 0; i < rows; i++) CCX_REQUIRE(ctx, row_seq_host[i] >= 0 && row_seq_host[i] < n_seq, "cross_attention_xa: row_seq[%d] out of range", i);
  else CCX_REQUIRE(ctx, rows <= n_seq, "cross_attention_xa: more rows than sequences without a row map");
  std::vector<bf16_t> wkt((size_t)D * D), wv((size_t)D * D);
  for (int hh = 0; hh < H; hh++)
    for (int f = 0; f < D; f++)
      for (int dd = 0; dd < 64; dd++) wkt[((size_t)hh * D + f) * 64 + dd] = xs_host_bf16(wk_host[(size_t)(hh * 64 + dd) * D + f]);
  for (size_t i = 0; i < wv.size(); i++) wv[i] = xs_host_bf16(wv_host[i]);
  bf16_t *d_wkt = nullptr, *d_wv = nullptr, *d_xq = nullptr, *d_out = nullptr;
  float* d_bv = nullptr;
  int* d_rs = nullptr;
  auto cleanup = [&]() { hipFree(d_wkt); hipFree(d_wv); hipFree(d_xq); hipFree(d_out); hipFree(d_bv); hipFree(d_rs); };
#define XS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return ccx_fail(ctx, CCX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
  XS_TRY(hipMalloc(&d_wkt, wkt.size() * 2)); XS_TRY(hipMalloc(&d_wv, wv.size() * 2)); XS_TRY(hipMalloc(&d_bv, (size_t)D * 4));
  XS_TRY(hipMalloc(&d_xq, (size_t)rows * H * D * 2)); XS_TRY(hipMalloc(&d_out, (size_t)rows * D * 2));
  XS_TRY(hipMemcpy(d_wkt, wkt.data(), wkt.size() * 2, hipMemcpyHostToDevice));
  XS_TRY(hipMemcpy(d_wv, wv.data(), wv.size() * 2, hipMemcpyHostToDevice));
  XS_TRY(hipMemcpy(d_bv, bv_host, (size_t)D * 4, hipMemcpyHostToDevice));
  if (row_seq_host) {
    XS_TRY(hipMalloc(&d_rs, (size_t)rows * 4));
    XS_TRY(hipMemcpy(d_rs, row_seq_host, (size_t)rows * 4, hipMemcpyHostToDevice));
  }
  XsParams p;
  memset(&p, 0, sizeof(p));
  p.q = q_dev; p.WkT = d_wkt; p.xq = d_xq; p.X = xa_dev; p.x_seq_stride = (long)S * D; p.row_seq = d_rs; p.Wv = d_wv; p.bv = d_bv; p.out = d_out;
  p.rows = rows; p.H = H; p.S = S; p.D = D; p.scale_log2e = 0.125f * 1.4426950408889634f;
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
