// decoder.hip -- Whisper TextDecoder single-token step kernels (K9/K10 in SURVEY.md section 2a).
//
// One decode step for B sequences is a fixed chain of small kernels that the host captures in a
// hipGraph (whisper.hip).  All per-sequence variation (current token, position, prompt vs
// sampling phase, timestamp-rule state, finished flag) lives in device arrays, so the captured
// graph is replayed unchanged for every step and no host round trip is needed inside the loop.
//
//  * dec_linear_kernel: "skinny" GEMM out[M<=64 rows, N] = act * W^T for the decoder's
//    weight-streaming linears.  W fragments go HBM -> VGPR directly (each weight byte is used
//    once per step, an LDS round trip would be pure overhead: cdna_hip_programming.md, "GEMV /
//    M <= 16 decode weights"); activations are LayerNorm-ed (or combined from split-KV attention
//    partials) into LDS once per block; v_mfma_f32_16x16x32_bf16 with the sequence index on the
//    MFMA column; the 4 waves of a block split K and reduce through LDS.
//  * dec_attention_kernel: single-query attention streaming K and V rows with 16-byte loads
//    (8 keys x 128 B per wave instruction), lane-group-local online softmax, split-KV partials
//    for the 1500-key cross attention.
//  * dec_select_kernel: SuppressBlank / SuppressTokens / ApplyTimestampRules + greedy argmax +
//    running sum-logprob + no-speech probability (openai-whisper decoding.py, restated in
//    oracle/whisper_ref.py) -- one block per sequence, device-side state machine.
#include "decoder.h"

// ------------------------------------------------------------------------------------------
// Embedding: x[b] = tok_emb[token[b]] + pos_emb[pos[b]]
// ------------------------------------------------------------------------------------------
__global__ void dec_embed_kernel(const float* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                                 const int* __restrict__ cur_tok, const int* __restrict__ pos, float* __restrict__ x,
                                 int D) {
  const int b = blockIdx.x;
  const float4* te = (const float4*)(tok_emb + (long)cur_tok[b] * D);
  const float4* pe = (const float4*)(pos_emb + (long)pos[b] * D);
  float4* xo = (float4*)(x + (long)b * D);
  for (int i = threadIdx.x; i < D / 4; i += blockDim.x) {
    const float4 a = te[i], c = pe[i];
    xo[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
  }
}

// ------------------------------------------------------------------------------------------
// Skinny linear
// ------------------------------------------------------------------------------------------
template <int MT, int NT, int ACT, int EPI>
__global__ __launch_bounds__(256) void dec_linear_kernel(DecLinearParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MROWS = 16 * MT;
  constexpr int BN = 16 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * MROWS;
  const int K = p.K;
  const int lds_ld = K + 8;  // bf16 elements per LDS activation row (ACT_LN / ACT_COMBINE only)
  bf16_t* act_s = (bf16_t*)smem;
  // reduction scratch aliases the activation image when ACT_BF16 (no image) else sits after it
  float* red = (float*)(smem + ((ACT == ACT_BF16) ? 0 : ccx_align((size_t)MROWS * lds_ld * 2, 16)));

  // ---------------- activation staging ----------------
  if (ACT == ACT_LN) {
    // one wave per row: LayerNorm(x[m]) -> bf16 LDS row (K == d_model <= 1024)
    for (int r = wave; r < MROWS; r += 4) {
      const int m = m0 + r;
      const int nv = K >> 2;
      float4 v[4];
      float s = 0.f;
      const float4* xr = (const float4*)(p.x + (long)(m < p.M ? m : 0) * K);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int idx = lane + 64 * i;
        if (idx < nv) { v[i] = xr[idx]; s += v[i].x + v[i].y + v[i].z + v[i].w; }
      }
      const float mean = wave_reduce_sum(s) / (float)K;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
          const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
          q += a * a + b * b + c * c + d * d;
        }
      }
      const float rstd = rsqrtf(wave_reduce_sum(q) / (float)K + p.eps);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
          const float4 g = ((const float4*)p.ln_g)[idx], bb = ((const float4*)p.ln_b)[idx];
          uint2 o;
          if (m < p.M) {
            o.x = pack_bf16x2((v[i].x - mean) * rstd * g.x + bb.x, (v[i].y - mean) * rstd * g.y + bb.y);
            o.y = pack_bf16x2((v[i].z - mean) * rstd * g.z + bb.z, (v[i].w - mean) * rstd * g.w + bb.w);
          } else {
            o.x = 0; o.y = 0;
          }
          *(uint2*)(act_s + (long)r * lds_ld + 4 * idx) = o;
        }
      }
    }
    __syncthreads();
  } else if (ACT == ACT_COMBINE) {
    // combine split-KV attention partials: act[m][h*64+d] = sum_s w_s o_s[d] / sum_s w_s l_s
    const int H = K >> 6;
    for (int e = tid; e < MROWS * H * 8; e += 256) {  // one thread per (row, head, 8-wide d chunk)
      const int c = e & 7, h = (e >> 3) % H, r = e / (8 * H);
      const int m = m0 + r;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = 0.f;
      if (m < p.M) {
        const float* ml = p.part_ml + ((long)(m * H + h) * p.nsplit) * 2;
        float mx = -1e30f;
        for (int s = 0; s < p.nsplit; s++) mx = fmaxf(mx, ml[2 * s]);
        float den = 0.f;
        for (int s = 0; s < p.nsplit; s++) {
          const float w = __builtin_amdgcn_exp2f(ml[2 * s] - mx);
          den += w * ml[2 * s + 1];
          const float4* op = (const float4*)(p.part_o + ((long)(m * H + h) * p.nsplit + s) * 64 + 8 * c);
          const float4 a = op[0], b4 = op[1];
          o[0] += w * a.x; o[1] += w * a.y; o[2] += w * a.z; o[3] += w * a.w;
          o[4] += w * b4.x; o[5] += w * b4.y; o[6] += w * b4.z; o[7] += w * b4.w;
        }
        const float inv = 1.0f / den;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] *= inv;
      }
      uint4 pk;
      pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]);
      pk.z = pack_bf16x2(o[4], o[5]); pk.w = pack_bf16x2(o[6], o[7]);
      *(uint4*)(act_s + (long)r * lds_ld + h * 64 + 8 * c) = pk;
    }
    __syncthreads();
  }

  // ---------------- main loop: this wave's K slice ----------------
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, h4 = lane >> 4;
  const int ksteps = K >> 5;
  const int ks_per_wave = (ksteps + 3) >> 2;
  const int ks0 = wave * ks_per_wave;
  const int ks1 = (ks0 + ks_per_wave < ksteps) ? ks0 + ks_per_wave : ksteps;

  const bf16_t* wrow[NT];
#pragma unroll
  for (int i = 0; i < NT; i++) {
    int n = n0 + 16 * i + l15;
    n = n < p.N ? n : p.N - 1;
    wrow[i] = p.W + (long)n * p.ldw + 8 * h4;
  }
  const bf16_t* arow[MT];
#pragma unroll
  for (int j = 0; j < MT; j++) {
    if (ACT == ACT_BF16) {
      int m = m0 + 16 * j + l15;
      m = m < p.M ? m : p.M - 1;  // rows >= M compute garbage that is never stored
      arow[j] = p.act + (long)m * p.lda + 8 * h4;
    } else {
      arow[j] = act_s + (long)(16 * j + l15) * lds_ld + 8 * h4;
    }
  }

#pragma unroll 4
  for (int ks = ks0; ks < ks1; ks++) {
    bf16x8 wf[NT], af[MT];
#pragma unroll
    for (int i = 0; i < NT; i++) wf[i] = __builtin_nontemporal_load((const bf16x8*)(wrow[i] + 32 * ks));
#pragma unroll
    for (int j = 0; j < MT; j++) af[j] = *(const bf16x8*)(arow[j] + 32 * ks);
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
      for (int j = 0; j < MT; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
  }

  // ---------------- cross-wave reduction through LDS ----------------
  __syncthreads();  // everyone is done reading the activation image (red may alias nothing, but keep order)
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < MT; j++) *(f32x4*)(red + (((wave * NT + i) * MT + j) * 64 + lane) * 4) = acc[i][j];
  __syncthreads();

  for (int e = tid; e < BN * MROWS; e += 256) {
    const int nl = e % BN, r = e / BN;
    const int n = n0 + nl, m = m0 + r;
    if (n >= p.N || m >= p.M) continue;
    const int i = nl >> 4, j = r >> 4;
    const int ln = (r & 15) + 16 * ((nl & 15) >> 2), rg = nl & 3;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) v += red[(((w * NT + i) * MT + j) * 64 + ln) * 4 + rg];
    if (p.bias) v += p.bias[n];
    if (EPI == DEPI_BF16) {
      ((bf16_t*)p.out)[(long)m * p.ldo + n] = f32_to_bf16(v);
    } else if (EPI == DEPI_BF16_GELU) {
      ((bf16_t*)p.out)[(long)m * p.ldo + n] = f32_to_bf16(gelu_erf(v));
    } else if (EPI == DEPI_F32_ACCUM) {
      ((float*)p.out)[(long)m * p.ldo + n] += v;
    } else if (EPI == DEPI_F32) {
      ((float*)p.out)[(long)m * p.ldo + n] = v;
    } else if (EPI == DEPI_SELF_QKV) {
      const int D = p.K;  // d_model
      if (n < D) {
        ((float*)p.out)[(long)m * D + n] = v;
      } else {
        const int which = (n >= 2 * D);
        const int nn = n - (which ? 2 * D : D);
        const int hh = nn >> 6, d = nn & 63;
        const int H = D >> 6;
        bf16_t* cache = which ? p.cache_v : p.cache_k;
        cache[(((long)m * H + hh) * p.cache_T + p.pos[m]) * 64 + d] = f32_to_bf16(v);
      }
    }
  }
}

template <int MT, int NT, int ACT, int EPI>
static int launch_dec_linear_inst(ccx_ctx* ctx, const DecLinearParams& p, hipStream_t stream) {
  const int MROWS = 16 * MT, BN = 16 * NT;
  size_t act_bytes = (ACT == ACT_BF16) ? 0 : ccx_align((size_t)MROWS * (p.K + 8) * 2, 16);
  size_t red_bytes = (size_t)4 * NT * MT * 64 * 4 * 4;
  size_t lds = act_bytes + red_bytes;
  CCX_REQUIRE(ctx, lds <= 160 * 1024, "dec_linear: LDS %zu too large", lds);
  static size_t attr_set = 0;
  if (lds > 64 * 1024 && lds > attr_set) {
    CCX_HIP(ctx, hipFuncSetAttribute((const void*)dec_linear_kernel<MT, NT, ACT, EPI>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = 160 * 1024;
  }
  dim3 grid(ccx_cdiv(p.N, BN), ccx_cdiv(p.M, MROWS));
  hipLaunchKernelGGL((dec_linear_kernel<MT, NT, ACT, EPI>), grid, dim3(256), lds, stream, p);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

template <int ACT, int EPI>
static int launch_dec_linear_mt(ccx_ctx* ctx, const DecLinearParams& p, int nt, hipStream_t stream) {
  const int M = p.M;
  if (nt == 4) {
    if (M <= 16) return launch_dec_linear_inst<1, 4, ACT, EPI>(ctx, p, stream);
    if (M <= 32) return launch_dec_linear_inst<2, 4, ACT, EPI>(ctx, p, stream);
    return launch_dec_linear_inst<4, 4, ACT, EPI>(ctx, p, stream);
  }
  if (M <= 16) return launch_dec_linear_inst<1, 1, ACT, EPI>(ctx, p, stream);
  if (M <= 32) return launch_dec_linear_inst<2, 1, ACT, EPI>(ctx, p, stream);
  return launch_dec_linear_inst<4, 1, ACT, EPI>(ctx, p, stream);
}

int ccx_launch_dec_linear(ccx_ctx* ctx, int act, int epi, const DecLinearParams& p, hipStream_t stream) {
  CCX_REQUIRE(ctx, p.M > 0 && p.N > 0 && p.K > 0 && p.K % 32 == 0, "dec_linear: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
  CCX_REQUIRE(ctx, act == ACT_BF16 || p.K <= 1024, "dec_linear: LN/combine activation needs K <= 1024");
  CCX_REQUIRE(ctx, p.ldw % 8 == 0, "dec_linear: ldw must be a multiple of 8");
  // wide-N layers use 64-row weight panels per block, narrow ones 16 to spread over more CUs
  const int nt = (p.N >= 8192) ? 4 : 1;
#define CASE(A, E) \
  if (act == A && epi == E) return launch_dec_linear_mt<A, E>(ctx, p, nt, stream);
  CASE(ACT_LN, DEPI_SELF_QKV)
  CASE(ACT_LN, DEPI_F32)
  CASE(ACT_LN, DEPI_BF16_GELU)
  CASE(ACT_BF16, DEPI_F32_ACCUM)
  CASE(ACT_BF16, DEPI_F32)
  CASE(ACT_COMBINE, DEPI_F32_ACCUM)
#undef CASE
  return ccx_fail(ctx, CCX_ERR_ARG, "dec_linear: unsupported act=%d epi=%d", act, epi);
}

// ------------------------------------------------------------------------------------------
// Single-query attention (self: T = pos+1 keys, FINAL output; cross: split-KV partials)
// ------------------------------------------------------------------------------------------
struct SoftState {
  float m, l, o[8];
};
__device__ __forceinline__ void soft_merge(SoftState& a, float bm, float bl, const float (&bo)[8]) {
  const float mx = fmaxf(a.m, bm);
  const float wa = __builtin_amdgcn_exp2f(a.m - mx), wb = __builtin_amdgcn_exp2f(bm - mx);
  a.l = a.l * wa + bl * wb;
#pragma unroll
  for (int j = 0; j < 8; j++) a.o[j] = a.o[j] * wa + bo[j] * wb;
  a.m = mx;
}

template <bool FINAL>
__global__ __launch_bounds__(256) void dec_attention_kernel(DecAttnParams p) {
  __shared__ float sm_m[4][8], sm_l[4][8], sm_o[4][8][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, split = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int g = lane >> 3, c = lane & 7;
  const int T = p.pos ? (p.pos[b] + 1) : p.T;
  // this block's key range
  const int per = (T + gridDim.y - 1) / gridDim.y;
  const int kbeg = split * per;
  const int kend = (kbeg + per < T) ? kbeg + per : T;

  float q[8];
  {
    const float4* qp = (const float4*)(p.q + ((long)b * p.H + h) * 64 + 8 * c);
    const float4 a = qp[0], d = qp[1];
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = d.x; q[5] = d.y; q[6] = d.z; q[7] = d.w;
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] *= p.scale_log2e;
  }
  const bf16_t* Kb = p.k + ((long)b * p.H + h) * p.kv_T * 64 + 8 * c;
  const bf16_t* Vb = p.v + ((long)b * p.H + h) * p.kv_T * 64 + 8 * c;

  SoftState st;
  st.m = -1e30f; st.l = 0.f;
#pragma unroll
  for (int j = 0; j < 8; j++) st.o[j] = 0.f;

  // each wave instruction covers 8 keys (one per lane group g); waves interleave in units of 8 keys
  for (int k0 = kbeg + wave * 8; k0 < kend; k0 += 32) {
    const int key = k0 + g;
    const bool ok = key < kend;
    const int kk = ok ? key : (kend - 1);
    const bf16x8 kv = *(const bf16x8*)(Kb + (long)kk * 64);
    const bf16x8 vv = *(const bf16x8*)(Vb + (long)kk * 64);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) s = fmaf(q[j], bf16_to_f32((bf16_t)kv[j]), s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (!ok) s = -INFINITY;
    const float mn = fmaxf(st.m, s);
    const float al = __builtin_amdgcn_exp2f(st.m - mn);
    const float pe = __builtin_amdgcn_exp2f(s - mn);
    st.l = st.l * al + pe;
#pragma unroll
    for (int j = 0; j < 8; j++) st.o[j] = st.o[j] * al + pe * bf16_to_f32((bf16_t)vv[j]);
    st.m = mn;
  }
  // merge the 8 lane groups (lanes with equal c): xor 8, 16, 32
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    const float bm = __shfl_xor(st.m, off, 64), bl = __shfl_xor(st.l, off, 64);
    float bo[8];
#pragma unroll
    for (int j = 0; j < 8; j++) bo[j] = __shfl_xor(st.o[j], off, 64);
    soft_merge(st, bm, bl, bo);
  }
  if (g == 0) {
    sm_m[wave][c] = st.m; sm_l[wave][c] = st.l;
#pragma unroll
    for (int j = 0; j < 8; j++) sm_o[wave][c][j] = st.o[j];
  }
  __syncthreads();
  if (tid < 64) {
    // thread -> d = tid; merge 4 waves
    const int cc = tid >> 3, j = tid & 7;
    float mx = fmaxf(fmaxf(sm_m[0][cc], sm_m[1][cc]), fmaxf(sm_m[2][cc], sm_m[3][cc]));
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const float ww = __builtin_amdgcn_exp2f(sm_m[w][cc] - mx);
      l += ww * sm_l[w][cc];
      o += ww * sm_o[w][cc][j];
    }
    if (FINAL) {
      p.out_bf16[((long)b * p.H + h) * 64 + tid] = f32_to_bf16(o / l);
    } else {
      const long base = ((long)b * p.H + h) * gridDim.y + split;
      p.part_o[base * 64 + tid] = o;
      if (tid == 0) { p.part_ml[base * 2] = mx; p.part_ml[base * 2 + 1] = l; }
    }
  }
}

int ccx_launch_dec_attention(ccx_ctx* ctx, const DecAttnParams& p, int B, int nsplit, bool final_out,
                             hipStream_t stream) {
  CCX_REQUIRE(ctx, B > 0 && p.H > 0 && nsplit >= 1, "dec_attention: bad shape");
  CCX_REQUIRE(ctx, !final_out || nsplit == 1, "dec_attention: final output needs nsplit == 1");
  dim3 grid(B * p.H, nsplit);
  if (final_out) hipLaunchKernelGGL(dec_attention_kernel<true>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(dec_attention_kernel<false>, grid, dim3(256), 0, stream, p);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

// ------------------------------------------------------------------------------------------
// Logit filters + greedy selection + per-sequence state machine
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
  v = wave_reduce_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; i++) r = fmaxf(r, sh[i]);
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
  v = wave_reduce_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; i++) r += sh[i];
  return r;
}

__global__ __launch_bounds__(1024) void dec_select_kernel(DecSelectParams p) {
  __shared__ float sh[16];
  __shared__ float sh_v[16];
  __shared__ int sh_i[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int V = p.n_vocab;
  const float* lg = p.logits + (long)b * p.ld_logits;
  DecSeqState s = p.state[b];  // every thread reads the same struct ...
  __syncthreads();             // ... before thread 0 may overwrite it in an early-exit branch

  // ---- prompt phase: feed the next prompt token, nothing is sampled ----
  if (s.pos < s.prompt_len - 1) {
    if (tid == 0) {
      s.pos += 1;
      p.cur_tok[b] = p.prompt[(long)b * p.max_prompt + s.pos];
      p.pos[b] = s.pos;
      p.state[b] = s;
    }
    return;
  }
  if (s.done) {  // finished rows keep emitting eot; nothing else changes
    if (tid == 0 && s.n_gen < p.sample_len) {
      p.gen[(long)b * p.sample_len + s.n_gen] = p.eot;
      s.n_gen += 1;
      p.state[b] = s;
    }
    return;
  }

  const int i_gen = s.n_gen;
  const int tsb = p.timestamp_begin;
  // ---- no-speech probability from the raw logits at the SOT position (first sampling step) ----
  if (i_gen == 0) {
    float mx = -INFINITY;
    for (int v = tid; v < V; v += blockDim.x) mx = fmaxf(mx, lg[v]);
    mx = block_reduce_max(mx, sh);
    float sum = 0.f;
    for (int v = tid; v < V; v += blockDim.x) sum += expf(lg[v] - mx);
    sum = block_reduce_sum(sum, sh);
    if (tid == 0) s.no_speech_prob = expf(lg[p.no_speech] - mx) / sum;
  }

  // ---- timestamp-rule state (ApplyTimestampRules) ----
  const bool last_ts = i_gen >= 1 && s.last_tok >= tsb;
  const bool pen_ts = i_gen < 2 || s.pen_tok >= tsb;
  int ts_floor = tsb;  // timestamps in [tsb, ts_floor) are banned
  if (s.last_ts_tok >= 0) ts_floor = (last_ts && !pen_ts) ? s.last_ts_tok : s.last_ts_tok + 1;

  auto masked = [&](int v) -> bool {
    if (p.suppress_mask[v]) return true;                      // SuppressTokens + <|notimestamps|>
    if (i_gen == 0 && (v == p.blank || v == p.eot)) return true;  // SuppressBlank
    if (last_ts) {
      if (pen_ts) { if (v >= tsb) return true; }
      else { if (v < p.eot) return true; }
    }
    if (v >= tsb && v < ts_floor) return true;
    if (i_gen == 0) {
      if (v < tsb) return true;
      if (p.max_initial_ts >= 0 && v > tsb + p.max_initial_ts) return true;
    }
    return false;
  };

  // pass 1: maxima (text / timestamp) over the filtered logits
  float mx_text = -INFINITY, mx_ts = -INFINITY;
  int am_text = 0x7fffffff, am_ts = 0x7fffffff;
  for (int v = tid; v < V; v += blockDim.x) {
    if (masked(v)) continue;
    const float x = lg[v];
    if (v < tsb) { if (x > mx_text) { mx_text = x; am_text = v; } }
    else { if (x > mx_ts) { mx_ts = x; am_ts = v; } }
  }
  // block argmax (ties -> lowest index, as torch.argmax on CPU)
  auto block_argmax = [&](float val, int idx, float& oval, int& oidx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(val, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { sh_v[tid >> 6] = val; sh_i[tid >> 6] = idx; }
    __syncthreads();
    oval = sh_v[0]; oidx = sh_i[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++)
      if (sh_v[w] > oval || (sh_v[w] == oval && sh_i[w] < oidx)) { oval = sh_v[w]; oidx = sh_i[w]; }
  };
  float bt, bs; int it, is;
  block_argmax(mx_text, am_text, bt, it);
  block_argmax(mx_ts, am_ts, bs, is);
  const float mx_all = fmaxf(bt, bs);

  // pass 2: sum exp over text and over timestamps (relative to mx_all)
  float se_text = 0.f, se_ts = 0.f;
  for (int v = tid; v < V; v += blockDim.x) {
    if (masked(v)) continue;
    const float e = expf(lg[v] - mx_all);
    if (v < tsb) se_text += e; else se_ts += e;
  }
  se_text = block_reduce_sum(se_text, sh);
  se_ts = block_reduce_sum(se_ts, sh);

  if (tid == 0) {
    // timestamp_logprob > max_text_token_logprob  <=>  lse_ts > max_text (same normaliser)
    const float lse_ts = (se_ts > 0.f) ? mx_all + logf(se_ts) : -INFINITY;
    const bool force_ts = lse_ts > bt;
    int next; float logprob;
    if (force_ts) {
      next = is;
      logprob = bs - lse_ts;  // log_softmax over the re-filtered logits (text banned)
    } else {
      if (bt > bs || (bt == bs && it < is)) next = it; else next = is;
      const float lse = mx_all + logf(se_text + se_ts);
      logprob = fmaxf(bt, bs) - lse;
    }
    s.sum_logprob += logprob;
    p.gen[(long)b * p.sample_len + i_gen] = next;
    s.n_gen = i_gen + 1;
    s.pen_tok = s.last_tok;
    s.last_tok = next;
    if (next >= tsb) s.last_ts_tok = next;
    if (next == p.eot) {
      s.done = 1;
      s.n_tokens = i_gen;  // tokens before the first eot
      atomicAdd(p.n_done, 1);
    } else if (s.n_gen >= p.sample_len) {
      s.done = 1;
      s.n_tokens = s.n_gen;
      atomicAdd(p.n_done, 1);
    } else {
      s.pos += 1;
      p.cur_tok[b] = next;
      p.pos[b] = s.pos;
    }
    p.state[b] = s;
  }
}

int ccx_launch_dec_embed(ccx_ctx* ctx, const float* tok_emb, const float* pos_emb, const int* cur_tok, const int* pos,
                         float* x, int B, int D, hipStream_t stream) {
  hipLaunchKernelGGL(dec_embed_kernel, dim3(B), dim3(192), 0, stream, tok_emb, pos_emb, cur_tok, pos, x, D);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

int ccx_launch_dec_select(ccx_ctx* ctx, const DecSelectParams& p, int B, hipStream_t stream) {
  hipLaunchKernelGGL(dec_select_kernel, dim3(B), dim3(1024), 0, stream, p);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}
