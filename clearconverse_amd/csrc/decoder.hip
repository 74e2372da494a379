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
#include <string>
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

#include "dec_ln.h"

// ------------------------------------------------------------------------------------------
// Skinny linear
// ------------------------------------------------------------------------------------------
// Every kernel of the decode chain is latency-bound (a 768x768 bf16 matrix is 4.6 KB per CU, and a
// slot of a dependent graph chain costs ~1.6 us even when empty -- tools/microbench_chain.hip), so
// the structure minimises instructions and dependent round trips on the one critical path:
//  1. each wave issues ALL weight loads of its K slice first (<= KMAX x NT 16-byte non-temporal
//     loads per lane: each weight byte is read once per step);
//  2. the activation prologue runs under that flight: LayerNorm of the residual rows (live rows
//     only, two at a time), or the split-KV attention combine, or nothing (bf16 activations are
//     loaded as fragments, also all up front);
//  3. MFMAs run out of registers; the 4 waves (K split) reduce through LDS.
// Residual adds are DEFERRED: out-proj / cross-out / FFN2 write split-K partial sums (DEPI_PARTIAL,
// grid.z = K split, so 4.7 MB matrices spread over 192 blocks instead of 48) and the NEXT
// LayerNorm prologue folds them in (x_eff = x_in + sum partials), block (0,*,0) writing x_eff to
// the other residual buffer (ping-pong: no block may see a half-updated stream).
// NW waves per block split K (4; 12 for the K = 3072 second MLP linear of the LayerNorm-free chain, which must not use split-K slabs).
template <int MT, int NT, int KMAX, int ACT, int EPI, int NW = 4>
__global__ __launch_bounds__(64 * NW) void dec_linear_kernel(DecLinearParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MROWS = 16 * MT;
  constexpr int BN = 16 * NT;
  constexpr int NTHR = 64 * NW;
  constexpr bool BF16IN = (ACT == ACT_BF16 || ACT == ACT_BF16_LN);
  static_assert(NW == 4 || ACT == ACT_BF16, "the LayerNorm / combine prologues are written for four waves");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * MROWS;
  const int K = p.K;
  const int lds_ld = K + 8;  // bf16 elements per LDS activation row (ACT_LN / ACT_COMBINE only)
  bf16_t* act_s = (bf16_t*)smem;
  float* red = (float*)(smem + (BF16IN ? 0 : ccx_align((size_t)MROWS * lds_ld * 2, 16)));
  float2* lnst = (float2*)(red + NW * NT * MT * 64 * 4);      // ACT_BF16_LN: (mean, rstd) of the block's rows

  const int l15 = lane & 15, h4 = lane >> 4;
  // K range of this block (grid.z split), then of this wave
  const int ksteps = K >> 5;
  const int per_z = (ksteps + gridDim.z - 1) / gridDim.z;
  const int kz0 = blockIdx.z * per_z;
  const int kz1 = (kz0 + per_z < ksteps) ? kz0 + per_z : ksteps;
  const int per_w = (kz1 - kz0 + NW - 1) / NW;
  const int ks0 = kz0 + wave * per_w;
  int nks = kz1 - ks0;
  nks = nks < 0 ? 0 : (nks > per_w ? per_w : nks);

  // Epilogue ownership: a thread finishes QUADS of 4 consecutive output columns of one row.  Quad qd = ((i * MT + j) * 64 + ln)
  // is exactly the f32x4 that lane ln of every wave holds for tile (i, j): row 16 j + (ln & 15), columns 16 i + 4 (ln >> 4) .. + 3.
  // The four waves' partial sums of a quad are therefore four ds_read_b128 at consecutive-lane addresses (conflict-free; the
  // former one-column-per-thread walk read single floats 64 apart: 4-way bank conflicts, 60 % of the kernel's LDS cycles).
  constexpr int QUADS = NT * MT * 64;
  constexpr int EPI_ITERS = (QUADS + NTHR - 1) / NTHR;
  // bias of this thread's columns and, for DEPI_SELF_QKV, the cache position of its rows: loaded up front so that the epilogue
  // has no dependent memory round trip of its own
  float4 bias_v[EPI_ITERS];
  float4 lns_v[EPI_ITERS];     // ACT_BF16_LN: s of this thread's columns
  float4 xold_v[EPI_ITERS];    // DEPI_RESOLVE: the residual tile this thread finishes
  int pos_v[EPI_ITERS];
#pragma unroll
  for (int it = 0; it < EPI_ITERS; it++) {
    const int qd = tid + NTHR * it;
    const int ln = qd & 63, tj = (qd >> 6) % MT, ti = (qd >> 6) / MT;
    const int n_ = n0 + 16 * ti + 4 * (ln >> 4);
    bias_v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && blockIdx.z == 0 && qd < QUADS) {
      if (n_ + 3 < p.N) bias_v[it] = *(const float4*)(p.bias + n_);
      else {
        if (n_ < p.N) bias_v[it].x = p.bias[n_];
        if (n_ + 1 < p.N) bias_v[it].y = p.bias[n_ + 1];
        if (n_ + 2 < p.N) bias_v[it].z = p.bias[n_ + 2];
      }
    }
    pos_v[it] = 0;
    if (EPI == DEPI_SELF_QKV) {
      const int m_ = m0 + 16 * tj + (ln & 15);
      pos_v[it] = p.pos[m_ < p.M ? m_ : p.M - 1];
    }
    lns_v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ACT == ACT_BF16_LN && qd < QUADS && n_ + 3 < p.N) lns_v[it] = *(const float4*)(p.ln_s + n_);
    xold_v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == DEPI_RESOLVE && qd < QUADS) {
      const int m_ = m0 + 16 * tj + (ln & 15);
      xold_v[it] = *(const float4*)(p.xres + (long)(m_ < p.M ? m_ : p.M - 1) * p.N + (n_ + 3 < p.N ? n_ : 0));
    }
  }

  // ---------------- 1. weight (and bf16 activation) prefetch ----------------
  // W is stored fragment-packed (whisper.hip pack_mfma_rows): tile (n/16, k/32) holds the 64 lanes'
  // 16-byte A fragments back to back, so one wave load is 1 KB contiguous and successive k-steps
  // are successive KBs.  Rows >= N are zero padding inside the packed image.
  bf16x8 wf[NT][KMAX];
#pragma unroll
  for (int i = 0; i < NT; i++) {
    const long tile = (long)(blockIdx.x * NT + i) * ksteps + ks0;
    const bf16_t* wr = p.W + (tile * 64 + lane) * 8;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      const int kk = k < nks ? k : 0;  // clamp instead of branching: keeps the loads back to back
      wf[i][k] = __builtin_nontemporal_load((const bf16x8*)(wr + 512 * kk));
    }
  }
  bf16x8 af[MT][KMAX];
  if (BF16IN) {
#pragma unroll
    for (int j = 0; j < MT; j++) {
      int m = m0 + 16 * j + l15;
      m = m < p.M ? m : p.M - 1;  // rows >= M compute garbage that is never stored
      const bf16_t* ar = p.act + (long)m * p.lda + 8 * h4 + 32 * ks0;
#pragma unroll
      for (int k = 0; k < KMAX; k++) {
        const int kk = k < nks ? k : 0;
        af[j][k] = *(const bf16x8*)(ar + 32 * kk);
      }
    }
  }

  // ---------------- 2. activation staging ----------------
  if (ACT == ACT_BF16_LN) {
    // (mean, rstd) of the block's rows from the producer's per-tile statistics: four threads per row, each a quarter of the K / 16
    // tiles in order, combined (q0 + q1) + (q2 + q3) -- a fixed order, whatever tile geometry this block has
    if (tid < MROWS * 4) {
      const int r = tid >> 2, qt = tid & 3;
      int m = m0 + r;
      m = m < p.M ? m : p.M - 1;
      const int t4 = (K >> 4) >> 2;
      const float2* sp = p.ln_stats + (long)m * (K >> 4) + qt * t4;
      float s1 = 0.f, s2 = 0.f;
      for (int i = 0; i < t4; i++) {
        const float2 v = sp[i];
        s1 += v.x; s2 += v.y;
      }
      s1 += dpp_mov<CCX_DPP_QUAD_XOR1>(s1); s2 += dpp_mov<CCX_DPP_QUAD_XOR1>(s2);
      s1 += dpp_mov<CCX_DPP_QUAD_XOR2>(s1); s2 += dpp_mov<CCX_DPP_QUAD_XOR2>(s2);
      const float mean = s1 / (float)K;
      const float var = fmaf(-mean, mean, s2 / (float)K);
      if (qt == 0) lnst[r] = make_float2(mean, rsqrtf(fmaxf(var, 0.f) + p.eps));
    }
  }
  if (ACT == ACT_LN) {
    // x_eff = x + sum of pending split-K partials; LayerNorm(x_eff) -> bf16 LDS rows (K <= 1024).
    // A wave owns live rows wave, wave+4, ... and handles two of them per pass.
    const int nv = K >> 2;
    int live = p.M - m0;
    live = live > MROWS ? MROWS : live;
    const bool writer = (blockIdx.x == 0 && blockIdx.z == 0 && p.x_out != nullptr);
    float4 g[4], bb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = lane + 64 * i;
      const int ic = idx < nv ? idx : 0;
      g[i] = ((const float4*)p.ln_g)[ic];
      bb[i] = ((const float4*)p.ln_b)[ic];
    }
    for (int r0 = wave; r0 < live; r0 += 8) {
      float4 v[2][4];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int r = r0 + 4 * u;
        ok[u] = r < live;
        const long row = (long)(m0 + (ok[u] ? r : r0)) * K;
        const float4* xr = (const float4*)(p.x + row);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int idx = lane + 64 * i;
          const int ic = idx < nv ? idx : 0;
          float4 a = xr[ic];
          float4 q[4];
#pragma unroll
          for (int s = 0; s < 4; s++)  // all slab loads issued together (clamped index, masked add)
            q[s] = ((const float4*)(p.pend + (long)(s < p.pend_n ? s : 0) * p.pend_stride + row))[ic];
#pragma unroll
          for (int s = 0; s < 4; s++) a = ln_add_pend(a, s < p.pend_n ? 1.f : 0.f, q[s]);
          if (idx >= nv) a = make_float4(0.f, 0.f, 0.f, 0.f);
          v[u][i] = a;
        }
      }
      float mean[2], rstd[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) sm += ln_sum4(v[u][i]);
        mean[u] = wave_reduce_sum(sm) / (float)K;
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float t = ln_sq4(v[u][i], mean[u]);
          sq += (lane + 64 * i < nv) ? t : 0.f;
        }
        rstd[u] = rsqrtf(wave_reduce_sum(sq) / (float)K + p.eps);
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (!ok[u]) continue;
        const int r = r0 + 4 * u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int idx = lane + 64 * i;
          if (idx < nv) {
            *(uint2*)(act_s + (long)r * lds_ld + 4 * idx) = ln_pack4(v[u][i], mean[u], rstd[u], g[i], bb[i]);
            if (writer) ((float4*)(p.x_out + (long)(m0 + r) * K))[idx] = v[u][i];
          }
        }
      }
    }
    // dead rows of the 16-row MFMA tile: zeros
    for (int r = live + wave; r < MROWS; r += 4)
      for (int idx = lane; idx < (K >> 2); idx += 64) *(uint2*)(act_s + (long)r * lds_ld + 4 * idx) = make_uint2(0, 0);
    __syncthreads();
  } else if (ACT == ACT_COMBINE) {
    // combine split-KV attention partials: act[m][h*64+d] = sum_s w_s o_s[d] / sum_s w_s l_s.
    // One thread per (row, head, 8-wide d chunk); all partial loads (nsplit <= 8) issued up front.
    const int H = K >> 6;
    for (int e = tid; e < MROWS * H * 8; e += 256) {
      const int c = e & 7, h = (e >> 3) % H, r = e / (8 * H);
      const int m = m0 + r;
      uint4 pk = make_uint4(0, 0, 0, 0);
      if (m < p.M) {
        const float2* ml = (const float2*)(p.part_ml + ((long)(m * H + h) * p.nsplit) * 2);
        const float* ob = p.part_o + ((long)(m * H + h) * p.nsplit) * 64 + 8 * c;
        float2 mlv[8];
        float4 oa[8], oc[8];
#pragma unroll
        for (int s = 0; s < 8; s++) {
          const int sc = s < p.nsplit ? s : 0;
          mlv[s] = ml[sc];
          oa[s] = *(const float4*)(ob + sc * 64);
          oc[s] = *(const float4*)(ob + sc * 64 + 4);
          if (s >= p.nsplit) mlv[s] = make_float2(-1e30f, 0.f);
        }
        float mx = -1e30f;
#pragma unroll
        for (int s = 0; s < 8; s++) mx = fmaxf(mx, mlv[s].x);
        float den = 0.f, o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; s++) {
          const float w = (s < p.nsplit) ? __builtin_amdgcn_exp2f(mlv[s].x - mx) : 0.f;
          den += w * mlv[s].y;
          o[0] += w * oa[s].x; o[1] += w * oa[s].y; o[2] += w * oa[s].z; o[3] += w * oa[s].w;
          o[4] += w * oc[s].x; o[5] += w * oc[s].y; o[6] += w * oc[s].z; o[7] += w * oc[s].w;
        }
        const float inv = 1.0f / den;
        pk.x = pack_bf16x2(o[0] * inv, o[1] * inv); pk.y = pack_bf16x2(o[2] * inv, o[3] * inv);
        pk.z = pack_bf16x2(o[4] * inv, o[5] * inv); pk.w = pack_bf16x2(o[6] * inv, o[7] * inv);
      }
      *(uint4*)(act_s + (long)r * lds_ld + h * 64 + 8 * c) = pk;
    }
    __syncthreads();
  }

  // ---------------- 3. MFMAs out of the prefetched registers ----------------
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KMAX; k++) {
    if (k < nks) {
      if (!BF16IN) {
#pragma unroll
        for (int j = 0; j < MT; j++)
          af[j][k] = *(const bf16x8*)(act_s + (long)(16 * j + l15) * lds_ld + 8 * h4 + 32 * (ks0 + k));
      }
#pragma unroll
      for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][k], af[j][k], acc[i][j], 0, 0, 0);
    }
  }

  // ---------------- 4. cross-wave reduction through LDS + epilogue ----------------
  __syncthreads();  // everyone is done reading the activation image
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < MT; j++) *(f32x4*)(red + (((wave * NT + i) * MT + j) * 64 + lane) * 4) = acc[i][j];
  __syncthreads();

#pragma unroll
  for (int it = 0; it < EPI_ITERS; it++) {
    const int qd = tid + NTHR * it;
    if (qd >= QUADS) break;
    const int ln = qd & 63, j = (qd >> 6) % MT, i = (qd >> 6) / MT;
    const int m = m0 + 16 * j + (ln & 15), n = n0 + 16 * i + 4 * (ln >> 4);
    if (EPI == DEPI_RESOLVE) {
      // whole waves stay together here (the statistics go through lane swaps); N is a multiple of the block's columns
      f32x4 v = *(const f32x4*)(red + (((0 * NT + i) * MT + j) * 64 + ln) * 4);
#pragma unroll
      for (int w = 1; w < NW; w++) v += *(const f32x4*)(red + (((w * NT + i) * MT + j) * 64 + ln) * 4);
      v[0] = (v[0] + bias_v[it].x) + xold_v[it].x; v[1] = (v[1] + bias_v[it].y) + xold_v[it].y;
      v[2] = (v[2] + bias_v[it].z) + xold_v[it].z; v[3] = (v[3] + bias_v[it].w) + xold_v[it].w;
      float s1 = (v[0] + v[1]) + (v[2] + v[3]);
      float s2 = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
      s1 += lane_xor16(s1); s2 += lane_xor16(s2);        // the four quads of a row's 16-column tile sit 16 lanes apart
      s1 += lane_xor32(s1); s2 += lane_xor32(s2);
      if (m < p.M && n < p.N) {
        *(float4*)(p.xres + (long)m * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)(p.xb + (long)m * p.N + n) = pk;
        if ((ln >> 4) == 0) p.st_out[(long)m * (p.N >> 4) + (n >> 4)] = make_float2(s1, s2);
      }
      continue;
    }
    if (n >= p.N || m >= p.M) continue;
    f32x4 v = *(const f32x4*)(red + (((0 * NT + i) * MT + j) * 64 + ln) * 4);
#pragma unroll
    for (int w = 1; w < NW; w++) v += *(const f32x4*)(red + (((w * NT + i) * MT + j) * 64 + ln) * 4);      // same order as before: w = 0..3
    if (ACT == ACT_BF16_LN) {
      // LN(x) W^T + b = rstd (x (gamma o W)^T - mean s) + c, every multiply-add an explicit fmaf
      const float2 ms = lnst[16 * j + (ln & 15)];
      v[0] = fmaf(ms.y, fmaf(-ms.x, lns_v[it].x, v[0]), bias_v[it].x); v[1] = fmaf(ms.y, fmaf(-ms.x, lns_v[it].y, v[1]), bias_v[it].y);
      v[2] = fmaf(ms.y, fmaf(-ms.x, lns_v[it].z, v[2]), bias_v[it].z); v[3] = fmaf(ms.y, fmaf(-ms.x, lns_v[it].w, v[3]), bias_v[it].w);
    } else {
      v[0] += bias_v[it].x; v[1] += bias_v[it].y; v[2] += bias_v[it].z; v[3] += bias_v[it].w;
    }
    const bool full = n + 3 < p.N;       // N is a multiple of 4 everywhere but a guard costs nothing
    if (EPI == DEPI_BF16_GELU) {
      bf16_t* o = (bf16_t*)p.out + (long)m * p.ldo + n;
      if (full) {
        uint2 pk;
        pk.x = pack_bf16x2(gelu_erf(v[0]), gelu_erf(v[1]));
        pk.y = pack_bf16x2(gelu_erf(v[2]), gelu_erf(v[3]));
        *(uint2*)o = pk;
      } else {
        for (int c = 0; c < 4 && n + c < p.N; c++) o[c] = f32_to_bf16(gelu_erf(v[c]));
      }
    } else if (EPI == DEPI_PARTIAL || EPI == DEPI_F32) {
      float* o = (float*)p.out + (EPI == DEPI_PARTIAL ? (long)blockIdx.z * p.pend_stride : 0L) + (long)m * p.ldo + n;
      if (full) *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      else
        for (int c = 0; c < 4 && n + c < p.N; c++) o[c] = v[c];
    } else if (EPI == DEPI_SELF_QKV) {
      const int D = p.K;  // d_model (a multiple of 64: a quad never straddles q / k / v or two heads)
      if (n < D) {
        *(float4*)((float*)p.out + (long)m * D + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        const int which = (n >= 2 * D);
        const int nn = n - (which ? 2 * D : D);
        const int hh = nn >> 6, d = nn & 63;
        const int H = D >> 6;
        bf16_t* cache = which ? p.cache_v : p.cache_k;
        const int sq = p.row_seq ? p.row_seq[m] : m;
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)(cache + (((long)sq * H + hh) * p.cache_T + pos_v[it]) * 64 + d) = pk;
      }
    }
  }
}

// Stand-alone residual resolve + LayerNorm -> bf16 (input of the logits GEMV): one wave per row.
__global__ __launch_bounds__(256) void dec_resolve_ln_kernel(const float* __restrict__ x, const float* __restrict__ pend,
                                                             int pend_n, long pend_stride, const float* __restrict__ g,
                                                             const float* __restrict__ b, bf16_t* __restrict__ out,
                                                             float* __restrict__ x_out, int M, int K, float eps) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int nv = K >> 2;
  float4 v[4], gg[4], bb[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {   // affine parameters first: no dependent load after the reductions
    const int idx = lane + 64 * i;
    const int ic = idx < nv ? idx : 0;
    gg[i] = ((const float4*)g)[ic];
    bb[i] = ((const float4*)b)[ic];
  }
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int idx = lane + 64 * i;
    const int ic = idx < nv ? idx : 0;
    float4 a = ((const float4*)(x + (long)m * K))[ic];
    float4 q[4];
#pragma unroll
    for (int s = 0; s < 4; s++) q[s] = ((const float4*)(pend + (long)(s < pend_n ? s : 0) * pend_stride + (long)m * K))[ic];
#pragma unroll
    for (int s = 0; s < 4; s++) a = ln_add_pend(a, s < pend_n ? 1.f : 0.f, q[s]);
    if (idx >= nv) a = make_float4(0.f, 0.f, 0.f, 0.f);
    v[i] = a;
    if (x_out && idx < nv) ((float4*)(x_out + (long)m * K))[idx] = a;   // resolved residual stream (ping-pong buffer)
    sm += ln_sum4(a);
  }
  const float mean = wave_reduce_sum(sm) / (float)K;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float t = ln_sq4(v[i], mean);
    sq += (lane + 64 * i < nv) ? t : 0.f;
  }
  const float rstd = rsqrtf(wave_reduce_sum(sq) / (float)K + eps);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int idx = lane + 64 * i;
    if (idx < nv) ((uint2*)(out + (long)m * K))[idx] = ln_pack4(v[i], mean, rstd, gg[i], bb[i]);
  }
}

// Stand-alone resolve for the LayerNorm-free chain: x += sum of the pending split-K slabs (in place: a wave owns its row and reads all of
// it before it writes), a bf16 copy of the new row and (sum, sum of squares) per 16-column tile -- what DEPI_RESOLVE leaves behind,
// for the one producer that keeps its split-K slabs (the K = 3072 second MLP linear).  One wave per row; the statistics of a tile
// are the four lanes' quads in the order (q0 + q1) + (q2 + q3), as in the linear's epilogue.
__global__ __launch_bounds__(256) void dec_resolve_stats_kernel(float* __restrict__ x, const float* __restrict__ pend, int pend_n,
                                                                long pend_stride, bf16_t* __restrict__ xb, float2* __restrict__ st,
                                                                int M, int K) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int nv = K >> 2;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int idx = lane + 64 * i;
    const int ic = idx < nv ? idx : 0;
    float4 a = ((const float4*)(x + (long)m * K))[ic];
    float4 q[4];
#pragma unroll
    for (int s = 0; s < 4; s++) q[s] = ((const float4*)(pend + (long)(s < pend_n ? s : 0) * pend_stride + (long)m * K))[ic];
#pragma unroll
    for (int s = 0; s < 4; s++) a = ln_add_pend(a, s < pend_n ? 1.f : 0.f, q[s]);
    float s1 = (a.x + a.y) + (a.z + a.w);
    float s2 = fmaf(a.w, a.w, fmaf(a.z, a.z, fmaf(a.y, a.y, a.x * a.x)));
    // lanes 4 t .. 4 t + 3 hold the four quads of tile t of this pass
    s1 += dpp_mov<CCX_DPP_QUAD_XOR1>(s1); s2 += dpp_mov<CCX_DPP_QUAD_XOR1>(s2);
    s1 += dpp_mov<CCX_DPP_QUAD_XOR2>(s1); s2 += dpp_mov<CCX_DPP_QUAD_XOR2>(s2);
    if (idx < nv) {
      ((float4*)(x + (long)m * K))[idx] = a;
      uint2 pk;
      pk.x = pack_bf16x2(a.x, a.y); pk.y = pack_bf16x2(a.z, a.w);
      ((uint2*)(xb + (long)m * K))[idx] = pk;
      if ((lane & 3) == 0) st[(long)m * (K >> 4) + (idx >> 2)] = make_float2(s1, s2);
    }
  }
}

int ccx_launch_dec_resolve_stats(ccx_ctx* ctx, float* x, const float* pend, int pend_n, long pend_stride, bf16_t* xb, float2* st, int M,
                                 int K, hipStream_t stream) {
  CCX_REQUIRE(ctx, K % 16 == 0 && K <= 1024 && pend_n >= 0 && pend_n <= 4, "dec_resolve_stats: K=%d / %d slabs unsupported", K, pend_n);
  ccx_prof_scope ps(ctx, stream, "dec_resolve_stats_kernel", 0.0, (double)M * K * (4.0 * (2 + pend_n) + 2.0));
  hipLaunchKernelGGL(dec_resolve_stats_kernel, dim3(ccx_cdiv(M, 4)), dim3(256), 0, stream, x, pend, pend_n, pend_stride, xb, st, M, K);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

// Stand-alone split-KV combine -> bf16 attention output [M][H*64] (used instead of the in-GEMV prologue when
// many sequences are decoded: every weight-panel block would otherwise redo the whole combine).
__global__ __launch_bounds__(256) void dec_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                          int nsplit, bf16_t* __restrict__ out, int M, int H) {
  const int e = blockIdx.x * 256 + threadIdx.x;   // (row, head, 8-wide d chunk)
  if (e >= M * H * 8) return;
  const int c = e & 7, h = (e >> 3) % H, m = e / (8 * H);
  const float2* ml = (const float2*)(part_ml + ((long)(m * H + h) * nsplit) * 2);
  const float* ob = part_o + ((long)(m * H + h) * nsplit) * 64 + 8 * c;
  float2 mlv[8];
  float4 oa[8], oc[8];
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const int sc = s < nsplit ? s : 0;
    mlv[s] = ml[sc];
    oa[s] = *(const float4*)(ob + sc * 64);
    oc[s] = *(const float4*)(ob + sc * 64 + 4);
    if (s >= nsplit) mlv[s] = make_float2(-1e30f, 0.f);
  }
  float mx = -1e30f;
#pragma unroll
  for (int s = 0; s < 8; s++) mx = fmaxf(mx, mlv[s].x);
  float den = 0.f, o[8];
#pragma unroll
  for (int j = 0; j < 8; j++) o[j] = 0.f;
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const float w = (s < nsplit) ? __builtin_amdgcn_exp2f(mlv[s].x - mx) : 0.f;
    den += w * mlv[s].y;
    o[0] += w * oa[s].x; o[1] += w * oa[s].y; o[2] += w * oa[s].z; o[3] += w * oa[s].w;
    o[4] += w * oc[s].x; o[5] += w * oc[s].y; o[6] += w * oc[s].z; o[7] += w * oc[s].w;
  }
  const float inv = 1.0f / den;
  uint4 pk;
  pk.x = pack_bf16x2(o[0] * inv, o[1] * inv); pk.y = pack_bf16x2(o[2] * inv, o[3] * inv);
  pk.z = pack_bf16x2(o[4] * inv, o[5] * inv); pk.w = pack_bf16x2(o[6] * inv, o[7] * inv);
  *(uint4*)(out + ((long)m * H + h) * 64 + 8 * c) = pk;
}

__global__ void dec_gather_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx, bf16_t* __restrict__ dst, int D) {
  if (idx[blockIdx.x] < 0) return;                 // row not taken in this pass (chunked prefill): dst keeps what it has
  const uint4* s4 = (const uint4*)(src + (long)idx[blockIdx.x] * D);
  uint4* d4 = (uint4*)(dst + (long)blockIdx.x * D);
  for (int i = threadIdx.x; i < D / 8; i += blockDim.x) d4[i] = s4[i];
}

int ccx_launch_dec_gather_rows(ccx_ctx* ctx, const bf16_t* src, const int* idx, bf16_t* dst, int n, int D, hipStream_t stream) {
  CCX_REQUIRE(ctx, src && idx && dst && n >= 1 && D % 8 == 0, "dec_gather_rows: bad arguments");
  hipLaunchKernelGGL(dec_gather_rows_kernel, dim3(n), dim3(128), 0, stream, src, idx, dst, D);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

int ccx_launch_dec_combine(ccx_ctx* ctx, const float* part_o, const float* part_ml, int nsplit, bf16_t* out, int M, int H,
                           hipStream_t stream) {
  CCX_REQUIRE(ctx, nsplit >= 1 && nsplit <= 8, "dec_combine: nsplit out of range");
  ccx_prof_scope ps(ctx, stream, "dec_combine_kernel", 0.0, (double)M * H * nsplit * 66.0 * 4 + (double)M * H * 64 * 2);
  hipLaunchKernelGGL(dec_combine_kernel, dim3(ccx_cdiv(M * H * 8, 256)), dim3(256), 0, stream, part_o, part_ml, nsplit, out, M, H);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

int ccx_launch_dec_resolve_ln(ccx_ctx* ctx, const float* x, const float* pend, int pend_n, long pend_stride, const float* g,
                              const float* b, bf16_t* out, float* x_out, int M, int K, float eps, hipStream_t stream) {
  CCX_REQUIRE(ctx, K % 4 == 0 && K <= 1024, "dec_resolve_ln: K=%d unsupported", K);
  CCX_REQUIRE(ctx, x_out != x, "dec_resolve_ln: x_out must not alias x");
  // bytes: the residual rows and their pending slabs in, the bf16 rows (and the resolved fp32 rows) out
  ccx_prof_scope ps(ctx, stream, "dec_resolve_ln_kernel", 0.0, (double)M * K * (4.0 * (1 + pend_n) + 2.0 + (x_out ? 4.0 : 0.0)));
  hipLaunchKernelGGL(dec_resolve_ln_kernel, dim3(ccx_cdiv(M, 4)), dim3(256), 0, stream, x, pend, pend_n, pend_stride, g, b,
                     out, x_out, M, K, eps);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

template <int MT, int NT, int KMAX, int ACT, int EPI, int NW = 4>
static int launch_dec_linear_inst(ccx_ctx* ctx, const DecLinearParams& p, int ksplit, hipStream_t stream) {
  const int MROWS = 16 * MT, BN = 16 * NT;
  size_t act_bytes = (ACT == ACT_BF16 || ACT == ACT_BF16_LN) ? 0 : ccx_align((size_t)MROWS * (p.K + 8) * 2, 16);
  size_t red_bytes = (size_t)NW * NT * MT * 64 * 4 * 4;
  size_t lds = act_bytes + red_bytes + (ACT == ACT_BF16_LN ? (size_t)MROWS * 8 : 0);
  CCX_REQUIRE(ctx, lds <= 160 * 1024, "dec_linear: LDS %zu too large", lds);
  CCX_REQUIRE(ctx, ccx_cdiv(ccx_cdiv(p.K / 32, ksplit), NW) <= KMAX, "dec_linear: K=%d / split %d exceeds the prefetch depth %d", p.K, ksplit, KMAX);
  if (ACT == ACT_BF16_LN) CCX_REQUIRE(ctx, p.ln_stats && p.ln_s && p.bias && ksplit == 1 && p.K % 64 == 0 && p.N % 4 == 0, "dec_linear: bad LayerNorm-algebra arguments");
  if (EPI == DEPI_RESOLVE) CCX_REQUIRE(ctx, p.xres && p.xb && p.st_out && ksplit == 1 && p.N % BN == 0, "dec_linear: bad resolve arguments (N=%d, %d columns per block)", p.N, BN);
  static ccx_lds_optin optin;
  if (lds > 64 * 1024) CCX_HIP(ctx, optin.ensure(ctx->device, (const void*)dec_linear_kernel<MT, NT, KMAX, ACT, EPI, NW>));
  dim3 grid(ccx_cdiv(p.N, BN), ccx_cdiv(p.M, MROWS), ksplit);
  {
    // weight-streaming GEMV: algorithmic bytes = the weight matrix once (+ small activations)
    // labelled like rocprofv3 labels the instantiation, so that both rank the same kernels
    static const std::string label = "dec_linear_kernel<" + std::to_string(MT) + ", " + std::to_string(NT) + ", " + std::to_string(KMAX) +
                                     ", " + std::to_string(ACT) + ", " + std::to_string(EPI) + (NW == 4 ? "" : ", " + std::to_string(NW)) + ">";
    ccx_prof_scope ps(ctx, stream, label.c_str(), 2.0 * p.M * (double)p.N * p.K,
                      2.0 * (double)p.N * p.K + 2.0 * p.M * ((double)p.K + p.N));
    hipLaunchKernelGGL((dec_linear_kernel<MT, NT, KMAX, ACT, EPI, NW>), grid, dim3(64 * NW), lds, stream, p);
  }
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

template <int NT, int ACT, int EPI>
static int launch_dec_linear_mt(ccx_ctx* ctx, const DecLinearParams& p, int ksplit, hipStream_t stream) {
  const int M = p.M;
  // prefetch depth = k-steps per wave: 6 covers K = 768 (24 k-steps over 4 waves) without the two clamped duplicate
  // loads per fragment row and the 25 % idle registers of the depth-8 instantiation, which K = 1024 slices need
  const int need = ccx_cdiv(ccx_cdiv(p.K / 32, ksplit), 4);
  if (need <= 6) {
    if (M <= 16) return launch_dec_linear_inst<1, NT, 6, ACT, EPI>(ctx, p, ksplit, stream);
    if (M <= 32) return launch_dec_linear_inst<2, NT, 6, ACT, EPI>(ctx, p, ksplit, stream);
    return launch_dec_linear_inst<4, NT, 6, ACT, EPI>(ctx, p, ksplit, stream);
  }
  if (M <= 16) return launch_dec_linear_inst<1, NT, 8, ACT, EPI>(ctx, p, ksplit, stream);
  if (M <= 32) return launch_dec_linear_inst<2, NT, 8, ACT, EPI>(ctx, p, ksplit, stream);
  return launch_dec_linear_inst<4, NT, 8, ACT, EPI>(ctx, p, ksplit, stream);
}

int ccx_dec_linear_ksplit(int K, int epi) {
  // prefetch depth is 8 k-steps (256 elements) per wave, 4 waves per block
  if (epi == DEPI_RESOLVE) return 1;   // the residual is added in place: one block owns all of K (12 waves beyond K = 1024)
  int ks = ccx_cdiv(K, 1024);
  if (epi != DEPI_PARTIAL) return ks;  // only partial outputs can be split across blocks
  return ks < 1 ? 1 : ks;
}

int ccx_launch_dec_linear(ccx_ctx* ctx, int act, int epi, const DecLinearParams& p, hipStream_t stream) {
  CCX_REQUIRE(ctx, p.M > 0 && p.N > 0 && p.K > 0 && p.K % 32 == 0, "dec_linear: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
  CCX_REQUIRE(ctx, act == ACT_BF16 || p.K <= 1024, "dec_linear: LN/combine activation needs K <= 1024");
  CCX_REQUIRE(ctx, p.pend_n >= 0 && p.pend_n <= 4, "dec_linear: at most 4 pending slabs");
  const int ksplit = ccx_dec_linear_ksplit(p.K, epi);
  CCX_REQUIRE(ctx, epi == DEPI_PARTIAL || ksplit == 1, "dec_linear: K=%d needs a split-K (partial) epilogue", p.K);
  CCX_REQUIRE(ctx, epi != DEPI_PARTIAL || p.pend_stride >= (long)p.M * p.ldo, "dec_linear: pend_stride too small");
  // ---- the LayerNorm-free chain (X-stream path): consumers with the LayerNorm algebra in the epilogue, producers that resolve
  // the residual themselves.  ONE tile geometry rule for every row count (MT by rows, 32-column blocks from 128 rows on): per
  // output the arithmetic is the same in all of them, and the statistics are kept per 16-column tile, so a row's numbers do not
  // depend on its lane
  if (act == ACT_BF16_LN || epi == DEPI_RESOLVE) {
    CCX_REQUIRE(ctx, (act == ACT_BF16_LN) != (epi == DEPI_RESOLVE) && (act == ACT_BF16_LN || act == ACT_BF16), "dec_linear: LayerNorm-free modes do not combine");
    const bool wide_rows_ = p.M >= 128;
    if (epi == DEPI_RESOLVE && p.K > 1024) {
      // K = 3072 without split-K slabs: 12 waves x 8 k-steps, 32 rows x 32 columns per block (170 registers per thread at three waves per SIMD)
      CCX_REQUIRE(ctx, p.K <= 12 * 8 * 32, "dec_linear: K=%d too deep for the 12-wave resolve kernel", p.K);
      if (p.M <= 16) return launch_dec_linear_inst<1, 2, 8, ACT_BF16, DEPI_RESOLVE, 12>(ctx, p, 1, stream);
      return launch_dec_linear_inst<2, 2, 8, ACT_BF16, DEPI_RESOLVE, 12>(ctx, p, 1, stream);
    }
#define CCX_LNFREE(ACT_, EPI_)                                                                                           \
    do {                                                                                                                   \
      if (wide_rows_) return launch_dec_linear_mt<2, ACT_, EPI_>(ctx, p, 1, stream);                                       \
      return launch_dec_linear_mt<1, ACT_, EPI_>(ctx, p, 1, stream);                                                       \
    } while (0)
    if (epi == DEPI_RESOLVE) CCX_LNFREE(ACT_BF16, DEPI_RESOLVE);
    if (epi == DEPI_SELF_QKV) CCX_LNFREE(ACT_BF16_LN, DEPI_SELF_QKV);
    if (epi == DEPI_BF16_GELU) CCX_LNFREE(ACT_BF16_LN, DEPI_BF16_GELU);
    if (epi == DEPI_F32) CCX_LNFREE(ACT_BF16_LN, DEPI_F32);
#undef CCX_LNFREE
    return ccx_fail(ctx, CCX_ERR_ARG, "dec_linear: unsupported LayerNorm-free act=%d epi=%d", act, epi);
  }
  // wide-N layers (logits) use 64-row weight panels per block, narrow ones 16 to spread over more CUs
  const bool wide = p.N >= 8192;
  if (act == ACT_BF16 && epi == DEPI_F32 && wide) return launch_dec_linear_mt<4, ACT_BF16, DEPI_F32>(ctx, p, 1, stream);
  // Lanes of 128 rows and more (the decode groups of the pipelined schedule): a block of 16 output columns re-reads all of its 64
  // activation rows for 24 KB of weights, so the launch is bound by activation reads out of L2 (85 MB for the 1.2 MB QKV matrix
  // at 384 rows); 32 columns per block halve that: pipeline step 678.3 -> 653.8 ms with 2 lanes x 384 rows, 698 -> 683 with
  // 3 x 128 (64 columns per block: 659.9).  CCX_DEC_WIDE_ROWS sets the row count from which it applies (0 = never),
  // CCX_DEC_WIDE_NT=4 selects 64 columns.
  static const int wide_rows = [] { const char* e = getenv("CCX_DEC_WIDE_ROWS"); return e ? atoi(e) : 128; }();
  static const int wide_nt = [] { const char* e = getenv("CCX_DEC_WIDE_NT"); return e ? atoi(e) : 2; }();
  static const int nt4_min_n = [] { const char* e = getenv("CCX_DEC_NT4_MIN_N"); return e ? atoi(e) : 1 << 30; }();
  if (wide_rows > 0 && p.M >= wide_rows && act == ACT_BF16 && (wide_nt == 4 || p.N >= nt4_min_n)) {
    if (epi == DEPI_PARTIAL) return launch_dec_linear_mt<4, ACT_BF16, DEPI_PARTIAL>(ctx, p, ksplit, stream);
    if (epi == DEPI_F32) return launch_dec_linear_mt<4, ACT_BF16, DEPI_F32>(ctx, p, 1, stream);
    if (epi == DEPI_SELF_QKV) return launch_dec_linear_mt<4, ACT_BF16, DEPI_SELF_QKV>(ctx, p, 1, stream);
    if (epi == DEPI_BF16_GELU) return launch_dec_linear_mt<4, ACT_BF16, DEPI_BF16_GELU>(ctx, p, 1, stream);
  }
  if (wide_rows > 0 && p.M >= wide_rows && act == ACT_BF16) {
    if (epi == DEPI_PARTIAL) return launch_dec_linear_mt<2, ACT_BF16, DEPI_PARTIAL>(ctx, p, ksplit, stream);
    if (epi == DEPI_F32) return launch_dec_linear_mt<2, ACT_BF16, DEPI_F32>(ctx, p, 1, stream);
    if (epi == DEPI_SELF_QKV) return launch_dec_linear_mt<2, ACT_BF16, DEPI_SELF_QKV>(ctx, p, 1, stream);
    if (epi == DEPI_BF16_GELU) return launch_dec_linear_mt<2, ACT_BF16, DEPI_BF16_GELU>(ctx, p, 1, stream);
  }
  if (act == ACT_LN && epi == DEPI_SELF_QKV) return launch_dec_linear_mt<1, ACT_LN, DEPI_SELF_QKV>(ctx, p, 1, stream);
  if (act == ACT_LN && epi == DEPI_F32) return launch_dec_linear_mt<1, ACT_LN, DEPI_F32>(ctx, p, 1, stream);
  if (act == ACT_LN && epi == DEPI_BF16_GELU) return launch_dec_linear_mt<1, ACT_LN, DEPI_BF16_GELU>(ctx, p, 1, stream);
  if (act == ACT_COMBINE && epi == DEPI_PARTIAL) return launch_dec_linear_mt<1, ACT_COMBINE, DEPI_PARTIAL>(ctx, p, ksplit, stream);
  if (act == ACT_BF16 && epi == DEPI_PARTIAL) return launch_dec_linear_mt<1, ACT_BF16, DEPI_PARTIAL>(ctx, p, ksplit, stream);
  if (act == ACT_BF16 && epi == DEPI_F32) return launch_dec_linear_mt<1, ACT_BF16, DEPI_F32>(ctx, p, 1, stream);
  if (act == ACT_BF16 && epi == DEPI_SELF_QKV) return launch_dec_linear_mt<1, ACT_BF16, DEPI_SELF_QKV>(ctx, p, 1, stream);
  if (act == ACT_BF16 && epi == DEPI_BF16_GELU) return launch_dec_linear_mt<1, ACT_BF16, DEPI_BF16_GELU>(ctx, p, 1, stream);
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

// A wave instruction covers 8 keys x 128 B (lane group g = key, lane&7 = 16-byte d chunk).  A wave
// owns 64-key chunks and issues all 16 K/V loads of a chunk before touching the data, so one
// HBM round trip covers the chunk.
template <bool FINAL>
__global__ __launch_bounds__(256) void dec_attention_kernel(DecAttnParams p) {
  __shared__ float sm_m[4][8], sm_l[4][8], sm_o[4][8][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, split = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;        // b: row (q / out); its K/V belong to sequence sq
  const int sq = p.row_seq ? p.row_seq[b] : b;
  const int g = lane >> 3, c = lane & 7;
  const int T = p.pos ? (p.pos[b] + 1) : p.T;
  const int per = (T + gridDim.y - 1) / gridDim.y;
  const int kbeg = split * per;
  const int kend = (kbeg + per < T) ? kbeg + per : T;

  const bf16_t* Kb = p.k + ((long)sq * p.H + h) * p.kv_T * 64 + 8 * c;
  const bf16_t* Vb = p.v + ((long)sq * p.H + h) * p.kv_T * 64 + 8 * c;

  float sm = -1e30f, sl = 0.f, so[8];
#pragma unroll
  for (int j = 0; j < 8; j++) so[j] = 0.f;
  float q[8];
  {
    const float4* qp = (const float4*)(p.q + ((long)b * p.H + h) * 64 + 8 * c);
    const float4 a = qp[0], d = qp[1];
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = d.x; q[5] = d.y; q[6] = d.z; q[7] = d.w;
  }

  for (int base = kbeg + wave * 64; base < kend; base += 256) {
    bf16x8 kf[8], vf[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
      int key = base + it * 8 + g;
      key = key < kend ? key : kend - 1;
      if (FINAL) {   // self attention: the cache rows were written moments ago, keep them cacheable
        kf[it] = *(const bf16x8*)(Kb + (long)key * 64);
        vf[it] = *(const bf16x8*)(Vb + (long)key * 64);
      } else {       // cross attention: 0.9 GB per layer and step, read exactly once -> non-temporal
        kf[it] = __builtin_nontemporal_load((const bf16x8*)(Kb + (long)key * 64));
        vf[it] = __builtin_nontemporal_load((const bf16x8*)(Vb + (long)key * 64));
      }
    }
    float s[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) a = fmaf(q[j], bf16_to_f32((bf16_t)kf[it][j]), a);
      a = group8_sum(a) * p.scale_log2e;
      s[it] = (base + it * 8 + g < kend) ? a : -INFINITY;
    }
    float mn = sm;
#pragma unroll
    for (int it = 0; it < 8; it++) mn = fmaxf(mn, s[it]);
    const float al = __builtin_amdgcn_exp2f(sm - mn);
    sl *= al;
#pragma unroll
    for (int j = 0; j < 8; j++) so[j] *= al;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const float pe = __builtin_amdgcn_exp2f(s[it] - mn);
      sl += pe;
#pragma unroll
      for (int j = 0; j < 8; j++) so[j] = fmaf(pe, bf16_to_f32((bf16_t)vf[it][j]), so[j]);
    }
    sm = mn;
  }
  // merge the 8 lane groups (lanes with equal c) in registers: xor 8 via DPP row_ror:8, xor 16 /
  // xor 32 via v_permlane16_swap / v_permlane32_swap (VALU speed, no LDS round trips)
  {
    auto merge_with = [&](float om, float ol, const float (&oo)[8]) {
      const float mx = fmaxf(sm, om);
      const float wa = __builtin_amdgcn_exp2f(sm - mx), wb = __builtin_amdgcn_exp2f(om - mx);
      sl = sl * wa + ol * wb;
#pragma unroll
      for (int j = 0; j < 8; j++) so[j] = so[j] * wa + oo[j] * wb;
      sm = mx;
    };
    float om, ol, oo[8];
    om = dpp_mov<0x128>(sm); ol = dpp_mov<0x128>(sl);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = dpp_mov<0x128>(so[j]);
    merge_with(om, ol, oo);
    om = lane_xor16(sm); ol = lane_xor16(sl);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = lane_xor16(so[j]);
    merge_with(om, ol, oo);
    om = lane_xor32(sm); ol = lane_xor32(sl);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = lane_xor32(so[j]);
    merge_with(om, ol, oo);
  }
  if (g == 0) {
    sm_m[wave][c] = sm; sm_l[wave][c] = sl;
#pragma unroll
    for (int j = 0; j < 8; j++) sm_o[wave][c][j] = so[j];
  }
  __syncthreads();
  if (tid < 64) {
    // thread -> d = tid = 8*cc + j; merge the 4 waves
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
      const long pbase = ((long)b * p.H + h) * gridDim.y + split;
      p.part_o[pbase * 64 + tid] = o;
      if (tid == 0) { p.part_ml[pbase * 2] = mx; p.part_ml[pbase * 2 + 1] = l; }
    }
  }
}

// Small-batch cross attention with the query projection inside (B <= 16 rows: the reference's own calling pattern, one file per task
// and one window per decode, back/api.py:1286-1292).  The decode chain of a small batch is a string of latency-bound launches
// (~5-7 us each against a ~1 us HBM floor); this removes one of the eight per layer.  A block = (row, head, key split) as in
// dec_attention_kernel<false>; before it touches q it
//   1. requests its FIRST 64-key chunk of K and V (they do not depend on q: one HBM round trip now runs under the prologue),
//   2. requests the 4 x 6 weight fragments of its head's 64 query columns (wave w: k-steps 6 w .. 6 w + 5, the slice the same wave
//      of dec_linear_kernel<1, 1, 6, ACT_LN, DEPI_F32> owns),
//   3. normalises its row (every wave for itself into its own LDS row: no barrier) with the shared ln_* pieces,
//   4. runs the same 24 MFMAs, reduces the four waves' partial sums in the same order (w = 0..3) and adds the bias.
// Steps 3-4 repeat dec_linear's operations one for one, so q -- and with it every token and log-probability -- is bit-identical
// to the two-launch path (tests/test_whisper_gpu.py::test_fused_cross_query_equals_two_launches).  K == 768 only (24 k-steps).
// Register budget.  EARLY_V = false: <= 168 VGPRs (three blocks per CU, so that the 576 blocks of 8 rows x 12 heads x 6 splits are all
// resident): the K chunk (32) and two weight groups (64) are in flight across the LayerNorm (64) and the V chunk is requested only when
// the first MFMAs have freed a weight group -- its HBM round trip is then partly exposed.  EARLY_V = true (grids of <= 512 blocks, i.e.
// up to 7 rows: two blocks per CU hold them all): V is requested right behind K, 32 registers more.
template <bool EARLY_V>
__global__ __launch_bounds__(256, EARLY_V ? 2 : 3) void dec_cross_fused_q_kernel(DecAttnParams p) {
  __shared__ float sm_m[4][8], sm_l[4][8], sm_o[4][8][8];
  __shared__ __attribute__((aligned(16))) bf16_t ln_row[4][776];
  __shared__ __attribute__((aligned(16))) float red[4][4][4][4];     // [wave][n-tile][lane >> 4][4 outputs]: column 0 of every tile
  __shared__ __attribute__((aligned(16))) float q_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, split = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int g = lane >> 3, c = lane & 7;
  const int T = p.T;
  const int per = (T + gridDim.y - 1) / gridDim.y;
  const int kbeg = split * per;
  const int kend = (kbeg + per < T) ? kbeg + per : T;
  const int K = p.q_K;                       // 768
  // block-uniform bases + 32-bit byte offsets per lane (one SGPR pair + one VGPR per address instead of a 64-bit VGPR pair)
  const char* Ku = (const char*)(p.k + ((long)b * p.H + h) * p.kv_T * 64);
  const char* Vu = (const char*)(p.v + ((long)b * p.H + h) * p.kv_T * 64);
  auto kv_off = [&](int key) -> unsigned { return (unsigned)(key * 64 + 8 * c) * 2u; };

  // vmcnt retires IN ORDER, so whatever the LayerNorm waits for must be requested BEFORE the long HBM round trip of the K chunk:
  // ---- 1. the row, its pending split-K slabs and the LayerNorm's affine parameters (L2 hits) ----
  const int nv = K >> 2;
  const long row = (long)b * K;
  float4 v[4], pq[4], gg[4], bb[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int idx = lane + 64 * i;
    const int ic = idx < nv ? idx : 0;
    const unsigned o16 = (unsigned)ic * 16u;
    v[i] = *(const float4*)((const char*)(p.qx + row) + o16);
    // (dec_linear adds all four slab slots, the unused ones with weight 0: fmaf(0, finite, a) == a, so skipping them is bit-identical;
    //  two or more pending slabs take the generic loop below)
    pq[i] = *(const float4*)((const char*)(p.q_pend + row) + o16);
    gg[i] = *(const float4*)((const char*)p.q_ln_g + o16);
    bb[i] = *(const float4*)((const char*)p.q_ln_b + o16);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 2. first K chunk of this wave (clamped like the loop below: a wave without keys loads row kend - 1 and masks it) ----
  bf16x8 kf[8], vf[8];
  {
    const int base = kbeg + wave * 64;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      int key = base + it * 8 + g;
      key = key < kend ? key : kend - 1;
      kf[it] = __builtin_nontemporal_load((const bf16x8*)(Ku + kv_off(key)));
      if (EARLY_V) vf[it] = __builtin_nontemporal_load((const bf16x8*)(Vu + kv_off(key)));
    }
  }
  // ---- 3. weight fragments: n-tiles 4 h .. 4 h + 3, k-steps 6 wave .. 6 wave + 5 (packed image: tile (n / 16, k / 32) = 1 KB), in three
  // groups of two k-steps (32 registers each, at most two groups in flight).  Default cache policy: the 48 blocks of a head share them.
  const int ksteps = K >> 5;                 // 24
  const int ks0 = wave * 6;
  const char* wu = (const char*)(p.q_W + ((long)(4 * h) * ksteps + ks0) * 512);       // block- and wave-uniform
  auto load_group = [&](int gk, bf16x8 (&w)[4][2]) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int k = 0; k < 2; k++) w[i][k] = *(const bf16x8*)(wu + (unsigned)(((i * ksteps + 2 * gk + k) * 64 + lane) * 16));
  };
  bf16x8 wa[4][2], wb[4][2];
  load_group(0, wa);
  __builtin_amdgcn_sched_barrier(0);
  // ---- 4. LayerNorm of row b (x + pending slabs), every wave for itself ----
  {
    const bool writer = (h == 0 && split == 0 && wave == 0 && p.q_x_out != nullptr);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = lane + 64 * i;
      const int ic = idx < nv ? idx : 0;
      float4 a = v[i];
      a = ln_add_pend(a, p.q_pend_n > 0 ? 1.f : 0.f, pq[i]);
      for (int s = 1; s < p.q_pend_n; s++)       // (rare: only K > 1024 producers leave more than one slab)
        a = ln_add_pend(a, 1.f, ((const float4*)(p.q_pend + (long)s * p.q_pend_stride + row))[ic]);
      if (idx >= nv) a = make_float4(0.f, 0.f, 0.f, 0.f);
      v[i] = a;
    }
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) sm += ln_sum4(v[i]);
    const float mean = wave_reduce_sum(sm) / (float)K;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float t = ln_sq4(v[i], mean);
      sq += (lane + 64 * i < nv) ? t : 0.f;
    }
    const float rstd = rsqrtf(wave_reduce_sum(sq) / (float)K + p.q_eps);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = lane + 64 * i;
      if (idx < nv) {
        *(uint2*)(&ln_row[wave][4 * idx]) = ln_pack4(v[i], mean, rstd, gg[i], bb[i]);
        if (writer) ((float4*)(p.q_x_out + row))[idx] = v[i];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  load_group(1, wb);
  // the wave's own LDS row: its stores are complete once lgkmcnt is 0, no other wave reads it
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // ---- 5. q = row * Wq^T: the activation fragment is the same row in every MFMA column; k-steps in the order 0..5 ----
  float4 bias4;
  {
    const int h4 = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma_group = [&](int gk, const bf16x8 (&w)[4][2]) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const bf16x8 af = *(const bf16x8*)(&ln_row[wave][8 * h4 + 32 * (ks0 + 2 * gk + k)]);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][k], af, acc[i], 0, 0, 0);
      }
    };
    mma_group(0, wa);
    __builtin_amdgcn_sched_barrier(0);
    load_group(2, wa);
    bias4 = *(const float4*)(p.q_bias + h * 64 + 4 * (tid & 15));     // (no branch around the load: hipcc would wait at it)
    // first V chunk behind the last weight group (its HBM round trip runs under the remaining MFMAs, the reduction and q . k)
    if (!EARLY_V) {
      const int base = kbeg + wave * 64;
#pragma unroll
      for (int it = 0; it < 8; it++) {
        int key = base + it * 8 + g;
        key = key < kend ? key : kend - 1;
        vf[it] = __builtin_nontemporal_load((const bf16x8*)(Vu + kv_off(key)));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_group(1, wb);
    mma_group(2, wa);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) *(f32x4*)(&red[wave][i][h4][0]) = acc[i];
    }
  }
  __syncthreads();
  if (tid < 16) {
    // thread t finishes outputs 4 t .. 4 t + 3 of the head: n-tile t >> 2, rows 4 (t & 3) .. + 3; waves summed in the order 0..3
    const int i = tid >> 2, r4 = tid & 3;
    f32x4 v = *(const f32x4*)(&red[0][i][r4][0]);
#pragma unroll
    for (int w = 1; w < 4; w++) v += *(const f32x4*)(&red[w][i][r4][0]);
    v[0] += bias4.x; v[1] += bias4.y; v[2] += bias4.z; v[3] += bias4.w;
    *(float4*)(&q_s[4 * tid]) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  float q[8];
  {
    const float4 a = *(const float4*)(&q_s[8 * c]), d = *(const float4*)(&q_s[8 * c + 4]);
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = d.x; q[5] = d.y; q[6] = d.z; q[7] = d.w;
  }

  // ---- attention over this block's keys: dec_attention_kernel<false>'s loop, its first chunk already in registers ----
  float sm = -1e30f, sl = 0.f, so[8];
#pragma unroll
  for (int j = 0; j < 8; j++) so[j] = 0.f;
  for (int base = kbeg + wave * 64; base < kend; base += 256) {
    if (base != kbeg + wave * 64) {
#pragma unroll
      for (int it = 0; it < 8; it++) {
        int key = base + it * 8 + g;
        key = key < kend ? key : kend - 1;
        kf[it] = __builtin_nontemporal_load((const bf16x8*)(Ku + kv_off(key)));
        vf[it] = __builtin_nontemporal_load((const bf16x8*)(Vu + kv_off(key)));
      }
    }
    float s[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) a = fmaf(q[j], bf16_to_f32((bf16_t)kf[it][j]), a);
      a = group8_sum(a) * p.scale_log2e;
      s[it] = (base + it * 8 + g < kend) ? a : -INFINITY;
    }
    float mn = sm;
#pragma unroll
    for (int it = 0; it < 8; it++) mn = fmaxf(mn, s[it]);
    const float al = __builtin_amdgcn_exp2f(sm - mn);
    sl *= al;
#pragma unroll
    for (int j = 0; j < 8; j++) so[j] *= al;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const float pe = __builtin_amdgcn_exp2f(s[it] - mn);
      sl += pe;
#pragma unroll
      for (int j = 0; j < 8; j++) so[j] = fmaf(pe, bf16_to_f32((bf16_t)vf[it][j]), so[j]);
    }
    sm = mn;
  }
  {
    auto merge_with = [&](float om, float ol, const float (&oo)[8]) {
      const float mx = fmaxf(sm, om);
      const float wa = __builtin_amdgcn_exp2f(sm - mx), wb = __builtin_amdgcn_exp2f(om - mx);
      sl = sl * wa + ol * wb;
#pragma unroll
      for (int j = 0; j < 8; j++) so[j] = so[j] * wa + oo[j] * wb;
      sm = mx;
    };
    float om, ol, oo[8];
    om = dpp_mov<0x128>(sm); ol = dpp_mov<0x128>(sl);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = dpp_mov<0x128>(so[j]);
    merge_with(om, ol, oo);
    om = lane_xor16(sm); ol = lane_xor16(sl);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = lane_xor16(so[j]);
    merge_with(om, ol, oo);
    om = lane_xor32(sm); ol = lane_xor32(sl);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = lane_xor32(so[j]);
    merge_with(om, ol, oo);
  }
  if (g == 0) {
    sm_m[wave][c] = sm; sm_l[wave][c] = sl;
#pragma unroll
    for (int j = 0; j < 8; j++) sm_o[wave][c][j] = so[j];
  }
  __syncthreads();
  if (tid < 64) {
    const int cc = tid >> 3, j = tid & 7;
    float mx = fmaxf(fmaxf(sm_m[0][cc], sm_m[1][cc]), fmaxf(sm_m[2][cc], sm_m[3][cc]));
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const float ww = __builtin_amdgcn_exp2f(sm_m[w][cc] - mx);
      l += ww * sm_l[w][cc];
      o += ww * sm_o[w][cc][j];
    }
    const long pbase = ((long)b * p.H + h) * gridDim.y + split;
    p.part_o[pbase * 64 + tid] = o;
    if (tid == 0) { p.part_ml[pbase * 2] = mx; p.part_ml[pbase * 2 + 1] = l; }
  }
}

int ccx_launch_dec_cross_fused_q(ccx_ctx* ctx, const DecAttnParams& p, int B, int nsplit, hipStream_t stream) {
  CCX_REQUIRE(ctx, B > 0 && B <= 16 && p.H > 0 && nsplit >= 1 && nsplit <= 8, "dec_cross_fused_q: bad shape");
  CCX_REQUIRE(ctx, p.q_K == 768 && p.H * 64 == p.q_K && !p.pos && !p.row_seq, "dec_cross_fused_q: needs d_model 768 and one row per sequence");
  CCX_REQUIRE(ctx, p.qx && p.q_W && p.q_bias && p.q_ln_g && p.q_ln_b && p.q_pend && p.q_pend_n >= 0 && p.q_pend_n <= 4, "dec_cross_fused_q: missing operands");
  CCX_REQUIRE(ctx, p.q_x_out != p.qx, "dec_cross_fused_q: x_out must not alias x");
  {
    // algorithmic bytes: the K/V of the batch once + the query weights once
    ccx_prof_scope ps(ctx, stream, "dec_cross_fused_q_kernel", 4.0 * B * p.H * (double)p.T * 64 + 2.0 * B * (double)p.q_K * p.q_K,
                      (double)B * p.H * p.T * 64 * 2 * 2 + 2.0 * p.q_K * p.q_K);
    if (B * p.H * nsplit <= 512) hipLaunchKernelGGL(dec_cross_fused_q_kernel<true>, dim3(B * p.H, nsplit), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(dec_cross_fused_q_kernel<false>, dim3(B * p.H, nsplit), dim3(256), 0, stream, p);
  }
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

// Cross attention for decode lanes that overlap other lanes' latency-bound chains ("lean streaming").  The chain kernels of
// the other lanes are slowed mostly by the memory queue of the CU they share with streaming waves (MI355X_MICROARCH.md,
// handoff-1to1: 0.8 us idle, 2.3-2.8 us beside 8 streaming waves, 1.2-1.5 beside 2-5), so this variant keeps FEW waves per CU
// and FEW bytes per wave in flight while still covering HBM latency: a wave owns one contiguous key range and rolls through it
// in 32-key pieces (4 K + 4 V wave loads = 8 KB), the next piece always requested before the current one is reduced
// (two named register sets; hipcc leaves the younger 8 loads in flight: s_waitcnt vmcnt(8)).
// NP = 32-key pieces per wave (compile-time: the loop is fully unrolled into straight-line code, because with a real loop
// hipcc keeps the loop-carried piece in other registers than it loads into and copies it at the loop end behind a vmcnt(0)).
// PRE (prompt prefill): a sequence has rows_per_seq consecutive rows (one per prompt position) that all attend to ITS K/V.  Logical
// block L = (sequence * H + head) * rows_per_seq + t, and the launch order is remapped (the GEMM's XCD-aware bijective map) so
// that the rows of one (sequence, head) run on one XCD at about the same time: the first one brings the 384 KB of K/V into
// that XCD's L2, the others hit it -- the prompt costs about one step of HBM traffic instead of one per prompt token.  Loads are
// cacheable here (non-temporal for the decode steps, where every byte is used once).
template <bool FINAL, int NP, bool PRE = false>
__global__ __launch_bounds__(256) void dec_cross_stream_kernel(DecAttnParams p) {
  __shared__ float sm_m[4][8], sm_l[4][8], sm_o[4][8][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.y;
  int b, h, sq;                                      // row (q / out), head, sequence (K/V)
  if (PRE) {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    const int t = L % p.rows_per_seq, sh = L / p.rows_per_seq;
    h = sh % p.H; sq = sh / p.H;
    b = sq * p.rows_per_seq + t;
  } else {
    const int bh = blockIdx.x;
    b = bh / p.H; h = bh - b * p.H; sq = b;
  }
  const int g = lane >> 3, c = lane & 7;
  const int T = p.T;
  const int per = (T + gridDim.y - 1) / gridDim.y;
  const int kbeg = split * per;
  const int kend = (kbeg + per < T) ? kbeg + per : T;
  // contiguous range of this wave, in multiples of 32 keys
  const int per_w = (((kend - kbeg) + 3) / 4 + 31) & ~31;
  // wave-uniform by construction; readfirstlane makes it provable, so the loop below branches on scalars (s_cbranch_scc)
  // instead of masking lanes, and hipcc can count its loads (vmcnt(8)) instead of draining them
  // (the four waves taking alternating pieces of ONE sweep over the block's keys instead of a contiguous quarter each was
  // measured too: 92.2 against 93.7 us per launch alone, 706.0 against 704.4 ms per pipeline step -- no difference)
  const int w0 = __builtin_amdgcn_readfirstlane(kbeg + wave * per_w);
  const int w1 = __builtin_amdgcn_readfirstlane((w0 + per_w < kend) ? w0 + per_w : kend);
  constexpr int PSTEP = 32;

  const bf16_t* Kb = p.k + ((long)sq * p.H + h) * p.kv_T * 64 + 8 * c;
  const bf16_t* Vb = p.v + ((long)sq * p.H + h) * p.kv_T * 64 + 8 * c;

  float sm = -1e30f, sl = 0.f, so[8];
#pragma unroll
  for (int j = 0; j < 8; j++) so[j] = 0.f;
  float q[8];
  {
    const float4* qp = (const float4*)(p.q + ((long)b * p.H + h) * 64 + 8 * c);
    const float4 a = qp[0], d = qp[1];
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = d.x; q[5] = d.y; q[6] = d.z; q[7] = d.w;
  }
  auto load = [&](bf16x8 (&kf)[4], bf16x8 (&vf)[4], int base) {
    long off[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
      int key = base + it * 8 + g;
      key = key < w1 ? key : kend - 1;    // clamped rows (one cache line) are masked below
      off[it] = (long)key * 64;
    }
#pragma unroll
    for (int it = 0; it < 4; it++)   // K first: the scores need it first
      kf[it] = PRE ? *(const bf16x8*)(Kb + off[it]) : __builtin_nontemporal_load((const bf16x8*)(Kb + off[it]));
#pragma unroll
    for (int it = 0; it < 4; it++) vf[it] = PRE ? *(const bf16x8*)(Vb + off[it]) : __builtin_nontemporal_load((const bf16x8*)(Vb + off[it]));
  };
  auto reduce = [&](const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], int base) {
    float s[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) a = fmaf(q[j], bf16_to_f32((bf16_t)kf[it][j]), a);
      a = group8_sum(a) * p.scale_log2e;
      s[it] = (base + it * 8 + g < w1) ? a : -INFINITY;
    }
    float mn = sm;
#pragma unroll
    for (int it = 0; it < 4; it++) mn = fmaxf(mn, s[it]);
    const float al = __builtin_amdgcn_exp2f(sm - mn);
    sl *= al;
#pragma unroll
    for (int j = 0; j < 8; j++) so[j] *= al;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const float pe = __builtin_amdgcn_exp2f(s[it] - mn);
      sl += pe;
#pragma unroll
      for (int j = 0; j < 8; j++) so[j] = fmaf(pe, bf16_to_f32((bf16_t)vf[it][j]), so[j]);
    }
    sm = mn;
  };
  {
    // no branch around a request or a reduce: one code path, exact load counts.  Pieces past the end of the wave's range
    // re-read its last key row (one cache line) and are masked to -inf scores.  sched_barrier pins the order
    // request(i+1) -> reduce(i): at most two pieces (16 KB) and at least one (8 KB) in flight per wave.
    bf16x8 kA[4], vA[4], kB[4], vB[4];
    load(kA, vA, w0);
    // q is consumed once HERE: behind the first piece's requests, so that its round trip and theirs overlap (the wait is a
    // counted vmcnt(8): q is older than the piece), and before the loop (a wait for q inside it would sit on every iteration)
    __builtin_amdgcn_sched_barrier(0);
    {
      float qs = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) qs += q[j];
      asm volatile("" ::"v"(qs));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; i += 2) {
      const int base = w0 + PSTEP * i;
      if (i + 1 < NP) load(kB, vB, base + PSTEP);
      __builtin_amdgcn_sched_barrier(0);
      reduce(kA, vA, base);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 2 < NP) load(kA, vA, base + 2 * PSTEP);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < NP) reduce(kB, vB, base + PSTEP);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {
    auto merge_with = [&](float om, float ol, const float (&oo)[8]) {
      const float mx = fmaxf(sm, om);
      const float wa = __builtin_amdgcn_exp2f(sm - mx), wb = __builtin_amdgcn_exp2f(om - mx);
      sl = sl * wa + ol * wb;
#pragma unroll
      for (int j = 0; j < 8; j++) so[j] = so[j] * wa + oo[j] * wb;
      sm = mx;
    };
    float om, ol, oo[8];
    om = dpp_mov<0x128>(sm); ol = dpp_mov<0x128>(sl);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = dpp_mov<0x128>(so[j]);
    merge_with(om, ol, oo);
    om = lane_xor16(sm); ol = lane_xor16(sl);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = lane_xor16(so[j]);
    merge_with(om, ol, oo);
    om = lane_xor32(sm); ol = lane_xor32(sl);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = lane_xor32(so[j]);
    merge_with(om, ol, oo);
  }
  if (g == 0) {
    sm_m[wave][c] = sm; sm_l[wave][c] = sl;
#pragma unroll
    for (int j = 0; j < 8; j++) sm_o[wave][c][j] = so[j];
  }
  __syncthreads();
  if (tid < 64) {
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
      const long pbase = ((long)b * p.H + h) * gridDim.y + split;
      p.part_o[pbase * 64 + tid] = o;
      if (tid == 0) { p.part_ml[pbase * 2] = mx; p.part_ml[pbase * 2 + 1] = l; }
    }
  }
}

// Prompt prefill of the cross attention, RB prompt rows per block.  A sequence's prompt rows all attend to ITS K/V: with one block
// per (sequence, head, row) (dec_cross_stream_kernel<.., PRE>) the 384 KB of a head came out of L2 once per row (35 GB per
// 768-sequence prefill at 10 rows, ~2 TB/s effective).  Here a block streams the K/V of one (sequence, head) ONCE per RB rows:
// every 32-key piece is reduced against RB queries.  Same wave / piece structure as the streaming kernel (a wave owns a contiguous
// quarter of the keys, the next piece is requested before the current one is reduced), cacheable loads, XCD-aware block order.
template <int NP, int RB>
__global__ __launch_bounds__(256) void dec_cross_prefill_kernel(DecAttnParams p) {
  __shared__ float sm_m[RB][4][8], sm_l[RB][4][8], sm_o[RB][4][8][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ngrp = (p.rows_per_seq + RB - 1) / RB;                  // row groups per (sequence, head)
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
  const int L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
  const int tg = L % ngrp, sh = L / ngrp;
  const int h = sh % p.H, sq = sh / p.H;
  const int t0 = tg * RB;
  const int g = lane >> 3, c = lane & 7;
  const int T = p.T;
  const int per_w = ((T + 3) / 4 + 31) & ~31;
  const int w0 = __builtin_amdgcn_readfirstlane(wave * per_w);
  const int w1 = __builtin_amdgcn_readfirstlane((w0 + per_w < T) ? w0 + per_w : T);
  const bf16_t* Kb = p.k + ((long)sq * p.H + h) * p.kv_T * 64 + 8 * c;
  const bf16_t* Vb = p.v + ((long)sq * p.H + h) * p.kv_T * 64 + 8 * c;

  float q[RB][8], sm[RB], sl[RB], so[RB][8];
#pragma unroll
  for (int r = 0; r < RB; r++) {
    const int t = t0 + r < p.rows_per_seq ? t0 + r : p.rows_per_seq - 1;      // rows past the prompt repeat the last one (not stored)
    const float4* qp = (const float4*)(p.q + ((long)(sq * p.rows_per_seq + t) * p.H + h) * 64 + 8 * c);
    const float4 a = qp[0], d = qp[1];
    q[r][0] = a.x; q[r][1] = a.y; q[r][2] = a.z; q[r][3] = a.w; q[r][4] = d.x; q[r][5] = d.y; q[r][6] = d.z; q[r][7] = d.w;
    sm[r] = -1e30f; sl[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) so[r][j] = 0.f;
  }
  auto load = [&](bf16x8 (&kf)[4], bf16x8 (&vf)[4], int base) {
    long off[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
      int key = base + it * 8 + g;
      key = key < w1 ? key : T - 1;
      off[it] = (long)key * 64;
    }
#pragma unroll
    for (int it = 0; it < 4; it++) kf[it] = *(const bf16x8*)(Kb + off[it]);
#pragma unroll
    for (int it = 0; it < 4; it++) vf[it] = *(const bf16x8*)(Vb + off[it]);
  };
  auto reduce = [&](const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], int base) {
#pragma unroll
    for (int r = 0; r < RB; r++) {
      float s[4];
#pragma unroll
      for (int it = 0; it < 4; it++) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) a = fmaf(q[r][j], bf16_to_f32((bf16_t)kf[it][j]), a);
        a = group8_sum(a) * p.scale_log2e;
        s[it] = (base + it * 8 + g < w1) ? a : -INFINITY;
      }
      float mn = sm[r];
#pragma unroll
      for (int it = 0; it < 4; it++) mn = fmaxf(mn, s[it]);
      const float al = __builtin_amdgcn_exp2f(sm[r] - mn);
      sl[r] *= al;
#pragma unroll
      for (int j = 0; j < 8; j++) so[r][j] *= al;
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const float pe = __builtin_amdgcn_exp2f(s[it] - mn);
        sl[r] += pe;
#pragma unroll
        for (int j = 0; j < 8; j++) so[r][j] = fmaf(pe, bf16_to_f32((bf16_t)vf[it][j]), so[r][j]);
      }
      sm[r] = mn;
    }
  };
  {
    // a real loop here (the decode-step kernel is straight-line code for exact load counts; this one runs once per decode
    // and lives within 256 registers only as a loop): the next piece is requested, the current one reduced against RB rows
    bf16x8 kA[4], vA[4], kB[4], vB[4];
    load(kA, vA, w0);
#pragma unroll 1
    for (int i = 0; i < NP; i++) {
      const int base = w0 + 32 * i;
      load(kB, vB, base + 32 < w1 ? base + 32 : base);
      reduce(kA, vA, base);
#pragma unroll
      for (int it = 0; it < 4; it++) { kA[it] = kB[it]; vA[it] = vB[it]; }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; r++) {
    auto merge_with = [&](float om, float ol, const float (&oo)[8]) {
      const float mx = fmaxf(sm[r], om);
      const float wa = __builtin_amdgcn_exp2f(sm[r] - mx), wb = __builtin_amdgcn_exp2f(om - mx);
      sl[r] = sl[r] * wa + ol * wb;
#pragma unroll
      for (int j = 0; j < 8; j++) so[r][j] = so[r][j] * wa + oo[j] * wb;
      sm[r] = mx;
    };
    float om, ol, oo[8];
    om = dpp_mov<0x128>(sm[r]); ol = dpp_mov<0x128>(sl[r]);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = dpp_mov<0x128>(so[r][j]);
    merge_with(om, ol, oo);
    om = lane_xor16(sm[r]); ol = lane_xor16(sl[r]);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = lane_xor16(so[r][j]);
    merge_with(om, ol, oo);
    om = lane_xor32(sm[r]); ol = lane_xor32(sl[r]);
#pragma unroll
    for (int j = 0; j < 8; j++) oo[j] = lane_xor32(so[r][j]);
    merge_with(om, ol, oo);
    if (g == 0) {
      sm_m[r][wave][c] = sm[r]; sm_l[r][wave][c] = sl[r];
#pragma unroll
      for (int j = 0; j < 8; j++) sm_o[r][wave][c][j] = so[r][j];
    }
  }
  __syncthreads();
  for (int e = tid; e < RB * 64; e += 256) {
    const int r = e >> 6, d = e & 63;
    if (t0 + r >= p.rows_per_seq) continue;
    const int cc = d >> 3, j = d & 7;
    const float mx = fmaxf(fmaxf(sm_m[r][0][cc], sm_m[r][1][cc]), fmaxf(sm_m[r][2][cc], sm_m[r][3][cc]));
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const float ww = __builtin_amdgcn_exp2f(sm_m[r][w][cc] - mx);
      l += ww * sm_l[r][w][cc];
      o += ww * sm_o[r][w][cc][j];
    }
    p.out_bf16[((long)(sq * p.rows_per_seq + t0 + r) * p.H + h) * 64 + d] = f32_to_bf16(o / l);
  }
}

int ccx_launch_dec_attention(ccx_ctx* ctx, const DecAttnParams& p, int B, int nsplit, bool final_out,
                             hipStream_t stream) {
  CCX_REQUIRE(ctx, B > 0 && p.H > 0 && nsplit >= 1, "dec_attention: bad shape");
  CCX_REQUIRE(ctx, !final_out || nsplit == 1, "dec_attention: final output needs nsplit == 1");
  dim3 grid(B * p.H, nsplit);
  {
    // self-attention length varies per row and is known on the device only: priced at one key (q in, one K/V row, out), so that
    // the launch shows up in the per-kernel times without claiming traffic it may not have moved
    const double keys = p.pos ? 1.0 : (double)p.T;
    // pieces per wave: keys per block / 4 waves, rounded up to 32 (the kernel's own formula)
    const int np_need = p.pos ? 0 : ((ccx_cdiv(ccx_cdiv(p.T, nsplit), 4) + 31) / 32);
    static const int rb_env = [] { const char* e = getenv("CCX_PREFILL_ROWS_PER_BLOCK"); return e ? atoi(e) : 4; }();
    // profile label = the symbol that runs (rocprofv3's kernel trace shows the same name), self attention marked as such
    const char* label;
    const int npi = np_need <= 4 ? 0 : (np_need <= 6 ? 1 : 2);
    if (p.pos) label = final_out ? "dec_attention_kernel<true> (self)" : "dec_attention_kernel<false> (self)";
    else if (p.rows_per_seq > 1) {
      static const char* const pf4[3] = {"dec_cross_prefill_kernel<4,4>", "dec_cross_prefill_kernel<6,4>", "dec_cross_prefill_kernel<12,4>"};
      static const char* const pf1[3] = {"dec_cross_stream_kernel<true,4,true>", "dec_cross_stream_kernel<true,6,true>", "dec_cross_stream_kernel<true,12,true>"};
      label = (rb_env == 4 && p.rows_per_seq >= 3) ? pf4[npi] : pf1[npi];
    } else if (p.stream_mode && np_need <= 12) {
      static const char* const st[2][3] = {{"dec_cross_stream_kernel<false,4,false>", "dec_cross_stream_kernel<false,6,false>", "dec_cross_stream_kernel<false,12,false>"},
                                           {"dec_cross_stream_kernel<true,4,false>", "dec_cross_stream_kernel<true,6,false>", "dec_cross_stream_kernel<true,12,false>"}};
      label = st[final_out ? 1 : 0][npi];
    } else label = final_out ? "dec_attention_kernel<true> (cross)" : "dec_attention_kernel<false> (cross)";
    ccx_prof_scope ps(ctx, stream, label,
                      4.0 * B * p.H * keys * 64 * (p.rows_per_seq > 1 && !p.pos ? p.rows_per_seq : 1),
                      (double)B * p.H * keys * 64 * 2 * 2);
    if (p.rows_per_seq > 1 && !p.pos) {
      // prompt prefill: B = sequences here, one block per (sequence, head, prompt row), whole key range per block
      CCX_REQUIRE(ctx, nsplit == 1 && final_out && np_need <= 12, "dec_attention: the prefill cross attention takes the whole key range (T <= 1536)");
      if (rb_env == 4 && p.rows_per_seq >= 3) {
        // four prompt rows per block: the K/V of a (sequence, head) comes out of L2 once per four rows
        dim3 g4(B * p.H * ccx_cdiv(p.rows_per_seq, 4), 1);
        if (np_need <= 4) hipLaunchKernelGGL((dec_cross_prefill_kernel<4, 4>), g4, dim3(256), 0, stream, p);
        else if (np_need <= 6) hipLaunchKernelGGL((dec_cross_prefill_kernel<6, 4>), g4, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((dec_cross_prefill_kernel<12, 4>), g4, dim3(256), 0, stream, p);
      } else {
        dim3 pgrid(B * p.H * p.rows_per_seq, 1);
        if (np_need <= 4) hipLaunchKernelGGL((dec_cross_stream_kernel<true, 4, true>), pgrid, dim3(256), 0, stream, p);
        else if (np_need <= 6) hipLaunchKernelGGL((dec_cross_stream_kernel<true, 6, true>), pgrid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((dec_cross_stream_kernel<true, 12, true>), pgrid, dim3(256), 0, stream, p);
      }
    } else if (p.stream_mode && !p.pos && np_need <= 12) {
      const int pad = p.lds_pad > 0 ? (p.lds_pad < 128 * 1024 ? p.lds_pad : 128 * 1024) : 0;
#define CCX_CROSS_STREAM_LAUNCH(F, ...)                                                                                      \
  do {                                                                                                                       \
    static ccx_lds_optin optin_;                                                                                             \
    CCX_HIP(ctx, optin_.ensure(ctx->device, (const void*)dec_cross_stream_kernel<F, __VA_ARGS__>, 128 * 1024));              \
    hipLaunchKernelGGL((dec_cross_stream_kernel<F, __VA_ARGS__>), grid, dim3(256), pad, stream, p);                         \
  } while (0)
      if (final_out) {
        if (np_need <= 4) CCX_CROSS_STREAM_LAUNCH(true, 4); else if (np_need <= 6) CCX_CROSS_STREAM_LAUNCH(true, 6); else CCX_CROSS_STREAM_LAUNCH(true, 12);
      } else {
        if (np_need <= 4) CCX_CROSS_STREAM_LAUNCH(false, 4); else if (np_need <= 6) CCX_CROSS_STREAM_LAUNCH(false, 6); else CCX_CROSS_STREAM_LAUNCH(false, 12);
      }
#undef CCX_CROSS_STREAM_LAUNCH
    } else if (final_out) hipLaunchKernelGGL(dec_attention_kernel<true>, grid, dim3(256), 0, stream, p);
    else {
      static ccx_lds_optin optin;
      if (p.lds_pad > 0) CCX_HIP(ctx, optin.ensure(ctx->device, (const void*)dec_attention_kernel<false>, 128 * 1024));   // static + dynamic LDS beyond 64 KB
      hipLaunchKernelGGL(dec_attention_kernel<false>, grid, dim3(256), p.lds_pad > 0 ? (p.lds_pad < 128 * 1024 ? p.lds_pad : 128 * 1024) : 0, stream, p);
    }
  }
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

// One block (1024 threads) per sequence.  The whole logit row (<= 53248 values) is loaded ONCE
// into registers (13 float4 per thread, all loads issued before any use) together with the
// suppress mask; every reduction then runs out of registers.  The kernel also writes the input
// embedding of the NEXT step (token + position), so the step chain needs no separate embed launch.
// Philox4x32-10 (Salmon et al., SC'11): counter-based, so every (seed, sequence, step, vocabulary quad) has its own
// four 32-bit draws regardless of launch geometry.  oracle/whisper_ref.py::philox4x32 is the same function in numpy.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * ctr.x, p1 = (unsigned long long)0xCD9E8D57u * ctr.z;
    ctr = make_uint4((unsigned)(p1 >> 32) ^ ctr.y ^ key.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ ctr.w ^ key.y, (unsigned)p0);
    key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
  }
  return ctr;
}
// standard Gumbel noise from a 32-bit draw: u = (x + 0.5) / 2^32 in (0, 1), g = -log(-log(u))
__device__ __forceinline__ float gumbel_from_u32(unsigned x) {
  const float u = ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);   // 24 bits: exact in fp32, never 0 or 1
  return -logf(-logf(u));
}

#define SEL_V4 13
// SAMPLE = false is the greedy kernel (no noise code, no extra registers: the sampling branch costs the 1024-thread
// block its register budget and spills); SAMPLE = true adds the temperature > 0 draw.
template <bool SAMPLE>
__global__ __launch_bounds__(1024) void dec_select_kernel(DecSelectParams p) {
  __shared__ float sh[16];
  __shared__ float sh_v[16];
  __shared__ int sh_i[16];
  __shared__ int sh_next[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int V = p.n_vocab;
  const float* lg = p.logits + (long)b * p.ld_logits;
  DecSeqState s = p.state[b];  // every thread reads the same struct ...
  __syncthreads();             // ... before thread 0 may overwrite it below

  auto embed_next = [&](int tok, int pos) {
    // x[b] = tok_emb[tok] + pos_emb[pos]  (TextDecoder.forward input of the next step)
    const float4* te = (const float4*)(p.tok_emb + (long)tok * p.D);
    const float4* pe = (const float4*)(p.pos_emb + (long)pos * p.D);
    float4* xo = (float4*)(p.x + (long)b * p.D);
    for (int i = tid; i < p.D / 4; i += blockDim.x) {
      const float4 a = te[i], c = pe[i];
      xo[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    }
  };

  // ---- prompt phase: feed the next prompt token, nothing is sampled ----
  if (s.pos < s.prompt_len - 1) {
    const int np = s.pos + 1;
    const int tok = p.prompt[(long)b * p.max_prompt + np];
    embed_next(tok, np);
    if (tid == 0) {
      s.pos = np;
      p.cur_tok[b] = tok;
      p.pos[b] = np;
      p.state[b] = s;
    }
    return;
  }
  if (s.done) {  // finished rows keep emitting eot; the model input stays as it is
    if (tid == 0 && s.n_gen < p.sample_len) {
      p.gen[(long)b * p.sample_len + s.n_gen] = p.eot;
      s.n_gen += 1;
      p.state[b] = s;
    }
    return;
  }

  // ---- load the row + mask once ----
  float val[SEL_V4 * 4];
  unsigned long long sup = 0;  // bit i: element i of this thread is suppressed / out of range
#pragma unroll
  for (int i = 0; i < SEL_V4; i++) {
    const int v0 = tid * 4 + i * 4096;
    if (v0 + 3 < V) {
      const float4 f = *(const float4*)(lg + v0);
      const uchar4 m = *(const uchar4*)(p.suppress_mask + v0);
      val[4 * i] = f.x; val[4 * i + 1] = f.y; val[4 * i + 2] = f.z; val[4 * i + 3] = f.w;
      sup |= ((unsigned long long)((m.x ? 1 : 0) | (m.y ? 2 : 0) | (m.z ? 4 : 0) | (m.w ? 8 : 0))) << (4 * i);
    } else {  // n_vocab % 4 == 0 (checked on the host): a float4 is either fully inside or fully outside
      val[4 * i] = val[4 * i + 1] = val[4 * i + 2] = val[4 * i + 3] = -INFINITY;
      sup |= 0xFull << (4 * i);
    }
  }
  const int i_gen = s.n_gen;
  const int tsb = p.timestamp_begin;

  // ---- no-speech probability from the raw logits at the SOT position (first sampling step) ----
  if (i_gen == 0) {
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SEL_V4 * 4; i++) mx = fmaxf(mx, val[i]);
    mx = block_reduce_max(mx, sh);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SEL_V4 * 4; i++) sum += __builtin_amdgcn_exp2f((val[i] - mx) * 1.4426950408889634f);  // -inf -> 0
    sum = block_reduce_sum(sum, sh);
    if (tid == 0) s.no_speech_prob = expf(lg[p.no_speech] - mx) / sum;
  }

  // ---- timestamp-rule state (ApplyTimestampRules) folded into two allowed id ranges ----
  // text ids   [t_lo, tsb):  empty on the first step (must start with a timestamp); only >= eot after
  //                          "text, timestamp" (a timestamp must be paired or followed by eot)
  // timestamps [s_lo, s_hi): >= the last timestamp (+1 unless it is still unpaired); empty after a
  //                          closed pair "timestamp, timestamp"; <= max_initial_timestamp on the first step
  // SuppressBlank only matters on the first step, where every text id is banned anyway.
  const bool last_ts = i_gen >= 1 && s.last_tok >= tsb;
  const bool pen_ts = i_gen < 2 || s.pen_tok >= tsb;
  int s_lo = tsb;
  if (s.last_ts_tok >= 0) s_lo = (last_ts && !pen_ts) ? s.last_ts_tok : s.last_ts_tok + 1;
  int s_hi = V;
  if (i_gen == 0 && p.max_initial_ts >= 0) s_hi = tsb + p.max_initial_ts + 1;
  if (last_ts && pen_ts) s_hi = s_lo;
  const int t_lo = (i_gen == 0) ? tsb : ((last_ts && !pen_ts) ? p.eot : 0);

  // pass 1 (registers): apply the filters in place (-inf) and take text / timestamp maxima
  float mx_text = -INFINITY, mx_ts = -INFINITY;
  int am_text = 0x7fffffff, am_ts = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < SEL_V4 * 4; i++) {
    const int v = tid * 4 + (i >> 2) * 4096 + (i & 3);
    const bool is_text = v < tsb;
    const bool allowed = !((sup >> i) & 1) && (is_text ? (v >= t_lo) : (v >= s_lo && v < s_hi));
    const float x = allowed ? val[i] : -INFINITY;
    val[i] = x;
    const bool bt_ = is_text && x > mx_text, bs_ = !is_text && x > mx_ts;
    mx_text = bt_ ? x : mx_text; am_text = bt_ ? v : am_text;
    mx_ts = bs_ ? x : mx_ts; am_ts = bs_ ? v : am_ts;
  }
  // block argmax (ties -> lowest index, as torch.argmax on CPU)
  auto block_argmax = [&](float v_, int idx, float& oval, int& oidx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v_, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (ov > v_ || (ov == v_ && oi < idx)) { v_ = ov; idx = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { sh_v[tid >> 6] = v_; sh_i[tid >> 6] = idx; }
    __syncthreads();
    oval = sh_v[0]; oidx = sh_i[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++)
      if (sh_v[w] > oval || (sh_v[w] == oval && sh_i[w] < oidx)) { oval = sh_v[w]; oidx = sh_i[w]; }
  };
  float bt, bs; int it, is;
  block_argmax(mx_text, am_text, bt, it);
  block_argmax(mx_ts, am_ts, bs, is);
  const float mx_all = fmaxf(bt, bs);

  // pass 2 (registers): sum exp over text and over timestamps (relative to mx_all)
  float se_text = 0.f, se_ts = 0.f;
#pragma unroll
  for (int i = 0; i < SEL_V4 * 4; i++) {
    const int v = tid * 4 + (i >> 2) * 4096 + (i & 3);
    const float e = __builtin_amdgcn_exp2f((val[i] - mx_all) * 1.4426950408889634f);  // banned entries are -inf -> 0
    se_text += (v < tsb) ? e : 0.f;
    se_ts += (v < tsb) ? 0.f : e;
  }
  se_text = block_reduce_sum(se_text, sh);
  se_ts = block_reduce_sum(se_ts, sh);

  // timestamp_logprob > max_text_token_logprob  <=>  lse_ts > max_text (same normaliser)
  const float lse_ts = (se_ts > 0.f) ? mx_all + logf(se_ts) : -INFINITY;
  const bool force_ts = lse_ts > bt;
  const float lse = force_ts ? lse_ts : mx_all + logf(se_text + se_ts);   // log-normaliser of the re-filtered logits

  // ---- temperature > 0: Categorical(logits / T) by the Gumbel-max trick (argmax of logits / T + Gumbel noise) ----
  const float temperature = SAMPLE ? __uint_as_float(p.sample_cfg[0]) : 0.f;
  int samp = -1; float samp_logit = 0.f;
  if (SAMPLE && temperature > 0.f) {
    const uint2 key = make_uint2(p.sample_cfg[1], p.sample_cfg[2]);
    const float inv_t = 1.0f / temperature;
    float best = -INFINITY, best_logit = -INFINITY; int best_i = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < SEL_V4; i++) {
      const int v0 = tid * 4 + i * 4096;
      const uint4 r = philox4x32_10(make_uint4((unsigned)(v0 >> 2), (unsigned)(p.row0 + b), (unsigned)i_gen, 0u), key);
      const unsigned rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int v = v0 + j;
        const float x = (force_ts && v < tsb) ? -INFINITY : val[4 * i + j];
        const float sc = x * inv_t + gumbel_from_u32(rr[j]);      // -inf stays -inf
        if (sc > best) { best = sc; best_i = v; best_logit = x; }
      }
    }
    float ov; int oi;
    block_argmax(best, best_i, ov, oi);
    samp = oi;
    __syncthreads();
    if (best_i == samp && best == ov) sh[0] = best_logit;         // exactly one thread owns the winner
    __syncthreads();
    samp_logit = sh[0];
  }

  if (tid == 0) {
    int next; float logprob;
    if (samp >= 0) {
      next = samp;
      logprob = samp_logit - lse;
    } else if (force_ts) {
      next = is;
      logprob = bs - lse;   // log_softmax over the re-filtered logits (text banned)
    } else {
      if (bt > bs || (bt == bs && it < is)) next = it; else next = is;
      logprob = fmaxf(bt, bs) - lse;
    }
    s.sum_logprob += logprob;
    p.gen[(long)b * p.sample_len + i_gen] = next;
    s.n_gen = i_gen + 1;
    s.pen_tok = s.last_tok;
    s.last_tok = next;
    if (next >= tsb) s.last_ts_tok = next;
    int cont = 0;
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
      cont = 1;
    }
    p.state[b] = s;
    sh_next[0] = cont ? next : -1;
    sh_next[1] = s.pos;
  }
  __syncthreads();
  if (sh_next[0] >= 0) embed_next(sh_next[0], sh_next[1]);
}

int ccx_launch_dec_embed(ccx_ctx* ctx, const float* tok_emb, const float* pos_emb, const int* cur_tok, const int* pos,
                         float* x, int B, int D, hipStream_t stream) {
  hipLaunchKernelGGL(dec_embed_kernel, dim3(B), dim3(192), 0, stream, tok_emb, pos_emb, cur_tok, pos, x, D);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

int ccx_launch_dec_select(ccx_ctx* ctx, const DecSelectParams& p, int B, hipStream_t stream) {
  ccx_prof_scope ps(ctx, stream, "dec_select_kernel", 0.0, (double)B * p.n_vocab * 5.0);     // the logit row + its suppress mask
  if (p.sample) hipLaunchKernelGGL(dec_select_kernel<true>, dim3(B), dim3(1024), 0, stream, p);
  else hipLaunchKernelGGL(dec_select_kernel<false>, dim3(B), dim3(1024), 0, stream, p);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}
