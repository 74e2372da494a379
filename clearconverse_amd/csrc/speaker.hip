// speaker.hip -- pyannote-style speaker networks of libccx (K18-K20 in SURVEY.md section 2a):
//   * SincNet front end (instance-norm'd waveform -> 80 sinc band-pass filters k251 s10 -> |.| ->
//     maxpool3 -> instance norm -> leaky ReLU -> conv k5 -> pool -> IN -> conv k5 -> pool -> IN)
//   * XVectorSincNet embedder (`pyannote/embedding`, reference back/api.py:776-780, called at 869):
//     5 TDNN layers (dilated Conv1d + LeakyReLU + BatchNorm) -> mean||std pooling -> Linear(3000,512)
//   * PyanNet segmentation net (inside the VAD / diarization pipelines, reference back/api.py:782-792):
//     SincNet -> 4 x BiLSTM(128) -> 2 x Linear+LeakyReLU -> classifier -> log-softmax / sigmoid
// Semantics follow pyannote.audio 3.x [UPSTREAM-RECALL]; CPU restatement: oracle/pyannote_ref.py.
//
// Batching: a call processes n crops of arbitrary lengths.  Every stage keeps the crops concatenated
// as rows of one channel-last matrix, so each valid (un-padded) convolution is ONE bf16 MFMA GEMM over
// all crops: a k-tap conv over channel-last rows is a GEMM whose A matrix is a strided VIEW of the
// activations (row m = k consecutive frames, lda = channels), dilated taps are accumulated GEMMs.
// Rows whose window crosses a crop end compute garbage that later stages never read.
#include <map>
#include <math.h>
#include "../../include/ccx.h"
#include "ccx_common.h"
#include "gemm_bf16.h"

namespace {

#define SN_K 251
#define SN_STRIDE 10
#define SN_F 80

// per-crop mean / rstd of the raw waveform (InstanceNorm1d(1, affine) -> a*x + c folded into the sinc conv)
__global__ __launch_bounds__(256) void wav_stats_kernel(const float* __restrict__ wav, const long* __restrict__ crop_off,
                                                        const int* __restrict__ crop_len, const float* __restrict__ nw,
                                                        const float* __restrict__ nb, float2* __restrict__ ac) {
  __shared__ float red[8];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* x = wav + crop_off[i];
  const int n = crop_len[i];
  // (crop offsets are sample positions: no 16-byte alignment to build float4 loads on; eight independent 4-byte loads per
  // thread and pass keep enough bytes in flight instead)
  float s = 0.f;
  {
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int t = tid;
    for (; t + 7 * 256 < n; t += 8 * 256) {
#pragma unroll
      for (int u = 0; u < 8; u++) p[u] += x[t + 256 * u];
    }
    for (; t < n; t += 256) p[0] += x[t];
    s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  }
  s = wave_reduce_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)n;
  float q = 0.f;
  {
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int t = tid;
    for (; t + 7 * 256 < n; t += 8 * 256) {
#pragma unroll
      for (int u = 0; u < 8; u++) { const float d = x[t + 256 * u] - mean; p[u] += d * d; }
    }
    for (; t < n; t += 256) { const float d = x[t] - mean; p[0] += d * d; }
    q = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  }
  q = wave_reduce_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  if (tid == 0) {
    const float var = (red[4] + red[5] + red[6] + red[7]) / (float)n;
    const float a = nw[0] * rsqrtf(var + 1e-5f);
    ac[i] = make_float2(a, nb[0] - mean * a);
  }
}

// ---------------------------------------------------------------------------------------------
// Sinc convolution (80 filters x 251 taps, stride 10) + |.| + max-pool(3) on the matrix cores with
// fp32-grade operands: x = xh + xl and w = wh + wl (two bf16 each), out = xh wh + xl wh + xh wl (the
// dropped xl wl term is 2^-18 relative; bf16 products are exact in the fp32 accumulator).
//   * k axis: stride 10 divides the taps into groups of 10 consecutive samples; an MFMA step takes 30 taps
//     = the 30 consecutive samples x[10 p + 30 t ..] of position p (k slots 30, 31 carry zero weights),
//     9 steps cover taps 0..269 (>= 251: zero weights);
//   * rows: M tile r (0..2) holds positions 3 i + r of the wave's 16 pooled frames i, so the max-pool is
//     a max over the three accumulators of a lane;
//   * a wave = 16 pooled frames x all 80 filters (15 accumulators); the filter fragments (9 steps x 5
//     tiles x hi/lo, 90 KB) sit lane-linear in LDS for the life of the (persistent) block, the block's
//     samples are split into hi / lo bf16 arrays once per work item (128 pooled frames).
// The waveform's instance norm (a x + c) is folded as before: conv(a x + c) = a conv(x) + c sum(w).
// ---------------------------------------------------------------------------------------------
#define SM_FRAMES 128
#define SM_STEPS 9
#define SM_NS (SM_FRAMES * 3 * SN_STRIDE + 30 * (SM_STEPS - 1) + 32)
#define SM_BBYTES (SM_STEPS * 5 * 2 * 1024)
#define SM_LDS (SM_BBYTES + 2 * SM_NS * 2)
__global__ __launch_bounds__(512) void sinc_conv_pool_kernel(const float* __restrict__ wav, const long* __restrict__ crop_off,
                                                             const int* __restrict__ crop_len, const int* __restrict__ row_off,
                                                             const int* __restrict__ n_pool, const float2* __restrict__ ac,
                                                             const bf16_t* __restrict__ bfrag,   // [9][5][hi|lo][64 lanes][8]
                                                             const float* __restrict__ filt_sum, // [80]
                                                             float* __restrict__ out, int n_crops, int chunks_per_crop) {
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  bf16_t* Bs = (bf16_t*)sm_raw;
  bf16_t* xh = (bf16_t*)(sm_raw + SM_BBYTES);
  bf16_t* xl = xh + SM_NS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, hq = lane >> 4;
  for (int i = tid; i < SM_BBYTES / 16; i += 512) ((uint4*)Bs)[i] = ((const uint4*)bfrag)[i];
  const int n_items = n_crops * chunks_per_crop;
  const int abase = 3 * SN_STRIDE * (16 * wave + l15) + 8 * hq;    // + 10 r + 30 t  (bf16 elements, even)
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int crop = item / chunks_per_crop, p0 = (item - crop * chunks_per_crop) * SM_FRAMES;
    const int np = n_pool[crop];
    if (p0 >= np) continue;                                         // uniform over the block
    const float* x = wav + crop_off[crop];
    const int n = crop_len[crop], s0 = p0 * 3 * SN_STRIDE;
    __syncthreads();                                                // the previous item's fragment reads are done
    for (int i = tid; i < SM_NS; i += 512) {
      const float v = (s0 + i < n) ? x[s0 + i] : 0.f;
      const bf16_t h = f32_to_bf16(v);
      xh[i] = h;
      xl[i] = f32_to_bf16(v - bf16_to_f32(h));
    }
    __syncthreads();
    f32x4 acc[3][5];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int nt = 0; nt < 5; nt++) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t = 0; t < SM_STEPS; t++) {
      bf16x8 ah[3], al[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const uint32_t* ph = (const uint32_t*)(xh + abase + SN_STRIDE * r + 30 * t);   // 4-byte aligned: ds_read2_b32
        const uint32_t* pl = (const uint32_t*)(xl + abase + SN_STRIDE * r + 30 * t);
        union { bf16x8 v; uint32_t u[4]; } ch, cl;
#pragma unroll
        for (int e = 0; e < 4; e++) { ch.u[e] = ph[e]; cl.u[e] = pl[e]; }
        ah[r] = ch.v; al[r] = cl.v;
      }
#pragma unroll
      for (int nt = 0; nt < 5; nt++) {
        const bf16x8 bh = *(const bf16x8*)(Bs + (((t * 5 + nt) * 2 + 0) * 64 + lane) * 8);
        const bf16x8 bl = *(const bf16x8*)(Bs + (((t * 5 + nt) * 2 + 1) * 64 + lane) * 8);
#pragma unroll
        for (int r = 0; r < 3; r++) {
          acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[r], bh, acc[r][nt], 0, 0, 0);
          acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bl, acc[r][nt], 0, 0, 0);
          acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bh, acc[r][nt], 0, 0, 0);
        }
      }
    }
    const float2 a_c = ac[crop];
    const long orow = (long)row_off[crop] + p0 + 16 * wave + 4 * hq;
#pragma unroll
    for (int nt = 0; nt < 5; nt++) {
      const int f = 16 * nt + l15;
      const float cs = a_c.y * filt_sum[f];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        if (p0 + 16 * wave + 4 * hq + reg < np) {
          const float v0 = fabsf(fmaf(a_c.x, acc[0][nt][reg], cs)), v1 = fabsf(fmaf(a_c.x, acc[1][nt][reg], cs)),
                      v2 = fabsf(fmaf(a_c.x, acc[2][nt][reg], cs));
          out[(orow + reg) * SN_F + f] = fmaxf(v0, fmaxf(v1, v2));
        }
      }
    }
  }
}

// (optional maxpool3 over rows) + InstanceNorm over the crop's frames per channel + LeakyReLU -> bf16.
// in: f32 rows [.., ld_in] at in_off[crop]; out: bf16 [.., ld_out] at out_off[crop], n_out[crop] frames,
// channels >= C written as zeros.  Block = one crop, 256 threads = 64 channels x 4 frame groups.
// (max-pool by POOL) -> InstanceNorm1d(affine) -> LeakyReLU -> bf16, in two fully parallel launches: every crop's time
// axis is cut into INORM_CHUNKS pieces.  Launch 1 leaves (count, mean, M2) of every (crop, piece, channel); launch 2
// combines the pieces of its crop (Chan et al. pairwise update: numerically the two-pass variance) and normalises its
// own piece.  grid (crops, INORM_CHUNKS, channel groups of 64), 256 threads = 64 channels x 4 interleaved frame groups.
#define INORM_CHUNKS 16
// InstanceNorm1d (+ the MaxPool1d(3) in front of it for POOL == 3) + LeakyReLU over ragged crops, two kernels:
//   partial: per (crop, piece of its frames): mean and sum of squared deviations of every channel (two passes over the piece),
//   apply:   combines the 16 pieces of a crop (Chan et al.) and writes the normalised, activated bf16 rows.
// A thread owns FOUR consecutive channels (float4 loads, one 8-byte store) of every 8th frame: 256 threads = 8 frame groups x
// 32 channel quads (20 live for 80 channels, 15 for 60).  [one channel per lane in two 64-channel blocks, the second 3/4 idle,
// read the rows with 4-byte loads: 6.4 ms per pipeline step for 0.6 ms worth of HBM traffic]
template <int POOL>
__device__ __forceinline__ float4 inorm_val4(const float* __restrict__ x, int ld_in, int c, int t) {
  if (POOL == 1) return *(const float4*)(x + (long)t * ld_in + c);
  const float* p = x + (long)(3 * t) * ld_in + c;
  const float4 a = *(const float4*)p, b = *(const float4*)(p + ld_in), d = *(const float4*)(p + 2 * ld_in);
  return make_float4(fmaxf(a.x, fmaxf(b.x, d.x)), fmaxf(a.y, fmaxf(b.y, d.y)), fmaxf(a.z, fmaxf(b.z, d.z)), fmaxf(a.w, fmaxf(b.w, d.w)));
}

template <int POOL>
__global__ __launch_bounds__(256) void inorm_partial_kernel(const float* __restrict__ in, int ld_in, const int* __restrict__ in_off,
                                                            const int* __restrict__ n_out, float* __restrict__ part, int C, int Cpad) {
  __shared__ float4 red[8][32];
  const int crop = blockIdx.x, piece = blockIdx.y;
  const int q = threadIdx.x & 31, c = 4 * q, grp = threadIdx.x >> 5;
  const int n = n_out[crop];
  const int per = (n + INORM_CHUNKS - 1) / INORM_CHUNKS;
  const int t0 = piece * per, t1 = min(n, t0 + per);
  const float* x = in + (long)in_off[crop] * ld_in;
  const bool live = c < C;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) for (int t = t0 + grp; t < t1; t += 8) { const float4 v = inorm_val4<POOL>(x, ld_in, c, t); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  red[grp][q] = s;
  __syncthreads();
  const int cnt = max(t1 - t0, 0);
  float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cnt > 0) {
#pragma unroll
    for (int g = 0; g < 8; g++) { const float4 r = red[g][q]; mean.x += r.x; mean.y += r.y; mean.z += r.z; mean.w += r.w; }
    const float inv = 1.f / (float)cnt;
    mean.x *= inv; mean.y *= inv; mean.z *= inv; mean.w *= inv;
  }
  __syncthreads();
  float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) for (int t = t0 + grp; t < t1; t += 8) {
    const float4 v = inorm_val4<POOL>(x, ld_in, c, t);
    const float a = v.x - mean.x, b = v.y - mean.y, d = v.z - mean.z, e = v.w - mean.w;
    m2.x += a * a; m2.y += b * b; m2.z += d * d; m2.w += e * e;
  }
  red[grp][q] = m2;
  __syncthreads();
  if (grp == 0 && live) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < 8; g++) { const float4 r = red[g][q]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
    float4* o = (float4*)(part + (((long)crop * INORM_CHUNKS + piece) * Cpad + c) * 2);     // (mean, m2) pairs of 4 channels
    o[0] = make_float4(mean.x, t.x, mean.y, t.y);
    o[1] = make_float4(mean.z, t.z, mean.w, t.w);
  }
}

template <int POOL>
__global__ __launch_bounds__(256) void inorm_apply_kernel(const float* __restrict__ in, int ld_in, const int* __restrict__ in_off,
                                                          bf16_t* __restrict__ out, int ld_out, const int* __restrict__ out_off,
                                                          const int* __restrict__ n_out, const float* __restrict__ part,
                                                          const float* __restrict__ g, const float* __restrict__ b, int C, int Cpad) {
  const int crop = blockIdx.x, piece = blockIdx.y;
  const int q = threadIdx.x & 31, c = 4 * q, grp = threadIdx.x >> 5;
  const int n = n_out[crop];
  const int per = (n + INORM_CHUNKS - 1) / INORM_CHUNKS;
  const int t0 = piece * per, t1 = min(n, t0 + per);
  if (t0 >= t1) return;                                  // (block-uniform)
  const bool live = c < C;
  // combine the pieces of this crop ONCE per block (frame group 0), hand the result to the other seven groups through LDS:
  // mean = sum n_i mean_i / n, M2 = sum (M2_i + n_i (mean_i - mean)^2)
  __shared__ float4 s_mean[32], s_sc[32], s_sh[32];
  if (grp == 0) {
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      const float* pp = part + ((long)crop * INORM_CHUNKS * Cpad + c) * 2;
      float4 pa[INORM_CHUNKS], pd[INORM_CHUNKS];
#pragma unroll
      for (int i = 0; i < INORM_CHUNKS; i++) { pa[i] = *(const float4*)(pp + (long)i * Cpad * 2); pd[i] = *(const float4*)(pp + (long)i * Cpad * 2 + 4); }
#pragma unroll
      for (int i = 0; i < INORM_CHUNKS; i++) {
        const float ni = (float)max(min(n, (i + 1) * per) - i * per, 0);
        mean[0] += ni * pa[i].x; mean[1] += ni * pa[i].z; mean[2] += ni * pd[i].x; mean[3] += ni * pd[i].z;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) mean[k] /= (float)n;
#pragma unroll
      for (int i = 0; i < INORM_CHUNKS; i++) {
        const float ni = (float)max(min(n, (i + 1) * per) - i * per, 0);
        const float d0 = pa[i].x - mean[0], d1 = pa[i].z - mean[1], d2 = pd[i].x - mean[2], d3 = pd[i].z - mean[3];
        m2[0] += pa[i].y + ni * d0 * d0; m2[1] += pa[i].w + ni * d1 * d1; m2[2] += pd[i].y + ni * d2 * d2; m2[3] += pd[i].w + ni * d3 * d3;
      }
    }
    float scv[4], shv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float rstd = rsqrtf(m2[k] / (float)n + 1e-5f);
      const float gg = live ? g[c + k] : 0.f, bb = live ? b[c + k] : 0.f;
      scv[k] = rstd * gg; shv[k] = bb;                            // (v - mean) * (rstd * g) + b
    }
    s_mean[q] = make_float4(mean[0], mean[1], mean[2], mean[3]);
    s_sc[q] = make_float4(scv[0], scv[1], scv[2], scv[3]);
    s_sh[q] = make_float4(shv[0], shv[1], shv[2], shv[3]);
  }
  __syncthreads();
  if (c >= ld_out) return;
  const float4 mean4 = s_mean[q], sc4 = s_sc[q], sh4 = s_sh[q];
  const float mean[4] = {mean4.x, mean4.y, mean4.z, mean4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
  const float* x = in + (long)in_off[crop] * ld_in;
  bf16_t* o = out + (long)out_off[crop] * ld_out;
  for (int t = t0 + grp; t < t1; t += 8) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      const float4 r = inorm_val4<POOL>(x, ld_in, c, t);
      v[0] = (r.x - mean[0]) * sc[0] + sh[0]; v[1] = (r.y - mean[1]) * sc[1] + sh[1];
      v[2] = (r.z - mean[2]) * sc[2] + sh[2]; v[3] = (r.w - mean[3]) * sc[3] + sh[3];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = v[k] >= 0.f ? v[k] : 0.01f * v[k];
    }
    uint2 pk;
    pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
    *(uint2*)(o + (long)t * ld_out + c) = pk;
  }
}

// mean || std over the crop's valid frames -> bf16 [n][ld] (mean at [0,C), std at [C,2C), zero pad).
// Unweighted: unbiased std (StatsPool without weights).  Weighted (pyannote StatsPool with weights, used
// for masked per-speaker embeddings in diarization): w_t = weights[nearest(t)], mean = sum(w x)/v1,
// var = sum(w (x-mean)^2) / (v1 - v2/v1), v1 = sum w (+1e-8), v2 = sum w^2.
__global__ __launch_bounds__(256) void stats_pool_kernel(const bf16_t* __restrict__ x, int ld_in, const int* __restrict__ row_off,
                                                         const int* __restrict__ n_valid, const float* __restrict__ weights,
                                                         const long* __restrict__ w_off, const int* __restrict__ w_len,
                                                         bf16_t* __restrict__ out, int ld_out, int C) {
  const int crop = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const int n = n_valid[crop];
  const bf16_t* p = x + (long)row_off[crop] * ld_in + c;
  float mean, sd;
  if (weights == nullptr) {
    float s = 0.f;
    for (int t = 0; t < n; t++) s += bf16_to_f32(p[(long)t * ld_in]);
    mean = s / (float)n;
    float q = 0.f;
    for (int t = 0; t < n; t++) { const float d = bf16_to_f32(p[(long)t * ld_in]) - mean; q += d * d; }
    sd = sqrtf(q / (float)(n - 1));
  } else {
    const float* w = weights + w_off[crop];
    const int nw = w_len[crop];
    float v1 = 1e-8f, v2 = 0.f, s = 0.f;
    for (int t = 0; t < n; t++) {
      const float wt = w[(int)(((long)t * nw) / n)];   // F.interpolate(mode="nearest")
      v1 += wt; v2 += wt * wt; s += wt * bf16_to_f32(p[(long)t * ld_in]);
    }
    mean = s / v1;
    float q = 0.f;
    for (int t = 0; t < n; t++) {
      const float wt = w[(int)(((long)t * nw) / n)];
      const float d = bf16_to_f32(p[(long)t * ld_in]) - mean;
      q += wt * d * d;
    }
    sd = sqrtf(q / (v1 - v2 / v1 + 1e-8f));
  }
  out[(long)crop * ld_out + c] = f32_to_bf16(mean);
  out[(long)crop * ld_out + C + c] = f32_to_bf16(sd);
}

// ---------------------------------------------------------------------------------------------
// LSTM recurrence on the matrix cores.  grid (ceil(crops / 16), direction); a block advances 16 crops
// together: gates[16 x 512] = H[16 x 128] x W_hh^T as v_mfma_f32_16x16x32_bf16 with fp32 accumulate.
//   * wave w owns hidden units 16w..16w+15 and all four of their gates (i|f|g|o): its 16 W_hh fragments
//     (4 gates x K = 128) stay in registers for the whole sequence, and the cell update of a unit is
//     lane-local (the accumulators of the four gates line up register by register);
//   * H lives in LDS as bf16 (two buffers, 288-byte rows: conflict-free ds_read_b128 per tools/lds_bank_sim.py; 272 was two-way), one
//     barrier per step;
//   * the accumulators start from gx (x W_ih^T + b_ih + b_hh, fp32, from the GEMM), fetched three
//     steps ahead of use.
// Rows of the MFMA are independent, so a crop's result does not depend on its block mates.
// gx: [rows][1024] f32 (fwd 0..511 | rev 512..1023);  whh: [2][512][128] bf16;  hout: [rows][256] bf16 (fwd h | rev h).
// ---------------------------------------------------------------------------------------------
#define LSTM_SEQS 16
__global__ __launch_bounds__(512) void lstm_recurrent_kernel(const float* __restrict__ gx, const bf16_t* __restrict__ whh,
                                                             const int* __restrict__ row_off, const int* __restrict__ n_rows, int n_crops,
                                                             bf16_t* __restrict__ hout) {
  __shared__ __attribute__((aligned(16))) bf16_t Hs[2][LSTM_SEQS][144];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, hq = lane >> 4;
  const int dir = blockIdx.y, c0 = blockIdx.x * LSTM_SEQS;
  const int unit = 16 * wave + l15;

  bf16x8 wf[4][4];   // [gate][k step]: B operand, column = unit, k = 32 kk + 8 hq ..
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
      wf[g][kk] = *(const bf16x8*)(whh + ((long)dir * 512 + g * 128 + unit) * 128 + 32 * kk + 8 * hq);

  // this lane's four crops are the accumulator rows 4 hq + r
  int nr[4];
  const float* gbase[4];
  bf16_t* hbase[4];
  int nmax = 0;
  for (int q = 0; q < LSTM_SEQS; q++) {
    const int n = c0 + q < n_crops ? n_rows[c0 + q] : 0;
    nmax = n > nmax ? n : nmax;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int crop = c0 + 4 * hq + r;
    nr[r] = crop < n_crops ? n_rows[crop] : 0;
    const long r0 = crop < n_crops ? row_off[crop] : 0;
    gbase[r] = gx + r0 * 1024 + dir * 512 + unit;
    hbase[r] = hout + r0 * 256 + dir * 128 + unit;
  }
  for (int i = tid; i < 2 * LSTM_SEQS * 144; i += 512) (&Hs[0][0][0])[i] = 0;
  float c[4] = {0.f, 0.f, 0.f, 0.f};

  float gq[4][4][4];   // [ring slot][gate][row]: gx of steps s .. s+3
  auto fetch = [&](int step, int slot) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (step < nr[r]) {
        const int t = dir == 0 ? step : nr[r] - 1 - step;
        const float* gp = gbase[r] + (long)t * 1024;
#pragma unroll
        for (int g = 0; g < 4; g++) gq[slot][g][r] = gp[g * 128];
      }
    }
  };
  fetch(0, 0); fetch(1, 1); fetch(2, 2);
  __syncthreads();
  for (int s0 = 0; s0 < nmax; s0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int step = s0 + u;
      if (step >= nmax) break;            // uniform over the block
      fetch(step + 3, (u + 3) & 3);
      const bf16_t(*Hc)[144] = Hs[step & 1];
      bf16_t(*Hn)[144] = Hs[(step + 1) & 1];
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; g++) acc[g] = (f32x4){gq[u][g][0], gq[u][g][1], gq[u][g][2], gq[u][g][3]};
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const bf16x8 a = *(const bf16x8*)&Hc[l15][32 * kk + 8 * hq];   // A operand: row = crop l15
#pragma unroll
        for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[g][kk], acc[g], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (step < nr[r]) {
          // sigmoid(x) = 1 / (1 + e^-x), tanh(x) = (1 - e^-2x) / (1 + e^-2x), over ONE common denominator per
          // product: c' = f c + i tanh(g) = [c (1+ei)(1+eg) + (1-eg)(1+ef)] / [(1+ef)(1+ei)(1+eg)] and
          // h = o tanh(c') = (1-ec) / [(1+eo)(1+ec)]  -- 5 v_exp + 2 v_rcp per cell (the gate math, not the MFMAs,
          // is the serial part of a step).  Exponents are clamped to +-25 so the triple product stays finite
          // (sigmoid(-25) = 1.4e-11: below fp32 resolution of the sums it enters).
          constexpr float L2E = 1.4426950408889634f, CL = 25.f;
          const float ei = __builtin_amdgcn_exp2f(-L2E * fminf(fmaxf(acc[0][r], -CL), CL));
          const float ef = __builtin_amdgcn_exp2f(-L2E * fminf(fmaxf(acc[1][r], -CL), CL));
          const float eg = __builtin_amdgcn_exp2f(-L2E * fminf(fmaxf(2.f * acc[2][r], -CL), CL));
          const float eo = __builtin_amdgcn_exp2f(-L2E * fminf(fmaxf(acc[3][r], -CL), CL));
          const float ab = (1.f + ei) * (1.f + eg), ff = 1.f + ef;
          c[r] = fmaf(c[r], ab, (1.f - eg) * ff) * __builtin_amdgcn_rcpf(ff * ab);
          const float ec = __builtin_amdgcn_exp2f(-L2E * fminf(fmaxf(2.f * c[r], -CL), CL));
          const bf16_t hb = f32_to_bf16((1.f - ec) * __builtin_amdgcn_rcpf((1.f + eo) * (1.f + ec)));
          const int t = dir == 0 ? step : nr[r] - 1 - step;
          Hn[4 * hq + r][unit] = hb;
          hbase[r][(long)t * 256] = hb;
        }
      }
      __syncthreads();
    }
  }
}

// per-frame activation of the classifier scores: log-softmax (powerset) or sigmoid (multi-label)
__global__ void seg_activation_kernel(const float* __restrict__ logits, int ld, float* __restrict__ out, int C, int rows, int powerset) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* p = logits + (long)r * ld;
  if (powerset) {
    float mx = -INFINITY;
    for (int c = 0; c < C; c++) mx = fmaxf(mx, p[c]);
    float s = 0.f;
    for (int c = 0; c < C; c++) s += expf(p[c] - mx);
    const float lse = mx + logf(s);
    for (int c = 0; c < C; c++) out[(long)r * C + c] = p[c] - lse;
  } else {
    for (int c = 0; c < C; c++) out[(long)r * C + c] = 1.f / (1.f + expf(-p[c]));
  }
}

struct HostT { std::vector<float> data; };

inline bf16_t h2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

struct SincNetW {
  float *nw, *nb, *filt_sum, *n0g, *n0b, *n1g, *n1b, *n2g, *n2b, *b1, *b2;
  bf16_t* bfrag;    // sinc filters as MFMA B fragments [9 steps][5 tiles][hi|lo][64 lanes][8]
  bf16_t *W1, *W2;  // conv k5: [60][448] (5 x 80 padded) and [60][320] (5 x 64)
};

struct Plan {
  int n = 0;
  long R1 = 0, R2 = 0, R3 = 0;
  std::vector<long> coff;
  std::vector<int> clen, off1, off2, off3, f1, f2, f3, fv;
};

}  // namespace

struct ccx_speaker {
  ccx_ctx* ctx = nullptr;
  int kind = 0;            // 0 = x-vector embedder, 1 = PyanNet segmentation
  int n_classes = 0, powerset = 1;
  int max_crops = 0;
  long max_samples = 0;
  bool finalized = false;
  std::map<std::string, HostT> staged;
  std::vector<void*> allocs;
  SincNetW sn{};
  // x-vector
  bf16_t* Wt[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // TDNN weights, tap-major [taps][N][Kpad]
  float *bt[5] = {}, *sct[5] = {}, *sht[5] = {};
  bf16_t* Wemb = nullptr; float* bemb = nullptr;
  // pyannet
  bf16_t* Wih[4] = {}; float* bih[4] = {}; bf16_t* whh[4] = {};
  bf16_t *Wl0 = nullptr, *Wl1 = nullptr, *Wcls = nullptr; float *bl0 = nullptr, *bl1 = nullptr, *bcls = nullptr;
  // workspaces
  long R1cap = 0;
  float2* ac = nullptr;
  float *s1 = nullptr, *c2 = nullptr, *c3 = nullptr, *acc = nullptr, *gx = nullptr, *logit = nullptr, *inorm_part = nullptr;
  bf16_t *s1n = nullptr, *s2n = nullptr, *s3n = nullptr, *a1 = nullptr, *a2 = nullptr, *a5 = nullptr, *pooled = nullptr, *hA = nullptr, *hB = nullptr;
  // host copies of the per-call tables: the uploads are asynchronous, so the last few calls' tables stay alive here
  Plan plan_ring[4];
  int plan_slot = 0;
  long* crop_off = nullptr; long* w_off = nullptr; int* w_len = nullptr;
  int *crop_len = nullptr, *off1 = nullptr, *off2 = nullptr, *off3 = nullptr, *nF1 = nullptr, *nF2 = nullptr, *nF3 = nullptr, *nFv = nullptr;
};

namespace {

#define PTRY(expr)        \
  do {                    \
    int _rc = (expr);     \
    if (_rc) return _rc;  \
  } while (0)

template <typename T>
int palloc(ccx_speaker* s, T** out, size_t count) {
  void* p = nullptr;
  const size_t bytes = ccx_align(count * sizeof(T), 256);
  CCX_HIP(s->ctx, hipMalloc(&p, bytes));
  CCX_HIP(s->ctx, hipMemset(p, 0, bytes));
  s->allocs.push_back(p);
  *out = (T*)p;
  return CCX_OK;
}
int pup_f32(ccx_speaker* s, float** out, const float* src, size_t n) {
  PTRY(palloc(s, out, n));
  CCX_HIP(s->ctx, hipMemcpy(*out, src, n * 4, hipMemcpyHostToDevice));
  return CCX_OK;
}
int pup_bf16(ccx_speaker* s, bf16_t** out, const std::vector<float>& src) {
  std::vector<bf16_t> tmp(src.size());
  for (size_t i = 0; i < src.size(); i++) tmp[i] = h2bf(src[i]);
  PTRY(palloc(s, out, src.size()));
  CCX_HIP(s->ctx, hipMemcpy(*out, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
  return CCX_OK;
}
int pneed(ccx_speaker* s, const std::string& name, size_t numel, const HostT** out) {
  auto it = s->staged.find(name);
  if (it == s->staged.end()) return ccx_fail(s->ctx, CCX_ERR_MISSING, "speaker: tensor '%s' was never set", name.c_str());
  if (it->second.data.size() != numel)
    return ccx_fail(s->ctx, CCX_ERR_ARG, "speaker: tensor '%s' has %zu elements, expected %zu", name.c_str(), it->second.data.size(), numel);
  *out = &it->second;
  return CCX_OK;
}
#define PNEED(var, name, numel) \
  const HostT* var = nullptr;   \
  PTRY(pneed(s, (name), (size_t)(numel), &var));

// Conv1d weight [O][I][k] -> GEMM weight [O][Kpad], column tap*Ipad + c
std::vector<float> conv_w(const std::vector<float>& w, int O, int I, int k, int Ipad, int Kpad) {
  std::vector<float> r((size_t)O * Kpad, 0.f);
  for (int o = 0; o < O; o++)
    for (int c = 0; c < I; c++)
      for (int t = 0; t < k; t++) r[(size_t)o * Kpad + t * Ipad + c] = w[((size_t)o * I + c) * k + t];
  return r;
}
// one tap of a Conv1d weight [O][I][k] -> [O][Ipad]
std::vector<float> conv_tap(const std::vector<float>& w, int O, int I, int k, int tap, int Ipad) {
  std::vector<float> r((size_t)O * Ipad, 0.f);
  for (int o = 0; o < O; o++)
    for (int c = 0; c < I; c++) r[(size_t)o * Ipad + c] = w[((size_t)o * I + c) * k + tap];
  return r;
}

int load_sincnet(ccx_speaker* s, const std::string& pre) {
  SincNetW& n = s->sn;
  PNEED(nw, pre + "wav_norm1d.weight", 1); PNEED(nb, pre + "wav_norm1d.bias", 1);
  PNEED(fl, pre + "conv1d.0.filters", SN_F * SN_K);
  PNEED(w1, pre + "conv1d.1.weight", 60 * 80 * 5); PNEED(b1, pre + "conv1d.1.bias", 60);
  PNEED(w2, pre + "conv1d.2.weight", 60 * 60 * 5); PNEED(b2, pre + "conv1d.2.bias", 60);
  PNEED(g0, pre + "norm1d.0.weight", 80); PNEED(be0, pre + "norm1d.0.bias", 80);
  PNEED(g1, pre + "norm1d.1.weight", 60); PNEED(be1, pre + "norm1d.1.bias", 60);
  PNEED(g2, pre + "norm1d.2.weight", 60); PNEED(be2, pre + "norm1d.2.bias", 60);
  std::vector<float> fs(SN_F, 0.f), frag((size_t)SM_STEPS * 5 * 2 * 64 * 8, 0.f);
  for (int f = 0; f < SN_F; f++)
    for (int k = 0; k < SN_K; k++) fs[f] += fl->data[(size_t)f * SN_K + k];
  for (int t = 0; t < SM_STEPS; t++)
    for (int nt = 0; nt < 5; nt++)
      for (int lane = 0; lane < 64; lane++)
        for (int e = 0; e < 8; e++) {
          const int kk = 8 * (lane >> 4) + e, tap = 30 * t + kk, f = 16 * nt + (lane & 15);
          const float w = (kk < 30 && tap < SN_K) ? fl->data[(size_t)f * SN_K + tap] : 0.f;
          const bf16_t hb = h2bf(w);
          uint32_t hu = (uint32_t)hb << 16; float hf; memcpy(&hf, &hu, 4);
          frag[((((size_t)t * 5 + nt) * 2 + 0) * 64 + lane) * 8 + e] = hf;       // exactly representable: pup_bf16 keeps it
          frag[((((size_t)t * 5 + nt) * 2 + 1) * 64 + lane) * 8 + e] = w - hf;   // rounded to bf16 by pup_bf16
        }
  PTRY(pup_f32(s, &n.nw, nw->data.data(), 1)); PTRY(pup_f32(s, &n.nb, nb->data.data(), 1));
  PTRY(pup_bf16(s, &n.bfrag, frag)); PTRY(pup_f32(s, &n.filt_sum, fs.data(), fs.size()));
  PTRY(pup_bf16(s, &n.W1, conv_w(w1->data, 60, 80, 5, 80, 448))); PTRY(pup_f32(s, &n.b1, b1->data.data(), 60));
  PTRY(pup_bf16(s, &n.W2, conv_w(w2->data, 60, 60, 5, 64, 320))); PTRY(pup_f32(s, &n.b2, b2->data.data(), 60));
  PTRY(pup_f32(s, &n.n0g, g0->data.data(), 80)); PTRY(pup_f32(s, &n.n0b, be0->data.data(), 80));
  PTRY(pup_f32(s, &n.n1g, g1->data.data(), 60)); PTRY(pup_f32(s, &n.n1b, be1->data.data(), 60));
  PTRY(pup_f32(s, &n.n2g, g2->data.data(), 60)); PTRY(pup_f32(s, &n.n2b, be2->data.data(), 60));
  return CCX_OK;
}

int make_plan(ccx_speaker* s, const int* n_samples, const int64_t* offsets, int n, Plan& P) {
  ccx_ctx* ctx = s->ctx;
  P.n = n;
  P.coff.resize(n); P.clen.resize(n); P.off1.resize(n); P.off2.resize(n); P.off3.resize(n);
  P.f1.resize(n); P.f2.resize(n); P.f3.resize(n); P.fv.resize(n);
  for (int i = 0; i < n; i++) {
    const int T = n_samples[i];
    const int f1c = (T - SN_K) / SN_STRIDE + 1, f1 = T >= SN_K ? f1c / 3 : 0;
    const int f2 = (f1 - 4) / 3, f3 = (f2 - 4) / 3;
    CCX_REQUIRE(ctx, T >= SN_K && f1 >= 5 && f2 >= 5 && f3 >= 1, "speaker: crop %d (%d samples) is too short for the SincNet front end", i, T);
    const int fv = f3 - 14;  // after the TDNN's valid convolutions (k5 d1, k3 d2, k3 d3)
    CCX_REQUIRE(ctx, s->kind != 0 || fv >= 2, "speaker: crop %d (%d samples) is too short for the x-vector TDNN", i, T);
    P.coff[i] = offsets[i]; P.clen[i] = T; P.f1[i] = f1; P.f2[i] = f2; P.f3[i] = f3; P.fv[i] = fv;
    P.off1[i] = (int)P.R1; P.off2[i] = (int)P.R2; P.off3[i] = (int)P.R3;
    P.R1 += f1; P.R2 += f2; P.R3 += f3;
  }
  CCX_REQUIRE(ctx, P.R1 <= s->R1cap, "speaker: %ld front-end frames exceed capacity %ld", P.R1, s->R1cap);
  return CCX_OK;
}

int upload_plan(ccx_speaker* s, const Plan& P, hipStream_t st) {
  ccx_ctx* ctx = s->ctx;
#define UPI(dst, vec) CCX_HIP(ctx, hipMemcpyAsync(dst, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice, st))
  UPI(s->crop_off, P.coff); UPI(s->crop_len, P.clen); UPI(s->off1, P.off1); UPI(s->off2, P.off2); UPI(s->off3, P.off3);
  UPI(s->nF1, P.f1); UPI(s->nF2, P.f2); UPI(s->nF3, P.f3); UPI(s->nFv, P.fv);
#undef UPI
  return CCX_OK;   // no sync: P lives in the handle's ring, and the device tables are rewritten in stream order
}

// SincNet over all crops: leaves s3n [R3][64] bf16 (60 channels + zero pad)
int run_sincnet(ccx_speaker* s, const float* wav, const Plan& P, hipStream_t st) {
  ccx_ctx* ctx = s->ctx;
  const SincNetW& n = s->sn;
  hipLaunchKernelGGL(wav_stats_kernel, dim3(P.n), dim3(256), 0, st, wav, s->crop_off, s->crop_len, n.nw, n.nb, s->ac);
  CCX_CHECK_LAUNCH(ctx);
  int maxp = 0;
  for (int i = 0; i < P.n; i++) maxp = P.f1[i] > maxp ? P.f1[i] : maxp;
  {
    static ccx_lds_optin optin;
    CCX_HIP(ctx, optin.ensure(ctx->device, (const void*)sinc_conv_pool_kernel, SM_LDS));
    const int chunks = ccx_cdiv(maxp, SM_FRAMES);
    const long items = (long)chunks * P.n;
    long positions = 0;
    for (int i = 0; i < P.n; i++) positions += 3L * P.f1[i];
    ccx_prof_scope ps(ctx, st, "sinc_conv_pool_kernel", 2.0 * positions * SN_F * SN_K, 0.0);
    hipLaunchKernelGGL(sinc_conv_pool_kernel, dim3((unsigned)(items < 256 ? items : 256)), dim3(512), SM_LDS, st, wav, s->crop_off, s->crop_len,
                       s->off1, s->nF1, s->ac, n.bfrag, n.filt_sum, s->s1, P.n, chunks);
  }
  CCX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(inorm_partial_kernel<1>, dim3(P.n, INORM_CHUNKS, 1), dim3(256), 0, st, s->s1, 80, s->off1, s->nF1, s->inorm_part, 80, 128);
  CCX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(inorm_apply_kernel<1>, dim3(P.n, INORM_CHUNKS, 1), dim3(256), 0, st, s->s1, 80, s->off1, s->s1n, 80, s->off1, s->nF1,
                     s->inorm_part, n.n0g, n.n0b, 80, 128);
  CCX_CHECK_LAUNCH(ctx);
  GemmParams p;
  memset(&p, 0, sizeof(p));   // conv1d(80 -> 60, k5) as a GEMM over a strided view: row m = frames m..m+4
  p.A = s->s1n; p.lda = 80; p.W = n.W1; p.ldw = 448; p.M = (int)P.R1; p.N = 60; p.K = 448; p.bias = n.b1; p.out = s->c2; p.ldo = 128;
  PTRY(ccx_launch_gemm(ctx, EPI_F32, p, st));
  hipLaunchKernelGGL(inorm_partial_kernel<3>, dim3(P.n, INORM_CHUNKS, 1), dim3(256), 0, st, s->c2, 128, s->off1, s->nF2, s->inorm_part, 60, 128);
  CCX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(inorm_apply_kernel<3>, dim3(P.n, INORM_CHUNKS, 1), dim3(256), 0, st, s->c2, 128, s->off1, s->s2n, 64, s->off2, s->nF2,
                     s->inorm_part, n.n1g, n.n1b, 60, 128);
  CCX_CHECK_LAUNCH(ctx);
  memset(&p, 0, sizeof(p));   // conv1d(60 -> 60, k5), channels padded to 64
  p.A = s->s2n; p.lda = 64; p.W = n.W2; p.ldw = 320; p.M = (int)P.R2; p.N = 60; p.K = 320; p.bias = n.b2; p.out = s->c3; p.ldo = 128;
  PTRY(ccx_launch_gemm(ctx, EPI_F32, p, st));
  hipLaunchKernelGGL(inorm_partial_kernel<3>, dim3(P.n, INORM_CHUNKS, 1), dim3(256), 0, st, s->c3, 128, s->off2, s->nF3, s->inorm_part, 60, 128);
  CCX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(inorm_apply_kernel<3>, dim3(P.n, INORM_CHUNKS, 1), dim3(256), 0, st, s->c3, 128, s->off2, s->s3n, 64, s->off3, s->nF3,
                     s->inorm_part, n.n2g, n.n2b, 60, 128);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

}  // namespace

extern "C" {

int ccx_speaker_create(ccx_ctx* ctx, int kind, int n_classes, int powerset, int max_crops, int64_t max_samples, ccx_speaker** out) {
  if (!ctx) return CCX_ERR_ARG;
  CCX_REQUIRE(ctx, out && (kind == 0 || kind == 1), "ccx_speaker_create: kind must be 0 (x-vector) or 1 (PyanNet)");
  CCX_REQUIRE(ctx, kind == 0 || (n_classes >= 1 && n_classes <= 64), "ccx_speaker_create: n_classes out of range");
  CCX_REQUIRE(ctx, max_crops >= 1 && max_samples >= 1024, "ccx_speaker_create: capacity too small");
  ccx_speaker* s = new ccx_speaker();
  s->ctx = ctx; s->kind = kind; s->n_classes = n_classes; s->powerset = powerset; s->max_crops = max_crops; s->max_samples = max_samples;
  *out = s;
  return CCX_OK;
}

void ccx_speaker_destroy(ccx_speaker* s) {
  if (!s) return;
  for (void* p : s->allocs) hipFree(p);
  delete s;
}

int ccx_speaker_set_tensor(ccx_speaker* s, const char* name, const float* data, int64_t numel) {
  if (!s) return CCX_ERR_ARG;
  CCX_REQUIRE(s->ctx, !s->finalized && name && data && numel > 0, "speaker: set_tensor bad arguments");
  HostT t;
  t.data.resize((size_t)numel);
  CCX_HIP(s->ctx, hipMemcpy(t.data.data(), data, (size_t)numel * 4, hipMemcpyDefault));
  s->staged[std::string(name)] = std::move(t);
  return CCX_OK;
}

int ccx_speaker_finalize(ccx_speaker* s) {
  if (!s) return CCX_ERR_ARG;
  CCX_REQUIRE(s->ctx, !s->finalized, "speaker: finalize called twice");
  PTRY(load_sincnet(s, "sincnet."));
  if (s->kind == 0) {
    const int outc[5] = {512, 512, 512, 512, 1500}, ks[5] = {5, 3, 3, 1, 1}, inc[5] = {60, 512, 512, 512, 512};
    for (int l = 0; l < 5; l++) {
      const std::string p = "tdnns." + std::to_string(l);
      PNEED(w, p + ".0.weight", (size_t)outc[l] * inc[l] * ks[l]); PNEED(b, p + ".0.bias", outc[l]);
      PNEED(g, p + ".2.weight", outc[l]); PNEED(be, p + ".2.bias", outc[l]);
      PNEED(rm, p + ".2.running_mean", outc[l]); PNEED(rv, p + ".2.running_var", outc[l]);
      std::vector<float> sc(outc[l]), sh(outc[l]);
      for (int c = 0; c < outc[l]; c++) { sc[c] = g->data[c] / sqrtf(rv->data[c] + 1e-5f); sh[c] = be->data[c] - rm->data[c] * sc[c]; }
      PTRY(pup_f32(s, &s->bt[l], b->data.data(), outc[l])); PTRY(pup_f32(s, &s->sct[l], sc.data(), outc[l])); PTRY(pup_f32(s, &s->sht[l], sh.data(), outc[l]));
      std::vector<float> all;
      if (l == 0) all = conv_w(w->data, 512, 60, 5, 64, 320);   // dilation 1: one GEMM over the 5-frame view
      else
        for (int t = 0; t < ks[l]; t++) { auto tap = conv_tap(w->data, outc[l], inc[l], ks[l], t, 512); all.insert(all.end(), tap.begin(), tap.end()); }
      PTRY(pup_bf16(s, &s->Wt[l], all));
    }
    PNEED(we, "embedding.weight", 512 * 3000); PNEED(bemb, "embedding.bias", 512);
    std::vector<float> wp((size_t)512 * 3072, 0.f);
    for (int o = 0; o < 512; o++) for (int k = 0; k < 3000; k++) wp[(size_t)o * 3072 + k] = we->data[(size_t)o * 3000 + k];
    PTRY(pup_bf16(s, &s->Wemb, wp)); PTRY(pup_f32(s, &s->bemb, bemb->data.data(), 512));
  } else {
    for (int l = 0; l < 4; l++) {
      const int in = l == 0 ? 60 : 256, inp = l == 0 ? 64 : 256;
      std::vector<float> wih((size_t)1024 * inp, 0.f), bias(1024), whh((size_t)2 * 512 * 128);
      for (int dir = 0; dir < 2; dir++) {
        const std::string sfx = "_l" + std::to_string(l) + (dir ? "_reverse" : "");
        PNEED(wi, "lstm.weight_ih" + sfx, (size_t)512 * in); PNEED(wh, "lstm.weight_hh" + sfx, 512 * 128);
        PNEED(bi, "lstm.bias_ih" + sfx, 512); PNEED(bh, "lstm.bias_hh" + sfx, 512);
        for (int r = 0; r < 512; r++) {
          for (int k = 0; k < in; k++) wih[(size_t)(dir * 512 + r) * inp + k] = wi->data[(size_t)r * in + k];
          bias[dir * 512 + r] = bi->data[r] + bh->data[r];
          for (int k = 0; k < 128; k++) whh[((size_t)dir * 512 + r) * 128 + k] = wh->data[(size_t)r * 128 + k];
        }
      }
      PTRY(pup_bf16(s, &s->Wih[l], wih)); PTRY(pup_f32(s, &s->bih[l], bias.data(), 1024)); PTRY(pup_bf16(s, &s->whh[l], whh));
    }
    PNEED(l0w, "linear.0.weight", 128 * 256); PNEED(l0b, "linear.0.bias", 128);
    PNEED(l1w, "linear.1.weight", 128 * 128); PNEED(l1b, "linear.1.bias", 128);
    PNEED(cw, "classifier.weight", (size_t)s->n_classes * 128); PNEED(cb, "classifier.bias", s->n_classes);
    PTRY(pup_bf16(s, &s->Wl0, l0w->data)); PTRY(pup_f32(s, &s->bl0, l0b->data.data(), 128));
    PTRY(pup_bf16(s, &s->Wl1, l1w->data)); PTRY(pup_f32(s, &s->bl1, l1b->data.data(), 128));
    PTRY(pup_bf16(s, &s->Wcls, cw->data)); PTRY(pup_f32(s, &s->bcls, cb->data.data(), s->n_classes));
  }
  s->staged.clear();
  // capacity: pooled sinc frames <= samples / 30
  s->R1cap = s->max_samples / 30 + 8 * (long)s->max_crops;
  const size_t R1 = (size_t)s->R1cap + 16, R2 = R1 / 3 + 16, R3 = R2 / 3 + 16, C = (size_t)s->max_crops;
  PTRY(palloc(s, &s->ac, C)); PTRY(palloc(s, &s->s1, R1 * 80)); PTRY(palloc(s, &s->s1n, R1 * 80 + 1024));
  PTRY(palloc(s, &s->c2, R1 * 128)); PTRY(palloc(s, &s->s2n, R2 * 64 + 1024)); PTRY(palloc(s, &s->c3, R2 * 128)); PTRY(palloc(s, &s->s3n, R3 * 64 + 1024));
  PTRY(palloc(s, &s->inorm_part, (size_t)C * INORM_CHUNKS * 128 * 2));
  PTRY(palloc(s, &s->crop_off, C)); PTRY(palloc(s, &s->crop_len, C)); PTRY(palloc(s, &s->w_off, C)); PTRY(palloc(s, &s->w_len, C));
  PTRY(palloc(s, &s->off1, C)); PTRY(palloc(s, &s->off2, C)); PTRY(palloc(s, &s->off3, C));
  PTRY(palloc(s, &s->nF1, C)); PTRY(palloc(s, &s->nF2, C)); PTRY(palloc(s, &s->nF3, C)); PTRY(palloc(s, &s->nFv, C));
  if (s->kind == 0) {
    PTRY(palloc(s, &s->a1, (R3 + 16) * 512)); PTRY(palloc(s, &s->a2, (R3 + 16) * 512)); PTRY(palloc(s, &s->acc, (R3 + 16) * 512));
    PTRY(palloc(s, &s->a5, (R3 + 16) * 1536)); PTRY(palloc(s, &s->pooled, C * 3072 + 1024));
  } else {
    PTRY(palloc(s, &s->gx, R3 * 1024)); PTRY(palloc(s, &s->hA, R3 * 256 + 1024)); PTRY(palloc(s, &s->hB, R3 * 256 + 1024));
    PTRY(palloc(s, &s->logit, R3 * 128));
  }
  s->finalized = true;
  return CCX_OK;
}

// x-vector: wav_dev holds the crops at `offsets[i]` (in samples) with n_samples[i] each -> out_dev [n][512] f32
int ccx_speaker_embed(ccx_speaker* s, const float* wav, const int64_t* offsets, const int* n_samples, int n, const float* weights,
                      const int64_t* w_offsets, const int* w_lens, float* out, void* stream_) {
  if (!s) return CCX_ERR_ARG;
  ccx_ctx* ctx = s->ctx;
  hipStream_t st = (hipStream_t)stream_;
  CCX_REQUIRE(ctx, s->finalized && s->kind == 0 && wav && offsets && n_samples && out && n >= 1 && n <= s->max_crops, "speaker_embed: bad arguments (n=%d, max %d)", n, s->max_crops);
  Plan& P = s->plan_ring[s->plan_slot++ & 3];
  P = Plan();
  PTRY(make_plan(s, n_samples, offsets, n, P));
  PTRY(upload_plan(s, P, st));
  PTRY(run_sincnet(s, wav, P, st));
  const int R3 = (int)P.R3;
  GemmParams p;
  // tdnn 0: k5 d1 over the 5-frame view of s3n
  memset(&p, 0, sizeof(p));
  p.A = s->s3n; p.lda = 64; p.W = s->Wt[0]; p.ldw = 320; p.M = R3; p.N = 512; p.K = 320; p.bias = s->bt[0]; p.out = s->a1; p.ldo = 512;
  p.scale = s->sct[0]; p.shift = s->sht[0]; p.slope = 0.01f;
  PTRY(ccx_launch_gemm(ctx, EPI_BF16_LRELU_AFFINE, p, st));
  // tdnn 1 (d2) and 2 (d3): three accumulated taps each
  bf16_t* src = s->a1; bf16_t* dst = s->a2;
  for (int l = 1; l <= 2; l++) {
    const int dil = l + 1;
    for (int tap = 0; tap < 3; tap++) {
      memset(&p, 0, sizeof(p));
      p.A = src + (long)tap * dil * 512; p.lda = 512; p.W = s->Wt[l] + (long)tap * 512 * 512; p.ldw = 512; p.M = R3; p.N = 512; p.K = 512;
      if (tap == 0) { p.bias = s->bt[l]; p.out = s->acc; p.ldo = 512; PTRY(ccx_launch_gemm(ctx, EPI_F32, p, st)); }
      else if (tap == 1) { p.out = s->acc; p.ldo = 512; p.resid = s->acc; p.ldr = 512; PTRY(ccx_launch_gemm(ctx, EPI_F32_RESID, p, st)); }
      else {
        p.out = dst; p.ldo = 512; p.resid = s->acc; p.ldr = 512; p.scale = s->sct[l]; p.shift = s->sht[l]; p.slope = 0.01f;
        PTRY(ccx_launch_gemm(ctx, EPI_BF16_LRELU_AFFINE, p, st));
      }
    }
    bf16_t* t = src; src = dst; dst = t;
  }
  // tdnn 3 (k1) and 4 (k1, 1500 channels)
  memset(&p, 0, sizeof(p));
  p.A = src; p.lda = 512; p.W = s->Wt[3]; p.ldw = 512; p.M = R3; p.N = 512; p.K = 512; p.bias = s->bt[3]; p.out = dst; p.ldo = 512;
  p.scale = s->sct[3]; p.shift = s->sht[3]; p.slope = 0.01f;
  PTRY(ccx_launch_gemm(ctx, EPI_BF16_LRELU_AFFINE, p, st));
  memset(&p, 0, sizeof(p));
  p.A = dst; p.lda = 512; p.W = s->Wt[4]; p.ldw = 512; p.M = R3; p.N = 1500; p.K = 512; p.bias = s->bt[4]; p.out = s->a5; p.ldo = 1536;
  p.scale = s->sct[4]; p.shift = s->sht[4]; p.slope = 0.01f;
  PTRY(ccx_launch_gemm(ctx, EPI_BF16_LRELU_AFFINE, p, st));
  if (weights) {
    CCX_REQUIRE(ctx, w_offsets && w_lens, "speaker_embed: weights need offsets and lengths");
    std::vector<long> wo(n); std::vector<int> wl(n);
    for (int i = 0; i < n; i++) { CCX_REQUIRE(ctx, w_lens[i] >= 1, "speaker_embed: empty weight row %d", i); wo[i] = w_offsets[i]; wl[i] = w_lens[i]; }
    CCX_HIP(ctx, hipMemcpyAsync(s->w_off, wo.data(), n * sizeof(long), hipMemcpyHostToDevice, st));
    CCX_HIP(ctx, hipMemcpyAsync(s->w_len, wl.data(), n * 4, hipMemcpyHostToDevice, st));
    CCX_HIP(ctx, hipStreamSynchronize(st));
  }
  hipLaunchKernelGGL(stats_pool_kernel, dim3(n, ccx_cdiv(1500, 256)), dim3(256), 0, st, s->a5, 1536, s->off3, s->nFv, weights, s->w_off,
                     s->w_len, s->pooled, 3072, 1500);
  CCX_CHECK_LAUNCH(ctx);
  // embedding Linear(3000 -> 512); out rows are 512 wide
  memset(&p, 0, sizeof(p));
  p.A = s->pooled; p.lda = 3072; p.W = s->Wemb; p.ldw = 3072; p.M = n; p.N = 512; p.K = 3072; p.bias = s->bemb; p.out = out; p.ldo = 512;
  PTRY(ccx_launch_gemm(ctx, EPI_F32, p, st));
  return CCX_OK;
}

// PyanNet: scores for every SincNet frame of every crop.  frames_out[i] (host) receives the frame count of
// crop i; out_dev is [sum frames][n_classes] f32 in crop order.
int ccx_speaker_segment(ccx_speaker* s, const float* wav, const int64_t* offsets, const int* n_samples, int n, float* out,
                        int64_t out_capacity_rows, int* frames_out, void* stream_) {
  if (!s) return CCX_ERR_ARG;
  ccx_ctx* ctx = s->ctx;
  hipStream_t st = (hipStream_t)stream_;
  CCX_REQUIRE(ctx, s->finalized && s->kind == 1 && wav && offsets && n_samples && out && frames_out && n >= 1 && n <= s->max_crops, "speaker_segment: bad arguments");
  Plan& P = s->plan_ring[s->plan_slot++ & 3];
  P = Plan();
  PTRY(make_plan(s, n_samples, offsets, n, P));
  CCX_REQUIRE(ctx, P.R3 <= out_capacity_rows, "speaker_segment: output needs %ld rows, capacity %ld", P.R3, (long)out_capacity_rows);
  for (int i = 0; i < n; i++) frames_out[i] = P.f3[i];
  PTRY(upload_plan(s, P, st));
  PTRY(run_sincnet(s, wav, P, st));
  const int R3 = (int)P.R3;
  GemmParams p;
  const bf16_t* x = s->s3n;
  long ldx = 64; int K = 64;
  bf16_t* hbuf[2] = {s->hA, s->hB};
  for (int l = 0; l < 4; l++) {
    memset(&p, 0, sizeof(p));
    p.A = x; p.lda = ldx; p.W = s->Wih[l]; p.ldw = K; p.M = R3; p.N = 1024; p.K = K; p.bias = s->bih[l]; p.out = s->gx; p.ldo = 1024;
    PTRY(ccx_launch_gemm(ctx, EPI_F32, p, st));
    {
      ccx_prof_scope ps(ctx, st, "lstm_recurrent_kernel", 0.0, 0.0);
      hipLaunchKernelGGL(lstm_recurrent_kernel, dim3(ccx_cdiv(n, LSTM_SEQS), 2), dim3(512), 0, st, s->gx, s->whh[l], s->off3, s->nF3, n, hbuf[l & 1]);
    }
    CCX_CHECK_LAUNCH(ctx);
    x = hbuf[l & 1]; ldx = 256; K = 256;
  }
  bf16_t* t0 = hbuf[0];  // layer 3 wrote hbuf[1]; reuse hbuf[0] for the linear outputs ([R3][128] fits in [R3][256])
  memset(&p, 0, sizeof(p));
  p.A = x; p.lda = 256; p.W = s->Wl0; p.ldw = 256; p.M = R3; p.N = 128; p.K = 256; p.bias = s->bl0; p.out = t0; p.ldo = 128; p.slope = 0.01f;
  PTRY(ccx_launch_gemm(ctx, EPI_BF16_LRELU_AFFINE, p, st));
  bf16_t* t1 = hbuf[1];
  memset(&p, 0, sizeof(p));
  p.A = t0; p.lda = 128; p.W = s->Wl1; p.ldw = 128; p.M = R3; p.N = 128; p.K = 128; p.bias = s->bl1; p.out = t1; p.ldo = 128; p.slope = 0.01f;
  PTRY(ccx_launch_gemm(ctx, EPI_BF16_LRELU_AFFINE, p, st));
  memset(&p, 0, sizeof(p));
  p.A = t1; p.lda = 128; p.W = s->Wcls; p.ldw = 128; p.M = R3; p.N = s->n_classes; p.K = 128; p.bias = s->bcls; p.out = s->logit; p.ldo = 128;
  PTRY(ccx_launch_gemm(ctx, EPI_F32, p, st));
  hipLaunchKernelGGL(seg_activation_kernel, dim3(ccx_cdiv(R3, 256)), dim3(256), 0, st, s->logit, 128, out, s->n_classes, R3, s->powerset);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

}  // extern "C"
