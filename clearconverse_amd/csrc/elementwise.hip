// elementwise.hip -- LayerNorm (fp32 statistics, bf16 output for the following MFMA GEMM) and
// conversion helpers.  HBM-bound: one wave per row, 16-byte loads, 8/16-byte stores.
#include "elementwise.h"
#include <map>
#include <mutex>
#include <string>

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ ob,
                                                        float* __restrict__ of, long ldo, int M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float4* xr = (const float4*)(x + (long)row * ldx);
  const int nv = D >> 2;  // D % 4 == 0
  // D <= 1024 -> at most 4 float4 per lane kept in registers
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int idx = lane + 64 * i;
    if (idx < nv) { v[i] = xr[idx]; s += v[i].x + v[i].y + v[i].z + v[i].w; }
  }
  const float mean = wave_reduce_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int idx = lane + 64 * i;
    if (idx < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_reduce_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int idx = lane + 64 * i;
    if (idx < nv) {
      const float4 g = ((const float4*)gamma)[idx], bb = ((const float4*)beta)[idx];
      float4 y;
      y.x = (v[i].x - mean) * rstd * g.x + bb.x;
      y.y = (v[i].y - mean) * rstd * g.y + bb.y;
      y.z = (v[i].z - mean) * rstd * g.z + bb.z;
      y.w = (v[i].w - mean) * rstd * g.w + bb.w;
      if (ob) {
        uint2 o;
        o.x = pack_bf16x2(y.x, y.y);
        o.y = pack_bf16x2(y.z, y.w);
        ((uint2*)(ob + (long)row * ldo))[idx] = o;
      }
      if (of) ((float4*)(of + (long)row * ldo))[idx] = y;
    }
  }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    const float4 v = *(const float4*)(src + i);
    uint2 o;
    o.x = pack_bf16x2(v.x, v.y);
    o.y = pack_bf16x2(v.z, v.w);
    *(uint2*)(dst + i) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long j = n & ~3L; j < n; j++) dst[j] = f32_to_bf16(src[j]);
}

__global__ void fill_u16_kernel(bf16_t* dst, bf16_t v, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = v;
}

// K3 peak normalisation: y[b][:n_b] = x / (max|x| + eps); eps == 0 -> divide only when the peak is > 0
// (the two flavours of reference back/api.py:834 and 350-351).  One block per row, two passes.
__global__ __launch_bounds__(1024) void peak_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, long stride,
                                                              const int* __restrict__ n_samples, float eps) {
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = n_samples[b];
  const float* xr = x + (long)b * stride;
  float* yr = y + (long)b * stride;
  float m = 0.f;
  for (int i = tid; i < n; i += 1024) m = fmaxf(m, fabsf(xr[i]));
  m = wave_reduce_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int w = 1; w < 16; w++) m = fmaxf(m, red[w]);
  const float inv = (eps > 0.f) ? 1.0f / (m + eps) : (m > 0.f ? 1.0f / m : 1.0f);
  for (int i = tid; i < n; i += 1024) yr[i] = xr[i] * inv;
}

int ccx_launch_peak_normalize(ccx_ctx* ctx, const float* x, float* y, long stride, const int* n_samples_dev, int B, float eps,
                              hipStream_t stream) {
  CCX_REQUIRE(ctx, x && y && n_samples_dev && B >= 1, "peak_normalize: bad arguments");
  hipLaunchKernelGGL(peak_normalize_kernel, dim3(B), dim3(1024), 0, stream, x, y, stride, n_samples_dev, eps);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

// Unbiased variance of the first n_b samples of every row -- `torch.var(segment_audio)` of the reference's embedding-quality
// weight (back/api.py:939).  One block per row, two passes (mean, then squared deviations), fp64 accumulators in a FIXED order:
// the value does not depend on which rows share the launch.  n_b < 2 gives NaN, like torch.
__global__ __launch_bounds__(1024) void row_variance_kernel(const float* __restrict__ x, long stride, const int* __restrict__ n_samples,
                                                            float* __restrict__ out) {
  __shared__ double red[16];
  __shared__ double bc;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = n_samples[b];
  const float* xr = x + (long)b * stride;
  auto block_sum = [&](double v) -> double {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < 16; w++) t += red[w];
      bc = t;
    }
    __syncthreads();
    return bc;
  };
  double s = 0.0;
  for (int i = tid; i < n; i += 1024) s += (double)xr[i];
  const double mean = block_sum(s) / (double)n;
  double q = 0.0;
  for (int i = tid; i < n; i += 1024) { const double d = (double)xr[i] - mean; q += d * d; }
  const double ss = block_sum(q);
  if (tid == 0) out[b] = (float)(ss / (double)(n - 1));
}

int ccx_launch_row_variance(ccx_ctx* ctx, const float* x, long stride, const int* n_samples_dev, int B, float* out, hipStream_t stream) {
  CCX_REQUIRE(ctx, x && n_samples_dev && out && B >= 1, "row_variance: bad arguments");
  hipLaunchKernelGGL(row_variance_kernel, dim3(B), dim3(1024), 0, stream, x, stride, n_samples_dev, out);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

// Row-wise cosine similarity, `torch.nn.functional.cosine_similarity(a, b, dim=...)` as ATen computes it since 2.x
// (reference back/api.py:878-879): sum_i (a_i / max(|a|, eps)) * (b_i / max(|b|, eps)), eps = 1e-8.  b row r = b + (r % b_rows) * D
// (b_rows < R: every a row against a repeating set of b rows).  One wave per row, fixed reduction tree.
__global__ __launch_bounds__(256) void cosine_rows_kernel(const float* __restrict__ a, const float* __restrict__ bm, int R, int D, int b_rows,
                                                          float eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* ar = a + (long)r * D;
  const float* br = bm + (long)(r % b_rows) * D;
  float na = 0.f, nb = 0.f;
  for (int i = lane; i < D; i += 64) { na = fmaf(ar[i], ar[i], na); nb = fmaf(br[i], br[i], nb); }
  na = fmaxf(sqrtf(wave_reduce_sum(na)), eps);
  nb = fmaxf(sqrtf(wave_reduce_sum(nb)), eps);
  float dot = 0.f;
  for (int i = lane; i < D; i += 64) dot += (ar[i] / na) * (br[i] / nb);
  dot = wave_reduce_sum(dot);
  if (lane == 0) out[r] = dot;
}

int ccx_launch_cosine_rows(ccx_ctx* ctx, const float* a, const float* b, int R, int D, int b_rows, float* out, hipStream_t stream) {
  CCX_REQUIRE(ctx, a && b && out && R >= 1 && D >= 1 && b_rows >= 1, "cosine_rows: bad arguments");
  hipLaunchKernelGGL(cosine_rows_kernel, dim3(ccx_cdiv(R, 4)), dim3(256), 0, stream, a, b, R, D, b_rows, 1e-8f, out);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

// Speaker profiles of the reference's `_build_speaker_profiles` (back/api.py:946-953): per clip c and speaker s the sum of the
// turn embeddings emb[c][t] of that speaker's turns weighted by w[c][t] / (sum of the speaker's weights) -- not re-normalised.
// emb [C, T, D], w [C, T], spk [T] (speaker index of turn t, the same for every clip) -> out [C, S, D].  Turns in index order.
__global__ __launch_bounds__(256) void speaker_profiles_kernel(const float* __restrict__ emb, const float* __restrict__ w,
                                                               const int* __restrict__ spk, int T, int D, int S, float* __restrict__ out) {
  const int c = blockIdx.x, s = blockIdx.y;
  float tot = 0.f;
  for (int t = 0; t < T; t++) tot += (spk[t] == s) ? w[(long)c * T + t] : 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    float acc = 0.f;
    for (int t = 0; t < T; t++)
      if (spk[t] == s) acc += emb[((long)c * T + t) * D + i] * (w[(long)c * T + t] / tot);
    out[((long)c * S + s) * D + i] = acc;
  }
}

int ccx_launch_speaker_profiles(ccx_ctx* ctx, const float* emb, const float* w, const int* spk_dev, int C, int T, int D, int S, float* out,
                                hipStream_t stream) {
  CCX_REQUIRE(ctx, emb && w && spk_dev && out && C >= 1 && T >= 1 && D >= 1 && S >= 1, "speaker_profiles: bad arguments");
  hipLaunchKernelGGL(speaker_profiles_kernel, dim3(C, S), dim3(256), 0, stream, emb, w, spk_dev, T, D, S, out);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

// dst[i][0 .. len_i) = src_i[0 .. len_i): ragged crops (row pointers and lengths in device tables) into a padded batch,
// one launch instead of one copy per crop.  Columns past len_i are left untouched.
__global__ __launch_bounds__(256) void gather_rows_kernel(const long* __restrict__ src_ptrs, const int* __restrict__ lens,
                                                          float* __restrict__ dst, long stride) {
  const int row = blockIdx.y;
  const int n = lens[row];
  const float* src = (const float*)src_ptrs[row];
  float* d = dst + (long)row * stride;
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int j = i + 256 * k;
      if (j < n) d[j] = src[j];
    }
  }
}

int ccx_launch_gather_rows(ccx_ctx* ctx, const long* src_ptrs_dev, const int* lens_dev, int n_rows, int max_len, float* dst,
                           long stride, hipStream_t stream) {
  CCX_REQUIRE(ctx, src_ptrs_dev && lens_dev && dst && n_rows >= 1 && max_len >= 0 && stride >= max_len, "gather_rows: bad arguments");
  if (max_len == 0) return CCX_OK;
  int bx = ccx_cdiv(max_len, 1024);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(bx, n_rows), dim3(256), 0, stream, src_ptrs_dev, lens_dev, dst, stride);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

int ccx_launch_layernorm(ccx_ctx* ctx, const float* x, long ldx, const float* gamma, const float* beta,
                         bf16_t* out_bf16, float* out_f32, long ldo, int M, int D, float eps, hipStream_t stream) {
  CCX_REQUIRE(ctx, M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "layernorm: D=%d must be a multiple of 4 and <= 1024", D);
  CCX_REQUIRE(ctx, ldx % 4 == 0 && ldo % 4 == 0, "layernorm: ld must be a multiple of 4");
  static const bool by_shape = getenv("CCX_PROF_SHAPES") != nullptr;
  const char* label = "layernorm_kernel";
  if (by_shape && ctx->prof_on) {
    static std::mutex mu;
    static std::map<std::string, std::string> names;
    char buf[96];
    snprintf(buf, sizeof(buf), "layernorm_kernel M=%d D=%d", M, D);
    std::lock_guard<std::mutex> lk(mu);
    label = names.emplace(buf, buf).first->second.c_str();
  }
  ccx_prof_scope ps(ctx, stream, label, 0.0, (double)M * D * (4.0 + (out_bf16 ? 2.0 : 0.0) + (out_f32 ? 4.0 : 0.0)));
  hipLaunchKernelGGL(layernorm_kernel, dim3(ccx_cdiv(M, 4)), dim3(256), 0, stream, x, ldx, gamma, beta, out_bf16,
                     out_f32, ldo, M, D, eps);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

int ccx_launch_f32_to_bf16(ccx_ctx* ctx, const float* src, bf16_t* dst, long n, hipStream_t stream) {
  if (n <= 0) return CCX_OK;
  CCX_REQUIRE(ctx, ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "f32_to_bf16: misaligned");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((int)blocks), dim3(256), 0, stream, src, dst, n);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

int ccx_launch_fill_u16(ccx_ctx* ctx, bf16_t* dst, bf16_t v, long n, hipStream_t stream) {
  if (n <= 0) return CCX_OK;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_u16_kernel, dim3((int)blocks), dim3(256), 0, stream, dst, v, n);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}
