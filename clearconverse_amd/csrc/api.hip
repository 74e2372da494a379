// api.hip -- context management, error reporting and the primitive-operator entry points of
// the C ABI declared in include/ccx.h.
#include <stdarg.h>
#include "../../include/ccx.h"
#include "attention.h"
#include "ccx_common.h"
#include "elementwise.h"
#include "gemm_bf16.h"

static thread_local std::string g_create_error;

int ccx_fail(ccx_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->last_error = buf;
  } else g_create_error = buf;
  return code;
}

extern "C" {

const char* ccx_version(void) { return "ccx 0.1 (gfx950)"; }

int ccx_ctx_create(int device, ccx_ctx** out) {
  if (!out) return ccx_fail(nullptr, CCX_ERR_ARG, "ccx_ctx_create: out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return ccx_fail(nullptr, CCX_ERR_HIP, "ccx_ctx_create: no HIP device available (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n) return ccx_fail(nullptr, CCX_ERR_ARG, "ccx_ctx_create: device %d out of range [0,%d)", device, n);
  e = hipSetDevice(device);
  if (e != hipSuccess) return ccx_fail(nullptr, CCX_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return ccx_fail(nullptr, CCX_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return ccx_fail(nullptr, CCX_ERR_HIP, "libccx is built for gfx950 only; device %d is %s", device, prop.gcnArchName);
  ccx_ctx* c = new ccx_ctx();
  c->device = device;
  *out = c;
  return CCX_OK;
}

static void prof_clear(ccx_ctx* ctx) {
  for (auto& r : ctx->prof) { hipEventDestroy(r.start); hipEventDestroy(r.stop); }
  ctx->prof.clear();
}

void ccx_ctx_destroy(ccx_ctx* ctx) {
  if (!ctx) return;
  prof_clear(ctx);
  delete ctx;
}

int ccx_prof_enable(ccx_ctx* ctx, int on) {
  if (!ctx) return CCX_ERR_ARG;
  prof_clear(ctx);
  ctx->prof_on = on != 0;
  return CCX_OK;
}

int ccx_prof_count(ccx_ctx* ctx) { return ctx ? (int)ctx->prof.size() : 0; }

int ccx_prof_get(ccx_ctx* ctx, int i, char* name_out, int name_cap, double* flops, double* bytes, float* ms) {
  if (!ctx) return CCX_ERR_ARG;
  CCX_REQUIRE(ctx, i >= 0 && i < (int)ctx->prof.size() && name_out && name_cap > 1, "ccx_prof_get: bad index %d", i);
  const ccx_prof_rec& r = ctx->prof[i];
  CCX_HIP(ctx, hipEventSynchronize(r.stop));
  float t = 0.f;
  CCX_HIP(ctx, hipEventElapsedTime(&t, r.start, r.stop));
  snprintf(name_out, name_cap, "%s", r.name);
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  if (ms) *ms = t;
  return CCX_OK;
}

const char* ccx_last_error(const ccx_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

int ccx_gemm_bf16(ccx_ctx* ctx, int epi, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                  void* out, int64_t ldo, const float* resid, int64_t ldr, int M, int N, int K, void* stream) {
  if (!ctx) return CCX_ERR_ARG;
  CCX_REQUIRE(ctx, epi == EPI_BF16 || epi == EPI_BF16_GELU || epi == EPI_F32_RESID || epi == EPI_F32 || epi == EPI_BF16_RELU,
              "ccx_gemm_bf16: epilogue %d not available through this entry point", epi);
  CCX_REQUIRE(ctx, epi != EPI_F32_RESID || resid != nullptr, "ccx_gemm_bf16: epilogue 2 needs resid");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw;
  p.M = M; p.N = N; p.K = K; p.bias = bias; p.out = out; p.ldo = ldo; p.resid = resid; p.ldr = ldr;
  return ccx_launch_gemm(ctx, epi, p, (hipStream_t)stream);
}

int ccx_layernorm(ccx_ctx* ctx, const float* x, const float* gamma, const float* beta, void* out_bf16, float* out_f32,
                  int M, int D, float eps, void* stream) {
  if (!ctx) return CCX_ERR_ARG;
  return ccx_launch_layernorm(ctx, x, D, gamma, beta, (bf16_t*)out_bf16, out_f32, D, M, D, eps, (hipStream_t)stream);
}

int ccx_peak_normalize(ccx_ctx* ctx, const float* x, float* y, int64_t stride, const int* n_samples_dev, int B, float eps, void* stream) {
  if (!ctx) return CCX_ERR_ARG;
  return ccx_launch_peak_normalize(ctx, x, y, (long)stride, n_samples_dev, B, eps, (hipStream_t)stream);
}

int ccx_row_variance(ccx_ctx* ctx, const float* x, int64_t stride, const int* n_samples_dev, int B, float* out, void* stream) {
  if (!ctx) return CCX_ERR_ARG;
  return ccx_launch_row_variance(ctx, x, (long)stride, n_samples_dev, B, out, (hipStream_t)stream);
}

int ccx_cosine_rows(ccx_ctx* ctx, const float* a, const float* b, int R, int D, int b_rows, float* out, void* stream) {
  if (!ctx) return CCX_ERR_ARG;
  return ccx_launch_cosine_rows(ctx, a, b, R, D, b_rows, out, (hipStream_t)stream);
}

int ccx_speaker_profiles(ccx_ctx* ctx, const float* emb, const float* w, const int* spk_dev, int C, int T, int D, int S, float* out, void* stream) {
  if (!ctx) return CCX_ERR_ARG;
  return ccx_launch_speaker_profiles(ctx, emb, w, spk_dev, C, T, D, S, out, (hipStream_t)stream);
}

int ccx_gather_rows(ccx_ctx* ctx, const int64_t* src_ptrs_dev, const int* lens_dev, int n_rows, int max_len, float* dst_dev,
                    int64_t stride, void* stream) {
  if (!ctx) return CCX_ERR_ARG;
  return ccx_launch_gather_rows(ctx, (const long*)src_ptrs_dev, lens_dev, n_rows, max_len, dst_dev, (long)stride, (hipStream_t)stream);
}

int ccx_enc_attention(ccx_ctx* ctx, const void* q, const void* k, const void* vt, void* o, int B, int H, int S, int Spad,
                      void* stream) {
  if (!ctx) return CCX_ERR_ARG;
  return ccx_launch_enc_attention(ctx, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)o, B, H, S, Spad,
                                  (hipStream_t)stream);
}

}  // extern "C"
