// ccx_common.h -- shared device/host helpers for libccx (gfx950 / CDNA4 only).
//
// libccx is the MI355X-native replacement for the model objects ClearConverse's
// EnhancedAudioProcessor calls (reference back/api.py:657-797 creates them,
// back/api.py:1298-1549 drives them).  No torch types appear anywhere in the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#define CCX_OK 0
#define CCX_ERR_ARG 1
#define CCX_ERR_HIP 2
#define CCX_ERR_STATE 3
#define CCX_ERR_MISSING 4

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// One timed launch (HIP events on the launch stream) -- filled only while profiling is enabled.
struct ccx_prof_rec {
  const char* name;
  double flops, bytes;  // algorithmic work of this launch
  hipEvent_t start, stop;
};

struct ccx_ctx {
  int device;
  std::string last_error;
  bool prof_on = false;
  std::vector<ccx_prof_rec> prof;
  std::mutex mu;   // guards prof and last_error: the software-pipelined batch driver calls into one context from two host threads
};

// RAII: records start/stop events around a kernel launch when ctx->prof_on (never inside a
// stream capture: the caller passes capturing=true there).
struct ccx_prof_scope {
  ccx_ctx* ctx; hipStream_t stream; bool active; hipEvent_t stop;
  ccx_prof_scope(ccx_ctx* c, hipStream_t s, const char* name, double flops, double bytes) : ctx(c), stream(s), active(false), stop(nullptr) {
    if (!c || !c->prof_on) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
    ccx_prof_rec r{name, flops, bytes, nullptr, nullptr};
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
    hipEventRecord(r.start, s);
    stop = r.stop;                       // kept here: another thread's push_back may move the vector's storage
    {
      std::lock_guard<std::mutex> lk(c->mu);
      c->prof.push_back(r);
    }
    active = true;
  }
  ~ccx_prof_scope() { if (active) hipEventRecord(stop, stream); }
};

// One-time opt-in of a kernel to more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize), PER DEVICE and safe
// for the two host threads that may drive one context (BatchPipeline.run_pinned_pipelined): one bit per device in an atomic, the
// call itself is idempotent, so two threads racing to be first both make it and both succeed.  A `static ccx_lds_optin` lives next to
// each launcher (one per kernel instantiation).
struct ccx_lds_optin {
  std::atomic<unsigned long long> done{0};
  hipError_t ensure(int device, const void* fn, int bytes = 160 * 1024) {
    const unsigned long long bit = 1ull << (device & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

// Set ctx error text and return the code (host side).
int ccx_fail(ccx_ctx* ctx, int code, const char* fmt, ...);

#define CCX_HIP(ctx, expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ccx_fail((ctx), CCX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                      __FILE__, __LINE__);                                                   \
  } while (0)

#define CCX_CHECK_LAUNCH(ctx)                                                                \
  do {                                                                                       \
    hipError_t _e = hipGetLastError();                                                       \
    if (_e != hipSuccess)                                                                    \
      return ccx_fail((ctx), CCX_ERR_HIP, "kernel launch failed: %s (%s:%d)",                \
                      hipGetErrorString(_e), __FILE__, __LINE__);                            \
  } while (0)

#define CCX_REQUIRE(ctx, cond, ...)                                       \
  do {                                                                    \
    if (!(cond)) return ccx_fail((ctx), CCX_ERR_ARG, __VA_ARGS__);        \
  } while (0)

// ---- device helpers -------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// Round-to-nearest-even f32 -> bf16 via the hardware convert (keeps NaN a NaN, see
// MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

// two round-to-nearest-even conversions in ONE v_cvt_pk_bf16_f32 (the scalar form costs a convert, a shift and an or per value)
typedef __attribute__((ext_vector_type(2))) float ccx_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 ccx_bf16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const ccx_bf16x2 v = __builtin_convertvector((ccx_f32x2){lo, hi}, ccx_bf16x2);
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float gelu_erf(float x) {
  // erf GELU (torch.nn.GELU() default: Whisper's conv stem and MLP).  erf by Abramowitz-Stegun 7.1.26 (|error| <=
  // 1.5e-7, one rcp + one exp): the libm erff costs ~3x the VALU work, and the encoder evaluates 885 M GELUs per layer
  // in the GEMM epilogue.
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  // every multiply-add is an explicit fmaf: no contraction freedom, so all kernels round alike
  const float e = fmaf(-(poly * t), __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f), 1.0f);
  const float hx = 0.5f * x;
  return fmaf(hx, copysignf(e, x), hx);
}

// ---- cross-lane reductions on the DPP path (VALU speed; __shfl_xor lowers to ds_bpermute, which
// costs an LDS round trip per step and dominated the latency of the small decode kernels) ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
#define CCX_DPP_QUAD_XOR1 0xB1    // quad_perm [1,0,3,2]
#define CCX_DPP_QUAD_XOR2 0x4E    // quad_perm [2,3,0,1]
#define CCX_DPP_HALF_MIRROR 0x141 // lane i <-> 7-i inside each 8-lane half row
#define CCX_DPP_ROW_MIRROR 0x140  // lane i <-> 15-i inside each 16-lane row

// value held by lane (l ^ 16) / (l ^ 32): gfx950 v_permlane16_swap / v_permlane32_swap
__device__ __forceinline__ float lane_xor16(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // r[0] = rows [0,0,2,2], r[1] = rows [1,1,3,3]
  return __uint_as_float(((threadIdx.x >> 4) & 1) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0] = [lo,lo], r[1] = [hi,hi]
  return __uint_as_float(((threadIdx.x >> 5) & 1) ? r[0] : r[1]);
}

// Sum over aligned groups of 8 lanes; every lane of the group gets the sum.
__device__ __forceinline__ float group8_sum(float v) {
  v += dpp_mov<CCX_DPP_QUAD_XOR1>(v);
  v += dpp_mov<CCX_DPP_QUAD_XOR2>(v);
  v += dpp_mov<CCX_DPP_HALF_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float wave_reduce_sum(float v) {
  v = group8_sum(v);
  v += dpp_mov<CCX_DPP_ROW_MIRROR>(v);  // 16-lane row sums
  const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
  const float b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
  const float c = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
  const float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
  return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_reduce_max(float v) {
  v = fmaxf(v, dpp_mov<CCX_DPP_QUAD_XOR1>(v));
  v = fmaxf(v, dpp_mov<CCX_DPP_QUAD_XOR2>(v));
  v = fmaxf(v, dpp_mov<CCX_DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_mov<CCX_DPP_ROW_MIRROR>(v));
  const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
  const float b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
  const float c = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
  const float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

__host__ __device__ static inline int ccx_cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline size_t ccx_align(size_t x, size_t a) { return (x + a - 1) / a * a; }
