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

struct ccx_ctx {
  int device;
  std::string last_error;
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

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ float gelu_erf(float x) {
  // exact (erf) GELU, as torch.nn.GELU() default used by Whisper's conv stem and MLP
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__host__ __device__ static inline int ccx_cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline size_t ccx_align(size_t x, size_t a) { return (x + a - 1) / a * a; }
