// microbench_overlap.hip -- can an HBM-streaming kernel run BESIDE the phased GEMM on the same CUs, or do they only time-slice?
// Stream A: the encoder's fc1 GEMM (ccx_gemm_bf16, M = 288000, N = 3072, K = 768: one 512-thread block per CU, 128 KB of LDS, 2 x 224
// VGPRs per SIMD).  Stream B: a read-only streaming kernel over a 8 GB buffer with a SMALL footprint (256 threads, <= 32 VGPRs, no LDS,
// 4 x 16 B per lane in flight) -- small enough to be resident next to a GEMM block.  Times: each alone, then both together.
// build (GPU box, repo root): hipcc -O3 --offload-arch=gfx950 -Iinclude tools/microbench_overlap.hip -Lclearconverse_amd -lccx -Wl,-rpath,$PWD/clearconverse_amd -o /tmp/mb_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "ccx.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  unsigned acc = 0;
  for (; i + 256 * (UNROLL - 1) < n16; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = __builtin_nontemporal_load(src + i + 256 * u);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
  ccx_ctx* ctx = nullptr;
  if (ccx_ctx_create(0, &ctx)) { printf("ctx failed\n"); return 1; }
  const int M = 288000, N = 3072, K = 768;
  void *A, *W, *out; float* bias; unsigned* sink; u32x4* big;
  const size_t big_bytes = (size_t)8 << 30;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&out, (size_t)M * N * 2));
  CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&big, big_bytes));
  CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(W, 0x3c, (size_t)N * K * 2)); CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(big, 1, big_bytes));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t a0, a1, b0, b1;
  CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
  const int NG = 20, NS = 24;
  auto gemms = [&]() {
    for (int i = 0; i < NG; i++)
      if (ccx_gemm_bf16(ctx, 1, A, K, W, K, bias, out, N, nullptr, 0, M, N, K, sa)) { printf("gemm: %s\n", ccx_last_error(ctx)); exit(1); }
  };
  for (int blocks_per_cu : {2, 4, 8}) {
    auto streams = [&]() {
      for (int i = 0; i < NS; i++) hipLaunchKernelGGL(stream_read<4>, dim3(256 * blocks_per_cu), dim3(256), 0, sb, big, big_bytes / 16, sink);
    };
    gemms(); streams(); CK(hipDeviceSynchronize());
    float tg, ts, tga, tsa;
    CK(hipEventRecord(a0, sa)); gemms(); CK(hipEventRecord(a1, sa)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tg, a0, a1));
    CK(hipEventRecord(b0, sb)); streams(); CK(hipEventRecord(b1, sb)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ts, b0, b1));
    CK(hipEventRecord(a0, sa)); CK(hipEventRecord(b0, sb)); gemms(); streams(); CK(hipEventRecord(a1, sa)); CK(hipEventRecord(b1, sb));
    CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tga, a0, a1)); CK(hipEventElapsedTime(&tsa, b0, b1));
    printf("stream kernel %d blocks/CU: GEMM alone %.2f ms (%.0f TFLOP/s) | stream alone %.2f ms (%.2f TB/s) | together: GEMM %.2f ms, stream %.2f ms  (sum of alone %.2f, max %.2f)\n",
           blocks_per_cu, tg, 2.0 * M * N * K * NG / tg / 1e9, ts, (double)big_bytes * NS / ts / 1e9, tga, tsa, tg + ts, tg > ts ? tg : ts);
  }
  // the same streaming kernel on launches the size of one lane's cross attention (589.8 MB): ramp-up and tail included
  for (int blocks_per_cu : {2, 4, 8}) {
    const size_t small = (size_t)589824000;
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(stream_read<4>, dim3(256 * blocks_per_cu), dim3(256), 0, sb, big, small / 16, sink);
    CK(hipDeviceSynchronize());
    float ts;
    CK(hipEventRecord(b0, sb));
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(stream_read<4>, dim3(256 * blocks_per_cu), dim3(256), 0, sb, big + (size_t)(i % 8) * (small / 16), small / 16, sink);
    CK(hipEventRecord(b1, sb)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ts, b0, b1));
    printf("589.8 MB launches, %d blocks/CU: %.1f us per launch, %.2f TB/s\n", blocks_per_cu, ts / 50 * 1e3, (double)small * 50 / ts / 1e9);
  }
  return 0;
}
