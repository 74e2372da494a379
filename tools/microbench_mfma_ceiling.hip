// microbench_mfma_ceiling.hip -- what can the matrix cores of THIS part sustain at the register / LDS footprint of the encoder GEMM?
//
// bench.py prices the Whisper encoder against the nominal 2.5 PFLOP/s of dense bf16 (MI355X_MICROARCH.md).  Under sustained MFMA load on
// non-zero operands the part does not hold its boost clock (DVFS), so the nominal figure is not reachable by ANY kernel; this
// microbenchmark measures the ceiling a kernel with the phased 256 x 256 GEMM's footprint has (csrc/gemm_bf16.hip: 512 threads = 8
// waves per CU, 128 accumulator registers per lane = an 8 x 4 grid of 16 x 16 tiles per wave, 128 KB of LDS claimed -> one block per
// CU), with NOTHING but the matrix instructions in the loop: v_mfma_f32_16x16x32_bf16 on register operands holding random bf16 values
// (zeros would let the clock rise), no loads, no LDS traffic, no barriers.  Variants: (a) bare MFMAs; (b) MFMAs with the main loop's
// LDS operand reads beside them (ds_read_b128 of the 64 KB K-tile image, one per MFMA pair).  The held shader clock is measured
// inside the kernel: s_memtime (shader clock) against s_memrealtime (100 MHz).
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/microbench_mfma_ceiling.hip -o /tmp/mb_mfma && /tmp/mb_mfma [out.json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool LDS_READS>
__global__ __launch_bounds__(512, 1) void mfma_loop(const bf16x8* __restrict__ ops, float* __restrict__ sink, long long* __restrict__ clocks, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  // operands: 8 A fragments and 4 W fragments per wave and k-step, as in the GEMM's 128 x 64 wave tile
  bf16x8 a[8], w[4];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = ops[(threadIdx.x + 512 * i) & 4095];
#pragma unroll
  for (int j = 0; j < 4; j++) w[j] = ops[(threadIdx.x + 512 * (8 + j)) & 4095];
  if (LDS_READS) {
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) ((bf16x8*)smem)[i] = ops[i & 4095];
    __syncthreads();
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {                       // one K tile of 64 = two k-steps of 32: 64 MFMAs per wave
      if (LDS_READS) {
        // the main loop's operand reads: 12 ds_read_b128 per k-step (8 A + 4 W fragments), conflict-free (consecutive lanes)
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = *(const bf16x8*)(smem + ((ks * 12 + i) * 1024 + lane * 16 + (it & 1) * 32768));
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = *(const bf16x8*)(smem + ((ks * 12 + 8 + j) * 1024 + lane * 16 + (it & 1) * 32768));
      }
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[i], acc[i][j], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

// The lever DESIGN.md section 4 names for the next round: FOUR waves per CU (one per SIMD), each 128 x 128 outputs = an 8 x 8 grid of
// accumulators (256 registers), 16 ds_read_b128 per 64 MFMAs (8 A + 8 W fragments per k-step) instead of 12 per 32, the fragments
// of the next k-step requested before the MFMAs of the current one (two register sets).
__global__ __launch_bounds__(256, 1) void mfma_loop_4w(const bf16x8* __restrict__ ops, float* __restrict__ sink, long long* __restrict__ clocks, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 16; i += 256) ((bf16x8*)smem)[i] = ops[i & 4095];
  __syncthreads();
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a[2][8], w[2][8];
  auto rd = [&](int set, int slot) {
#pragma unroll
    for (int i = 0; i < 8; i++) a[set][i] = *(const bf16x8*)(smem + ((slot * 16 + i) * 1024 + lane * 16));
#pragma unroll
    for (int j = 0; j < 8; j++) w[set][j] = *(const bf16x8*)(smem + ((slot * 16 + 8 + j) * 1024 + lane * 16));
  };
  rd(0, 0);
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      rd(ks ^ 1, (2 * it + ks + 1) & 3);                  // next k-step's fragments fly under this one's MFMAs
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks][j], a[ks][i], acc[i][j], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

// The same four-wave geometry on v_mfma_f32_32x32x16_bf16 (a 4 x 4 grid of 32 x 32 accumulators = 256 registers; 8 ds_read_b128 per
// 16 MFMAs = the same bytes per flop): one wave per SIMD cannot issue the 16-cycle 16x16x32 form back to back (variant above), the
// 32-cycle form it can.
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256, 1) void mfma_loop_4w32(const bf16x8* __restrict__ ops, float* __restrict__ sink, long long* __restrict__ clocks, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 16; i += 256) ((bf16x8*)smem)[i] = ops[i & 4095];
  __syncthreads();
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  bf16x8 a[2][4], w[2][4];
  auto rd = [&](int set, int slot) {
#pragma unroll
    for (int i = 0; i < 4; i++) a[set][i] = *(const bf16x8*)(smem + ((slot * 8 + i) * 1024 + lane * 16));
#pragma unroll
    for (int j = 0; j < 4; j++) w[set][j] = *(const bf16x8*)(smem + ((slot * 8 + 4 + j) * 1024 + lane * 16));
  };
  rd(0, 0);
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {                       // a K tile of 64 = four k-steps of 16
      rd((ks & 1) ^ 1, (4 * it + ks + 1) & 7);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks & 1][j], a[ks & 1][i], acc[i][j], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) s += acc[i][j][r];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

static unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool LDS_READS>
static void run(const char* name, const bf16x8* ops, float* sink, long long* clocks, int blocks, FILE* js, bool last) {
  hipFuncSetAttribute((const void*)mfma_loop<LDS_READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 20000;                                  // ~1 ms and more per launch: long enough for the clock to settle
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((mfma_loop<LDS_READS>), dim3(blocks), dim3(512), 131072, 0, ops, sink, clocks, iters);
  hipEventRecord(e0, 0);
  const int n = 10;
  for (int i = 0; i < n; i++) hipLaunchKernelGGL((mfma_loop<LDS_READS>), dim3(blocks), dim3(512), 131072, 0, ops, sink, clocks, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= n;
  std::vector<long long> c(2 * blocks);
  hipMemcpy(c.data(), clocks, c.size() * 8, hipMemcpyDeviceToHost);
  double mhz = 0.0;
  for (int b = 0; b < blocks; b++) mhz += (double)c[2 * b] / (double)c[2 * b + 1] * 100.0;
  mhz /= blocks;
  const double flops = (double)blocks * 8 /*waves*/ * iters * 64.0 * (2.0 * 16 * 16 * 32);
  const double tf = flops / (ms * 1e-3) / 1e12;
  printf("%-44s %d blocks x 8 waves: %.3f ms per launch  %.1f TFLOP/s  (%.3f of 2500)  shader clock held %.0f MHz\n", name, blocks, ms, tf, tf / 2500.0, mhz);
  if (js) fprintf(js, "  \"%s\": {\"tflops\": %.1f, \"frac_of_nominal\": %.4f, \"shader_clock_mhz\": %.0f, \"ms_per_launch\": %.3f}%s\n", name, tf, tf / 2500.0, mhz, ms, last ? "" : ",");
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount;              // one 8-wave block per CU, as the phased GEMM
  std::vector<unsigned short> h(4096 * 8);
  srand(1);
  for (auto& v : h) v = f2bf((float)rand() / RAND_MAX * 2.f - 1.f);      // random bf16 in [-1, 1]: the clock under real data
  bf16x8* ops;
  float* sink;
  long long* clocks;
  hipMalloc(&ops, h.size() * 2);
  hipMalloc(&sink, (size_t)blocks * 512 * 4);
  hipMalloc(&clocks, (size_t)blocks * 16);
  hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (js) fprintf(js, "{\n  \"what\": \"tools/microbench_mfma_ceiling.hip: v_mfma_f32_16x16x32_bf16 only, 8 waves per CU, 128 accumulator registers per lane, 128 KB of LDS claimed, random bf16 operands, %d CUs\",\n", blocks);
  run<false>("bare_mfma", ops, sink, clocks, blocks, js, false);
  run<true>("mfma_with_lds_operand_reads", ops, sink, clocks, blocks, js, false);
  auto run4 = [&](const char* name, void (*kern)(const bf16x8*, float*, long long*, int), bool last) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 131072, 0, ops, sink, clocks, iters);
    hipEventRecord(e0, 0);
    const int n = 10;
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 131072, 0, ops, sink, clocks, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= n;
    std::vector<long long> c(2 * blocks);
    hipMemcpy(c.data(), clocks, c.size() * 8, hipMemcpyDeviceToHost);
    double mhz = 0.0;
    for (int b = 0; b < blocks; b++) mhz += (double)c[2 * b] / (double)c[2 * b + 1] * 100.0;
    mhz /= blocks;
    const double flops = (double)blocks * 4 * iters * 128.0 * (2.0 * 16 * 16 * 32);
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-44s %d blocks x 4 waves: %.3f ms per launch  %.1f TFLOP/s  (%.3f of 2500)  shader clock held %.0f MHz\n", name, blocks, ms, tf, tf / 2500.0, mhz);
    if (js) fprintf(js, "  \"%s\": {\"tflops\": %.1f, \"frac_of_nominal\": %.4f, \"shader_clock_mhz\": %.0f, \"ms_per_launch\": %.3f}%s\n", name, tf, tf / 2500.0, mhz, ms, last ? "" : ",");
  };
  run4("four_waves_128x128_16x16x32_with_lds_reads", mfma_loop_4w, false);
  run4("four_waves_128x128_32x32x16_with_lds_reads", mfma_loop_4w32, true);
  if (js) { fprintf(js, "}\n"); fclose(js); }
  return 0;
}
