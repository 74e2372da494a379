// microbench_store.hip -- which lane -> address map should a GEMM epilogue use?
// A 256 x 256 fp32 (or bf16) tile per block, 512 threads, every wave a 128 x 64 sub-tile as 8 row tiles of 16 rows:
//   pattern 0 ("lane-contiguous"): lane (l15, h) owns row l15 and 16 consecutive columns 16h .. 16h+15  -> 4 x dwordx4 per row tile,
//                                  one instruction = 16-byte pieces at a 64-byte pitch
//   pattern 1 ("wave-contiguous"): lane owns columns 16i + 4h .. +3 for i = 0..3                         -> same 4 x dwordx4,
//                                  one instruction = 64 contiguous bytes per row
// mode 0: store only; mode 1: load + add + store (the fp32 residual epilogue).
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_store.hip -o /tmp/mb_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PAT, int MODE>
__global__ __launch_bounds__(512) void tile_store(float* __restrict__ out, int M, int N, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, h = lane >> 4;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * 256 + wr * 128, n0 = tn * 256 + wc * 64;
  float4 r[8][4];
  if (MODE == 1) {
#pragma unroll
    for (int mt = 0; mt < 8; mt++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int col = PAT == 0 ? 16 * h + 4 * i : 16 * i + 4 * h;
        r[mt][i] = *(const float4*)(out + (long)(m0 + mt * 16 + l15) * N + n0 + col);
      }
  }
#pragma unroll
  for (int mt = 0; mt < 8; mt++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int col = PAT == 0 ? 16 * h + 4 * i : 16 * i + 4 * h;
      float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
      if (MODE == 1) { v.x += r[mt][i].x; v.y += r[mt][i].y; v.z += r[mt][i].z; v.w += r[mt][i].w; }
      *(float4*)(out + (long)(m0 + mt * 16 + l15) * N + n0 + col) = v;
    }
}

template <int PAT, int MODE>
static void run(float* buf, int M, int N, const char* name) {
  const int tiles_n = N / 256, tiles = (M / 256) * tiles_n;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((tile_store<PAT, MODE>), dim3(tiles), dim3(512), 0, 0, buf, M, N, tiles_n);
  hipEventRecord(e0, 0);
  const int n = 20;
  for (int i = 0; i < n; i++) hipLaunchKernelGGL((tile_store<PAT, MODE>), dim3(tiles), dim3(512), 0, 0, buf, M, N, tiles_n);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
  const double bytes = (double)M * N * 4 * (MODE == 1 ? 2 : 1);
  printf("%-34s M=%d N=%d: %.3f ms  %.2f TB/s\n", name, M, N, ms, bytes / ms / 1e9);
}

int main() {
  const int M = 288000 / 256 * 256;
  for (int N : {768, 3072}) {
    float* buf;
    hipMalloc(&buf, (size_t)M * N * 4);
    hipMemset(buf, 0, (size_t)M * N * 4);
    run<0, 0>(buf, M, N, "lane-contiguous store");
    run<1, 0>(buf, M, N, "wave-contiguous store");
    run<0, 1>(buf, M, N, "lane-contiguous load+add+store");
    run<1, 1>(buf, M, N, "wave-contiguous load+add+store");
    hipFree(buf);
  }
  return 0;
}
