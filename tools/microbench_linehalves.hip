// Does it matter for the per-CU pull rate whether a wave instruction asks for HALF cache lines (16 rows x 64 B: the MFMA A-operand image
// dec_xs_stream_kernel loads, the other half of every line coming with the next instruction) or for FULL lines (8 rows x 128 B)?
// One block of 4 waves per row of a [rows][1500][768] bf16 array (2.3 MB per block), wave w streams the 384-byte quarter of every key
// row, 24 x 16-byte loads per lane in flight (4 tiles of 16 keys), cacheable loads, values xor-ed into a sink.
// build: hipcc -w --offload-arch=gfx950 -O3 tools/microbench_linehalves.hip -o /tmp/linehalves && /tmp/linehalves
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int FULL>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned short* X, unsigned* sink, int S) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned short* base = X + (long)blockIdx.x * S * 768 + wave * 192;
  u4 acc = {0, 0, 0, 0};
  const int NT = S / 16;
  u4 img[4][6];
  auto load = [&](u4 (&r)[6], int t) {
    if (FULL) {
      // 8 rows x 128 B per instruction: lane -> row l/8, chunk l%8; 6 instructions cover 16 rows x 384 B
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int row = t * 16 + (i & 1) * 8 + (lane >> 3), line = i >> 1;
        r[i] = *(const u4*)(base + (long)row * 768 + line * 64 + (lane & 7) * 8);
      }
    } else {
      // 16 rows x 64 B per instruction: lane -> row l%16, chunk l/16 (+4 per instruction)
#pragma unroll
      for (int i = 0; i < 6; i++) r[i] = *(const u4*)(base + (long)(t * 16 + (lane & 15)) * 768 + i * 32 + (lane >> 4) * 8);
    }
  };
#pragma unroll
  for (int j = 0; j < 4; j++) load(img[j], j);
  for (int t0 = 0; t0 < NT; t0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (t0 + j < NT) {
#pragma unroll
        for (int i = 0; i < 6; i++) acc ^= img[j][i];
        if (t0 + j + 4 < NT) load(img[j], t0 + j + 4);
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

int main() {
  const int S = 1488, rows_max = 512;           // 93 tiles
  unsigned short* X; unsigned* sink;
  hipMalloc(&X, (size_t)rows_max * S * 768 * 2); hipMalloc(&sink, 4);
  hipMemset(X, 1, (size_t)rows_max * S * 768 * 2);
  for (int rows : {64, 256, 512}) {
    for (int full = 0; full < 2; full++) {
      float best = 1e9;
      for (int it = 0; it < 5; it++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (full) hipLaunchKernelGGL(stream_kernel<1>, dim3(rows), dim3(256), 0, 0, X, sink, S);
        else hipLaunchKernelGGL(stream_kernel<0>, dim3(rows), dim3(256), 0, 0, X, sink, S);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      const double bytes = (double)rows * S * 768 * 2;
      printf("rows %3d  %s  %7.1f us  %5.2f TB/s  %5.1f GB/s per busy CU\n", rows, full ? "full lines (8 rows x 128 B)" : "half lines (16 rows x 64 B)",
             best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e9 / (rows < 256 ? rows : 256));
    }
  }
  // the same kernel on a stream restricted to N CUs (hipExtStreamCreateWithCUMask, the first N mask bits = N / 8 CUs of every XCD):
  // how many CUs does it take to pull the whole HBM rate, with half and with full lines, at 1 - 4 blocks per CU?
  for (int ncu : {64, 96, 128, 160}) {
    unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ncu; i++) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
    for (int per : {1, 2, 3, 4}) {
      const int rows = ncu * per;
      if (rows > rows_max) continue;
      for (int full = 0; full < 2; full++) {
        float best = 1e9;
        for (int it = 0; it < 5; it++) {
          hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
          hipEventRecord(a, st);
          if (full) hipLaunchKernelGGL(stream_kernel<1>, dim3(rows), dim3(256), 0, st, X, sink, S);
          else hipLaunchKernelGGL(stream_kernel<0>, dim3(rows), dim3(256), 0, st, X, sink, S);
          hipEventRecord(b, st); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
        }
        const double bytes = (double)rows * S * 768 * 2;
        printf("%3d CUs x %d blocks  %s  %7.1f us  %5.2f TB/s  %5.1f GB/s per CU\n", ncu, per, full ? "full lines" : "half lines", best * 1e3,
               bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e9 / ncu);
      }
    }
  }
  return 0;
}
