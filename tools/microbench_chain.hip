// microbench_chain.hip -- what does one slot of a dependent hipGraph kernel chain cost on MI355X?
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_chain.hip -o gpurun_out/mb ; run on the GPU box.
// Design aid for csrc/decoder.hip (the decode step is a chain of ~100 tiny kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float* a) {}
__global__ void k_store(float* a) { a[blockIdx.x * blockDim.x + threadIdx.x] = 1.f; }
__global__ void k_load_store(const float* __restrict__ in, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[i] + 1.f;
}
__global__ void k_dep2(const int* __restrict__ idx, const float* __restrict__ in, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = idx[i];
  out[i] = in[j] + 1.f;
}
// stream `bytes_per_block` of weights (fresh every kernel) with 16-byte loads, all issued up front
typedef __attribute__((ext_vector_type(4))) float vf4;
template <int NL>
__global__ void k_stream(const vf4* __restrict__ w, const float* __restrict__ in, float* __restrict__ out, long stride4) {
  const vf4* p = w + (long)blockIdx.x * stride4 + threadIdx.x;
  vf4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) v[i] = __builtin_nontemporal_load(p + i * 256);
  float s = in[threadIdx.x];
#pragma unroll
  for (int i = 0; i < NL; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_chain(hipStream_t st, int n, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; i++) launch(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, st);
  for (int r = 0; r < reps; r++) hipGraphLaunch(ge, st);
  hipEventRecord(b, st); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3 / (reps * n);
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int N = 100, R = 20;
  float *a, *b; int* idx; vf4* w;
  const size_t WB = (size_t)1 << 30;
  CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 22)); CK(hipMalloc(&idx, 1 << 22)); CK(hipMalloc(&w, WB));
  CK(hipMemset(a, 0, 1 << 22)); CK(hipMemset(b, 0, 1 << 22)); CK(hipMemset(idx, 0, 1 << 22)); CK(hipMemset(w, 0, WB));
  for (int grid : {8, 48, 96, 576, 2048}) {
    double t0 = time_chain(st, N, R, [&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, a); });
    double t1 = time_chain(st, N, R, [&](int) { hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, st, a); });
    double t2 = time_chain(st, N, R, [&](int i) { hipLaunchKernelGGL(k_load_store, dim3(grid), dim3(256), 0, st, (i & 1) ? a : b, (i & 1) ? b : a); });
    double t3 = time_chain(st, N, R, [&](int i) { hipLaunchKernelGGL(k_dep2, dim3(grid), dim3(256), 0, st, idx, (i & 1) ? a : b, (i & 1) ? b : a); });
    printf("grid %5d: empty %.2f us | store %.2f | load->store (ping-pong) %.2f | idx->load->store %.2f\n", grid, t0, t1, t2, t3);
  }
  // weight streaming: 1.2 MB and 4.7 MB per kernel, fresh bytes each kernel of the chain
  for (int grid : {48, 96, 192, 384}) {
    for (long mb10 : {12L, 47L}) {
      const long bytes = mb10 * 100000;
      const long per_block = bytes / grid / 4096 * 4096;  // multiple of 256 lanes * 16 B
      const long stride4 = per_block / 16;
      const int nl = (int)(per_block / 4096);
      double t = -1;
      auto L = [&](int i) {
        const vf4* wp = w + (long)i * (bytes / 16 + 4096);
        const float* in = (i & 1) ? a : b; float* out = (i & 1) ? b : a;
        if (nl <= 2) hipLaunchKernelGGL(k_stream<2>, dim3(grid), dim3(256), 0, st, wp, in, out, stride4);
        else if (nl <= 4) hipLaunchKernelGGL(k_stream<4>, dim3(grid), dim3(256), 0, st, wp, in, out, stride4);
        else if (nl <= 8) hipLaunchKernelGGL(k_stream<8>, dim3(grid), dim3(256), 0, st, wp, in, out, stride4);
        else if (nl <= 16) hipLaunchKernelGGL(k_stream<16>, dim3(grid), dim3(256), 0, st, wp, in, out, stride4);
        else hipLaunchKernelGGL(k_stream<32>, dim3(grid), dim3(256), 0, st, wp, in, out, stride4);
      };
      t = time_chain(st, N, R, L);
      printf("stream grid %4d, %.1f MB/kernel (%d x 4 KB per block): %.2f us/kernel\n", grid, bytes / 1e6, nl, t);
    }
  }
  return 0;
}
