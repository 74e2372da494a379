// How much does a software grid barrier cost on MI355X?  256 / 512 persistent blocks, N barriers: every block's thread 0 adds 1 to a
// device-scope counter and spins (bounded) until it reaches blocks * (k + 1); the other threads wait at __syncthreads.  Compare with
// the ~2.8 us a launch boundary of a dependent hipGraph chain costs at batch 1 (profiles/r03_chain_gaps_whisper_b1.txt).
// build: hipcc -w --offload-arch=gfx950 -O3 tools/microbench_gridbar.hip -o /tmp/gridbar && /tmp/gridbar
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void bar_kernel(unsigned* counter, int n, int* fail, float* sink, const float* src) {
  float acc = 0.f;
  for (int k = 0; k < n; k++) {
    // a little dependent work per phase: one cached load of a value another block wrote in the previous phase
    acc += src[(blockIdx.x * 37 + k) & 1023];
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(counter, 1u);
      const unsigned target = gridDim.x * (unsigned)(k + 1);
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000) { *fail = 1; break; }       // never hang the box
      }
      __threadfence();
    }
    __syncthreads();
  }
  if (acc == 12345.f) sink[0] = acc;
}

int main() {
  unsigned* counter; int* fail; float *sink, *src;
  hipMalloc(&counter, 4); hipMalloc(&fail, 4); hipMalloc(&sink, 4); hipMalloc(&src, 4096);
  hipMemset(src, 0, 4096);
  for (int blocks : {64, 256, 512}) {
    for (int n : {200, 2000}) {
      hipMemset(counter, 0, 4); hipMemset(fail, 0, 4);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(bar_kernel, dim3(blocks), dim3(256), 0, 0, counter, n, fail, sink, src);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      int f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
      printf("blocks %3d  barriers %4d  %8.3f ms  %6.2f us per barrier%s\n", blocks, n, ms, ms * 1e3 / n, f ? "  (SPIN LIMIT HIT)" : "");
    }
  }
  return 0;
}
