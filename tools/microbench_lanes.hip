// microbench_lanes.hip -- do hipGraph replays on two streams overlap on MI355X, and what does a replay cost the host?
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -pthread tools/microbench_lanes.hip -o /tmp/mbl && /tmp/mbl
// Design aid for the decode lanes in csrc/whisper.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// spin for ~`cycles` shader clocks, then touch memory so the chain is a real dependency
__global__ void k_spin(float* a, long cycles) {
  const long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (threadIdx.x == 0) a[blockIdx.x] += 1.f;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static hipGraphExec_t make_graph(hipStream_t st, float* buf, int n, int blocks, long cycles) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, buf, cycles);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  return ge;
}

int main() {
  const int N = 135, REPS = 40;
  float* buf[2];
  hipStream_t st[2];
  for (int i = 0; i < 2; i++) {
    CK(hipMalloc(&buf[i], 1 << 20));
    CK(hipMemset(buf[i], 0, 1 << 20));
    CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  }
  for (int blocks : {1, 64, 1024}) {
    for (long cycles : {2000L, 20000L}) {   // ~1 us and ~10 us kernels
      hipGraphExec_t ge[2] = {make_graph(st[0], buf[0], N, blocks, cycles), make_graph(st[1], buf[1], N, blocks, cycles)};
      for (int i = 0; i < 2; i++) { hipGraphLaunch(ge[i], st[i]); hipStreamSynchronize(st[i]); }
      // one stream
      double t0 = now_us();
      for (int r = 0; r < REPS; r++) hipGraphLaunch(ge[0], st[0]);
      double t_host1 = now_us() - t0;
      hipStreamSynchronize(st[0]);
      double t_one = now_us() - t0;
      // two streams, one host thread
      t0 = now_us();
      for (int r = 0; r < REPS; r++) { hipGraphLaunch(ge[0], st[0]); hipGraphLaunch(ge[1], st[1]); }
      double t_host2 = now_us() - t0;
      hipStreamSynchronize(st[0]); hipStreamSynchronize(st[1]);
      double t_two = now_us() - t0;
      // two streams, two host threads
      t0 = now_us();
      std::thread th([&] { hipSetDevice(0); for (int r = 0; r < REPS; r++) hipGraphLaunch(ge[1], st[1]); hipStreamSynchronize(st[1]); });
      for (int r = 0; r < REPS; r++) hipGraphLaunch(ge[0], st[0]);
      hipStreamSynchronize(st[0]);
      th.join();
      double t_thr = now_us() - t0;
      printf("blocks %5d cycles %6ld | 1 stream: %8.1f us/graph (host %6.1f) | 2 streams 1 thread: %8.1f us/pair (host %6.1f) | 2 threads: %8.1f us/pair\n",
             blocks, cycles, t_one / REPS, t_host1 / REPS, t_two / REPS, t_host2 / REPS, t_thr / REPS);
      // eager, two streams interleaved kernel by kernel
      t0 = now_us();
      for (int r = 0; r < 4; r++)
        for (int i = 0; i < N; i++) {
          hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st[0], buf[0], cycles);
          hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st[1], buf[1], cycles);
        }
      hipStreamSynchronize(st[0]); hipStreamSynchronize(st[1]);
      printf("   eager interleaved 2 streams: %8.1f us/pair-of-chains\n", (now_us() - t0) / 4);
      hipGraphExecDestroy(ge[0]); hipGraphExecDestroy(ge[1]);
    }
  }
  return 0;
}
