// microbench_lanes2.hip -- which stream pairs overlap their hipGraph replays? (HW queue / pipe mapping probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_spin(float* a, long cycles) {
  const long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (threadIdx.x == 0) a[blockIdx.x] += 1.f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static hipGraphExec_t make_graph(hipStream_t st, float* buf, int n, int blocks, long cycles) {
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, buf, cycles);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  return ge;
}
int main() {
  const int NS = 8, N = 135, REPS = 10;
  float* buf[NS]; hipStream_t st[NS]; hipGraphExec_t ge[NS];
  for (int i = 0; i < NS; i++) {
    (void)hipMalloc(&buf[i], 1 << 20); (void)hipMemset(buf[i], 0, 1 << 20);
    (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    ge[i] = make_graph(st[i], buf[i], N, 64, 20000);
    (void)hipGraphLaunch(ge[i], st[i]); (void)hipStreamSynchronize(st[i]);
  }
  double t0 = now_us();
  for (int r = 0; r < REPS; r++) (void)hipGraphLaunch(ge[0], st[0]);
  (void)hipStreamSynchronize(st[0]);
  printf("single: %.1f us/graph\n", (now_us() - t0) / REPS);
  for (int i = 0; i < NS; i++)
    for (int j = i + 1; j < NS; j++) {
      t0 = now_us();
      for (int r = 0; r < REPS; r++) { (void)hipGraphLaunch(ge[i], st[i]); (void)hipGraphLaunch(ge[j], st[j]); }
      (void)hipStreamSynchronize(st[i]); (void)hipStreamSynchronize(st[j]);
      printf("pair (%d,%d): %.1f us/pair\n", i, j, (now_us() - t0) / REPS);
    }
  return 0;
}
