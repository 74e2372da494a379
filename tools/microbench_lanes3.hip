// microbench_lanes3.hip -- when do hipGraph replays on two concurrent-capable streams stop overlapping?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
struct Big { long cycles; float* a; int pad[40]; };
__global__ void k_spin(float* a, long cycles) {
  const long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (threadIdx.x == 0) a[blockIdx.x] += 1.f;
}
__global__ void k_spin_big(Big p) {
  const long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < p.cycles) {}
  if (threadIdx.x == 0) p.a[blockIdx.x] += 1.f + p.pad[3];
}
// occupies a whole CU's LDS: at most one block per CU -> a 256-block launch fills the chip
__global__ void k_spin_lds(float* a, long cycles) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = 1.f;
  const long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (threadIdx.x == 0) a[blockIdx.x] += sm[5];
}
// streams memory: HBM-bound like the cross attention
__global__ void k_stream(const float4* __restrict__ src, float* a, long n4) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.f) a[0] = s;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F>
static hipGraphExec_t make_graph(hipStream_t st, F body) {
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  body(st);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  return ge;
}
static double run(hipGraphExec_t a, hipStream_t sa, hipGraphExec_t b, hipStream_t sb, int reps) {
  (void)hipStreamSynchronize(sa); if (b) (void)hipStreamSynchronize(sb);
  const double t0 = now_us();
  for (int r = 0; r < reps; r++) { (void)hipGraphLaunch(a, sa); if (b) (void)hipGraphLaunch(b, sb); }
  (void)hipStreamSynchronize(sa); if (b) (void)hipStreamSynchronize(sb);
  return (now_us() - t0) / reps;
}
int main() {
  const int NS = 6;
  hipStream_t st[NS]; float* buf[NS];
  for (int i = 0; i < NS; i++) { (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); (void)hipMalloc(&buf[i], 1 << 22); (void)hipMemset(buf[i], 0, 1 << 22); }
  float4* big; const long n4 = (1L << 30) / 16; (void)hipMalloc(&big, n4 * 16); (void)hipMemset(big, 0, n4 * 16);
  (void)hipFuncSetAttribute((const void*)k_spin_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  // find a concurrent pair with the simple graph
  int A = 0, Bq = -1;
  {
    std::vector<hipGraphExec_t> ge(NS);
    for (int i = 0; i < NS; i++) ge[i] = make_graph(st[i], [&](hipStream_t s) { for (int k = 0; k < 50; k++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, buf[i], 20000L); });
    const double single = run(ge[0], st[0], nullptr, nullptr, 5);
    for (int j = 1; j < NS; j++) {
      const double pr = run(ge[0], st[0], ge[j], st[j], 5);
      printf("pair (0,%d): %.0f vs single %.0f\n", j, pr, single);
      if (Bq < 0 && pr < 1.4 * single) Bq = j;
    }
  }
  if (Bq < 0) { printf("no concurrent pair\n"); return 0; }
  printf("using streams 0 and %d\n", Bq);
  auto test = [&](const char* name, auto body) {
    hipGraphExec_t g0 = make_graph(st[A], [&](hipStream_t s) { body(s, buf[A]); });
    hipGraphExec_t g1 = make_graph(st[Bq], [&](hipStream_t s) { body(s, buf[Bq]); });
    (void)hipGraphLaunch(g0, st[A]); (void)hipGraphLaunch(g1, st[Bq]);
    const double s1 = run(g0, st[A], nullptr, nullptr, 8), p = run(g0, st[A], g1, st[Bq], 8);
    printf("%-48s single %8.1f us  pair %8.1f us  ratio %.2f\n", name, s1, p, p / s1);
  };
  test("135 x spin 1 block", [&](hipStream_t s, float* b) { for (int k = 0; k < 135; k++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, b, 30000L); });
  test("135 x spin 2304 blocks x 256", [&](hipStream_t s, float* b) { for (int k = 0; k < 135; k++) hipLaunchKernelGGL(k_spin, dim3(2304), dim3(256), 0, s, b, 30000L); });
  test("135 x spin big kernarg", [&](hipStream_t s, float* b) { Big p{}; p.cycles = 30000; p.a = b; for (int k = 0; k < 135; k++) hipLaunchKernelGGL(k_spin_big, dim3(64), dim3(256), 0, s, p); });
  test("135 x spin 256 blocks x 160KB LDS (chip full)", [&](hipStream_t s, float* b) { for (int k = 0; k < 135; k++) hipLaunchKernelGGL(k_spin_lds, dim3(256), dim3(256), 160 * 1024, s, b, 30000L); });
  test("135 x spin 128 blocks x 160KB LDS (half chip)", [&](hipStream_t s, float* b) { for (int k = 0; k < 135; k++) hipLaunchKernelGGL(k_spin_lds, dim3(128), dim3(256), 160 * 1024, s, b, 30000L); });
  test("12 x (10 spin + 1 GB stream)", [&](hipStream_t s, float* b) {
    for (int l = 0; l < 12; l++) {
      for (int k = 0; k < 10; k++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, b, 15000L);
      hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s, big, b, n4 / 2);
    }
  });
  return 0;
}
