// resample.hip -- K1 of SURVEY.md section 2a: band-limited sinc resampling to 16 kHz, the device side of
// `torchaudio.transforms.Resample(orig_freq, new_freq)(signal)` as the reference applies it to every file that is not already
// 16 kHz (/root/reference/back/api.py:824-830).  Algorithm = torchaudio.functional.resample with its defaults (sinc_interp_hann,
// lowpass_filter_width 6, rolloff 0.99) [UPSTREAM-RECALL]; CPU restatement: oracle/resample_ref.py.
//
// With o = orig / gcd, n = new / gcd, one input "frame" of o samples yields n output samples; output sample f*n + p is the dot
// product of phase p's filter (taps = 2*width + o values) with the zero-padded input starting at sample f*o - width.  The
// polyphase table is built once on the host (float64 -> float32, as upstream) and handed over TRANSPOSED, [taps][n], so that
// the threads of a wave -- consecutive output samples, i.e. consecutive phases -- read consecutive table entries.
// HBM-bound byte work: a block stages the input span of its 256 outputs in LDS once (coalesced, zero padded) and streams the
// table (76 K floats for 44.1 k -> 16 k: L2-resident) through it; nothing is reshaped into a GEMM.
#include "../../include/ccx.h"
#include "ccx_common.h"

namespace {

__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, long stride_in, const int* __restrict__ n_in,
                                                            const float* __restrict__ kernT, int o, int n, int width, int taps,
                                                            float* __restrict__ y, long stride_out, const int* __restrict__ n_out) {
  extern __shared__ float xs[];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int len_in = n_in[b], len_out = n_out[b];
  const int o0 = blockIdx.x * 256;
  if (o0 >= len_out) return;                       // uniform per block
  const int last = (o0 + 255 < len_out ? o0 + 255 : len_out - 1);
  const int f0 = o0 / n, f1 = last / n;
  const long s0 = (long)f0 * o - width;            // first input sample of the span (may be negative: left padding)
  const int span = (f1 - f0) * o + taps;
  const float* xr = x + (long)b * stride_in;
  for (int i = tid; i < span; i += 256) {
    const long s = s0 + i;
    xs[i] = (s >= 0 && s < len_in) ? xr[s] : 0.f;
  }
  __syncthreads();
  const int oi = o0 + tid;
  if (oi >= len_out) return;
  const int f = oi / n, p = oi - f * n;
  const float* xf = xs + (f - f0) * o;
  const float* kt = kernT + p;
  float acc = 0.f;
  int k = 0;
  for (; k + 4 <= taps; k += 4) {                  // 4 independent table loads in flight per thread
    const float k0 = kt[(long)k * n], k1 = kt[(long)(k + 1) * n], k2 = kt[(long)(k + 2) * n], k3 = kt[(long)(k + 3) * n];
    acc = fmaf(k0, xf[k], acc);
    acc = fmaf(k1, xf[k + 1], acc);
    acc = fmaf(k2, xf[k + 2], acc);
    acc = fmaf(k3, xf[k + 3], acc);
  }
  for (; k < taps; k++) acc = fmaf(kt[(long)k * n], xf[k], acc);
  y[(long)b * stride_out + oi] = acc;
}

}  // namespace

extern "C" int ccx_resample_sinc(ccx_ctx* ctx, const float* x_dev, int64_t stride_in, const int* n_in_dev, int B, int orig, int new_,
                                 int width, const float* kernT_dev, float* y_dev, int64_t stride_out, const int* n_out_dev,
                                 int max_out, void* stream) {
  if (!ctx) return CCX_ERR_ARG;
  CCX_REQUIRE(ctx, x_dev && n_in_dev && kernT_dev && y_dev && n_out_dev && B >= 1, "resample: null argument");
  CCX_REQUIRE(ctx, orig >= 1 && new_ >= 1 && width >= 1 && max_out >= 0 && stride_out >= max_out, "resample: bad geometry");
  if (max_out == 0) return CCX_OK;
  const int taps = 2 * width + orig;
  // widest span of a block: its 256 outputs touch at most 255 / new + 2 frames
  const long span = (long)(255 / new_ + 1) * orig + taps;
  CCX_REQUIRE(ctx, span * 4 <= 64 * 1024, "resample: %d -> %d needs %ld bytes of LDS per block (ratio not supported)", orig, new_, span * 4);
  dim3 grid(ccx_cdiv(max_out, 256), B);
  {
    ccx_prof_scope ps(ctx, (hipStream_t)stream, "resample_sinc_kernel", 2.0 * B * (double)max_out * taps,
                      4.0 * B * ((double)max_out * orig / new_ + max_out));
    hipLaunchKernelGGL(resample_sinc_kernel, grid, dim3(256), (size_t)span * 4, (hipStream_t)stream, x_dev, (long)stride_in, n_in_dev,
                       kernT_dev, orig, new_, width, taps, y_dev, (long)stride_out, n_out_dev);
  }
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}
