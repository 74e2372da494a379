// logmel.hip -- Whisper log-mel front end (K4 in SURVEY.md section 2a) on CDNA4.
//
// Follows openai-whisper audio.py::log_mel_spectrogram as the reference calls it through
// whisper.transcribe (reference back/api.py:1286, 1432, 1474) [UPSTREAM-RECALL, restated in
// oracle/whisper_ref.py::log_mel_spectrogram]:
//   pad 480000 zeros -> STFT(n_fft 400, hop 160, periodic Hann, center, reflect) -> drop last
//   frame -> |.|^2 -> mel[80x201] -> log10(clamp 1e-10) -> max(x, max-8) -> (x+4)/4.
//
// Pass 1 (logmel_power_kernel): a block owns 32 consecutive frames.  The 5360 samples they
// touch are staged once into LDS (coalesced HBM reads, reflect/zero padding resolved at stage
// time); each lane owns one DFT bin and accumulates all 32 frames from ds_read_b128 sample
// quads against a window-folded DFT table (L2-resident, coalesced along bins).  The power
// spectrum goes back through LDS for the sparse mel projection; the per-clip maximum is
// reduced in-block and merged with one atomicMax.
// Pass 2 (logmel_finalize_kernel): applies the max-8 floor and (x+4)/4 scaling for a 3000-frame
// window and writes (a) the fp32 mel for parity and (b) the bf16 im2col matrix
// [B*3000, 256] (3 taps x 80 mels, zero padded to K=256) consumed by the conv1 MFMA GEMM.
#include "logmel.h"

#define LM_FRAMES 32
#define LM_NFFT 400
#define LM_HOP 160
#define LM_BINS 201
#define LM_SPAN (LM_HOP * (LM_FRAMES - 1) + LM_NFFT)  // 5360 samples
#define LM_PSTRIDE 204

__global__ __launch_bounds__(256) void logmel_power_kernel(
    const float* __restrict__ audio, long audio_stride, const int* __restrict__ n_samples,
    const float* __restrict__ dft_cos,  // [400][208] window folded in
    const float* __restrict__ dft_sin,  // [400][208]
    const float* __restrict__ mel_fb,   // [80][208]
    const int* __restrict__ mel_range,  // [80][2]  first bin, one-past-last bin
    float* __restrict__ raw,            // [B][80][Fraw]  (only frames < gridDim.x*32 are written)
    unsigned int* __restrict__ gmax_bits, int Fraw) {
  __shared__ __attribute__((aligned(16))) float xs[LM_SPAN];
  __shared__ __attribute__((aligned(16))) float pw[LM_FRAMES * LM_PSTRIDE];
  __shared__ float red[4];
  const int b = blockIdx.y, f0 = blockIdx.x * LM_FRAMES, tid = threadIdx.x;
  const int n = n_samples[b];
  const long ntot = (long)n + 480000;
  const int total_frames = (int)(ntot / LM_HOP);  // after dropping the last STFT frame
  const float* x = audio + (long)b * audio_stride;
  // stage samples: padded index i -> original o = i - 200 (reflect at both ends of the padded signal)
  const long i0 = (long)f0 * LM_HOP;
  for (int i = tid; i < LM_SPAN; i += 256) {
    long o = i0 + i - 200;
    if (o < 0) o = -o;
    if (o >= ntot) o = 2 * (ntot - 1) - o;
    xs[i] = (o < n) ? x[o] : 0.f;
  }
  __syncthreads();

  float re[LM_FRAMES], im[LM_FRAMES];
#pragma unroll
  for (int f = 0; f < LM_FRAMES; f++) { re[f] = 0.f; im[f] = 0.f; }
  if (tid < LM_BINS) {
    for (int nn = 0; nn < LM_NFFT; nn += 4) {
      float c[4], s[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        c[j] = dft_cos[(nn + j) * 208 + tid];
        s[j] = dft_sin[(nn + j) * 208 + tid];
      }
#pragma unroll
      for (int f = 0; f < LM_FRAMES; f++) {
        const float4 xv = *(const float4*)(xs + f * LM_HOP + nn);
        re[f] = fmaf(xv.x, c[0], re[f]); im[f] = fmaf(xv.x, s[0], im[f]);
        re[f] = fmaf(xv.y, c[1], re[f]); im[f] = fmaf(xv.y, s[1], im[f]);
        re[f] = fmaf(xv.z, c[2], re[f]); im[f] = fmaf(xv.z, s[2], im[f]);
        re[f] = fmaf(xv.w, c[3], re[f]); im[f] = fmaf(xv.w, s[3], im[f]);
      }
    }
#pragma unroll
    for (int f = 0; f < LM_FRAMES; f++) pw[f * LM_PSTRIDE + tid] = re[f] * re[f] + im[f] * im[f];
  }
  __syncthreads();

  // mel projection: 80 x 32 outputs, 10 per thread; thread -> frame = tid & 31, mel = (tid >> 5) + 8*i
  float lmax = -10.f;
  const int f = tid & 31;
#pragma unroll 1
  for (int i = 0; i < 10; i++) {
    const int m = (tid >> 5) + 8 * i;
    const int k0 = mel_range[2 * m], k1 = mel_range[2 * m + 1];
    float acc = 0.f;
    for (int k = k0; k < k1; k++) acc = fmaf(mel_fb[m * 208 + k], pw[f * LM_PSTRIDE + k], acc);
    const float v = log10f(fmaxf(acc, 1e-10f));
    const int fr = f0 + f;
    if (fr < Fraw) raw[((long)b * 80 + m) * Fraw + fr] = v;
    if (fr < total_frames) lmax = fmaxf(lmax, v);
  }
  lmax = wave_reduce_max(lmax);
  if ((tid & 63) == 0) red[tid >> 6] = lmax;
  __syncthreads();
  if (tid == 0) {
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(gmax_bits + b, __float_as_uint(mx + 16.0f));  // mx >= -10 -> positive float, bit order == value order
  }
}

__global__ void logmel_init_max_kernel(unsigned int* gmax_bits, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) gmax_bits[i] = __float_as_uint(6.0f);  // -10 + 16
}

// One block = 64 frames of one clip.
__global__ __launch_bounds__(256) void logmel_finalize_kernel(
    const float* __restrict__ raw, const unsigned int* __restrict__ gmax_bits, const int* __restrict__ n_samples,
    const int* __restrict__ seek, const int* __restrict__ seg_len, int Fraw, int Fcomp,
    float* __restrict__ mel_out,   // [B][80][3000] or nullptr
    bf16_t* __restrict__ im2col) { // [B*3000][256] or nullptr
  __shared__ float tile[80][67];   // frames t0-1 .. t0+64 (66 used)
  const int b = blockIdx.y, t0 = blockIdx.x * 64, tid = threadIdx.x;
  const float floorv = __uint_as_float(gmax_bits[b]) - 16.0f - 8.0f;
  const int s0 = seek ? seek[b] : 0;
  const int n = n_samples[b];
  const int total_frames = (int)(((long)n + 480000) / LM_HOP);
  // transcribe.py: mel_segment = mel[:, seek : seek + segment_size]; pad_or_trim(mel_segment, 3000)
  // -> frames past segment_size are literal zeros, not log-floor values.
  const int valid = seg_len ? seg_len[b] : 3000;
  for (int i = tid; i < 80 * 66; i += 256) {
    const int c = i / 66, j = i - c * 66;
    const int t = t0 - 1 + j;  // frame inside the window
    float v = 0.f;             // conv zero padding outside [0, 3000)
    if (t >= 0 && t < valid) {
      const int fr = s0 + t;
      float r = -10.f;  // frames past the computed range are pure zero padding: log10(1e-10)
      if (fr < Fcomp && fr < total_frames) r = raw[((long)b * 80 + c) * Fraw + fr];
      // frames beyond the (padded) signal do not exist in the reference; pad_or_trim pads with 0
      v = (fr < total_frames) ? (fmaxf(r, floorv) + 4.0f) * 0.25f : 0.f;
    }
    tile[c][j] = v;
  }
  __syncthreads();
  if (mel_out) {
    for (int i = tid; i < 80 * 64; i += 256) {
      const int c = i >> 6, j = i & 63;
      const int t = t0 + j;
      if (t < 3000) mel_out[((long)b * 80 + c) * 3000 + t] = tile[c][j + 1];
    }
  }
  if (im2col) {
    for (int i = tid; i < 64 * 256; i += 256) {
      const int j = i >> 8, k = i & 255;
      const int t = t0 + j;
      if (t >= 3000) continue;
      float v = 0.f;
      if (k < 240) {
        const int tap = k / 80, c = k - tap * 80;
        v = tile[c][j + tap];  // frame t - 1 + tap
      }
      im2col[((long)b * 3000 + t) * 256 + k] = f32_to_bf16(v);
    }
  }
}

int ccx_launch_logmel(ccx_ctx* ctx, const LogmelTables& tb, const float* audio, long audio_stride,
                      const int* n_samples_dev, const int* seek_dev, const int* seg_len_dev, int B, int Fraw, int Fcomp, float* raw,
                      unsigned int* gmax_bits, float* mel_out, bf16_t* im2col, hipStream_t stream) {
  CCX_REQUIRE(ctx, B > 0 && Fraw > 0 && Fraw % LM_FRAMES == 0 && Fcomp > 0 && Fcomp <= Fraw && Fcomp % LM_FRAMES == 0,
              "logmel: bad B=%d / Fraw=%d / Fcomp=%d", B, Fraw, Fcomp);
  hipLaunchKernelGGL(logmel_init_max_kernel, dim3(ccx_cdiv(B, 64)), dim3(64), 0, stream, gmax_bits, B);
  CCX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(logmel_power_kernel, dim3(Fcomp / LM_FRAMES, B), dim3(256), 0, stream, audio, audio_stride,
                     n_samples_dev, tb.dft_cos, tb.dft_sin, tb.mel_fb, tb.mel_range, raw, gmax_bits, Fraw);
  CCX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(logmel_finalize_kernel, dim3(ccx_cdiv(3000, 64), B), dim3(256), 0, stream, raw, gmax_bits,
                     n_samples_dev, seek_dev, seg_len_dev, Fraw, Fcomp, mel_out, im2col);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}
