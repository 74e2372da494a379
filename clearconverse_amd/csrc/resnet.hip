// resnet.hip -- WeSpeaker ResNet-34 speaker embedder (K21 in SURVEY.md section 2a): the embedding model inside
// `pyannote/speaker-diarization-3.1`, which the reference loads at back/api.py:788-792 and calls at
// back/api.py:1056-1060 and 1124-1128.  Restated in oracle/wespeaker_ref.py.
//
//   waveform chunks -> Kaldi fbank (25 ms / 10 ms, 512-point FFT, 80 mel bins, log) -> per-chunk mean normalisation
//   -> conv1 (1 -> 32, 3x3) -> 16 basic blocks [3,4,6,3] x (32,64,128,256), stride 2 between stages
//   -> weighted statistics pooling over time (mean || std of 256 x 10 features) -> Linear(5120, 256)
//
// Layout.  Activations are NHWC bf16 with a one-pixel zero halo: [chunk][H+2][W+2][C], H = mel axis, W = time.  A 3x3
// convolution is then ONE launch of the bf16 MFMA GEMM (gemm_bf16.hip) with three accumulated taps (one per kernel
// row): for tap kh the three horizontal neighbours x C channels of an output pixel are 3C CONTIGUOUS elements, so the
// A operand is a strided view of the activation tensor itself (lda = stride*C, tap stride = one padded row) and no
// im2col buffer exists.  Rows of the view that fall on halo positions are dropped by the GEMM's two-level row remap,
// which also writes straight into the next layer's padded layout.  BatchNorm is folded into the weights/bias at load
// time, ReLU and the residual add are GEMM epilogues (EPI_BF16_RELU / EPI_BF16_ADD_RELU).  With 32 input channels a
// tap is padded from 96 to 128 columns with zero weights (the view then covers a fourth, ignored pixel).
//
// The convolutional trunk depends only on the chunk, not on the speaker mask, so it runs once per chunk and the
// pooling + linear run once per (chunk, local speaker) mask.
#include <map>
#include <string>
#include <vector>
#include <math.h>
#include "../../include/ccx.h"
#include "ccx_common.h"
#include "gemm_bf16.h"

namespace {

constexpr int FB_LEN = 400, FB_SHIFT = 160, FB_NFFT = 512, FB_MELS = 80, FB_MELW = 64;
constexpr int RN_C0 = 32, RN_EMB = 256, RN_STAGES = 4;
constexpr int RN_BLOCKS[RN_STAGES] = {3, 4, 6, 3};

#define RTRY(expr)        \
  do {                    \
    int _rc = (expr);     \
    if (_rc) return _rc;  \
  } while (0)

// ------------------------------------------------------------------------------------------------------------
// Kaldi fbank: one wave per frame, 4 frames per block.  512-point radix-2 FFT in LDS.
// feats_t [chunk][80][T] f32 (mel-major so that the mean pass and conv1 read along time).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wav, long stride, int T, const float* __restrict__ window,
                                                    const float2* __restrict__ twiddle, const float* __restrict__ melw,
                                                    const int* __restrict__ mel_start, const int* __restrict__ mel_len,
                                                    float* __restrict__ feats_t) {
  __shared__ float2 buf[4][FB_NFFT];
  __shared__ float2 tw[FB_NFFT / 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < FB_NFFT / 2; i += 256) tw[i] = twiddle[i];
  const int chunk = blockIdx.y;
  const int t = blockIdx.x * 4 + wave;
  const bool live = t < T;
  const float* x = wav + (long)chunk * stride + (long)(live ? t : 0) * FB_SHIFT;
  float2* b = buf[wave];
  // scaled samples, DC removal
  float v[7];
  float sm = 0.f;
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const int i = lane + 64 * j;
    v[j] = i < FB_LEN ? x[i] * 32768.f : 0.f;
    sm += v[j];
  }
  const float mean = wave_reduce_sum(sm) / (float)FB_LEN;
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const int i = lane + 64 * j;
    if (i < FB_LEN) b[i].y = v[j] - mean;     // park the DC-free frame in the imaginary lane of the buffer
  }
  __syncthreads();
  // pre-emphasis (replicate padding on the left) and Hamming window, then the bit-reversed FFT input
  float z[7];
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const int i = lane + 64 * j;
    z[j] = 0.f;
    if (i < FB_LEN) z[j] = (b[i].y - 0.97f * b[i > 0 ? i - 1 : 0].y) * window[i];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = lane + 64 * j;
    const int r = (int)(__brev((unsigned)i) >> 23);   // 9-bit reversal
    b[r] = make_float2(j < 7 ? z[j] : 0.f, 0.f);       // samples 400..511 are zero padding
  }
  __syncthreads();
  for (int s = 1; s <= 9; s++) {
    const int half = 1 << (s - 1);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int bf = lane + 64 * q;                    // butterfly index 0..255
      const int grp = bf >> (s - 1), pos = bf & (half - 1);
      const int i0 = (grp << s) + pos, i1 = i0 + half;
      const float2 w = tw[pos << (9 - s)];
      const float2 a = b[i0], c = b[i1];
      const float2 m = make_float2(c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x);
      b[i0] = make_float2(a.x + m.x, a.y + m.y);
      b[i1] = make_float2(a.x - m.x, a.y - m.y);
    }
    __syncthreads();
  }
  // power spectrum in place (bins 0..255; the Nyquist bin has zero mel weight)
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int k = lane + 64 * q;
    const float2 c = b[k];
    b[k].x = c.x * c.x + c.y * c.y;
  }
  __syncthreads();
  for (int m = lane; m < FB_MELS; m += 64) {
    const int st = mel_start[m], ln = mel_len[m];
    float acc = 0.f;
    for (int k = 0; k < ln; k++) acc = fmaf(melw[m * FB_MELW + k], b[st + k].x, acc);
    if (live) feats_t[((long)chunk * FB_MELS + m) * T + t] = logf(fmaxf(acc, 1.1920929e-07f));
  }
}

// mean over time of every (chunk, mel) row
__global__ __launch_bounds__(256) void fbank_mean_kernel(const float* __restrict__ feats_t, int T, float* __restrict__ mean) {
  __shared__ float sh[4];
  const float* p = feats_t + (long)blockIdx.x * T;
  float s = 0.f;
  for (int i = threadIdx.x; i < T; i += 256) s += p[i];
  s = wave_reduce_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) mean[blockIdx.x] = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)T;
}

// conv1: 1 -> 32 channels, 3x3, pad 1, folded BatchNorm, ReLU.  One thread per output pixel (all 32 channels).
// out: padded NHWC [chunk][82][T + 2][32] bf16.
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ feats_t, const float* __restrict__ mean, int T,
                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                    bf16_t* __restrict__ out) {
  __shared__ float sw[RN_C0 * 9 + RN_C0];
  for (int i = threadIdx.x; i < RN_C0 * 9 + RN_C0; i += 256) sw[i] = i < RN_C0 * 9 ? w[i] : bias[i - RN_C0 * 9];
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y, chunk = blockIdx.z;
  if (t >= T) return;
  float x[9];
#pragma unroll
  for (int kh = 0; kh < 3; kh++) {
    const int ff = f + kh - 1;
    const bool fin = ff >= 0 && ff < FB_MELS;
    const float mu = fin ? mean[chunk * FB_MELS + ff] : 0.f;
    const float* row = feats_t + ((long)chunk * FB_MELS + (fin ? ff : 0)) * T;
#pragma unroll
    for (int kw = 0; kw < 3; kw++) {
      const int tt = t + kw - 1;
      x[kh * 3 + kw] = (fin && tt >= 0 && tt < T) ? row[tt] - mu : 0.f;
    }
  }
  const long Wp = T + 2;
  bf16_t* dst = out + (((long)chunk * (FB_MELS + 2) + f + 1) * Wp + t + 1) * RN_C0;
#pragma unroll
  for (int c8 = 0; c8 < RN_C0 / 8; c8++) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int c = c8 * 8 + j;
      float a = sw[RN_C0 * 9 + c];
#pragma unroll
      for (int k = 0; k < 9; k++) a = fmaf(sw[c * 9 + k], x[k], a);
      o[j] = fmaxf(a, 0.f);
    }
    uint4 pk;
    pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]); pk.z = pack_bf16x2(o[4], o[5]); pk.w = pack_bf16x2(o[6], o[7]);
    ((uint4*)dst)[c8] = pk;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Stage-1 convolution (32 -> 32 channels, 3x3, stride 1) as a direct MFMA kernel.  Through the GEMM these six layers are
// the slowest of the network: N = 32 leaves half a tile idle, a tap is padded 96 -> 128 and every pixel is DMA-ed into
// LDS twelve times.  Here a block stages its (4 + 2) x (64 + 2) pixel halo tile ONCE (96-byte pixel pitch: the
// ds_read_b128 fragment reads are conflict-free, tools/lds_bank_sim.py), every wave keeps all 9 x 32 x 32 weights in 72
// VGPRs, and one output row of 64 pixels is 72 v_mfma_f32_16x16x32_bf16 (K = the 32 input channels of one tap).
// Operands are swapped (M = output channels, permuted so that a lane ends up with 8 consecutive channels of one pixel:
// one 16-byte store), folded-BatchNorm bias, optional residual, ReLU in registers.
// ------------------------------------------------------------------------------------------------------------
template <bool HAS_RESID>
__global__ __launch_bounds__(256) void conv3x3_c32_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ wpk,   // [32][3][128]
                                                          const float* __restrict__ bias, const bf16_t* __restrict__ resid,
                                                          bf16_t* __restrict__ out, int H, int W) {
  constexpr int TW = 64, TH = 4, PITCH = 96;                       // bytes per staged pixel (64 used)
  __shared__ __attribute__((aligned(16))) char tile[(TH + 2) * (TW + 2) * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, hq = lane >> 4;
  const long Wp = W + 2;
  const int y0 = blockIdx.y * TH, img = blockIdx.z;
  const bf16_t* src = in + ((long)img * (H + 2) + y0) * Wp * 32;
  // A block walks its 4-row strip in x tiles of 64 pixels: the weights are fetched once, and the halo tile (and the
  // residual pixels) of tile k+1 travel HBM -> registers while tile k is multiplied, so only the first tile of a
  // strip waits for memory (one short-lived block per tile spent most of its life in that wait).
  constexpr int NCH = (TH + 2) * (TW + 2) * 4, NIT = (NCH + 255) / 256;
  uint4 stg[NIT];
  uint4 rres[4];
  const long orow = ((long)img * (H + 2) + y0 + wave + 1) * Wp + 1;
  auto fetch = [&](int x0) {       // 6 rows x 66 pixels x 4 chunks of 16 B
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int idx = tid + 256 * it;
      const int row = idx / ((TW + 2) * 4), rem = idx - row * (TW + 2) * 4;
      const int px = rem >> 2, q = rem & 3;
      stg[it] = make_uint4(0, 0, 0, 0);
      if (idx < NCH && x0 + px < Wp) stg[it] = *(const uint4*)(src + ((long)row * Wp + x0 + px) * 32 + q * 8);
    }
    if (HAS_RESID) {
#pragma unroll
      for (int pt = 0; pt < 4; pt++) {
        const int x = x0 + pt * 16 + l15;
        rres[pt] = make_uint4(0, 0, 0, 0);
        if (x < W) rres[pt] = *(const uint4*)(resid + (orow + x) * 32 + 8 * hq);
      }
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int idx = tid + 256 * it;
      const int row = idx / ((TW + 2) * 4), rem = idx - row * (TW + 2) * 4;
      const int px = rem >> 2, q = rem & 3;
      if (idx < NCH) *(uint4*)(tile + (row * (TW + 2) + px) * PITCH + q * 16) = stg[it];
    }
  };
  fetch(0);
  // ---- weights: A operand of tap t, channel block j: row i = l15 -> output channel 8 (i >> 2) + 4 j + (i & 3) ----
  bf16x8 wf[9][2];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int co = 8 * (l15 >> 2) + 4 * j + (l15 & 3);
      wf[t][j] = *(const bf16x8*)(wpk + (long)co * 384 + (t / 3) * 128 + (t % 3) * 32 + hq * 8);
    }
  float bv[8];
#pragma unroll
  for (int i = 0; i < 8; i++) bv[i] = bias[8 * hq + i];
  const int n_tiles = (W + TW - 1) / TW;
  for (int k = 0; k < n_tiles; k++) {
    const int x0 = k * TW;
    put();
    uint4 rcur[4];
    if (HAS_RESID) {
#pragma unroll
      for (int pt = 0; pt < 4; pt++) rcur[pt] = rres[pt];
    }
    __syncthreads();
    if (k + 1 < n_tiles) fetch(x0 + TW);     // in flight during this tile's MFMAs
    f32x4 acc[4][2];
#pragma unroll
    for (int pt = 0; pt < 4; pt++)
#pragma unroll
      for (int j = 0; j < 2; j++) acc[pt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int kh = t / 3, kw = t % 3;
#pragma unroll
      for (int pt = 0; pt < 4; pt++) {
        // B operand: column = pixel pt*16 + l15, k-slice hq: 8 input channels of the tap's pixel
        const bf16x8 bfrag = *(const bf16x8*)(tile + ((wave + kh) * (TW + 2) + pt * 16 + l15 + kw) * PITCH + hq * 16);
#pragma unroll
        for (int j = 0; j < 2; j++) acc[pt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][j], bfrag, acc[pt][j], 0, 0, 0);
      }
    }
    // ---- epilogue: lane (pixel l15, quarter hq) holds output channels 8 hq + 4 j + r ----
#pragma unroll
    for (int pt = 0; pt < 4; pt++) {
      const int x = x0 + pt * 16 + l15;
      if (x >= W) continue;
      float v[8];
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) v[4 * j + r] = acc[pt][j][r] + bv[4 * j + r];
      const long off = (orow + x) * 32 + 8 * hq;
      if (HAS_RESID) {
        const uint32_t rw[4] = {rcur[pt].x, rcur[pt].y, rcur[pt].z, rcur[pt].w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          v[2 * i] += __uint_as_float(rw[i] << 16);
          v[2 * i + 1] += __uint_as_float(rw[i] & 0xffff0000u);
        }
      }
      uint4 o;
      o.x = pack_bf16x2(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)); o.y = pack_bf16x2(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
      o.z = pack_bf16x2(fmaxf(v[4], 0.f), fmaxf(v[5], 0.f)); o.w = pack_bf16x2(fmaxf(v[6], 0.f), fmaxf(v[7], 0.f));
      *(uint4*)(out + off) = o;
    }
    __syncthreads();     // every wave is done with the tile before the next one is written over it
  }
}

// Weighted statistics pooling (pyannote StatsPool / wespeaker TSTP) over the W4 frames of the last stage.
// grid (10, n_masks), 256 threads = channels.  pooled [mask][5120]: mean at c*10 + h, std at 2560 + c*10 + h.
__global__ __launch_bounds__(256) void tstp_kernel(const bf16_t* __restrict__ act, int W4, const int* __restrict__ mask_chunk,
                                                   const float* __restrict__ weights, int n_w, float* __restrict__ pooled) {
  const int h = blockIdx.x, j = blockIdx.y, c = threadIdx.x;
  const int chunk = mask_chunk ? mask_chunk[j] : j;
  const long Wp = W4 + 2;
  const bf16_t* base = act + (((long)chunk * 12 + h + 1) * Wp + 1) * 256 + c;
  const float* wj = weights ? weights + (long)j * n_w : nullptr;
  float v1 = 0.f, v2 = 0.f, sx = 0.f;
  const float scale = (float)n_w / (float)W4;   // F.interpolate(mode="nearest"): src = min(floor(dst * in/out), in - 1)
  for (int t = 0; t < W4; t++) {
    const float w = wj ? wj[min((int)floorf(t * scale), n_w - 1)] : 1.f;
    v1 += w; v2 += w * w;
    sx = fmaf(w, bf16_to_f32(base[(long)t * 256]), sx);
  }
  v1 += 1e-8f;
  const float mean = sx / v1;
  float sq = 0.f;
  for (int t = 0; t < W4; t++) {
    const float w = wj ? wj[min((int)floorf(t * scale), n_w - 1)] : 1.f;
    const float d = bf16_to_f32(base[(long)t * 256]) - mean;
    sq = fmaf(w, d * d, sq);
  }
  const float var = sq / (v1 - v2 / v1 + 1e-8f);
  float* o = pooled + (long)j * 5120;
  o[c * 10 + h] = mean;
  o[2560 + c * 10 + h] = sqrtf(var);
}

// embedding = pooled @ W^T + b, fp32.  One wave per output feature; grid (64, n_masks).
__global__ __launch_bounds__(256) void embed_linear_kernel(const float* __restrict__ pooled, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6), j = blockIdx.y;
  const float4* p = (const float4*)(pooled + (long)j * 5120);
  const float4* w = (const float4*)(W + (long)o * 5120);
  float s = 0.f;
  for (int i = lane; i < 5120 / 4; i += 64) {
    const float4 a = p[i], b = w[i];
    s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  s = wave_reduce_sum(s);
  if (lane == 0) out[(long)j * RN_EMB + o] = s + bias[o];
}

struct HostT { std::vector<float> data; };

inline bf16_t h2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

struct Conv {
  bf16_t* W = nullptr;   // [cout][ntaps * Kpad], tap = kernel row, columns kw * cin + c (zero padded to Kpad)
  float* b = nullptr;    // folded BatchNorm shift
  int cin = 0, cout = 0, k = 3, stride = 1, Kpad = 0;
};
struct Block {
  Conv c1, c2, sc;
  bool has_sc = false;
};

}  // namespace

struct ccx_resnet {
  ccx_ctx* ctx = nullptr;
  int max_chunks = 0, max_masks = 0, max_frames = 0;
  long max_samples = 0;
  bool finalized = false;
  std::map<std::string, HostT> staged;
  std::vector<void*> allocs;
  // fbank tables
  float *window = nullptr, *melw = nullptr;
  float2* twiddle = nullptr;
  int *mel_start = nullptr, *mel_len = nullptr;
  // weights
  float *c1w = nullptr, *c1b = nullptr, *segW = nullptr, *segb = nullptr;
  std::vector<Block> blocks;
  // workspaces
  float *feats_t = nullptr, *fmean = nullptr, *pooled = nullptr;
  bf16_t* act[RN_STAGES][3] = {};
  size_t act_elems[RN_STAGES] = {};
  int* mask_chunk_dev = nullptr;
  std::vector<int> mask_host[4];   // private copies of the caller's mask_chunk arrays (ring: the copy is asynchronous)
  int mask_slot = 0;
  int last_T = -1;
  int halo_chunks = 0;             // chunks [0, halo_chunks) have zero halo cells for the geometry of last_T
};

namespace {

template <typename T>
int ralloc(ccx_resnet* r, T** out, size_t count) {
  void* p = nullptr;
  const size_t bytes = ccx_align(count * sizeof(T), 256);
  CCX_HIP(r->ctx, hipMalloc(&p, bytes));
  CCX_HIP(r->ctx, hipMemset(p, 0, bytes));
  r->allocs.push_back(p);
  *out = (T*)p;
  return CCX_OK;
}
template <typename T>
int rup(ccx_resnet* r, T** out, const std::vector<T>& src) {
  RTRY(ralloc(r, out, src.size()));
  CCX_HIP(r->ctx, hipMemcpy(*out, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return CCX_OK;
}
int rneed(ccx_resnet* r, const std::string& name, size_t numel, const HostT** out) {
  auto it = r->staged.find(name);
  if (it == r->staged.end()) return ccx_fail(r->ctx, CCX_ERR_MISSING, "resnet: tensor '%s' was never set", name.c_str());
  if (it->second.data.size() != numel)
    return ccx_fail(r->ctx, CCX_ERR_ARG, "resnet: tensor '%s' has %zu elements, expected %zu", name.c_str(), it->second.data.size(), numel);
  *out = &it->second;
  return CCX_OK;
}
#define RNEED(var, name, numel) \
  const HostT* var = nullptr;   \
  RTRY(rneed(r, (name), (size_t)(numel), &var));

// BatchNorm (eval) folded into a per-output-channel scale and shift
int bn_fold(ccx_resnet* r, const std::string& name, int c, std::vector<float>& scale, std::vector<float>& shift) {
  RNEED(g, name + ".weight", c); RNEED(be, name + ".bias", c);
  RNEED(rm, name + ".running_mean", c); RNEED(rv, name + ".running_var", c);
  scale.resize(c); shift.resize(c);
  for (int i = 0; i < c; i++) {
    scale[i] = g->data[i] / sqrtf(rv->data[i] + 1e-5f);
    shift[i] = be->data[i] - rm->data[i] * scale[i];
  }
  return CCX_OK;
}

// torch Conv2d weight [cout][cin][k][k] (+ folded BN scale) -> [cout][k taps][Kpad] with columns kw * cin + c
int load_conv(ccx_resnet* r, const std::string& wname, const std::string& bnname, int cin, int cout, int k, int stride, Conv& cv) {
  RNEED(w, wname, (size_t)cout * cin * k * k);
  std::vector<float> sc, sh;
  RTRY(bn_fold(r, bnname, cout, sc, sh));
  cv.cin = cin; cv.cout = cout; cv.k = k; cv.stride = stride;
  cv.Kpad = ccx_align(k * cin, 64);
  std::vector<bf16_t> packed((size_t)cout * k * cv.Kpad, 0);
  for (int o = 0; o < cout; o++)
    for (int c = 0; c < cin; c++)
      for (int kh = 0; kh < k; kh++)
        for (int kw = 0; kw < k; kw++)
          packed[((size_t)o * k + kh) * cv.Kpad + kw * cin + c] = h2bf(w->data[(((size_t)o * cin + c) * k + kh) * k + kw] * sc[o]);
  RTRY(rup(r, &cv.W, packed));
  RTRY(rup(r, &cv.b, sh));
  return CCX_OK;
}

// Zero the halo cells (rows 0 and H + 1, columns 0 and W + 1) of chunks [c0, c0 + gridDim.y) in the three buffers of one stage.
// The convolutions write interior cells only, so this has to run when the geometry (frame count) changes or more chunks come into use;
// it touches ~3 % of what clearing the buffers would (which used to be 20 GB of writes per geometry change at 672 chunks).
__global__ __launch_bounds__(256) void halo_zero_kernel(bf16_t* b0, bf16_t* b1, bf16_t* b2, int c0, int H, long Wp, int C) {
  bf16_t* buf = blockIdx.z == 0 ? b0 : (blockIdx.z == 1 ? b1 : b2);
  const int y = blockIdx.x;                                   // 0 .. H + 1
  bf16_t* row = buf + (((long)(c0 + blockIdx.y) * (H + 2) + y) * Wp) * C;
  const uint4 z = make_uint4(0, 0, 0, 0);
  if (y == 0 || y == H + 1) {
    const long n = Wp * C / 8;                                // C is a multiple of 32
    for (long i = threadIdx.x; i < n; i += 256) ((uint4*)row)[i] = z;
  } else {
    const int n = C / 8;
    for (int i = threadIdx.x; i < 2 * n; i += 256) {
      bf16_t* cell = i < n ? row : row + (Wp - 1) * C;
      ((uint4*)cell)[i < n ? i : i - n] = z;
    }
  }
}

struct StageDims { int H, W, C; long Wp; };
void stage_dims(int T, StageDims (&d)[RN_STAGES]) {
  int H = FB_MELS, W = T, C = RN_C0;
  for (int s = 0; s < RN_STAGES; s++) {
    d[s] = {H, W, C, (long)W + 2};
    H /= 2; W = (W - 1) / 2 + 1; C *= 2;
  }
}

// one convolution as a GEMM over a strided view of `in` (stage si) writing the padded layout of `out` (stage so)
int run_conv(ccx_resnet* r, const Conv& cv, int epi, const bf16_t* in, const StageDims& di, bf16_t* out, const StageDims& dn,
             const bf16_t* resid, int n_chunks, hipStream_t st) {
  static const bool direct = getenv("CCX_RESNET_DIRECT") == nullptr || atoi(getenv("CCX_RESNET_DIRECT")) != 0;
  if (direct && cv.cin == 32 && cv.cout == 32 && cv.k == 3 && cv.stride == 1 && di.H % 4 == 0 && (epi == EPI_BF16_RELU || epi == EPI_BF16_ADD_RELU)) {
    const dim3 grid(1, di.H / 4, n_chunks);
    {
      ccx_prof_scope ps(r->ctx, st, "conv3x3_c32_kernel", 2.0 * 9 * 32 * 32 * (double)n_chunks * di.H * di.W,
                        (double)n_chunks * di.H * di.W * 64.0 * (resid ? 3 : 2));
      if (resid) hipLaunchKernelGGL(conv3x3_c32_kernel<true>, grid, dim3(256), 0, st, in, cv.W, cv.b, resid, out, di.H, di.W);
      else hipLaunchKernelGGL(conv3x3_c32_kernel<false>, grid, dim3(256), 0, st, in, cv.W, cv.b, resid, out, di.H, di.W);
    }
    CCX_CHECK_LAUNCH(r->ctx);
    return CCX_OK;
  }
  GemmParams p;
  memset(&p, 0, sizeof(p));
  const int s = cv.stride;
  const int RI = (di.H + 2) / s;
  p.A = in + (cv.k == 1 ? (di.Wp + 1) * di.C : 0);
  p.lda = (long)s * di.C;
  p.K = cv.Kpad; p.ntaps = cv.k; p.a_tap_stride = di.Wp * di.C;
  p.W = cv.W; p.ldw = (long)cv.k * cv.Kpad;
  p.M = (int)((long)n_chunks * RI * di.Wp); p.N = cv.cout;
  p.bias = cv.b; p.out = out; p.ldo = cv.cout;
  p.rpb_in = (int)di.Wp; p.rpb_valid = dn.W; p.rpb_out = (int)dn.Wp; p.roff = (int)dn.Wp + 1;
  p.img_rows_in = RI; p.img_rows_valid = dn.H; p.img_rows_out = dn.H + 2;
  p.resid_bf16 = resid; p.ldrb = cv.cout;
  return ccx_launch_gemm(r->ctx, epi, p, st);
}

}  // namespace

extern "C" {

int ccx_resnet_create(ccx_ctx* ctx, int max_chunks, int64_t max_samples, int max_masks, ccx_resnet** out) {
  if (!ctx || !out) return CCX_ERR_ARG;
  CCX_REQUIRE(ctx, max_chunks >= 1 && max_masks >= 1 && max_samples >= FB_LEN + 7 * FB_SHIFT, "resnet_create: bad capacities");
  ccx_resnet* r = new ccx_resnet();
  r->ctx = ctx; r->max_chunks = max_chunks; r->max_masks = max_masks; r->max_samples = max_samples;
  r->max_frames = 1 + (int)((max_samples - FB_LEN) / FB_SHIFT);
  CCX_REQUIRE(ctx, (long)max_chunks * 82 * (r->max_frames + 2) < (1L << 31), "resnet_create: max_chunks x frames too large for one launch");
  *out = r;
  return CCX_OK;
}

void ccx_resnet_destroy(ccx_resnet* r) {
  if (!r) return;
  for (void* p : r->allocs) hipFree(p);
  delete r;
}

int ccx_resnet_set_tensor(ccx_resnet* r, const char* name, const float* data, int64_t numel) {
  if (!r) return CCX_ERR_ARG;
  CCX_REQUIRE(r->ctx, !r->finalized && name && data && numel > 0, "resnet: set_tensor bad arguments");
  HostT t;
  t.data.resize((size_t)numel);
  CCX_HIP(r->ctx, hipMemcpy(t.data.data(), data, (size_t)numel * 4, hipMemcpyDefault));
  r->staged[std::string(name)] = std::move(t);
  return CCX_OK;
}

int ccx_resnet_finalize(ccx_resnet* r) {
  if (!r) return CCX_ERR_ARG;
  ccx_ctx* ctx = r->ctx;
  CCX_REQUIRE(ctx, !r->finalized, "resnet: finalize called twice");
  // ---- fbank tables (Kaldi mel scale, triangular filters linear in mel, 20 Hz .. Nyquist) ----
  {
    std::vector<float> win(FB_LEN);
    for (int i = 0; i < FB_LEN; i++) win[i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / (FB_LEN - 1)));
    std::vector<float2> tw(FB_NFFT / 2);
    for (int i = 0; i < FB_NFFT / 2; i++) tw[i] = make_float2((float)cos(-2.0 * M_PI * i / FB_NFFT), (float)sin(-2.0 * M_PI * i / FB_NFFT));
    auto mel = [](double f) { return 1127.0 * log(1.0 + f / 700.0); };
    const double mlo = mel(20.0), mhi = mel(8000.0), delta = (mhi - mlo) / (FB_MELS + 1);
    std::vector<float> mw((size_t)FB_MELS * FB_MELW, 0.f);
    std::vector<int> ms(FB_MELS, 0), ml(FB_MELS, 0);
    for (int m = 0; m < FB_MELS; m++) {
      const double left = mlo + m * delta, center = left + delta, right = center + delta;
      int first = -1, last = -1;
      std::vector<float> wts(FB_NFFT / 2, 0.f);
      for (int k = 0; k < FB_NFFT / 2; k++) {
        const double mk = mel(16000.0 / FB_NFFT * k);
        const double up = (mk - left) / (center - left), down = (right - mk) / (right - center);
        const double wv = fmax(0.0, fmin(up, down));
        wts[k] = (float)wv;
        if (wv > 0.0) { if (first < 0) first = k; last = k; }
      }
      if (first < 0) { first = 0; last = -1; }
      CCX_REQUIRE(ctx, last - first + 1 <= FB_MELW, "resnet: mel filter %d spans %d bins", m, last - first + 1);
      ms[m] = first; ml[m] = last - first + 1;
      for (int k = first; k <= last; k++) mw[(size_t)m * FB_MELW + (k - first)] = wts[k];
    }
    RTRY(rup(r, &r->window, win)); RTRY(rup(r, &r->twiddle, tw)); RTRY(rup(r, &r->melw, mw));
    RTRY(rup(r, &r->mel_start, ms)); RTRY(rup(r, &r->mel_len, ml));
  }
  // ---- conv1 (kept f32: one input channel) ----
  {
    RNEED(w, "resnet.conv1.weight", RN_C0 * 9);
    std::vector<float> sc, sh;
    RTRY(bn_fold(r, "resnet.bn1", RN_C0, sc, sh));
    std::vector<float> wf(RN_C0 * 9);
    for (int c = 0; c < RN_C0; c++)
      for (int k = 0; k < 9; k++) wf[c * 9 + k] = w->data[c * 9 + k] * sc[c];
    RTRY(rup(r, &r->c1w, wf)); RTRY(rup(r, &r->c1b, sh));
  }
  // ---- residual stages ----
  int cin = RN_C0;
  for (int s = 0; s < RN_STAGES; s++) {
    const int cout = RN_C0 << s;
    for (int b = 0; b < RN_BLOCKS[s]; b++) {
      const std::string p = "resnet.layer" + std::to_string(s + 1) + "." + std::to_string(b) + ".";
      const int stride = (s > 0 && b == 0) ? 2 : 1;
      Block blk;
      RTRY(load_conv(r, p + "conv1.weight", p + "bn1", cin, cout, 3, stride, blk.c1));
      RTRY(load_conv(r, p + "conv2.weight", p + "bn2", cout, cout, 3, 1, blk.c2));
      blk.has_sc = stride != 1 || cin != cout;
      if (blk.has_sc) RTRY(load_conv(r, p + "shortcut.0.weight", p + "shortcut.1", cin, cout, 1, stride, blk.sc));
      r->blocks.push_back(blk);
      cin = cout;
    }
  }
  {
    RNEED(w, "resnet.seg_1.weight", (size_t)RN_EMB * 5120); RNEED(b, "resnet.seg_1.bias", RN_EMB);
    RTRY(rup(r, &r->segW, w->data)); RTRY(rup(r, &r->segb, b->data));
  }
  r->staged.clear();
  // ---- workspaces ----
  const int T = r->max_frames;
  RTRY(ralloc(r, &r->feats_t, (size_t)r->max_chunks * FB_MELS * T));
  RTRY(ralloc(r, &r->fmean, (size_t)r->max_chunks * FB_MELS));
  RTRY(ralloc(r, &r->pooled, (size_t)r->max_masks * 5120));
  RTRY(ralloc(r, &r->mask_chunk_dev, (size_t)r->max_masks));
  StageDims d[RN_STAGES];
  stage_dims(T, d);
  for (int s = 0; s < RN_STAGES; s++) {
    // + slack: the strided views read up to two padded rows and one K tile past the last (dropped) row
    r->act_elems[s] = (size_t)r->max_chunks * (d[s].H + 2) * d[s].Wp * d[s].C + 4 * d[s].Wp * d[s].C + 1024;
    for (int k = 0; k < 3; k++) RTRY(ralloc(r, &r->act[s][k], r->act_elems[s]));
  }
  r->finalized = true;
  return CCX_OK;
}

int ccx_resnet_embed(ccx_resnet* r, const float* wav_dev, int64_t stride, int n_samples, int n_chunks, const float* weights_dev,
                     int n_w, const int* mask_chunk, int n_masks, float* out_dev, void* stream_) {
  if (!r) return CCX_ERR_ARG;
  ccx_ctx* ctx = r->ctx;
  hipStream_t st = (hipStream_t)stream_;
  CCX_REQUIRE(ctx, r->finalized, "resnet: not finalized");
  CCX_REQUIRE(ctx, wav_dev && out_dev && n_chunks >= 1 && n_chunks <= r->max_chunks, "resnet_embed: bad chunk count %d (max %d)", n_chunks, r->max_chunks);
  CCX_REQUIRE(ctx, n_samples <= stride && n_samples <= r->max_samples, "resnet_embed: n_samples %d exceeds stride or capacity", n_samples);
  const int T = n_samples >= FB_LEN ? 1 + (n_samples - FB_LEN) / FB_SHIFT : 0;
  StageDims d[RN_STAGES];
  stage_dims(T > 0 ? T : 1, d);
  CCX_REQUIRE(ctx, T >= 1 && d[3].W >= 2, "resnet_embed: %d samples are too short (at least two pooled frames are needed)", n_samples);
  if (weights_dev) {
    CCX_REQUIRE(ctx, mask_chunk && n_masks >= 1 && n_masks <= r->max_masks && n_w >= 1, "resnet_embed: bad mask arguments");
    for (int j = 0; j < n_masks; j++)
      CCX_REQUIRE(ctx, mask_chunk[j] >= 0 && mask_chunk[j] < n_chunks, "resnet_embed: mask_chunk[%d]=%d out of range", j, mask_chunk[j]);
    std::vector<int>& keep = r->mask_host[r->mask_slot++ & 3];
    keep.assign(mask_chunk, mask_chunk + n_masks);            // the caller's array may go away before the copy runs
    CCX_HIP(ctx, hipMemcpyAsync(r->mask_chunk_dev, keep.data(), (size_t)n_masks * 4, hipMemcpyHostToDevice, st));
  } else {
    n_masks = n_chunks;
    CCX_REQUIRE(ctx, n_masks <= r->max_masks, "resnet_embed: %d chunks exceed the mask capacity %d", n_masks, r->max_masks);
  }
  static const bool full_clear = getenv("CCX_RESNET_FULL_CLEAR") != nullptr && atoi(getenv("CCX_RESNET_FULL_CLEAR")) != 0;   // A/B: round 2's behaviour
  if (T != r->last_T) {                                         // the halo cells move with the frame count
    r->last_T = T; r->halo_chunks = 0;
    if (full_clear) {
      for (int s = 0; s < RN_STAGES; s++)
        for (int k = 0; k < 3; k++) CCX_HIP(ctx, hipMemsetAsync(r->act[s][k], 0, r->act_elems[s] * 2, st));
      r->halo_chunks = r->max_chunks;
    }
  }
  if (n_chunks > r->halo_chunks) {
    for (int s = 0; s < RN_STAGES; s++)
      hipLaunchKernelGGL(halo_zero_kernel, dim3(d[s].H + 2, n_chunks - r->halo_chunks, 3), dim3(256), 0, st, r->act[s][0], r->act[s][1],
                         r->act[s][2], r->halo_chunks, d[s].H, d[s].Wp, d[s].C);
    CCX_CHECK_LAUNCH(ctx);
    r->halo_chunks = n_chunks;
  }
  {
    ccx_prof_scope ps(ctx, st, "fbank_kernel", 0.0, (double)n_chunks * n_samples * 4 + (double)n_chunks * T * FB_MELS * 4);
    hipLaunchKernelGGL(fbank_kernel, dim3(ccx_cdiv(T, 4), n_chunks), dim3(256), 0, st, wav_dev, (long)stride, T, r->window, r->twiddle,
                       r->melw, r->mel_start, r->mel_len, r->feats_t);
  }
  CCX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(fbank_mean_kernel, dim3(n_chunks * FB_MELS), dim3(256), 0, st, r->feats_t, T, r->fmean);
  CCX_CHECK_LAUNCH(ctx);
  {
    ccx_prof_scope ps(ctx, st, "conv1_kernel", 2.0 * 9 * RN_C0 * (double)n_chunks * FB_MELS * T,
                      (double)n_chunks * FB_MELS * T * (4 + RN_C0 * 2));
    hipLaunchKernelGGL(conv1_kernel, dim3(ccx_cdiv(T, 256), FB_MELS, n_chunks), dim3(256), 0, st, r->feats_t, r->fmean, T, r->c1w, r->c1b,
                       r->act[0][0]);
  }
  CCX_CHECK_LAUNCH(ctx);
  // residual stages: x lives in act[s][cur]; t = relu(conv1(x)); y = relu(conv2(t) + shortcut(x))
  int bi = 0, cur = 0;
  const bf16_t* x = r->act[0][0];
  int xs = 0;   // stage whose layout x has
  for (int s = 0; s < RN_STAGES; s++) {
    for (int b = 0; b < RN_BLOCKS[s]; b++, bi++) {
      const Block& blk = r->blocks[bi];
      bf16_t* tbuf; bf16_t* ybuf; bf16_t* scbuf = nullptr;
      if (xs == s) {
        tbuf = r->act[s][(cur + 1) % 3]; ybuf = r->act[s][(cur + 2) % 3];
      } else {  // first block of a stage: x is still in the previous stage's layout
        tbuf = r->act[s][0]; ybuf = r->act[s][1]; scbuf = r->act[s][2];
      }
      RTRY(run_conv(r, blk.c1, EPI_BF16_RELU, x, d[xs], tbuf, d[s], nullptr, n_chunks, st));
      const bf16_t* resid = x;
      if (blk.has_sc) {
        CCX_REQUIRE(ctx, scbuf != nullptr, "resnet: shortcut convolution inside a stage");
        RTRY(run_conv(r, blk.sc, EPI_BF16, x, d[xs], scbuf, d[s], nullptr, n_chunks, st));
        resid = scbuf;
      }
      RTRY(run_conv(r, blk.c2, EPI_BF16_ADD_RELU, tbuf, d[s], ybuf, d[s], resid, n_chunks, st));
      x = ybuf; xs = s;
      cur = (int)(ybuf == r->act[s][0] ? 0 : (ybuf == r->act[s][1] ? 1 : 2));
    }
  }
  hipLaunchKernelGGL(tstp_kernel, dim3(10, n_masks), dim3(256), 0, st, x, d[3].W, weights_dev ? r->mask_chunk_dev : nullptr, weights_dev,
                     n_w, r->pooled);
  CCX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(embed_linear_kernel, dim3(RN_EMB / 4, n_masks), dim3(256), 0, st, r->pooled, r->segW, r->segb, out_dev);
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

}  // extern "C"
