// whisper.hip -- Whisper model object of libccx: weight intake by openai-whisper state_dict names,
// encoder forward (conv stem as im2col MFMA GEMMs, 12 pre-LN transformer blocks, cross-KV
// projection) and greedy decoding (device-side state machine, step chain captured in a hipGraph).
//
// Replaces `self.whisper_model` of the reference (loaded at back/api.py:665-703, called at
// back/api.py:1286-1292, 1432-1438, 1474-1480).  Model semantics follow openai-whisper
// model.py / decoding.py [UPSTREAM-RECALL -- not vendored in the reference]; the CPU restatement
// the tests compare against is oracle/whisper_ref.py.
#include <array>
#include <atomic>
#include <map>
#include <chrono>
#include <math.h>
#include "../../include/ccx.h"
#include "attention.h"
#include "ccx_common.h"
#include "cross_x.h"
#include "decoder.h"
#include "elementwise.h"
#include "gemm_bf16.h"
#include "logmel.h"

namespace {

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

inline bf16_t host_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
inline float host_half_to_f32(uint16_t h) {
  const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1f, m = h & 0x3ff;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = s << 31;
    else {
      int ee = -1; uint32_t mm = m;
      do { ee++; mm <<= 1; } while (!(mm & 0x400));
      u = (s << 31) | ((uint32_t)(127 - 15 - ee) << 23) | ((mm & 0x3ff) << 13);
    }
  } else if (e == 31) u = (s << 31) | 0x7f800000u | (m << 13);
  else u = (s << 31) | ((e + 112) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct EncLayer {
  float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  bf16_t *Wqkv, *Wo, *W1, *W2;
  float *bqkv, *bo, *b1, *b2;
};
struct DecLayer {
  float *ln1_g, *ln1_b, *lnc_g, *lnc_b, *ln2_g, *ln2_b;
  bf16_t *Wqkv, *Wo, *Wcq, *Wckv, *Wco, *W1, *W2;
  bf16_t* W1_plain = nullptr;   // fc1 once more in plain row-major [F][D] for the tiled GEMM (lanes of >= 256 rows, prefill)
  bf16_t* WckT = nullptr;       // cross_attn.key.weight re-laid per head [H][D][64] for the expanded query (cross_x.hip)
  bf16_t* Wcq_plain = nullptr;  // cross_attn.query.weight in plain row-major [D][D] (dec_xq_fused_kernel)
  // LayerNorm-free chain (decoder.h, ACT_BF16_LN): gamma folded into the consumer weights, s_n = sum_k (gamma o W)_nk over the bf16
  // values the MFMAs see, c_n = sum_k beta_k W_nk + b_n.  Wqkv_g / W1_g fragment-packed, Wcq_g plain row-major
  bf16_t *Wqkv_g = nullptr, *Wcq_g = nullptr, *W1_g = nullptr;
  float *sqkv = nullptr, *cqkv = nullptr, *scq = nullptr, *ccq = nullptr, *s1 = nullptr, *c1 = nullptr;
  float *bqkv, *bo, *bcq, *bckv, *bco, *b1, *b2;
  bf16_t *crossK, *crossV, *selfK, *selfV;
};

}  // namespace

struct ccx_whisper {
  ccx_ctx* ctx = nullptr;
  ccx_whisper_dims d{};
  int max_batch = 0;
  bool finalized = false;
  ccx_whisper* scratch_donor = nullptr;   // ccx_whisper_share_encoder_scratch: log-mel / encoder workspaces of another instance
  int scratch_takers = 0;                 // instances that borrowed THIS instance's workspaces and are still alive
  bool destroy_pending = false;           // ccx_whisper_destroy was called while takers were alive: freed with the last taker
  std::atomic<ccx_whisper*> scratch_owner{nullptr};   // (donor) instance whose staged log-mel waits for its encode: a shared group has ONE user between logmel and encode
  hipEvent_t scratch_free = nullptr;      // (donor) recorded at the end of every encode of the group; every log-mel / set_mel of the
                                          // group waits for it, so that users of the shared workspaces are ordered on ANY streams
  std::map<std::string, HostTensor> staged;
  std::vector<void*> allocs;
  char* arena = nullptr;
  size_t arena_cap = 0, arena_off = 0;

  // derived sizes
  int Spad = 0;    // padded audio context (multiple of 128)
  int Vpad = 0;    // vocab rounded up to 128
  int Fraw = 3008; // frames computed by the log-mel pass (covers 30 s + reflect tail)
  static constexpr int kCrossSplitMax = 8;

  // tables
  LogmelTables lm{};
  // encoder weights
  bf16_t *Wc1 = nullptr, *Wc2 = nullptr;
  float *bc1 = nullptr, *bc2 = nullptr, *enc_pos = nullptr, *lnp_g = nullptr, *lnp_b = nullptr;
  std::vector<EncLayer> enc;
  // decoder weights
  float *tok_emb_f32 = nullptr, *dec_pos = nullptr, *lnd_g = nullptr, *lnd_b = nullptr;
  bf16_t* tok_emb_bf16 = nullptr;   // fragment-packed (dec_linear)
  bf16_t* tok_emb_rm = nullptr;     // row-major [n_vocab][D] (logits through the tiled GEMM)
  std::vector<DecLayer> dec;
  // rules
  ccx_decode_rules rules{};
  unsigned char* suppress_mask = nullptr;
  bool rules_set = false;

  // workspaces (encoder)
  float* lm_raw = nullptr; unsigned int* lm_max = nullptr; int *lm_n = nullptr, *lm_seek = nullptr, *lm_seg = nullptr;
  bf16_t *im2col = nullptr, *h1 = nullptr, *xn = nullptr, *qb = nullptr, *kb = nullptr, *vtb = nullptr, *attn = nullptr,
         *ffn = nullptr, *xa = nullptr;
  float* x = nullptr;
  // workspaces (decoder)
  float *dx = nullptr, *dx2 = nullptr, *pend = nullptr, *dq = nullptr, *dlogits = nullptr, *part_o = nullptr, *part_ml = nullptr;
  bf16_t *dattn = nullptr, *dffn = nullptr, *dxn = nullptr;
  int *cur_tok = nullptr, *pos = nullptr, *prompt = nullptr, *gen = nullptr, *n_done = nullptr;
  bool sampling = false;            // temperature > 0 in the current decode call (selects the kernel variant)
  unsigned* sample_cfg = nullptr;   // {temperature bits, seed lo, seed hi, 0}: read by the select kernel every step
  DecSeqState* state = nullptr;
  int max_prompt_cap = 0, sample_cap = 0;
  // prompt prefill: one pass over every prompt position of every sequence (rows = sequence * P + position) instead of one
  // decode step per prompt token; row-indexed copies of the step buffers and the row tables
  static constexpr int kPrefillMax = 16;     // prompt positions prefilled in one pass (longer prompts: several passes)
  float *pf_x = nullptr, *pf_x2 = nullptr, *pf_pend = nullptr, *pf_q = nullptr;
  bf16_t *pf_xn = nullptr, *pf_attn = nullptr, *pf_ffn = nullptr;
  int *pf_tok = nullptr, *pf_pos = nullptr, *pf_seq = nullptr, *pf_last = nullptr;
  // graph cache; decode runs on an internal stream when the caller hands over the legacy null
  // stream (stream capture is illegal there)
  std::map<std::array<int, 9>, hipGraphExec_t> graphs;
  hipStream_t own_stream = nullptr;
  hipEvent_t own_event = nullptr;
  // decode lanes: disjoint row ranges of one batch stepping concurrently on their own streams, staggered so
  // that one lane's HBM-bound cross attention overlaps the other lanes' latency-bound linears.  The gain is
  // modest (3-4 % at 192 sequences): the small kernels slow down 3-5x while HBM is saturated by another lane.
  static constexpr int kMaxLanes = 4;
  int cross_lds_pad = 0;                     // see ccx_whisper_decode: occupancy cap of the cross-attention blocks while lanes overlap
  int cross_stream = 0;                      // 1: lean-streaming cross attention (dec_cross_stream_kernel) for batches > 16
  int fuse_cross_q = 1;                      // 1: batches <= 16 compute the cross-attention query inside the attention blocks
  // Cross attention against the encoder output (cross_x.hip) for decodes of more than 80 sequences: no per-layer K/V caches beyond those
  // (42 GB at 768 sequences), half the bytes per step.  xs_on: the instance was built for it (widths cross_x.hip instantiates,
  // CCX_CROSS_X != 0 at finalize); the K/V caches then hold kv_cap <= 80 sequences and are filled from xa at the start of a decode
  // that takes the K/V path (kv_ready = sequences valid since the last encode).
  static constexpr int kKvSeqs = 80;         // sequences the K/V caches of an X-stream instance hold (4.5 GB at small.en)
  bool xs_on = false, xs_active = false, xs_fuse_q = true;
  int last_cross_path = -1;                  // ccx_whisper_last_cross_path: 0 kv16, 1 kv_stream, 2 xa_stream
  int kv_cap = 0, kv_ready = 0;
  bf16_t *xq = nullptr, *pf_xq = nullptr;    // expanded queries [rows][H][D] (step rows, prefill rows)
  // LayerNorm-free chain: bf16 copy of the resolved residual rows and their (sum, sum of squares) per 16-column tile
  bf16_t *dxb = nullptr, *pf_xb = nullptr;
  float2 *dst2 = nullptr, *pf_st2 = nullptr;
  bool lnfree = false;                       // CCX_DEC_LNFREE, read per decode
  bool lnfree_built = false;                 // the folded weights exist (CCX_DEC_LNFREE was set when the instance was finalized)
  int lnfree_mode = 0;                       // 1: every producer resolves in place (12-wave second MLP linear); 2: that one keeps split-K slabs
  float *xs_po = nullptr, *xs_pml = nullptr, *pf_xs_po = nullptr, *pf_xs_pml = nullptr;   // key-half partials (cross_x.h)
  static constexpr int kLanePool = 8;
  hipStream_t lane_pool[kLanePool] = {};     // candidates; HIP streams share a few hardware queues and two streams on one
                                             // queue run strictly one after the other, so lanes are picked by a probe
  std::map<hipStream_t, std::vector<hipStream_t>> lane_sets;   // lane-0 stream -> streams that overlap with it and each other
  int* probe_sink = nullptr;
  unsigned long long* stamps = nullptr;      // [kMaxLanes][kStampCap] diagnostic trace (ccx_whisper_trace_lanes / CCX_DEC_STAMPS)
  int* stamp_count = nullptr;                // [kMaxLanes]
  bool stamps_on = false;                    // stamp nodes are captured into the step graphs while set
  int stamp_level = 1;
  std::string stamp_path;
  static constexpr int kStampCap = 1 << 16;
  hipEvent_t lane_start[kMaxLanes] = {}, lane_poll[2][kMaxLanes] = {};
  int* poll_host = nullptr;                  // pinned [2][kMaxLanes]
};

namespace {

// Weights and per-step decode buffers are carved from ONE large allocation (bump pointer, in
// access order): large contiguous mappings keep the decode chain's weight stream on few, large
// TLB entries and make consecutive kernels touch consecutive addresses.
template <typename T>
int dev_alloc(ccx_whisper* w, T** out, size_t count, bool zero) {
  const size_t bytes = ccx_align(count * sizeof(T), 256);
  void* p = nullptr;
  if (w->arena && w->arena_off + bytes <= w->arena_cap) {
    p = w->arena + w->arena_off;
    w->arena_off += bytes;
  } else {
    CCX_HIP(w->ctx, hipMalloc(&p, bytes));
    w->allocs.push_back(p);
  }
  if (zero) CCX_HIP(w->ctx, hipMemset(p, 0, bytes));
  *out = (T*)p;
  return CCX_OK;
}

int up_f32(ccx_whisper* w, float** out, const float* src, size_t n) {
  int rc = dev_alloc(w, out, n, false);
  if (rc) return rc;
  CCX_HIP(w->ctx, hipMemcpy(*out, src, n * 4, hipMemcpyHostToDevice));
  return CCX_OK;
}
int up_bf16(ccx_whisper* w, bf16_t** out, const float* src, size_t n) {
  std::vector<bf16_t> tmp(n);
  for (size_t i = 0; i < n; i++) tmp[i] = host_f32_to_bf16(src[i]);
  int rc = dev_alloc(w, out, n, false);
  if (rc) return rc;
  CCX_HIP(w->ctx, hipMemcpy(*out, tmp.data(), n * 2, hipMemcpyHostToDevice));
  return CCX_OK;
}

// Decode-side weights are stored MFMA-fragment-packed for dec_linear_kernel: tile (n/16, k/32) is 64
// consecutive 16-byte chunks, chunk l = row n0 + (l & 15), columns k0 + 8*(l >> 4) .. +8.
// Rows are zero padded to a multiple of `row_pad`.
int up_bf16_packed(ccx_whisper* w, bf16_t** out, const float* src, int N, int K, int row_pad) {
  const int Np = (N + row_pad - 1) / row_pad * row_pad;
  const int kst = K / 32;
  std::vector<bf16_t> tmp((size_t)Np * K, 0);
  for (int nt = 0; nt < Np / 16; nt++)
    for (int ks = 0; ks < kst; ks++)
      for (int l = 0; l < 64; l++) {
        const int n = nt * 16 + (l & 15), k0 = ks * 32 + 8 * (l >> 4);
        bf16_t* dst = &tmp[(((size_t)nt * kst + ks) * 64 + l) * 8];
        if (n < N)
          for (int j = 0; j < 8; j++) dst[j] = host_f32_to_bf16(src[(size_t)n * K + k0 + j]);
      }
  int rc = dev_alloc(w, out, tmp.size(), false);
  if (rc) return rc;
  CCX_HIP(w->ctx, hipMemcpy(*out, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
  return CCX_OK;
}

const HostTensor* find(ccx_whisper* w, const std::string& name) {
  auto it = w->staged.find(name);
  return it == w->staged.end() ? nullptr : &it->second;
}

int need(ccx_whisper* w, const std::string& name, std::vector<int64_t> shape, const HostTensor** out) {
  const HostTensor* t = find(w, name);
  if (!t) return ccx_fail(w->ctx, CCX_ERR_MISSING, "whisper: tensor '%s' was never set", name.c_str());
  if (t->shape != shape) {
    std::string got, want;
    for (auto v : t->shape) got += std::to_string(v) + ",";
    for (auto v : shape) want += std::to_string(v) + ",";
    return ccx_fail(w->ctx, CCX_ERR_ARG, "whisper: tensor '%s' has shape [%s] expected [%s]", name.c_str(), got.c_str(),
                    want.c_str());
  }
  *out = t;
  return CCX_OK;
}

#define NEED(var, name, ...)                                              \
  const HostTensor* var = nullptr;                                        \
  {                                                                       \
    int _rc = need(w, (name), std::vector<int64_t>{__VA_ARGS__}, &var);   \
    if (_rc) return _rc;                                                  \
  }
#define TRY(expr)            \
  do {                       \
    int _rc = (expr);        \
    if (_rc) return _rc;     \
  } while (0)

// conv weight [out][in][3] -> GEMM weight [out][Kpad] with k = tap*in + c
std::vector<float> conv_to_gemm(const HostTensor& t, int Kpad) {
  const int64_t O = t.shape[0], I = t.shape[1], T = t.shape[2];
  std::vector<float> r((size_t)O * Kpad, 0.f);
  for (int64_t o = 0; o < O; o++)
    for (int64_t c = 0; c < I; c++)
      for (int64_t k = 0; k < T; k++) r[(size_t)o * Kpad + k * I + c] = t.data[(o * I + c) * T + k];
  return r;
}

int build_logmel_tables(ccx_whisper* w) {
  NEED(mf, "mel_filters", w->d.n_mels, 201);
  CCX_REQUIRE(w->ctx, w->d.n_mels == 80, "whisper: only n_mels == 80 is supported (got %d)", w->d.n_mels);
  std::vector<float> c(400 * 208, 0.f), s(400 * 208, 0.f), fb(80 * 208, 0.f);
  std::vector<int> rg(160);
  const double two_pi = 6.283185307179586476925286766559;
  for (int n = 0; n < 400; n++) {
    const double win = 0.5 - 0.5 * cos(two_pi * n / 400.0);  // periodic Hann (torch.hann_window default)
    for (int k = 0; k < 201; k++) {
      const int ph = (int)(((long)n * k) % 400);  // exact phase reduction
      const double a = two_pi * ph / 400.0;
      c[n * 208 + k] = (float)(cos(a) * win);
      s[n * 208 + k] = (float)(sin(a) * win);
    }
  }
  for (int m = 0; m < 80; m++) {
    int k0 = 201, k1 = 0;
    for (int k = 0; k < 201; k++) {
      const float v = mf->data[m * 201 + k];
      fb[m * 208 + k] = v;
      if (v != 0.f) { if (k < k0) k0 = k; k1 = k + 1; }
    }
    if (k1 == 0) { k0 = 0; k1 = 0; }
    rg[2 * m] = k0; rg[2 * m + 1] = k1;
  }
  float *dc, *ds, *dfb; int* drg;
  TRY(up_f32(w, &dc, c.data(), c.size()));
  TRY(up_f32(w, &ds, s.data(), s.size()));
  TRY(up_f32(w, &dfb, fb.data(), fb.size()));
  TRY(dev_alloc(w, &drg, 160, false));
  CCX_HIP(w->ctx, hipMemcpy(drg, rg.data(), 160 * 4, hipMemcpyHostToDevice));
  w->lm.dft_cos = dc; w->lm.dft_sin = ds; w->lm.mel_fb = dfb; w->lm.mel_range = drg;
  return CCX_OK;
}

std::vector<float> cat_rows(std::initializer_list<const std::vector<float>*> parts) {
  std::vector<float> r;
  for (auto p : parts) r.insert(r.end(), p->begin(), p->end());
  return r;
}

}  // namespace

extern "C" {

int ccx_whisper_create(ccx_ctx* ctx, const ccx_whisper_dims* dims, int max_batch, ccx_whisper** out) {
  if (!ctx) return CCX_ERR_ARG;
  CCX_REQUIRE(ctx, dims && out, "ccx_whisper_create: null argument");
  CCX_REQUIRE(ctx, max_batch >= 1 && max_batch <= 1536, "ccx_whisper_create: max_batch %d out of range", max_batch);
  const ccx_whisper_dims& d = *dims;
  CCX_REQUIRE(ctx, d.n_audio_state % 128 == 0 && d.n_text_state == d.n_audio_state, "whisper: n_state must be a multiple of 128 and equal for encoder/decoder");
  CCX_REQUIRE(ctx, d.n_audio_state / d.n_audio_head == 64 && d.n_text_state / d.n_text_head == 64, "whisper: head_dim must be 64");
  CCX_REQUIRE(ctx, d.n_audio_state <= 1024, "whisper: n_state > 1024 not supported yet");
  CCX_REQUIRE(ctx, d.n_audio_ctx == 1500 && d.n_mels == 80, "whisper: n_audio_ctx must be 1500 and n_mels 80");
  CCX_REQUIRE(ctx, d.n_audio_ctx % 4 == 0, "whisper: n_audio_ctx must be a multiple of 4");
  CCX_REQUIRE(ctx, d.n_vocab % 4 == 0 && d.n_vocab <= 13 * 4096, "whisper: n_vocab must be a multiple of 4 and <= 53248");
  ccx_whisper* w = new ccx_whisper();
  w->ctx = ctx;
  w->d = d;
  w->max_batch = max_batch;
  w->Spad = ccx_cdiv(d.n_audio_ctx, 128) * 128;
  w->Vpad = ccx_cdiv(d.n_vocab, 128) * 128;
  *out = w;
  return CCX_OK;
}

void ccx_whisper_destroy(ccx_whisper* w) {
  if (!w) return;
  if (w->scratch_takers > 0) {      // a taker still points into this instance's workspaces: keep everything until it is gone
    w->destroy_pending = true;
    return;
  }
  if (ccx_whisper* dn = w->scratch_donor) {
    w->scratch_donor = nullptr;
    ccx_whisper* me = w;
    dn->scratch_owner.compare_exchange_strong(me, nullptr);     // staged windows of a destroyed instance bind nobody
    if (--dn->scratch_takers == 0 && dn->destroy_pending) ccx_whisper_destroy(dn);
  }
  if (w->scratch_free) hipEventDestroy(w->scratch_free);
  for (auto& g : w->graphs) hipGraphExecDestroy(g.second);
  if (w->own_stream) hipStreamDestroy(w->own_stream);
  if (w->own_event) hipEventDestroy(w->own_event);
  for (int i = 0; i < ccx_whisper::kMaxLanes; i++) {
    if (w->lane_start[i]) hipEventDestroy(w->lane_start[i]);
    if (w->lane_poll[0][i]) hipEventDestroy(w->lane_poll[0][i]);
    if (w->lane_poll[1][i]) hipEventDestroy(w->lane_poll[1][i]);
  }
  for (int i = 0; i < ccx_whisper::kLanePool; i++)
    if (w->lane_pool[i]) hipStreamDestroy(w->lane_pool[i]);
  if (w->poll_host) hipHostFree(w->poll_host);
  for (void* p : w->allocs) hipFree(p);
  delete w;
}

int ccx_whisper_set_tensor(ccx_whisper* w, const char* name, const void* data, int dtype, int ndim, const int64_t* shape) {
  if (!w) return CCX_ERR_ARG;
  CCX_REQUIRE(w->ctx, !w->finalized, "whisper: set_tensor after finalize");
  CCX_REQUIRE(w->ctx, name && data && ndim >= 1 && ndim <= 4 && shape, "whisper: set_tensor bad arguments");
  const std::string nm(name);
  const bool known = nm == "mel_filters" || nm.rfind("encoder.", 0) == 0 || nm.rfind("decoder.", 0) == 0;
  CCX_REQUIRE(w->ctx, known, "whisper: unknown tensor name '%s'", name);
  size_t n = 1;
  HostTensor t;
  for (int i = 0; i < ndim; i++) { CCX_REQUIRE(w->ctx, shape[i] > 0, "whisper: '%s' has an empty dim", name); n *= (size_t)shape[i]; t.shape.push_back(shape[i]); }
  t.data.resize(n);
  if (dtype == CCX_DTYPE_F32) {
    CCX_HIP(w->ctx, hipMemcpy(t.data.data(), data, n * 4, hipMemcpyDefault));
  } else if (dtype == CCX_DTYPE_BF16 || dtype == CCX_DTYPE_F16) {
    std::vector<uint16_t> tmp(n);
    CCX_HIP(w->ctx, hipMemcpy(tmp.data(), data, n * 2, hipMemcpyDefault));
    for (size_t i = 0; i < n; i++) {
      if (dtype == CCX_DTYPE_BF16) { uint32_t u = (uint32_t)tmp[i] << 16; memcpy(&t.data[i], &u, 4); }
      else t.data[i] = host_half_to_f32(tmp[i]);
    }
  } else {
    return ccx_fail(w->ctx, CCX_ERR_ARG, "whisper: set_tensor dtype %d unsupported", dtype);
  }
  w->staged[nm] = std::move(t);
  return CCX_OK;
}

int ccx_whisper_set_max_audio(ccx_whisper* w, double seconds) {
  if (!w) return CCX_ERR_ARG;
  CCX_REQUIRE(w->ctx, !w->finalized, "whisper: set_max_audio after finalize");
  CCX_REQUIRE(w->ctx, seconds >= 1.0 && seconds <= 7200.0, "whisper: max audio %.1f s out of range [1, 7200]", seconds);
  const long frames = (long)(seconds * 100.0) + 8;
  w->Fraw = (int)((frames + 31) / 32 * 32);
  if (w->Fraw < 3008) w->Fraw = 3008;
  return CCX_OK;
}

int ccx_whisper_share_encoder_scratch(ccx_whisper* w, ccx_whisper* donor) {
  if (!w) return CCX_ERR_ARG;
  CCX_REQUIRE(w->ctx, donor && donor != w && donor->finalized && !w->finalized, "whisper: share_encoder_scratch needs a finalized donor and an unfinalized taker");
  CCX_REQUIRE(w->ctx, donor->max_batch >= w->max_batch && donor->Fraw >= w->Fraw && !memcmp(&donor->d, &w->d, sizeof(w->d)),
              "whisper: share_encoder_scratch: the donor must have the same dimensions and at least the taker's capacity");
  if (!donor->scratch_free) CCX_HIP(w->ctx, hipEventCreateWithFlags(&donor->scratch_free, hipEventDisableTiming));
  w->scratch_donor = donor;
  donor->scratch_takers++;
  return CCX_OK;
}

int ccx_whisper_set_rules(ccx_whisper* w, const ccx_decode_rules* r) {
  if (!w) return CCX_ERR_ARG;
  CCX_REQUIRE(w->ctx, r, "whisper: rules null");
  const int V = w->d.n_vocab;
  CCX_REQUIRE(w->ctx, r->eot >= 0 && r->eot < V && r->sot < V && r->no_speech < V && r->timestamp_begin <= V && r->blank < V &&
                          r->no_timestamps < V, "whisper: rule token id out of range");
  std::vector<unsigned char> mask((size_t)V + 4, 0);
  for (int i = 0; i < r->n_suppress; i++) {
    CCX_REQUIRE(w->ctx, r->suppress[i] >= 0 && r->suppress[i] < V, "whisper: suppress id %d out of range", r->suppress[i]);
    mask[r->suppress[i]] = 1;
  }
  if (r->no_timestamps >= 0) mask[r->no_timestamps] = 1;  // ApplyTimestampRules bans <|notimestamps|>
  if (!w->suppress_mask) TRY(dev_alloc(w, &w->suppress_mask, (size_t)V + 4, false));
  CCX_HIP(w->ctx, hipMemcpy(w->suppress_mask, mask.data(), (size_t)V + 4, hipMemcpyHostToDevice));
  w->rules = *r;
  w->rules.suppress = nullptr;
  w->rules_set = true;
  // captured step graphs bake the rule ids into the select kernel's parameters: drop them
  for (auto& g : w->graphs) hipGraphExecDestroy(g.second);
  w->graphs.clear();
  return CCX_OK;
}

int ccx_whisper_trace_lanes(ccx_whisper* w, const char* path, int level) {
  CCX_REQUIRE(w->ctx, w->finalized, "whisper: trace_lanes before finalize");
  w->stamps_on = path != nullptr && path[0] != 0;
  w->stamp_path = w->stamps_on ? path : "";
  w->stamp_level = level >= 2 ? 2 : 1;
  CCX_HIP(w->ctx, hipMemset(w->stamp_count, 0, ccx_whisper::kMaxLanes * 4));
  // the stamp kernels are nodes of the captured step graphs: drop the graphs so that the next decode captures the other kind
  for (auto& g : w->graphs) hipGraphExecDestroy(g.second);
  w->graphs.clear();
  return CCX_OK;
}

int ccx_whisper_finalize(ccx_whisper* w) {
  if (!w) return CCX_ERR_ARG;
  CCX_REQUIRE(w->ctx, !w->finalized, "whisper: finalize called twice");
  CCX_HIP(w->ctx, hipSetDevice(w->ctx->device));
  const ccx_whisper_dims& d = w->d;
  const int D = d.n_audio_state, F = 4 * D, B = w->max_batch, S = d.n_audio_ctx, H = d.n_audio_head;
  {
    // arena for all weights (bf16 copies + fp32 vectors/embedding) and the small decode buffers
    size_t staged = 0;
    for (auto& kv : w->staged) staged += kv.second.data.size() * 4;
    w->arena_cap = ccx_align(staged + ((size_t)64 << 20), (size_t)2 << 20);
    void* base = nullptr;
    CCX_HIP(w->ctx, hipMalloc(&base, w->arena_cap));
    w->allocs.push_back(base);
    w->arena = (char*)base;
    w->arena_off = 0;
  }
  TRY(build_logmel_tables(w));

  // ---------------- encoder ----------------
  {
    NEED(c1w, "encoder.conv1.weight", D, d.n_mels, 3);
    NEED(c1b, "encoder.conv1.bias", D);
    NEED(c2w, "encoder.conv2.weight", D, D, 3);
    NEED(c2b, "encoder.conv2.bias", D);
    NEED(pe, "encoder.positional_embedding", S, D);
    NEED(lg, "encoder.ln_post.weight", D);
    NEED(lb, "encoder.ln_post.bias", D);
    std::vector<float> g1 = conv_to_gemm(*c1w, 256), g2 = conv_to_gemm(*c2w, 3 * D);
    TRY(up_bf16(w, &w->Wc1, g1.data(), g1.size()));
    TRY(up_bf16(w, &w->Wc2, g2.data(), g2.size()));
    TRY(up_f32(w, &w->bc1, c1b->data.data(), D));
    TRY(up_f32(w, &w->bc2, c2b->data.data(), D));
    TRY(up_f32(w, &w->enc_pos, pe->data.data(), (size_t)S * D));
    TRY(up_f32(w, &w->lnp_g, lg->data.data(), D));
    TRY(up_f32(w, &w->lnp_b, lb->data.data(), D));
  }
  w->enc.resize(d.n_audio_layer);
  const std::vector<float> zerosD(D, 0.f);
  for (int l = 0; l < d.n_audio_layer; l++) {
    const std::string p = "encoder.blocks." + std::to_string(l) + ".";
    EncLayer& L = w->enc[l];
    NEED(qw, p + "attn.query.weight", D, D); NEED(qbias, p + "attn.query.bias", D);
    NEED(kw, p + "attn.key.weight", D, D);
    NEED(vw, p + "attn.value.weight", D, D); NEED(vbias, p + "attn.value.bias", D);
    NEED(ow, p + "attn.out.weight", D, D); NEED(obias, p + "attn.out.bias", D);
    NEED(l1g, p + "attn_ln.weight", D); NEED(l1b, p + "attn_ln.bias", D);
    NEED(m0w, p + "mlp.0.weight", F, D); NEED(m0b, p + "mlp.0.bias", F);
    NEED(m2w, p + "mlp.2.weight", D, F); NEED(m2b, p + "mlp.2.bias", D);
    NEED(l2g, p + "mlp_ln.weight", D); NEED(l2b, p + "mlp_ln.bias", D);
    std::vector<float> wqkv = cat_rows({&qw->data, &kw->data, &vw->data});
    std::vector<float> bqkv = cat_rows({&qbias->data, &zerosD, &vbias->data});
    TRY(up_bf16(w, &L.Wqkv, wqkv.data(), wqkv.size())); TRY(up_f32(w, &L.bqkv, bqkv.data(), bqkv.size()));
    TRY(up_bf16(w, &L.Wo, ow->data.data(), ow->data.size())); TRY(up_f32(w, &L.bo, obias->data.data(), D));
    TRY(up_bf16(w, &L.W1, m0w->data.data(), m0w->data.size())); TRY(up_f32(w, &L.b1, m0b->data.data(), F));
    TRY(up_bf16(w, &L.W2, m2w->data.data(), m2w->data.size())); TRY(up_f32(w, &L.b2, m2b->data.data(), D));
    TRY(up_f32(w, &L.ln1_g, l1g->data.data(), D)); TRY(up_f32(w, &L.ln1_b, l1b->data.data(), D));
    TRY(up_f32(w, &L.ln2_g, l2g->data.data(), D)); TRY(up_f32(w, &L.ln2_b, l2b->data.data(), D));
  }

  // ---------------- decoder ----------------
  {
    NEED(te, "decoder.token_embedding.weight", d.n_vocab, D);
    NEED(pe, "decoder.positional_embedding", d.n_text_ctx, D);
    NEED(lg, "decoder.ln.weight", D);
    NEED(lb, "decoder.ln.bias", D);
    TRY(up_f32(w, &w->tok_emb_f32, te->data.data(), te->data.size()));
    TRY(up_bf16_packed(w, &w->tok_emb_bf16, te->data.data(), d.n_vocab, D, 64));
    TRY(up_bf16(w, &w->tok_emb_rm, te->data.data(), (size_t)d.n_vocab * D));
    TRY(up_f32(w, &w->dec_pos, pe->data.data(), pe->data.size()));
    TRY(up_f32(w, &w->lnd_g, lg->data.data(), D));
    TRY(up_f32(w, &w->lnd_b, lb->data.data(), D));
  }
  w->dec.resize(d.n_text_layer);
  const int Tc = d.n_text_ctx;
  {
    const char* e = getenv("CCX_CROSS_X");      // 0: round 2's per-layer cross K/V caches for every sequence (A/B)
    w->xs_on = (!e || atoi(e) != 0) && d.n_text_state == D && d.n_text_head == H && ccx_xs_supported(D, H);
    w->kv_cap = w->xs_on ? (B < ccx_whisper::kKvSeqs ? B : ccx_whisper::kKvSeqs) : B;
  }
  for (int l = 0; l < d.n_text_layer; l++) {
    const std::string p = "decoder.blocks." + std::to_string(l) + ".";
    DecLayer& L = w->dec[l];
    NEED(qw, p + "attn.query.weight", D, D); NEED(qbias, p + "attn.query.bias", D);
    NEED(kw, p + "attn.key.weight", D, D);
    NEED(vw, p + "attn.value.weight", D, D); NEED(vbias, p + "attn.value.bias", D);
    NEED(ow, p + "attn.out.weight", D, D); NEED(obias, p + "attn.out.bias", D);
    NEED(l1g, p + "attn_ln.weight", D); NEED(l1b, p + "attn_ln.bias", D);
    NEED(cqw, p + "cross_attn.query.weight", D, D); NEED(cqb, p + "cross_attn.query.bias", D);
    NEED(ckw, p + "cross_attn.key.weight", D, D);
    NEED(cvw, p + "cross_attn.value.weight", D, D); NEED(cvb, p + "cross_attn.value.bias", D);
    NEED(cow, p + "cross_attn.out.weight", D, D); NEED(cob, p + "cross_attn.out.bias", D);
    NEED(lcg, p + "cross_attn_ln.weight", D); NEED(lcb, p + "cross_attn_ln.bias", D);
    NEED(m0w, p + "mlp.0.weight", F, D); NEED(m0b, p + "mlp.0.bias", F);
    NEED(m2w, p + "mlp.2.weight", D, F); NEED(m2b, p + "mlp.2.bias", D);
    NEED(l2g, p + "mlp_ln.weight", D); NEED(l2b, p + "mlp_ln.bias", D);
    std::vector<float> wqkv = cat_rows({&qw->data, &kw->data, &vw->data});
    std::vector<float> bqkv = cat_rows({&qbias->data, &zerosD, &vbias->data});
    std::vector<float> wckv = cat_rows({&ckw->data, &cvw->data});
    std::vector<float> bckv = cat_rows({&zerosD, &cvb->data});
    TRY(up_bf16_packed(w, &L.Wqkv, wqkv.data(), 3 * D, D, 16)); TRY(up_f32(w, &L.bqkv, bqkv.data(), bqkv.size()));
    TRY(up_bf16_packed(w, &L.Wo, ow->data.data(), D, D, 16)); TRY(up_f32(w, &L.bo, obias->data.data(), D));
    TRY(up_bf16_packed(w, &L.Wcq, cqw->data.data(), D, D, 16)); TRY(up_f32(w, &L.bcq, cqb->data.data(), D));
    TRY(up_bf16(w, &L.Wckv, wckv.data(), wckv.size())); TRY(up_f32(w, &L.bckv, bckv.data(), bckv.size()));
    TRY(up_bf16_packed(w, &L.Wco, cow->data.data(), D, D, 16)); TRY(up_f32(w, &L.bco, cob->data.data(), D));
    TRY(up_bf16_packed(w, &L.W1, m0w->data.data(), F, D, 16)); TRY(up_f32(w, &L.b1, m0b->data.data(), F));
    TRY(up_bf16(w, &L.W1_plain, m0w->data.data(), m0w->data.size()));
    TRY(up_bf16_packed(w, &L.W2, m2w->data.data(), D, F, 16)); TRY(up_f32(w, &L.b2, m2b->data.data(), D));
    TRY(up_f32(w, &L.ln1_g, l1g->data.data(), D)); TRY(up_f32(w, &L.ln1_b, l1b->data.data(), D));
    TRY(up_f32(w, &L.lnc_g, lcg->data.data(), D)); TRY(up_f32(w, &L.lnc_b, lcb->data.data(), D));
    TRY(up_f32(w, &L.ln2_g, l2g->data.data(), D)); TRY(up_f32(w, &L.ln2_b, l2b->data.data(), D));
    if (w->xs_on) {
      std::vector<float> wkt((size_t)D * D);
      for (int hh = 0; hh < H; hh++)
        for (int f = 0; f < D; f++)
          for (int dd = 0; dd < 64; dd++) wkt[((size_t)hh * D + f) * 64 + dd] = ckw->data[(size_t)(hh * 64 + dd) * D + f];
      TRY(up_bf16(w, &L.WckT, wkt.data(), wkt.size()));
      TRY(up_bf16(w, &L.Wcq_plain, cqw->data.data(), cqw->data.size()));
      // LayerNorm folded into the three consumers of a normalised row (q|k|v, the cross-attention query, the first MLP linear):
      // only built when the experiment is asked for at creation time (CCX_DEC_LNFREE set; 113 MB per small.en instance)
      w->lnfree_built = getenv("CCX_DEC_LNFREE") != nullptr;
      {
      auto fold = [&](const std::vector<float>& W, const std::vector<float>& bias, const std::vector<float>& g, const std::vector<float>& bt,
                      int N, std::vector<float>& Wg, std::vector<float>& sv, std::vector<float>& cv) {
        Wg.resize((size_t)N * D); sv.resize(N); cv.resize(N);
        for (int n = 0; n < N; n++) {
          double ss = 0.0, cc = 0.0;
          for (int k = 0; k < D; k++) {
            const float v = g[k] * W[(size_t)n * D + k];
            Wg[(size_t)n * D + k] = v;
            const uint32_t bits = (uint32_t)host_f32_to_bf16(v) << 16;
            float r;
            memcpy(&r, &bits, 4);
            ss += (double)r;                                   // what the MFMA sums: the bf16-rounded products
            cc += (double)bt[k] * (double)W[(size_t)n * D + k];
          }
          sv[n] = (float)ss;
          cv[n] = (float)(cc + (double)bias[n]);
        }
      };
      std::vector<float> Wg, sv, cv;
      // the cross-attention query's fold is part of the DEFAULT chain (mode 3); the other two only exist for the experiments
      fold(cqw->data, cqb->data, lcg->data, lcb->data, D, Wg, sv, cv);
      TRY(up_bf16(w, &L.Wcq_g, Wg.data(), Wg.size())); TRY(up_f32(w, &L.scq, sv.data(), sv.size())); TRY(up_f32(w, &L.ccq, cv.data(), cv.size()));
      if (w->lnfree_built) {
        fold(wqkv, bqkv, l1g->data, l1b->data, 3 * D, Wg, sv, cv);
        TRY(up_bf16_packed(w, &L.Wqkv_g, Wg.data(), 3 * D, D, 16)); TRY(up_f32(w, &L.sqkv, sv.data(), sv.size())); TRY(up_f32(w, &L.cqkv, cv.data(), cv.size()));
        fold(m0w->data, m0b->data, l2g->data, l2b->data, F, Wg, sv, cv);
        TRY(up_bf16_packed(w, &L.W1_g, Wg.data(), F, D, 16)); TRY(up_f32(w, &L.s1, sv.data(), sv.size())); TRY(up_f32(w, &L.c1, cv.data(), cv.size()));
      }
      }
    }
    const size_t ck = (size_t)w->kv_cap * H * w->Spad * 64, sk = (size_t)B * H * Tc * 64;
    TRY(dev_alloc(w, &L.crossK, ck, true)); TRY(dev_alloc(w, &L.crossV, ck, true));
    TRY(dev_alloc(w, &L.selfK, sk, true)); TRY(dev_alloc(w, &L.selfV, sk, true));
  }
  w->staged.clear();

  // ---------------- workspaces ----------------
  // log-mel and encoder: only live between ccx_whisper_logmel and the end of ccx_whisper_encode (the cross-KV it leaves behind is
  // per instance), so two instances whose log-mel / encode calls are ordered on ONE stream may share them (32 GB at 768 windows)
  if (ccx_whisper* dn = w->scratch_donor) {
    w->lm_raw = dn->lm_raw; w->lm_max = dn->lm_max; w->lm_n = dn->lm_n; w->lm_seek = dn->lm_seek; w->lm_seg = dn->lm_seg;
    w->im2col = dn->im2col; w->h1 = dn->h1; w->x = dn->x; w->xn = dn->xn;
    w->qb = dn->qb; w->kb = dn->kb; w->vtb = dn->vtb; w->attn = dn->attn; w->ffn = dn->ffn;
  } else {
    TRY(dev_alloc(w, &w->lm_raw, (size_t)B * 80 * w->Fraw, true));
    TRY(dev_alloc(w, &w->lm_max, (size_t)B, true));
    TRY(dev_alloc(w, &w->lm_n, (size_t)B, true));
    TRY(dev_alloc(w, &w->lm_seek, (size_t)B, true));
    TRY(dev_alloc(w, &w->lm_seg, (size_t)B, true));
    TRY(dev_alloc(w, &w->im2col, (size_t)B * 3000 * 256, true));
    TRY(dev_alloc(w, &w->h1, ((size_t)B * 3002 + 2) * D, true));
    TRY(dev_alloc(w, &w->x, (size_t)B * S * D, true));
    TRY(dev_alloc(w, &w->xn, (size_t)B * S * D, true));
    TRY(dev_alloc(w, &w->qb, (size_t)B * H * w->Spad * 64, true));
    TRY(dev_alloc(w, &w->kb, (size_t)B * H * w->Spad * 64, true));
    TRY(dev_alloc(w, &w->vtb, (size_t)B * H * 64 * w->Spad, true));
    TRY(dev_alloc(w, &w->attn, (size_t)B * S * D, true));
    TRY(dev_alloc(w, &w->ffn, (size_t)B * S * F, true));
  }

  // the encoder output stays with the instance: the decode streams it (cross_x.hip); + one key tile of slack
  TRY(dev_alloc(w, &w->xa, ((size_t)B * S + 16) * D, true));
  if (w->xs_on) {
    TRY(dev_alloc(w, &w->xq, (size_t)B * H * D, true));
    TRY(dev_alloc(w, &w->pf_xq, (size_t)B * ccx_whisper::kPrefillMax * H * D, true));
    TRY(dev_alloc(w, &w->xs_po, ccx_xs_part_o_elems(B, H, D), true));
    TRY(dev_alloc(w, &w->xs_pml, ccx_xs_part_ml_elems(B), true));
    TRY(dev_alloc(w, &w->pf_xs_po, ccx_xs_part_o_elems((size_t)B * ccx_whisper::kPrefillMax, H, D), true));
    TRY(dev_alloc(w, &w->pf_xs_pml, ccx_xs_part_ml_elems((size_t)B * ccx_whisper::kPrefillMax), true));
    TRY(dev_alloc(w, &w->dxb, (size_t)B * D, true));
    TRY(dev_alloc(w, &w->pf_xb, (size_t)B * ccx_whisper::kPrefillMax * D, true));
    TRY(dev_alloc(w, &w->dst2, (size_t)B * (D / 16), true));
    TRY(dev_alloc(w, &w->pf_st2, (size_t)B * ccx_whisper::kPrefillMax * (D / 16), true));
  }
  TRY(dev_alloc(w, &w->dx, (size_t)B * D, true));
  TRY(dev_alloc(w, &w->dx2, (size_t)B * D, true));
  TRY(dev_alloc(w, &w->pend, (size_t)4 * B * D, true));
  TRY(dev_alloc(w, &w->dxn, (size_t)B * D, true));
  TRY(dev_alloc(w, &w->dq, (size_t)B * D, true));
  TRY(dev_alloc(w, &w->dattn, (size_t)B * D, true));
  TRY(dev_alloc(w, &w->dffn, (size_t)B * F, true));
  TRY(dev_alloc(w, &w->dlogits, (size_t)B * w->Vpad, true));
  TRY(dev_alloc(w, &w->part_o, (size_t)B * H * ccx_whisper::kCrossSplitMax * 64, true));
  TRY(dev_alloc(w, &w->part_ml, (size_t)B * H * ccx_whisper::kCrossSplitMax * 2, true));
  TRY(dev_alloc(w, &w->cur_tok, (size_t)B, true));
  TRY(dev_alloc(w, &w->pos, (size_t)B, true));
  TRY(dev_alloc(w, &w->n_done, (size_t)ccx_whisper::kMaxLanes, true));
  TRY(dev_alloc(w, &w->sample_cfg, (size_t)4, true));
  TRY(dev_alloc(w, &w->state, (size_t)B, true));
  {
    const size_t R = (size_t)B * ccx_whisper::kPrefillMax;
    TRY(dev_alloc(w, &w->pf_x, R * D, true)); TRY(dev_alloc(w, &w->pf_x2, R * D, true)); TRY(dev_alloc(w, &w->pf_pend, 4 * R * D, true));
    TRY(dev_alloc(w, &w->pf_q, R * D, true)); TRY(dev_alloc(w, &w->pf_xn, R * D, true)); TRY(dev_alloc(w, &w->pf_attn, R * D, true));
    TRY(dev_alloc(w, &w->pf_ffn, R * F, true));
    TRY(dev_alloc(w, &w->pf_tok, R, true)); TRY(dev_alloc(w, &w->pf_pos, R, true)); TRY(dev_alloc(w, &w->pf_seq, R, true));
    TRY(dev_alloc(w, &w->pf_last, (size_t)B, true));
  }
  w->max_prompt_cap = d.n_text_ctx;
  w->sample_cap = d.n_text_ctx;
  TRY(dev_alloc(w, &w->prompt, (size_t)B * w->max_prompt_cap, true));
  TRY(dev_alloc(w, &w->gen, (size_t)B * w->sample_cap, true));
  // decode streams get the highest priority: when other work shares the GPU (the front end of the next batch on another
  // stream) the short, latency-bound chain kernels should get the next free wave slot.  CCX_LANE_PRIORITY=0 disables.
  int prio_least = 0, prio_greatest = 0;
  CCX_HIP(w->ctx, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  { const char* e = getenv("CCX_LANE_PRIORITY"); if (e && atoi(e) == 0) prio_greatest = 0; }
  CCX_HIP(w->ctx, hipStreamCreateWithPriority(&w->own_stream, hipStreamNonBlocking, prio_greatest));
  CCX_HIP(w->ctx, hipEventCreateWithFlags(&w->own_event, hipEventDisableTiming));
  for (int i = 0; i < ccx_whisper::kMaxLanes; i++) {
    CCX_HIP(w->ctx, hipEventCreateWithFlags(&w->lane_start[i], hipEventDisableTiming));
    CCX_HIP(w->ctx, hipEventCreateWithFlags(&w->lane_poll[0][i], hipEventDisableTiming));
    CCX_HIP(w->ctx, hipEventCreateWithFlags(&w->lane_poll[1][i], hipEventDisableTiming));
  }
  for (int i = 0; i < ccx_whisper::kLanePool; i++) CCX_HIP(w->ctx, hipStreamCreateWithPriority(&w->lane_pool[i], hipStreamNonBlocking, prio_greatest));
  TRY(dev_alloc(w, &w->probe_sink, (size_t)64, true));
  // lane trace buffers (2 MB): always there so that ccx_whisper_trace_lanes can switch the trace on later
  TRY(dev_alloc(w, &w->stamps, (size_t)ccx_whisper::kMaxLanes * ccx_whisper::kStampCap, true));
  TRY(dev_alloc(w, &w->stamp_count, (size_t)ccx_whisper::kMaxLanes, true));
  if (const char* e = getenv("CCX_DEC_STAMPS")) {
    w->stamps_on = true;
    w->stamp_path = e;
    if (const char* l = getenv("CCX_DEC_STAMP_LEVEL")) w->stamp_level = atoi(l);
  }
  CCX_HIP(w->ctx, hipHostMalloc((void**)&w->poll_host, 2 * ccx_whisper::kMaxLanes * sizeof(int), 0));
  w->finalized = true;
  return CCX_OK;
}

// Shared log-mel / encoder workspaces (ccx_whisper_share_encoder_scratch): every user waits for the end of the group's previous
// encode before it writes them, on whatever stream it runs (a no-op when all users share one stream).  The event only orders a
// logmel behind the previous ENCODE: between an instance's logmel / set_mel and its encode the workspaces hold its staged windows, so
// the host must issue that pair back to back -- enforced here: another instance's logmel in between is refused (include/ccx.h).
static int scratch_acquire(ccx_whisper* w, hipStream_t stream) {
  ccx_whisper* root = w->scratch_donor ? w->scratch_donor : w;
  if (root->scratch_free) {
    ccx_whisper* owner = root->scratch_owner.load(std::memory_order_acquire);
    CCX_REQUIRE(w->ctx, owner == nullptr || owner == w,
                "whisper: the shared encoder workspaces hold another instance's staged windows -- issue its ccx_whisper_encode before this logmel / set_mel");
    root->scratch_owner.store(w, std::memory_order_release);
    CCX_HIP(w->ctx, hipStreamWaitEvent(stream, root->scratch_free, 0));
  }
  return CCX_OK;
}
static int scratch_release(ccx_whisper* w, hipStream_t stream) {
  ccx_whisper* root = w->scratch_donor ? w->scratch_donor : w;
  if (root->scratch_free) {
    CCX_HIP(w->ctx, hipEventRecord(root->scratch_free, stream));
    ccx_whisper* me = w;
    root->scratch_owner.compare_exchange_strong(me, nullptr, std::memory_order_acq_rel);
  }
  return CCX_OK;
}

int ccx_whisper_logmel(ccx_whisper* w, const float* audio, int64_t stride, const int* n_samples, const int* seek, int B,
                       float* mel_out, void* stream_) {
  if (!w) return CCX_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  CCX_REQUIRE(w->ctx, w->finalized, "whisper: not finalized");
  CCX_REQUIRE(w->ctx, audio && n_samples && B >= 1 && B <= w->max_batch, "whisper_logmel: bad arguments (B=%d, max %d)", B, w->max_batch);
  TRY(scratch_acquire(w, stream));
  long fcomp = 32;
  for (int b = 0; b < B; b++) {
    CCX_REQUIRE(w->ctx, n_samples[b] >= 0 && n_samples[b] <= stride, "whisper_logmel: n_samples[%d]=%d exceeds stride", b, n_samples[b]);
    const int s = seek ? seek[b] : 0;
    // frames that touch audio content must lie inside the computed range
    const long need_frames = ((long)n_samples[b] + 200 + 159) / 160 + 1;
    CCX_REQUIRE(w->ctx, need_frames <= w->Fraw, "whisper_logmel: clip %d (%d samples) exceeds the configured maximum of %d frames (ccx_whisper_set_max_audio)", b, n_samples[b], w->Fraw);
    if (need_frames > fcomp) fcomp = need_frames;
    CCX_REQUIRE(w->ctx, s >= 0, "whisper_logmel: negative seek");
  }
  // segment_size = min(N_FRAMES, content_frames - seek), content_frames = n_samples // 160 (transcribe.py)
  std::vector<int> seg(B), sk(B);
  for (int b = 0; b < B; b++) {
    sk[b] = seek ? seek[b] : 0;
    int sl = n_samples[b] / 160 - sk[b];
    seg[b] = sl < 0 ? 0 : (sl > 3000 ? 3000 : sl);
  }
  CCX_HIP(w->ctx, hipMemcpyAsync(w->lm_n, n_samples, B * 4, hipMemcpyHostToDevice, stream));
  CCX_HIP(w->ctx, hipMemcpyAsync(w->lm_seek, sk.data(), B * 4, hipMemcpyHostToDevice, stream));
  CCX_HIP(w->ctx, hipMemcpyAsync(w->lm_seg, seg.data(), B * 4, hipMemcpyHostToDevice, stream));
  CCX_HIP(w->ctx, hipStreamSynchronize(stream));  // host staging vectors go out of scope
  fcomp = (fcomp + 31) / 32 * 32;   // only frames that touch audio are computed; everything later is log10(1e-10)
  return ccx_launch_logmel(w->ctx, w->lm, audio, stride, w->lm_n, w->lm_seek, w->lm_seg, B, w->Fraw, (int)fcomp, w->lm_raw,
                           w->lm_max, mel_out, w->im2col, stream);
}

__global__ void mel_to_im2col_kernel(const float* __restrict__ mel, bf16_t* __restrict__ im2col) {
  // mel [B][80][3000] -> im2col [B*3000][256], k = tap*80 + c holds frame t-1+tap
  const int b = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 8), k = threadIdx.x & 255;
  if (t >= 3000) return;
  float v = 0.f;
  if (k < 240) {
    const int tap = k / 80, c = k - tap * 80, fr = t - 1 + tap;
    if (fr >= 0 && fr < 3000) v = mel[((long)b * 80 + c) * 3000 + fr];
  }
  im2col[((long)b * 3000 + t) * 256 + k] = f32_to_bf16(v);
}

int ccx_whisper_set_mel(ccx_whisper* w, const float* mel, int B, void* stream_) {
  if (!w) return CCX_ERR_ARG;
  CCX_REQUIRE(w->ctx, w->finalized && mel && B >= 1 && B <= w->max_batch, "whisper_set_mel: bad arguments");
  TRY(scratch_acquire(w, (hipStream_t)stream_));
  hipLaunchKernelGGL(mel_to_im2col_kernel, dim3(750, B), dim3(1024), 0, (hipStream_t)stream_, mel, w->im2col);
  CCX_CHECK_LAUNCH(w->ctx);
  return CCX_OK;
}

// cross-attention K/V of sequences [0, n) for every decoder layer out of xa (head-major, not transposed: decode streams rows)
static int project_cross_kv(ccx_whisper* w, int n, hipStream_t stream) {
  const ccx_whisper_dims& d = w->d;
  const int D = d.n_audio_state, S = d.n_audio_ctx, H = d.n_audio_head;
  CCX_REQUIRE(w->ctx, n <= w->kv_cap, "whisper: cross K/V caches hold %d sequences, %d asked for", w->kv_cap, n);
  GemmParams p;
  for (int l = 0; l < d.n_text_layer; l++) {
    const DecLayer& L = w->dec[l];
    memset(&p, 0, sizeof(p));
    p.A = w->xa; p.lda = D; p.W = L.Wckv; p.ldw = D; p.M = n * S; p.N = 2 * D; p.K = D; p.bias = L.bckv;
    p.hk = L.crossK; p.hv = L.crossV; p.d_model = D; p.n_head = H; p.S = S; p.Spad = w->Spad; p.v_transposed = 0;
    p.first_block = 1;
    TRY(ccx_launch_gemm(w->ctx, EPI_HEADS, p, stream));
  }
  w->kv_ready = n;
  return CCX_OK;
}
// which cross attention a decode of B sequences uses, and the K/V it needs
static int select_cross_path(ccx_whisper* w, int B, hipStream_t stream) {
  // CCX_CROSS_X_MIN_ROWS (read per call: tests flip it): smallest decode that takes the X-stream path.  Default 81: the streaming kernel
  // runs two blocks per sequence, and below ~160 blocks they do not fill the chip -- Whisper only, 24 / 32 / 48 / 64 / 96 x 30 s:
  // 300.7 / 311.8 / 340.0 / 357.9 / 404.3 ms per batch on the X-stream path against 208.6 / 223.3 / 289.4 / 329.3 / 427.2 on per-layer
  // K / V (split-KV kernels with the fused query up to 16 sequences, the lean streaming kernel above).  Within one path a sequence's
  // numbers do not depend on its batch mates; across the paths they agree to rounding (tests/test_whisper_gpu.py).  Decodes the K / V
  // caches cannot hold (kv_cap sequences) always take the X-stream path.
  const char* e = getenv("CCX_CROSS_X_MIN_ROWS");
  const int min_rows = e ? atoi(e) : ccx_whisper::kKvSeqs + 1;
  w->xs_active = w->xs_on && (B >= min_rows || B > w->kv_cap);
  { const char* f = getenv("CCX_XS_FUSE_Q"); w->xs_fuse_q = !f || atoi(f) != 0; }      // read per call: tests flip it
  if (w->xs_on && !w->xs_active && w->kv_ready < B) TRY(project_cross_kv(w, B, stream));
  return CCX_OK;
}

int ccx_whisper_encode(ccx_whisper* w, int B, float* xa_out, void* stream_) {
  if (!w) return CCX_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  ccx_ctx* ctx = w->ctx;
  CCX_REQUIRE(ctx, w->finalized, "whisper: not finalized");
  CCX_REQUIRE(ctx, B >= 1 && B <= w->max_batch, "whisper_encode: B=%d out of range (max %d)", B, w->max_batch);
  const ccx_whisper_dims& d = w->d;
  const int D = d.n_audio_state, F = 4 * D, S = d.n_audio_ctx, H = d.n_audio_head;
  GemmParams p;
  // conv1 + GELU: [B*3000,256] x [D,256]^T -> h1 rows b*3002 + 1 + t
  memset(&p, 0, sizeof(p));
  p.A = w->im2col; p.lda = 256; p.W = w->Wc1; p.ldw = 256; p.M = B * 3000; p.N = D; p.K = 256;
  p.bias = w->bc1; p.out = w->h1; p.ldo = D; p.rpb_in = 3000; p.rpb_out = 3002; p.roff = 1; p.rpb_valid = 3000;
  TRY(ccx_launch_gemm(ctx, EPI_BF16_GELU, p, stream));
  // conv2 (stride 2) + GELU + positional embedding: row m' = b*1501 + t reads h1 rows 2m' .. 2m'+2
  memset(&p, 0, sizeof(p));
  p.A = w->h1; p.lda = 2 * D; p.W = w->Wc2; p.ldw = 3 * D; p.M = B * 1501; p.N = D; p.K = 3 * D;
  p.bias = w->bc2; p.out = w->x; p.ldo = D; p.resid = w->enc_pos; p.ldr = D; p.resid_mod = S;
  p.rpb_in = 1501; p.rpb_out = S; p.roff = 0; p.rpb_valid = S;
  TRY(ccx_launch_gemm(ctx, EPI_F32_GELU_POS, p, stream));

  const int M = B * S;
  for (int l = 0; l < d.n_audio_layer; l++) {
    const EncLayer& L = w->enc[l];
    TRY(ccx_launch_layernorm(ctx, w->x, D, L.ln1_g, L.ln1_b, w->xn, nullptr, D, M, D, 1e-5f, stream));
    memset(&p, 0, sizeof(p));
    p.A = w->xn; p.lda = D; p.W = L.Wqkv; p.ldw = D; p.M = M; p.N = 3 * D; p.K = D; p.bias = L.bqkv;
    p.hq = w->qb; p.hk = w->kb; p.hv = w->vtb; p.d_model = D; p.n_head = H; p.S = S; p.Spad = w->Spad; p.v_transposed = 1;
    TRY(ccx_launch_gemm(ctx, EPI_HEADS, p, stream));
    TRY(ccx_launch_enc_attention(ctx, w->qb, w->kb, w->vtb, w->attn, B, H, S, w->Spad, stream));
    memset(&p, 0, sizeof(p));
    p.A = w->attn; p.lda = D; p.W = L.Wo; p.ldw = D; p.M = M; p.N = D; p.K = D; p.bias = L.bo;
    p.out = w->x; p.ldo = D; p.resid = w->x; p.ldr = D;
    TRY(ccx_launch_gemm(ctx, EPI_F32_RESID, p, stream));
    TRY(ccx_launch_layernorm(ctx, w->x, D, L.ln2_g, L.ln2_b, w->xn, nullptr, D, M, D, 1e-5f, stream));
    memset(&p, 0, sizeof(p));
    p.A = w->xn; p.lda = D; p.W = L.W1; p.ldw = D; p.M = M; p.N = F; p.K = D; p.bias = L.b1; p.out = w->ffn; p.ldo = F;
    TRY(ccx_launch_gemm(ctx, EPI_BF16_GELU, p, stream));
    memset(&p, 0, sizeof(p));
    p.A = w->ffn; p.lda = F; p.W = L.W2; p.ldw = F; p.M = M; p.N = D; p.K = F; p.bias = L.b2;
    p.out = w->x; p.ldo = D; p.resid = w->x; p.ldr = D;
    TRY(ccx_launch_gemm(ctx, EPI_F32_RESID, p, stream));
  }
  TRY(ccx_launch_layernorm(ctx, w->x, D, w->lnp_g, w->lnp_b, w->xa, xa_out, D, M, D, 1e-5f, stream));
  TRY(scratch_release(w, stream));
  // cross-attention K/V: with the X-stream cross attention only decodes of <= 80 sequences use them and project them themselves
  w->kv_ready = 0;
  if (!w->xs_on) TRY(project_cross_kv(w, B, stream));
  return CCX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------------------------
namespace {

// ---- diagnostic time stamps (CCX_DEC_STAMPS=<file>, CCX_DEC_STAMP_LEVEL=1|2) ----
// A one-thread kernel that appends the 100 MHz wall clock (s_memrealtime) and a tag to the lane's trace; captured into the step
// graphs like any other node.  Level 1 brackets the cross attention of every layer (tags 1 / 2), level 2 also stamps after every
// chain kernel (tag 16 + kernel index in the layer).  Perturbs what it measures (one more ~2 us node per stamp): level 1 adds 24
// nodes to a step of ~150, and is what tools/decode_stamps.py reads.
__global__ void dec_stamp_kernel(unsigned long long* trace, int* count, int cap, int tag) {
  const int i = atomicAdd(count, 1);
  if (i < cap) trace[i] = (__builtin_amdgcn_s_memrealtime() << 8) | (unsigned long long)(tag & 255);
}

// ---- lane stream selection ----
// HIP multiplexes streams onto a handful of hardware queues, and streams that share a queue execute strictly one
// after the other (tools/microbench_lanes2.hip: some pairs of 8 fresh streams take 2x, the others 1x).  The mapping
// is fixed when a stream is created but not queryable, so it is measured: a few ~50 us spin kernels per stream.
__global__ void lane_probe_spin(int* sink, long cycles) {
  const long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (sink && threadIdx.x == 0) atomicAdd(sink, 1);
}

double lane_probe_time(ccx_whisper* w, hipStream_t a, hipStream_t b) {
  hipStreamSynchronize(a);
  if (b) hipStreamSynchronize(b);
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < 4; k++) {
    hipLaunchKernelGGL(lane_probe_spin, dim3(1), dim3(64), 0, a, w->probe_sink, 120000L);
    if (b) hipLaunchKernelGGL(lane_probe_spin, dim3(1), dim3(64), 0, b, w->probe_sink + 1, 120000L);
  }
  hipStreamSynchronize(a);
  if (b) hipStreamSynchronize(b);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

// streams (up to `want`) that run concurrently with s0 and with each other; fewer if the probe finds fewer
const std::vector<hipStream_t>& lane_streams_for(ccx_whisper* w, hipStream_t s0, int want) {
  auto it = w->lane_sets.find(s0);
  if (it != w->lane_sets.end()) return it->second;   // probed once per lane-0 stream (always for the maximum)
  (void)want;
  std::vector<hipStream_t>& set = w->lane_sets[s0];
  set.clear();
  lane_probe_time(w, s0, nullptr);                        // warm-up (code object load)
  double single = lane_probe_time(w, s0, nullptr);
  const double s2 = lane_probe_time(w, s0, nullptr);
  if (s2 < single) single = s2;
  for (int c = 0; c < ccx_whisper::kLanePool && (int)set.size() < ccx_whisper::kMaxLanes - 1; c++) {
    hipStream_t cand = w->lane_pool[c];
    bool ok = lane_probe_time(w, s0, cand) < 1.5 * single;
    for (size_t j = 0; ok && j < set.size(); j++) ok = lane_probe_time(w, set[j], cand) < 1.5 * single;
    if (ok) set.push_back(cand);
  }
  if (getenv("CCX_DEBUG_LANES")) fprintf(stderr, "[lanes] probe: single %.0f us, %zu concurrent streams found\n", single, set.size());
  return set;
}

int cross_split(int B, int H, bool capped, bool lean = false) {
  if (lean && B > 16) {
    // lean streaming: one block owns the whole key range of a (sequence, head) -- 12 rolling 32-key pieces per wave, FINAL output,
    // no partials and no combine launch; CCX_CROSS_SPLIT overrides
    const char* e = getenv("CCX_CROSS_SPLIT");
    const int forced = e ? atoi(e) : 0;
    return (forced >= 1 && forced <= ccx_whisper::kCrossSplitMax) ? forced : 1;
  }
  // enough blocks to fill the chip, and <= 256 keys per block (one 64-key chunk per wave)
  int ns = ccx_cdiv(512, B * H);
  if (ns < 6) ns = 6;
  if (ns > ccx_whisper::kCrossSplitMax) ns = ccx_whisper::kCrossSplitMax;
  if (B > 16) {
    // lanes with the two-blocks-per-CU cap: 3 splits, i.e. two 64-key chunks per wave and half as many, longer-lived blocks,
    // which suits the capped kernel better (pipeline step, ms: 2 splits 900, 3 891,
    // 4 893, 6 912; per 64-sequence launch 49.9 us against 52.0 at 6 splits)
    if (capped) ns = 3;
    // many sequences: CCX_CROSS_SPLIT=n overrides (1 = whole key range per block, no partials / combine).  Measured at
    // 64 sequences per lane: 6 splits 54.5 us per launch, 8 splits 56.8 us (a wave then owns 47 of a chunk's 64 keys).
    const char* e = getenv("CCX_CROSS_SPLIT");
    const int forced = e ? atoi(e) : 0;
    if (forced >= 1 && forced <= ccx_whisper::kCrossSplitMax) ns = forced;
  }
  return ns;
}

// Tail of a step for the sequences [b0, b0 + B): logits of the final-LayerNorm rows (w->dxn) against the tied embedding, then
// the select kernel (filters, argmax / sampling, state machine, next step's embedding).
int dec_head(ccx_whisper* w, int b0, int B, float* logits, long ld, bool select, int sample_len, int max_prompt, int* n_done,
             hipStream_t stream) {
  ccx_ctx* ctx = w->ctx;
  const ccx_whisper_dims& d = w->d;
  const int D = d.n_text_state;
  const long ro = b0;
  bf16_t* dxn = w->dxn + ro * D;
  int* pos = w->pos + b0;
  {
    // logits against the tied embedding through the tiled GEMM (one summation order for every batch size; 284 -> 282 ms
    // for 192 sequences x 65 steps, no change at 8 sequences); CCX_LOGITS_GEMM=0 selects the skinny kernel
    static const int gemm_logits = [] { const char* e = getenv("CCX_LOGITS_GEMM"); return e ? atoi(e) : 1; }();
    if (gemm_logits && D % 64 == 0) {
      GemmParams gp;
      memset(&gp, 0, sizeof(gp));
      gp.A = dxn; gp.lda = D; gp.W = w->tok_emb_rm; gp.ldw = D; gp.M = B; gp.N = d.n_vocab; gp.K = D; gp.out = logits; gp.ldo = ld;
      TRY(ccx_launch_gemm(ctx, EPI_F32, gp, stream));
    } else {
      DecLinearParams lp;
      memset(&lp, 0, sizeof(lp));
      lp.M = B; lp.N = d.n_vocab; lp.K = D; lp.W = w->tok_emb_bf16; lp.ldw = D; lp.bias = nullptr;
      lp.act = dxn; lp.lda = D; lp.out = logits; lp.ldo = ld;
      TRY(ccx_launch_dec_linear(ctx, ACT_BF16, DEPI_F32, lp, stream));
    }
  }
  if (select) {
    DecSelectParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.logits = logits; sp.ld_logits = ld; sp.n_vocab = d.n_vocab; sp.state = w->state + b0; sp.prompt = w->prompt + ro * max_prompt;
    sp.max_prompt = max_prompt; sp.cur_tok = w->cur_tok + b0; sp.pos = pos; sp.gen = w->gen + ro * sample_len; sp.sample_len = sample_len;
    sp.n_done = n_done; sp.suppress_mask = w->suppress_mask; sp.eot = w->rules.eot; sp.blank = w->rules.blank;
    sp.no_speech = w->rules.no_speech; sp.timestamp_begin = w->rules.timestamp_begin;
    sp.max_initial_ts = w->rules.max_initial_timestamp_index;
    sp.tok_emb = w->tok_emb_f32; sp.pos_emb = w->dec_pos; sp.x = w->dx + ro * D; sp.D = D;
    sp.sample_cfg = w->sample_cfg; sp.row0 = b0; sp.sample = w->sampling ? 1 : 0;
    TRY(ccx_launch_dec_select(ctx, sp, B, stream));
  }
  return CCX_OK;
}

// One decoder step for the B sequences [b0, b0 + B) on `stream` (a "lane": every per-sequence buffer is
// addressed through its row offset, so disjoint lanes can step concurrently on different streams).
// logits go to `logits` (row 0 = sequence b0) with row stride ld.  The residual stream ping-pongs between
// dx and dx2: out-proj / cross-out / FFN2 only write split-K partial slabs (pend) and the next LayerNorm
// folds them in (decoder.hip).  `stagger`, if set, is recorded right before layer 0's cross attention.
// `prefill_rows` = P > 0: the PROMPT PREFILL pass instead of a step -- B sequences x P prompt positions as B * P rows of the pf_*
// buffers (row = sequence * P + position), self-KV written at every position, cross attention with all rows of a sequence
// sharing its K/V, no logits / select (the caller takes the last prompt row of every sequence).
int dec_step(ccx_whisper* w, int b0, int B, float* logits, long ld, bool select, int sample_len, int max_prompt, int* n_done,
             hipStream_t stream, hipEvent_t stagger = nullptr, int lane_idx = 0, int prefill_rows = 0) {
  ccx_ctx* ctx = w->ctx;
  const ccx_whisper_dims& d = w->d;
  const int D = d.n_text_state, F = 4 * D, H = d.n_text_head, Tc = d.n_text_ctx;
  const float scale_log2e = 0.125f * 1.4426950408889634f;
  const int ns = cross_split(B, H, w->cross_lds_pad > 0, w->cross_stream != 0);
  long pstride = (long)B * D;
  const long ro = b0;
  const bool pre = prefill_rows > 0;
  const int nseq = B;                                  // sequences of this call
  if (pre) B = nseq * prefill_rows;                    // rows the chain works on
  pstride = (long)B * D;
  float* cur = pre ? w->pf_x : w->dx + ro * D;     // stream (minus the pending partials); the step's embedding is in dx
  float* other = pre ? w->pf_x2 : w->dx2 + ro * D;
  float* pend = pre ? w->pf_pend : w->pend + 4 * ro * D;
  float* dq = pre ? w->pf_q : w->dq + ro * D;
  bf16_t* dxn = pre ? w->pf_xn : w->dxn + ro * D;
  bf16_t* dattn = pre ? w->pf_attn : w->dattn + ro * D;
  bf16_t* dffn = pre ? w->pf_ffn : w->dffn + ro * F;
  float* part_o = w->part_o + ro * H * ccx_whisper::kCrossSplitMax * 64;
  float* part_ml = w->part_ml + ro * H * ccx_whisper::kCrossSplitMax * 2;
  int* pos = pre ? w->pf_pos : w->pos + b0;
  const int* row_seq = pre ? w->pf_seq : nullptr;
  const long cross_off = ro * H * w->Spad * 64, self_off = ro * H * Tc * 64;
  const int stamp_level = w->stamp_level;
  auto stamp = [&](int tag, int level) {
    if (w->stamps_on && level <= stamp_level)
      hipLaunchKernelGGL(dec_stamp_kernel, dim3(1), dim3(1), 0, stream, w->stamps + (size_t)lane_idx * ccx_whisper::kStampCap,
                         w->stamp_count + lane_idx, ccx_whisper::kStampCap, tag);
  };
  int pend_n = 0;
  auto ln_linear = [&](int epi, const bf16_t* W, const float* bias, int N, const float* g, const float* bta, void* out, long ldo,
                       DecLinearParams* extra) -> int {
    DecLinearParams lp;
    if (extra) lp = *extra; else memset(&lp, 0, sizeof(lp));
    lp.M = B; lp.N = N; lp.K = D; lp.W = W; lp.ldw = D; lp.bias = bias; lp.out = out; lp.ldo = ldo;
    int rc;
    if (B > 16) {
      // many sequences: normalise ONCE in a stand-alone kernel instead of redundantly in every weight-panel block
      rc = ccx_launch_dec_resolve_ln(ctx, cur, pend, pend_n, pstride, g, bta, dxn, pend_n > 0 ? other : nullptr, B, D, 1e-5f, stream);
      if (rc) return rc;
      lp.act = dxn; lp.lda = D;
      rc = ccx_launch_dec_linear(ctx, ACT_BF16, epi, lp, stream);
    } else {
      lp.x = cur; lp.pend = pend; lp.pend_n = pend_n; lp.pend_stride = pstride; lp.x_out = pend_n > 0 ? other : nullptr;
      lp.ln_g = g; lp.ln_b = bta; lp.eps = 1e-5f;
      rc = ccx_launch_dec_linear(ctx, ACT_LN, epi, lp, stream);
    }
    if (rc) return rc;
    if (pend_n > 0) { float* t = cur; cur = other; other = t; pend_n = 0; }
    return CCX_OK;
  };
  auto partial_linear = [&](int act, const bf16_t* W, const float* bias, int K, const bf16_t* a) -> int {
    DecLinearParams lp;
    memset(&lp, 0, sizeof(lp));
    lp.M = B; lp.N = D; lp.K = K; lp.W = W; lp.ldw = K; lp.bias = bias; lp.act = a; lp.lda = K;
    lp.part_o = part_o; lp.part_ml = part_ml; lp.nsplit = ns;
    lp.out = pend; lp.ldo = D; lp.pend_stride = pstride;
    int rc = ccx_launch_dec_linear(ctx, act, DEPI_PARTIAL, lp, stream);
    if (rc) return rc;
    pend_n = ccx_dec_linear_ksplit(K, DEPI_PARTIAL);
    return CCX_OK;
  };
  // diagnostic only (results are garbage): CCX_ABLATE=cross drops the cross attention launches, =chain everything else of a layer
  static const int ablate = [] { const char* e = getenv("CCX_ABLATE"); return !e ? 0 : (!strcmp(e, "cross") ? 1 : (!strcmp(e, "chain") ? 2 : 0)); }();
  // ---- the LayerNorm-free chain of the X-stream path (CCX_DEC_LNFREE=1): 9 launches per layer and no stand-alone resolve / LayerNorm.
  // The PRODUCERS of the residual stream (self-attention out, cross-attention out, second MLP linear: DEPI_RESOLVE) add their product
  // to the stream in place -- no split-K slabs, K = 3072 goes to 12 waves per block -- and leave a bf16 copy of the new rows plus
  // (sum, sum of squares) per 16-column tile; the CONSUMERS of a normalised row (q|k|v, the cross-attention query inside the
  // expansion, the first MLP linear: ACT_BF16_LN) read the bf16 rows as they are and apply the LayerNorm algebraically in their
  // epilogue (gamma folded into the weights at load time).  One set of kernels for every row count, statistics per tile in a fixed
  // order: a row's numbers do not depend on its lane.  Layer 0 normalises the step's embedding the old way (nothing produced it).
  // (CCX_DEC_LNFREE=3: only the self-attention output projection resolves in place and only the cross-attention query uses the algebra --
  //  the default chain without the twelve-fold resolve + LayerNorm inside dec_xq_fused_kernel; handled in the default loop below)
  //  CCX_DEC_LNFREE=4: only the cross-attention output projection resolves in place and only the first MLP linear uses the algebra; 5: both)
  const bool lnf_on = w->xs_active && ablate == 0;
  const bool lnfree3 = lnf_on && (w->lnfree_mode == 3 || w->lnfree_mode == 5);
  const bool lnfree4 = lnf_on && (w->lnfree_mode == 4 || w->lnfree_mode == 5);
  if (w->lnfree && w->lnfree_mode < 3 && w->xs_active && ablate == 0) {
    bf16_t* xb = pre ? w->pf_xb : w->dxb + ro * D;
    float2* st2 = pre ? w->pf_st2 : w->dst2 + ro * (D / 16);
    auto consumer = [&](int epi, const bf16_t* Wg, const float* sv, const float* cv, int N, void* out, long ldo, DecLinearParams* extra) -> int {
      DecLinearParams lp;
      if (extra) lp = *extra; else memset(&lp, 0, sizeof(lp));
      lp.M = B; lp.N = N; lp.K = D; lp.W = Wg; lp.ldw = D; lp.bias = cv; lp.ln_s = sv; lp.ln_stats = st2; lp.eps = 1e-5f;
      lp.act = xb; lp.lda = D; lp.out = out; lp.ldo = ldo;
      return ccx_launch_dec_linear(ctx, ACT_BF16_LN, epi, lp, stream);
    };
    auto producer = [&](const bf16_t* W, const float* bias, int K, const bf16_t* a) -> int {
      DecLinearParams lp;
      memset(&lp, 0, sizeof(lp));
      lp.M = B; lp.N = D; lp.K = K; lp.W = W; lp.ldw = K; lp.bias = bias; lp.act = a; lp.lda = K;
      lp.xres = cur; lp.xb = xb; lp.st_out = st2;
      return ccx_launch_dec_linear(ctx, ACT_BF16, DEPI_RESOLVE, lp, stream);
    };
    for (int l = 0; l < d.n_text_layer; l++) {
      const DecLayer& L = w->dec[l];
      DecLinearParams ex;
      memset(&ex, 0, sizeof(ex));
      ex.cache_k = L.selfK + self_off; ex.cache_v = L.selfV + self_off; ex.cache_T = Tc; ex.pos = pos; ex.row_seq = row_seq;
      if (l == 0) TRY(ln_linear(DEPI_SELF_QKV, L.Wqkv, L.bqkv, 3 * D, L.ln1_g, L.ln1_b, dq, D, &ex));
      else {
        // (CCX_DEC_LNFREE=2: the second MLP linear kept its split-K slabs -- fold them in here, one small launch)
        if (pend_n > 0) { TRY(ccx_launch_dec_resolve_stats(ctx, cur, pend, pend_n, pstride, xb, st2, B, D, stream)); pend_n = 0; }
        TRY(consumer(DEPI_SELF_QKV, L.Wqkv_g, L.sqkv, L.cqkv, 3 * D, dq, D, &ex));
      }
      stamp(16, 2);
      DecAttnParams ap;
      memset(&ap, 0, sizeof(ap));
      ap.q = dq; ap.k = L.selfK + self_off; ap.v = L.selfV + self_off; ap.H = H; ap.kv_T = Tc; ap.pos = pos; ap.scale_log2e = scale_log2e;
      ap.out_bf16 = dattn; ap.row_seq = row_seq;
      TRY(ccx_launch_dec_attention(ctx, ap, B, 1, true, stream));
      stamp(17, 2);
      TRY(producer(L.Wo, L.bo, D, dattn));
      stamp(18, 2);
      stamp(1, 1);
      if (l == 0 && stagger) CCX_HIP(ctx, hipEventRecord(stagger, stream));
      {
        XsParams xp;
        memset(&xp, 0, sizeof(xp));
        xp.xb = xb; xp.ln_stats = st2; xp.ln_s = L.scq; xp.Wq = L.Wcq_g; xp.bq = L.ccq; xp.eps = 1e-5f;
        xp.WkT = L.WckT; xp.xq = pre ? w->pf_xq : w->xq + ro * H * D;
        xp.part_o = pre ? w->pf_xs_po : w->xs_po + ccx_xs_part_o_elems(ro, H, D);
        xp.part_ml = pre ? w->pf_xs_pml : w->xs_pml + ccx_xs_part_ml_elems(ro);
        xp.X = pre ? w->xa : w->xa + ro * (long)d.n_audio_ctx * D; xp.x_seq_stride = (long)d.n_audio_ctx * D; xp.row_seq = row_seq;
        xp.Wv = L.Wckv + (long)D * D; xp.bv = L.bckv + D; xp.out = dattn;
        xp.rows = B; xp.H = H; xp.S = d.n_audio_ctx; xp.D = D; xp.scale_log2e = scale_log2e;
        xp.lds_pad = (w->cross_lds_pad > 0 && !pre) ? 65536 : 0;
        xp.rows_per_seq = pre ? prefill_rows : 0;
        TRY(ccx_launch_xs_cross_attention(ctx, xp, stream));
      }
      stamp(2, 1);
      TRY(producer(L.Wco, L.bco, D, dattn));
      stamp(19, 2);
      TRY(consumer(DEPI_BF16_GELU, L.W1_g, L.s1, L.c1, F, dffn, F, nullptr));
      stamp(20, 2);
      if (w->lnfree_mode == 2 && F > 1024) TRY(partial_linear(ACT_BF16, L.W2, L.b2, F, dffn));      // split-K slabs, resolved by the next consumer
      else TRY(producer(L.W2, L.b2, F, dffn));
      stamp(21, 2);
    }
    TRY(ccx_launch_dec_resolve_ln(ctx, cur, pend, pend_n, pstride, w->lnd_g, w->lnd_b, dxn, nullptr, B, D, 1e-5f, stream));
    if (pre) return CCX_OK;
    TRY(dec_head(w, b0, B, logits, ld, select, sample_len, max_prompt, n_done, stream));
    stamp(3, 1);
    return CCX_OK;
  }
  for (int l = 0; l < d.n_text_layer; l++) {
    const DecLayer& L = w->dec[l];
    if (ablate == 2 && w->xs_active) {
      XsParams xp;
      memset(&xp, 0, sizeof(xp));
      xp.q = dq; xp.WkT = L.WckT; xp.xq = w->xq + ro * H * D;
      xp.part_o = w->xs_po + ccx_xs_part_o_elems(ro, H, D); xp.part_ml = w->xs_pml + ccx_xs_part_ml_elems(ro);
      xp.X = w->xa + ro * (long)d.n_audio_ctx * D; xp.x_seq_stride = (long)d.n_audio_ctx * D;
      xp.Wv = L.Wckv + (long)D * D; xp.bv = L.bckv + D; xp.out = dattn;
      xp.rows = B; xp.H = H; xp.S = d.n_audio_ctx; xp.D = D; xp.scale_log2e = scale_log2e;
      xp.lds_pad = w->cross_lds_pad > 0 ? 65536 : 0;
      TRY(ccx_launch_xs_cross_attention(ctx, xp, stream));
      continue;
    }
    if (ablate == 2 && B > 16 && !w->xs_active) {
      DecAttnParams ap;
      memset(&ap, 0, sizeof(ap));
      ap.q = dq; ap.k = L.crossK + cross_off; ap.v = L.crossV + cross_off; ap.H = H; ap.kv_T = w->Spad; ap.pos = nullptr; ap.T = d.n_audio_ctx;
      ap.scale_log2e = scale_log2e; ap.part_o = part_o; ap.part_ml = part_ml; ap.out_bf16 = dattn;
      ap.lds_pad = w->cross_lds_pad; ap.stream_mode = w->cross_stream ? 1 : 0;
      TRY(ccx_launch_dec_attention(ctx, ap, B, ns, ns == 1, stream));
      continue;
    }
    // LN + QKV, k/v appended to the self cache at pos[b]
    {
      DecLinearParams ex;
      memset(&ex, 0, sizeof(ex));
      ex.cache_k = L.selfK + self_off; ex.cache_v = L.selfV + self_off; ex.cache_T = Tc; ex.pos = pos; ex.row_seq = row_seq;
      TRY(ln_linear(DEPI_SELF_QKV, L.Wqkv, L.bqkv, 3 * D, L.ln1_g, L.ln1_b, dq, D, &ex));
      stamp(16, 2);
    }
    DecAttnParams ap;
    memset(&ap, 0, sizeof(ap));
    ap.q = dq; ap.k = L.selfK + self_off; ap.v = L.selfV + self_off; ap.H = H; ap.kv_T = Tc; ap.pos = pos; ap.scale_log2e = scale_log2e;
    ap.out_bf16 = dattn; ap.row_seq = row_seq;
    TRY(ccx_launch_dec_attention(ctx, ap, B, 1, true, stream));
    stamp(17, 2);
    if (lnfree3) {
      DecLinearParams lp;
      memset(&lp, 0, sizeof(lp));
      lp.M = B; lp.N = D; lp.K = D; lp.W = L.Wo; lp.ldw = D; lp.bias = L.bo; lp.act = dattn; lp.lda = D;
      lp.xres = cur; lp.xb = pre ? w->pf_xb : w->dxb + ro * D; lp.st_out = pre ? w->pf_st2 : w->dst2 + ro * (D / 16);
      TRY(ccx_launch_dec_linear(ctx, ACT_BF16, DEPI_RESOLVE, lp, stream));
    } else {
      TRY(partial_linear(ACT_BF16, L.Wo, L.bo, D, dattn));
    }
    stamp(18, 2);
    // cross attention.  Small batches (<= 16 rows, d_model 768: the reference's one-window-per-call pattern): the query projection
    // LN(x) Wcq^T runs INSIDE the cross-attention blocks (ccx_launch_dec_cross_fused_q: one launch fewer per layer on a chain that
    // is latency-bound launch by launch; q is bit-identical to the two-launch path).  CCX_FUSE_CROSS_Q=0 restores the two launches.
    const bool fuse_q = w->fuse_cross_q && !pre && B <= 16 && D == 768 && ablate == 0 && !w->xs_active;
    // X-stream path: the query projection (with its resolve + LayerNorm) runs inside the expansion kernel -- one launch for three
    // (CCX_XS_FUSE_Q=0: the three launches)
    const bool xs_fused = w->xs_active && w->xs_fuse_q;
    if (!fuse_q && !xs_fused && !lnfree3) TRY(ln_linear(DEPI_F32, L.Wcq, L.bcq, D, L.lnc_g, L.lnc_b, dq, D, nullptr));
    stamp(1, 1);
    if (l == 0 && stagger) CCX_HIP(ctx, hipEventRecord(stagger, stream));
    if (w->xs_active) {
      // decodes of more than 80 sequences: one pass over the encoder output serves all heads (cross_x.hip)
      if (ablate != 1) {
        XsParams xp;
        memset(&xp, 0, sizeof(xp));
        if (lnfree3) {
          xp.xb = pre ? w->pf_xb : w->dxb + ro * D; xp.ln_stats = pre ? w->pf_st2 : w->dst2 + ro * (D / 16);
          xp.ln_s = L.scq; xp.Wq = L.Wcq_g; xp.bq = L.ccq; xp.eps = 1e-5f;
        } else if (xs_fused) {
          xp.x = cur; xp.pend = pend; xp.pend_n = pend_n; xp.pend_stride = pstride; xp.x_out = pend_n > 0 ? other : nullptr;
          xp.ln_g = L.lnc_g; xp.ln_b = L.lnc_b; xp.eps = 1e-5f; xp.Wq = L.Wcq_plain; xp.bq = L.bcq;
        }
        xp.q = dq; xp.WkT = L.WckT; xp.xq = pre ? w->pf_xq : w->xq + ro * H * D;
        xp.part_o = pre ? w->pf_xs_po : w->xs_po + ccx_xs_part_o_elems(ro, H, D);
        xp.part_ml = pre ? w->pf_xs_pml : w->xs_pml + ccx_xs_part_ml_elems(ro);
        xp.X = pre ? w->xa : w->xa + ro * (long)d.n_audio_ctx * D; xp.x_seq_stride = (long)d.n_audio_ctx * D; xp.row_seq = row_seq;
        xp.Wv = L.Wckv + (long)D * D; xp.bv = L.bckv + D; xp.out = dattn;
        xp.rows = B; xp.H = H; xp.S = d.n_audio_ctx; xp.D = D; xp.scale_log2e = scale_log2e;
        xp.lds_pad = (w->cross_lds_pad > 0 && !pre) ? 65536 : 0;
        xp.rows_per_seq = pre ? prefill_rows : 0;
        TRY(ccx_launch_xs_cross_attention(ctx, xp, stream));
        if (xs_fused && pend_n > 0) { float* t = cur; cur = other; other = t; pend_n = 0; }
      } else if (xs_fused) {
        TRY(ln_linear(DEPI_F32, L.Wcq, L.bcq, D, L.lnc_g, L.lnc_b, dq, D, nullptr));     // chain-only ablation: keep the resolve
      }
      stamp(2, 1);
      if (lnfree4) {
        DecLinearParams lp;
        memset(&lp, 0, sizeof(lp));
        lp.M = B; lp.N = D; lp.K = D; lp.W = L.Wco; lp.ldw = D; lp.bias = L.bco; lp.act = dattn; lp.lda = D;
        lp.xres = cur; lp.xb = pre ? w->pf_xb : w->dxb + ro * D; lp.st_out = pre ? w->pf_st2 : w->dst2 + ro * (D / 16);
        TRY(ccx_launch_dec_linear(ctx, ACT_BF16, DEPI_RESOLVE, lp, stream));
      } else {
        TRY(partial_linear(ACT_BF16, L.Wco, L.bco, D, dattn));
      }
      stamp(19, 2);
    } else {
    memset(&ap, 0, sizeof(ap));
    ap.q = dq; ap.k = L.crossK + cross_off; ap.v = L.crossV + cross_off; ap.H = H; ap.kv_T = w->Spad; ap.pos = nullptr; ap.T = d.n_audio_ctx;
    ap.scale_log2e = scale_log2e; ap.part_o = part_o; ap.part_ml = part_ml; ap.out_bf16 = dattn;
    ap.lds_pad = w->cross_lds_pad;
    ap.stream_mode = (w->cross_stream && B > 16) ? 1 : 0;
    if (fuse_q) {
      ap.qx = cur; ap.q_pend = pend; ap.q_pend_n = pend_n; ap.q_pend_stride = pstride; ap.q_x_out = pend_n > 0 ? other : nullptr;
      ap.q_ln_g = L.lnc_g; ap.q_ln_b = L.lnc_b; ap.q_eps = 1e-5f; ap.q_W = L.Wcq; ap.q_bias = L.bcq; ap.q_K = D;
      TRY(ccx_launch_dec_cross_fused_q(ctx, ap, B, ns, stream));
      if (pend_n > 0) { float* t = cur; cur = other; other = t; pend_n = 0; }
      TRY(partial_linear(ACT_COMBINE, L.Wco, L.bco, D, nullptr));
    } else if (pre) {
      ap.row_seq = row_seq; ap.rows_per_seq = prefill_rows; ap.lds_pad = 0; ap.stream_mode = 1;
      if (prefill_rows > 1) {
        TRY(ccx_launch_dec_attention(ctx, ap, nseq, 1, true, stream));
      } else {
        ap.row_seq = nullptr; ap.rows_per_seq = 0;
        TRY(ccx_launch_dec_attention(ctx, ap, nseq, 1, true, stream));
      }
      TRY(partial_linear(ACT_BF16, L.Wco, L.bco, D, dattn));
    } else if (ablate == 1 && B > 16) {
      TRY(partial_linear(ACT_BF16, L.Wco, L.bco, D, dattn));
    } else if (B > 16) {
      TRY(ccx_launch_dec_attention(ctx, ap, B, ns, ns == 1, stream));
      stamp(2, 1);
      if (ns > 1) TRY(ccx_launch_dec_combine(ctx, part_o, part_ml, ns, dattn, B, H, stream));
      TRY(partial_linear(ACT_BF16, L.Wco, L.bco, D, dattn));
      stamp(19, 2);
    } else {
      TRY(ccx_launch_dec_attention(ctx, ap, B, ns, false, stream));
      TRY(partial_linear(ACT_COMBINE, L.Wco, L.bco, D, nullptr));
    }
    }   // !xs_active
    // MLP.  OPTIONAL (CCX_DEC_FC1_GEMM_ROWS=n, off by default): with n rows and more the first linear runs through the tiled GEMM,
    // which shares the weight and activation tiles of a block through LDS where the skinny kernel re-reads them per 32-column
    // block out of L2: pipeline step 653.1 -> 647.6 ms at n = 256.  It is off because the GEMM sums K in another order than the
    // skinny kernel: a sequence's log-probabilities would then depend (in the last bits) on how many rows its lane or its
    // prefill pass has, and tests/test_pinned_parity_gpu.py holds the records of a clip bit-identical whatever its batch mates.
    // (The self-attention q|k|v projection the same way -- GEMM into an fp32 scratch plus a scatter kernel for the cache rows --
    // was measured too: 650.8 against 650.5 ms, removed.)
    static const int fc1_gemm_rows = [] { const char* e = getenv("CCX_DEC_FC1_GEMM_ROWS"); return e ? atoi(e) : 0; }();
    if (fc1_gemm_rows > 0 && B >= fc1_gemm_rows && L.W1_plain && D % 64 == 0) {
      TRY(ccx_launch_dec_resolve_ln(ctx, cur, pend, pend_n, pstride, L.ln2_g, L.ln2_b, dxn, pend_n > 0 ? other : nullptr, B, D, 1e-5f, stream));
      if (pend_n > 0) { float* t = cur; cur = other; other = t; pend_n = 0; }
      GemmParams gp;
      memset(&gp, 0, sizeof(gp));
      gp.A = dxn; gp.lda = D; gp.W = L.W1_plain; gp.ldw = D; gp.M = B; gp.N = F; gp.K = D; gp.bias = L.b1; gp.out = dffn; gp.ldo = F;
      TRY(ccx_launch_gemm(ctx, EPI_BF16_GELU, gp, stream));
    } else if (lnfree4) {
      DecLinearParams lp;
      memset(&lp, 0, sizeof(lp));
      lp.M = B; lp.N = F; lp.K = D; lp.W = L.W1_g; lp.ldw = D; lp.bias = L.c1; lp.ln_s = L.s1; lp.eps = 1e-5f;
      lp.ln_stats = pre ? w->pf_st2 : w->dst2 + ro * (D / 16); lp.act = pre ? w->pf_xb : w->dxb + ro * D; lp.lda = D; lp.out = dffn; lp.ldo = F;
      TRY(ccx_launch_dec_linear(ctx, ACT_BF16_LN, DEPI_BF16_GELU, lp, stream));
    } else {
      TRY(ln_linear(DEPI_BF16_GELU, L.W1, L.b1, F, L.ln2_g, L.ln2_b, dffn, F, nullptr));
    }
    stamp(20, 2);
    TRY(partial_linear(ACT_BF16, L.W2, L.b2, F, dffn));
    stamp(21, 2);
  }
  // resolve the last partials + final LN, then logits against the tied embedding
  TRY(ccx_launch_dec_resolve_ln(ctx, cur, pend, pend_n, pstride, w->lnd_g, w->lnd_b, dxn, nullptr, B, D, 1e-5f, stream));
  if (pre) return CCX_OK;         // the caller gathers the last prompt row of every sequence out of pf_xn
  TRY(dec_head(w, b0, B, logits, ld, select, sample_len, max_prompt, n_done, stream));
  stamp(3, 1);       // end of the step
  return CCX_OK;
}

// `prefilled`: the prompts go through the prefill pass, so every sequence starts at its LAST prompt position (the state machine's
// sampling phase) and the first embedding comes from the prefill, not from here.
int upload_decode_state(ccx_whisper* w, const int32_t* prompt_ids, const int32_t* prompt_lens, int max_prompt, int B,
                        float temperature, uint64_t seed, hipStream_t stream, bool prefilled = false) {
  std::vector<DecSeqState> st(B);
  std::vector<int> tok(B), ps(B, 0);
  for (int b = 0; b < B; b++) {
    memset(&st[b], 0, sizeof(DecSeqState));
    st[b].prompt_len = prompt_lens[b];
    st[b].last_tok = -1; st[b].pen_tok = -1; st[b].last_ts_tok = -1;
    tok[b] = prompt_ids[(size_t)b * max_prompt];
    if (prefilled) {
      st[b].pos = prompt_lens[b] - 1;
      ps[b] = prompt_lens[b] - 1;
      tok[b] = prompt_ids[(size_t)b * max_prompt + prompt_lens[b] - 1];
    }
  }
  CCX_HIP(w->ctx, hipMemcpyAsync(w->state, st.data(), B * sizeof(DecSeqState), hipMemcpyHostToDevice, stream));
  CCX_HIP(w->ctx, hipMemcpyAsync(w->cur_tok, tok.data(), B * 4, hipMemcpyHostToDevice, stream));
  CCX_HIP(w->ctx, hipMemcpyAsync(w->pos, ps.data(), B * 4, hipMemcpyHostToDevice, stream));
  CCX_HIP(w->ctx, hipMemcpyAsync(w->prompt, prompt_ids, (size_t)B * max_prompt * 4, hipMemcpyHostToDevice, stream));
  CCX_HIP(w->ctx, hipMemsetAsync(w->n_done, 0, ccx_whisper::kMaxLanes * 4, stream));
  w->sampling = temperature > 0.f;
  unsigned cfg[4] = {0u, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), 0u};
  memcpy(&cfg[0], &temperature, 4);
  CCX_HIP(w->ctx, hipMemcpyAsync(w->sample_cfg, cfg, sizeof(cfg), hipMemcpyHostToDevice, stream));
  // embedding of the first token; later steps get theirs from the select kernel
  if (!prefilled) TRY(ccx_launch_dec_embed(w->ctx, w->tok_emb_f32, w->dec_pos, w->cur_tok, w->pos, w->dx, B, w->d.n_text_state, stream));
  CCX_HIP(w->ctx, hipStreamSynchronize(stream));  // host vectors go out of scope
  return CCX_OK;
}

// Prompt prefill (openai-whisper's first forward over all initial tokens, decoding.py::_main_loop): every prompt position of every
// sequence through the layer chain in passes of up to kPrefillMax positions -- rows = sequence * Pc + (position - first position of
// the pass); a pass sees the self-K/V of the earlier passes in the caches, so a long prompt (the reference feeds the previous
// segment's transcript as `initial_prompt`, up to 223 tokens: back/api.py:1424-1426) costs one pass per 16 tokens instead of one
// decode step per token.  Positions past a shorter prompt are dead rows (their K/V land beyond the prompt and are overwritten by the
// tokens decoded there later).  Leaves the self-KV caches filled and the final-LayerNorm row of every sequence's last prompt
// position in w->dxn.
int run_prefill(ccx_whisper* w, const int32_t* prompt_ids, const int32_t* prompt_lens, int max_prompt, int B, int P, int sample_len,
                hipStream_t stream) {
  ccx_ctx* ctx = w->ctx;
  const int D = w->d.n_text_state, C = ccx_whisper::kPrefillMax;
  for (int t0 = 0; t0 < P; t0 += C) {
    const int Pc = P - t0 < C ? P - t0 : C, R = B * Pc;
    std::vector<int> tok(R), ps(R), sq(R), last(B);
    for (int b = 0; b < B; b++) {
      for (int t = 0; t < Pc; t++) {
        const int r = b * Pc + t, ta = t0 + t;
        tok[r] = ta < prompt_lens[b] ? prompt_ids[(size_t)b * max_prompt + ta] : w->rules.eot;
        ps[r] = ta; sq[r] = b;
      }
      const int tl = prompt_lens[b] - 1 - t0;        // the sequence's last prompt position, relative to this pass
      last[b] = (tl >= 0 && tl < Pc) ? b * Pc + tl : -1;
    }
    CCX_HIP(ctx, hipMemcpyAsync(w->pf_tok, tok.data(), (size_t)R * 4, hipMemcpyHostToDevice, stream));
    CCX_HIP(ctx, hipMemcpyAsync(w->pf_pos, ps.data(), (size_t)R * 4, hipMemcpyHostToDevice, stream));
    CCX_HIP(ctx, hipMemcpyAsync(w->pf_seq, sq.data(), (size_t)R * 4, hipMemcpyHostToDevice, stream));
    CCX_HIP(ctx, hipMemcpyAsync(w->pf_last, last.data(), (size_t)B * 4, hipMemcpyHostToDevice, stream));
    TRY(ccx_launch_dec_embed(ctx, w->tok_emb_f32, w->dec_pos, w->pf_tok, w->pf_pos, w->pf_x, R, D, stream));
    CCX_HIP(ctx, hipStreamSynchronize(stream));     // host tables go out of scope
    TRY(dec_step(w, 0, B, nullptr, 0, false, sample_len, max_prompt, nullptr, stream, nullptr, 0, Pc));
    TRY(ccx_launch_dec_gather_rows(ctx, w->pf_xn, w->pf_last, w->dxn, B, D, stream));
  }
  return CCX_OK;
}

}  // namespace

extern "C" {

int ccx_whisper_decoder_logits(ccx_whisper* w, const int32_t* tokens, int B, int T, float* logits_dev, void* stream_) {
  if (!w) return CCX_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  ccx_ctx* ctx = w->ctx;
  CCX_REQUIRE(ctx, w->finalized && w->rules_set, "whisper: not finalized or rules not set");
  CCX_REQUIRE(ctx, tokens && logits_dev && B >= 1 && B <= w->max_batch && T >= 1 && T <= w->d.n_text_ctx, "decoder_logits: bad arguments");
  for (long i = 0; i < (long)B * T; i++)
    CCX_REQUIRE(ctx, tokens[i] >= 0 && tokens[i] < w->d.n_vocab, "decoder_logits: token id %d out of range", tokens[i]);
  std::vector<int32_t> lens(B, T + 1);  // never leaves the prompt phase: every step feeds tokens[b][pos]
  // prompt buffer rows are `T` wide here
  TRY(select_cross_path(w, B, stream));
  TRY(upload_decode_state(w, tokens, lens.data(), T, B, 0.f, 0, stream));
  const long V = w->d.n_vocab;
  w->cross_lds_pad = 0;   // single lane: the cross attention runs uncapped
  w->cross_stream = 0;
  for (int t = 0; t < T; t++) {
    // the select kernel (prompt phase) advances cur_tok/pos; on the last step it would read prompt[T] -> skip it.
    // The logits GEMM stores whole 16-column groups, so it writes the padded workspace rows (ld = Vpad) and the n_vocab valid
    // columns are copied out: writing [B, T, V] in place would spill V % 16 columns into the next row / past the tensor.
    TRY(dec_step(w, 0, B, w->dlogits, w->Vpad, t + 1 < T, 1, T, w->n_done, stream));
    CCX_HIP(ctx, hipMemcpy2DAsync(logits_dev + (long)t * V, (size_t)T * V * 4, w->dlogits, (size_t)w->Vpad * 4, (size_t)V * 4, B,
                                  hipMemcpyDeviceToDevice, stream));
  }
  return CCX_OK;
}

int ccx_whisper_last_cross_path(ccx_whisper* w) { return w ? w->last_cross_path : -1; }

int ccx_whisper_prepare_lanes(ccx_whisper* w, void* stream_) {
  if (!w) return CCX_ERR_ARG;
  CCX_REQUIRE(w->ctx, w->finalized, "whisper: not finalized");
  hipStream_t stream = stream_ ? (hipStream_t)stream_ : w->own_stream;
  CCX_HIP(w->ctx, hipDeviceSynchronize());
  (void)lane_streams_for(w, stream, ccx_whisper::kMaxLanes - 1);
  return CCX_OK;
}

int ccx_whisper_decode_greedy(ccx_whisper* w, const int32_t* prompt_ids, const int32_t* prompt_lens, int max_prompt, int B,
                              int sample_len, int32_t* tokens_out, int32_t* n_tokens_out, float* sum_logprob_out,
                              float* no_speech_prob_out, void* stream_) {
  return ccx_whisper_decode(w, prompt_ids, prompt_lens, max_prompt, B, sample_len, 0.f, 0, tokens_out, n_tokens_out, sum_logprob_out,
                            no_speech_prob_out, stream_);
}

int ccx_whisper_decode(ccx_whisper* w, const int32_t* prompt_ids, const int32_t* prompt_lens, int max_prompt, int B, int sample_len,
                       float temperature, uint64_t seed, int32_t* tokens_out, int32_t* n_tokens_out, float* sum_logprob_out,
                       float* no_speech_prob_out, void* stream_) {
  if (!w) return CCX_ERR_ARG;
  CCX_REQUIRE(w->ctx, temperature >= 0.f && temperature == temperature, "decode: temperature must be >= 0");
  hipStream_t stream = (hipStream_t)stream_;
  ccx_ctx* ctx = w->ctx;
  CCX_REQUIRE(ctx, w->finalized && w->rules_set, "whisper: not finalized or rules not set");
  CCX_REQUIRE(ctx, prompt_ids && prompt_lens && tokens_out && B >= 1 && B <= w->max_batch, "decode_greedy: bad arguments");
  if (stream == nullptr) {
    // order after everything already queued on the caller's (null) stream, then run on our own
    CCX_HIP(ctx, hipEventRecord(w->own_event, nullptr));
    CCX_HIP(ctx, hipStreamWaitEvent(w->own_stream, w->own_event, 0));
    stream = w->own_stream;
  }
  CCX_REQUIRE(ctx, sample_len >= 1 && sample_len <= w->sample_cap && max_prompt >= 1 && max_prompt <= w->max_prompt_cap, "decode_greedy: sample_len/max_prompt out of range");
  int max_pl = 0;
  for (int b = 0; b < B; b++) {
    CCX_REQUIRE(ctx, prompt_lens[b] >= 1 && prompt_lens[b] <= max_prompt, "decode_greedy: prompt_lens[%d]=%d out of range", b, prompt_lens[b]);
    CCX_REQUIRE(ctx, prompt_lens[b] + sample_len - 1 <= w->d.n_text_ctx, "decode_greedy: prompt %d + sample_len %d exceeds n_text_ctx", prompt_lens[b], sample_len);
    for (int i = 0; i < prompt_lens[b]; i++) {
      const int t = prompt_ids[(size_t)b * max_prompt + i];
      CCX_REQUIRE(ctx, t >= 0 && t < w->d.n_vocab, "decode_greedy: prompt token %d out of range", t);
    }
    if (prompt_lens[b] > max_pl) max_pl = prompt_lens[b];
  }
  // prompts of 2 tokens and more are prefilled, kPrefillMax positions per pass (CCX_PREFILL=0: one decode step per prompt token, round
  // 1's way; CCX_PREFILL_MAX=n: prompts longer than n tokens stepwise -- 16 was round 2's limit)
  const int prefill_on = [] { const char* e = getenv("CCX_PREFILL"); return e ? atoi(e) : 1; }();     // read per call: tests flip it
  const int prefill_max = [] { const char* e = getenv("CCX_PREFILL_MAX"); return e ? atoi(e) : 1 << 30; }();
  const bool prefill = prefill_on && max_pl >= 2 && max_pl <= prefill_max;
  TRY(select_cross_path(w, B, stream));
  TRY(upload_decode_state(w, prompt_ids, prompt_lens, max_prompt, B, temperature, seed, stream, prefill));
  // steps still to run after the (eager) first one: the prefill already covers the prompt AND takes the first sample below
  const int total_steps = prefill ? sample_len : max_pl - 1 + sample_len;
  const bool use_graph = getenv("CCX_NO_GRAPH") == nullptr;
  const long ld = w->Vpad;

  // ---- lanes: row ranges [b0, b0 + Bl) stepping concurrently (CCX_DEC_LANES overrides the default) ----
  int nl = 1;
  {
    const char* e = getenv("CCX_DEC_LANES");
    const int forced = e ? atoi(e) : 0;
    if (forced >= 1) nl = forced;
    // measured at 192 sequences x 65 steps: 1 lane 292.5, 2 286.8, 3 284.9, 4 284.8 ms; pipeline step with 384-sequence groups: 2 lanes
    // 710.0, 3 lanes 698.0 ms; with 768-sequence groups: 1 lane 721.5, 2 lanes 676.6, 3 lanes 687.5, 4 lanes 704.0 ms
    // with the cross attention against the encoder output (half the bytes: the chain weighs more): 768-sequence groups 2 lanes 561.3,
    // 3 lanes 557.9 ms per pipeline step
    // ... 384-sequence groups 3 lanes 598.6, 2 lanes 616.6 ms; 192-sequence groups (the sequential schedule) 1 lane 768.1, 2 lanes
    // 779.5, 3 lanes 793.2 ms: below ~128 rows per lane the streaming launches no longer fill the chip
    else if (w->xs_on) nl = B >= 320 ? 3 : 1;
    else nl = B >= 640 ? 2 : (B >= 144 ? 3 : (B >= 96 ? 2 : 1));
    if (nl > ccx_whisper::kMaxLanes) nl = ccx_whisper::kMaxLanes;
    while (nl > 1 && B / nl < 16) nl--;
  }
  const std::vector<hipStream_t>* extra = nullptr;
  if (nl > 1) {
    extra = &lane_streams_for(w, stream, nl - 1);
    if ((int)extra->size() + 1 < nl) nl = (int)extra->size() + 1;
  }
  // While lanes overlap, the cross attention of one lane (4,608 short blocks that fill every wave slot) makes the other
  // lanes' 5-8 us kernels queue for a slot.  Its blocks therefore claim 64 KB of LDS they do not use: two blocks per CU
  // still keep HBM saturated (each wave has 16 KB of loads in flight; 49.4 -> 51.3 us per launch) and the pipeline step
  // drops 924 -> 897 ms (3 blocks per CU: 910; 4: 919; 1: 979).  CCX_CROSS_LDS_PAD overrides.
  {
    static const int forced_pad = [] { const char* e = getenv("CCX_CROSS_LDS_PAD"); return e ? atoi(e) : -1; }();
    static const int lean = [] { const char* e = getenv("CCX_CROSS_STREAM"); return e ? atoi(e) : 1; }();
    w->cross_stream = lean;
    { const char* e = getenv("CCX_FUSE_CROSS_Q"); w->fuse_cross_q = e ? (atoi(e) != 0) : 1; }      // read per decode: tests flip it
    // default 3: the self-attention output projection resolves the residual in place and the cross-attention query applies its
    // LayerNorm algebraically (no twelve-fold resolve + LayerNorm inside the expansion kernel; -1.0 ... -1.2 % per decode step);
    // 0: round 3's chain; 1 / 2 / 4 / 5: the experiments of DESIGN.md section 2 (need the instance created with CCX_DEC_LNFREE set)
    { const char* e = getenv("CCX_DEC_LNFREE"); w->lnfree_mode = e ? atoi(e) : 3;
      if (w->lnfree_mode != 0 && w->lnfree_mode != 3 && !w->lnfree_built) w->lnfree_mode = 3;
      w->lnfree = w->lnfree_mode != 0; }
    // lean streaming: ONE 4-wave block per CU (98 KB of claimed LDS), each wave with 8-16 KB in flight.  The claim only exists to
    // leave room for the OTHER lanes' chain kernels: a single lane runs uncapped.
    w->cross_lds_pad = forced_pad >= 0 ? forced_pad : (nl > 1 ? (lean ? 98304 : 65536) : 0);
  }
  struct Lane { int b0, B; hipStream_t s; hipGraphExec_t exec; };
  Lane lanes[ccx_whisper::kMaxLanes];
  {
    const int per = ccx_cdiv(ccx_cdiv(B, nl), 16) * 16;   // lane sizes in multiples of one MFMA row tile
    int b0 = 0, n = 0;
    for (; n < nl && b0 < B; n++) {
      lanes[n].b0 = b0; lanes[n].B = (b0 + per <= B) ? per : B - b0;
      lanes[n].s = n == 0 ? stream : (*extra)[n - 1];
      lanes[n].exec = nullptr;
      b0 += lanes[n].B;
    }
    nl = n;
  }
  // which cross attention the steps of this decode run (lane 0; a short last lane of <= 16 rows takes the <= 16-row kernels of its path)
  w->last_cross_path = w->xs_active ? 2 : ((w->cross_stream && lanes[0].B > 16) ? 1 : 0);
  if (prefill) {
    TRY(run_prefill(w, prompt_ids, prompt_lens, max_prompt, B, max_pl, sample_len, stream));
    // first sample of every sequence, lane by lane (each lane counts its own finished sequences)
    for (int i = 0; i < nl; i++)
      TRY(dec_head(w, lanes[i].b0, lanes[i].B, w->dlogits + (long)lanes[i].b0 * ld, ld, true, sample_len, max_prompt, w->n_done + i, stream));
  }
  // the state upload (and the prefill) was queued on `stream`: the other lanes start after it
  CCX_HIP(ctx, hipEventRecord(w->own_event, stream));
  for (int i = 1; i < nl; i++) CCX_HIP(ctx, hipStreamWaitEvent(lanes[i].s, w->own_event, 0));
  auto step_lane = [&](int i, hipEvent_t stagger) -> int {
    const Lane& L = lanes[i];
    return dec_step(w, L.b0, L.B, w->dlogits + (long)L.b0 * ld, ld, true, sample_len, max_prompt, w->n_done + i, L.s, stagger, i);
  };
  // first step runs eagerly (also performs one-time kernel attribute setup outside of capture).  Lane i + 1
  // starts when lane i reaches its first cross attention, which sets the stagger the later steps keep.
  int step = prefill ? 1 : 0;                   // the prefill's own sample counts as step 0
  if (step < total_steps) {
    for (int i = 0; i < nl; i++) {
      static const bool stagger_on = [] { const char* e = getenv("CCX_LANE_STAGGER"); return !e || atoi(e) != 0; }();
      if (i > 0 && stagger_on) CCX_HIP(ctx, hipStreamWaitEvent(lanes[i].s, w->lane_start[i - 1], 0));
      TRY(step_lane(i, (i + 1 < nl) ? w->lane_start[i] : nullptr));
      if (ctx->prof_on && !use_graph && nl > 1) CCX_HIP(ctx, hipStreamSynchronize(lanes[i].s));
    }
    step += 1;
  }
  if (use_graph && step < total_steps) {
    for (int i = 0; i < nl; i++) {
      // graphs are specific to (lane rows, sample_len, max_prompt)
      // ... and to everything else dec_step bakes into kernel parameters: the cross-attention LDS cap and split count
      const int ns_key = cross_split(lanes[i].B, w->d.n_text_head, w->cross_lds_pad > 0, w->cross_stream != 0);
      const std::array<int, 9> key = {lanes[i].b0, lanes[i].B, sample_len, max_prompt, w->sampling ? 1 : 0, ns_key, w->cross_lds_pad, w->cross_stream | (w->fuse_cross_q << 4) | ((w->xs_active ? 1 : 0) << 8) | ((w->xs_fuse_q ? 1 : 0) << 9) | (w->lnfree_mode << 16), i};
      auto it = w->graphs.find(key);
      if (it != w->graphs.end()) { lanes[i].exec = it->second; continue; }
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      CCX_HIP(ctx, hipStreamBeginCapture(lanes[i].s, hipStreamCaptureModeThreadLocal));
      int rc = step_lane(i, nullptr);
      hipError_t e = hipStreamEndCapture(lanes[i].s, &graph);
      if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
      if (e != hipSuccess) return ccx_fail(ctx, CCX_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
      e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      hipGraphDestroy(graph);
      if (e != hipSuccess) return ccx_fail(ctx, CCX_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
      w->graphs[key] = exec;
      lanes[i].exec = exec;
    }
  }
  // Steps are queued in chunks; the done counters of chunk c are polled only after chunk c + 1 is queued,
  // so no stream runs dry while the host waits.
  const int kChunk = 8;
  auto queue_chunk = [&](int c, int n) -> int {
    for (int k = 0; k < n; k++)
      for (int i = 0; i < nl; i++) {
        if (lanes[i].exec) CCX_HIP(ctx, hipGraphLaunch(lanes[i].exec, lanes[i].s));   // ~50 us of host time per replay
        else {
          TRY(step_lane(i, nullptr));
          // eager profiling runs (ccx_prof_enable + CCX_NO_GRAPH) time every kernel with an event pair: keep the lanes apart so
          // that the durations are those of the kernel alone, as rocprofv3 (which serialises replays) sees them
          if (ctx->prof_on && !use_graph && nl > 1) CCX_HIP(ctx, hipStreamSynchronize(lanes[i].s));
        }
      }
    for (int i = 0; i < nl; i++) {
      CCX_HIP(ctx, hipMemcpyAsync(&w->poll_host[(c & 1) * ccx_whisper::kMaxLanes + i], w->n_done + i, 4, hipMemcpyDeviceToHost, lanes[i].s));
      CCX_HIP(ctx, hipEventRecord(w->lane_poll[c & 1][i], lanes[i].s));
    }
    return CCX_OK;
  };
  auto chunk_done = [&](int c, bool* all) -> int {
    *all = true;
    for (int i = 0; i < nl; i++) {
      CCX_HIP(ctx, hipEventSynchronize(w->lane_poll[c & 1][i]));
      if (w->poll_host[(c & 1) * ccx_whisper::kMaxLanes + i] < lanes[i].B) *all = false;
    }
    return CCX_OK;
  };
  int c = 0;
  bool pending = false;   // chunk c - 1 queued but not polled yet
  while (step < total_steps) {
    const int n = (total_steps - step < kChunk) ? total_steps - step : kChunk;
    TRY(queue_chunk(c, n));
    step += n;
    if (pending) {
      bool all;
      TRY(chunk_done(c - 1, &all));
      if (all) break;
    }
    pending = true;
    c++;
  }
  // join the lanes back into `stream`
  for (int i = 1; i < nl; i++) {
    CCX_HIP(ctx, hipEventRecord(w->lane_start[i], lanes[i].s));
    CCX_HIP(ctx, hipStreamWaitEvent(stream, w->lane_start[i], 0));
  }
  if (w->stamps_on) {
    // append this decode's trace: one line per lane "lane <i> <n> <stamp> ..." (stamp = realtime << 8 | tag), then reset
    CCX_HIP(ctx, hipStreamSynchronize(stream));
    std::vector<int> cnt(ccx_whisper::kMaxLanes);
    CCX_HIP(ctx, hipMemcpy(cnt.data(), w->stamp_count, cnt.size() * 4, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(w->stamp_path.c_str(), "a")) {
      fprintf(f, "decode B %d lanes %d\n", B, nl);
      for (int i = 0; i < nl; i++) {
        const int n = cnt[i] < ccx_whisper::kStampCap ? cnt[i] : ccx_whisper::kStampCap;
        std::vector<unsigned long long> buf(n);
        CCX_HIP(ctx, hipMemcpy(buf.data(), w->stamps + (size_t)i * ccx_whisper::kStampCap, (size_t)n * 8, hipMemcpyDeviceToHost));
        fprintf(f, "lane %d %d", i, n);
        for (int k = 0; k < n; k++) fprintf(f, " %llu", buf[k]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
    CCX_HIP(ctx, hipMemset(w->stamp_count, 0, ccx_whisper::kMaxLanes * 4));
  }
  std::vector<DecSeqState> st(B);
  std::vector<int> gen((size_t)B * sample_len);
  CCX_HIP(ctx, hipMemcpyAsync(st.data(), w->state, B * sizeof(DecSeqState), hipMemcpyDeviceToHost, stream));
  CCX_HIP(ctx, hipMemcpyAsync(gen.data(), w->gen, gen.size() * 4, hipMemcpyDeviceToHost, stream));
  CCX_HIP(ctx, hipStreamSynchronize(stream));
  for (int b = 0; b < B; b++) {
    const int ng = st[b].n_gen < sample_len ? st[b].n_gen : sample_len;
    // not done after the step budget: treat everything sampled as text (no eot was produced)
    const int ntok = st[b].done ? st[b].n_tokens : ng;
    for (int i = 0; i < sample_len; i++) tokens_out[(size_t)b * sample_len + i] = (i < ntok) ? gen[(size_t)b * sample_len + i] : w->rules.eot;
    if (n_tokens_out) n_tokens_out[b] = ntok;
    if (sum_logprob_out) sum_logprob_out[b] = st[b].sum_logprob;
    if (no_speech_prob_out) no_speech_prob_out[b] = st[b].no_speech_prob;
  }
  return CCX_OK;
}

}  // extern "C"
