/* ccx.h -- C ABI of libccx, the MI355X-native (gfx950) compute library behind
 * ClearConverse's overlapped-speech transcription path.
 *
 * The reference has NO FFI for this path: EnhancedAudioProcessor calls duck-typed Python model
 * objects (reference back/api.py:657-797 creates them, back/api.py:1298-1549 drives them).  Each
 * entry point below states which of those Python calls it replaces; clearconverse_amd/_lib.py is
 * the ctypes binding and INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: every function returns 0 (CCX_OK) or a non-zero status and stores a message
 * retrievable with ccx_last_error(ctx).  All `*_dev` pointers are HIP device pointers owned by the
 * caller (torch allocates them); `stream` is a hipStream_t passed as void*.  The library owns only
 * weights, caches, workspaces and graphs inside its handles; it has no global state and starts no
 * threads, so it is safe to initialise after fork (reference back/api.py:2045-2049 forks per task).
 */
#ifndef CCX_H
#define CCX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CCX_DTYPE_F32 0
#define CCX_DTYPE_BF16 1
#define CCX_DTYPE_F16 2

typedef struct ccx_ctx ccx_ctx;
typedef struct ccx_whisper ccx_whisper;

const char* ccx_version(void);
int ccx_ctx_create(int device, ccx_ctx** out);
void ccx_ctx_destroy(ccx_ctx* ctx);
/* Last error text of this context ("" if none). ctx == NULL returns the text of the last failed
 * ccx_ctx_create on this thread. */
const char* ccx_last_error(const ccx_ctx* ctx);

/* ---- per-launch timing (HIP events on the launch stream) used by bench.py's roofline ----------
 * While enabled, every eagerly launched kernel of the library records a start/stop event pair and
 * its algorithmic flops/bytes.  Launches inside a stream capture (graph replay) are not recorded. */
int ccx_prof_enable(ccx_ctx* ctx, int on);  /* also clears previous records */
int ccx_prof_count(ccx_ctx* ctx);
int ccx_prof_get(ccx_ctx* ctx, int i, char* name_out, int name_cap, double* flops, double* bytes, float* ms);

/* ---- primitive operators (exposed so tests/ can check each kernel against oracle/) ------------ */

/* C[M,N] = A[M,K] * W[N,K]^T (+bias) with a fused epilogue; bf16 inputs, fp32 accumulate (MFMA).
 * epi: 0 bf16 out | 1 bf16 gelu | 2 f32 out = acc+bias+resid | 3 f32 out | 5 bf16 relu.
 * Replaces the torch.nn.Linear / Conv1d-as-GEMM calls inside whisper / speechbrain / pyannote
 * modules that the reference triggers at back/api.py:1077, 869, 1286. */
int ccx_gemm_bf16(ccx_ctx* ctx, int epi, const void* A_dev, int64_t lda, const void* W_dev, int64_t ldw,
                  const float* bias_dev, void* out_dev, int64_t ldo, const float* resid_dev, int64_t ldr,
                  int M, int N, int K, void* stream);

/* LayerNorm over the last dim with fp32 statistics; writes bf16 and/or fp32 (either may be NULL). */
int ccx_layernorm(ccx_ctx* ctx, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                  void* out_bf16_dev, float* out_f32_dev, int M, int D, float eps, void* stream);

/* Peak normalisation y = x / (max|x| + eps) per row (eps == 0: only when the peak is > 0) -- replaces
 * `signal_np / (np.max(np.abs(signal_np)) + 1e-8)` (reference back/api.py:834) and lines 350-351.
 * x,y [B, stride] f32 (may alias), n_samples_dev [B] int32 on the device. */
/* Ragged crops into a padded batch in one launch (the reference slices one crop at a time, `_extract_segment`,
 * back/api.py:840-860): dst[i][0 .. lens[i]) = row i, whose device address is src_ptrs_dev[i]; both tables in device memory. */
int ccx_gather_rows(ccx_ctx* ctx, const int64_t* src_ptrs_dev, const int* lens_dev, int n_rows, int max_len, float* dst_dev,
                    int64_t stride, void* stream);
int ccx_peak_normalize(ccx_ctx* ctx, const float* x_dev, float* y_dev, int64_t stride, const int* n_samples_dev, int B,
                       float eps, void* stream);
/* One decoder layer's cross attention as a stand-alone operator: out[r] = softmax(q[r] K^T / 8) V per head with K = xa Wk^T,
 * V = xa Wv^T + bv -- openai-whisper MultiHeadAttention.qkv_attention with xa (called per layer and decode step from
 * back/api.py:1286-1292 through transcribe()) -- computed the way the decode path of ccx_whisper_decode computes it for more than 80
 * sequences: against the encoder output itself (csrc/cross_x.hip), no K/V ever materialised.  q_dev [rows][64 n_head] f32 (bias
 * included), wk / wv [64 n_head][64 n_head] f32 and bv on the HOST, xa_dev bf16 [n_seq][n_ctx][64 n_head], row_seq (host, may be
 * null = identity) maps rows to sequences, out_dev [rows][64 n_head] f32.  rows_per_seq > 1: consecutive groups of that many rows
 * belong to one sequence each (the prompt prefill: four rows of a group then share one pass over the sequence's xa), else 0.
 * Synchronises the stream.  For kernel parity tests. */
int ccx_cross_attention_xa(ccx_ctx* ctx, const float* q_dev, const float* wk_host, const float* wv_host, const float* bv_host,
                           const uint16_t* xa_dev, const int* row_seq_host, int rows_per_seq, int rows, int n_seq, int n_head, int n_ctx,
                           float* out_dev, void* stream);
/* Embedding-quality weight of `_build_speaker_profiles`: out[b] = torch.var(x[b][0 .. n_b)) (unbiased; reference back/api.py:939).
 * fp64 accumulation in a fixed order: a row's value does not depend on its batch mates. */
int ccx_row_variance(ccx_ctx* ctx, const float* x_dev, int64_t stride, const int* n_samples_dev, int B, float* out_dev, void* stream);
/* `_calculate_embedding_similarity` (reference back/api.py:878-879: torch cosine_similarity(dim=0).item()), row-wise for a batch:
 * out[r] = sum_i (a[r][i] / max(|a[r]|, 1e-8)) * (b[r % b_rows][i] / max(|b[..]|, 1e-8)).  a [R, D], b [b_rows, D] f32. */
int ccx_cosine_rows(ccx_ctx* ctx, const float* a_dev, const float* b_dev, int R, int D, int b_rows, float* out_dev, void* stream);
/* The weighted sum of `_build_speaker_profiles` (reference back/api.py:946-953): out[c][s] = sum over turns t with spk[t] == s of
 * emb[c][t] * w[c][t] / (sum of those w[c][t]); NOT re-normalised.  emb [C, T, D], w [C, T], spk_dev [T] int32, out [C, S, D]. */
int ccx_speaker_profiles(ccx_ctx* ctx, const float* emb_dev, const float* w_dev, const int* spk_dev, int C, int T, int D, int S,
                         float* out_dev, void* stream);

/* K1: `torchaudio.transforms.Resample(orig_freq=sample_rate, new_freq=16000)(signal)` (reference back/api.py:824-830) with
 * torchaudio's default arguments.  orig / new_ are the two rates divided by their gcd, width = ceil(6 * orig / (min(orig, new_) * 0.99)),
 * kernT_dev the polyphase table TRANSPOSED, [2 * width + orig][new_] f32 (built by the caller exactly as upstream builds it:
 * clearconverse_amd/audio.py::sinc_resample_kernel).  x [B, stride_in] f32 rows of n_in_dev[b] samples -> y [B, stride_out] rows
 * of n_out_dev[b] = ceil(new_ * n_in / orig) samples; max_out = the largest of them.  All pointers are device pointers. */
int ccx_resample_sinc(ccx_ctx* ctx, const float* x_dev, int64_t stride_in, const int* n_in_dev, int B, int orig, int new_,
                      int width, const float* kernT_dev, float* y_dev, int64_t stride_out, const int* n_out_dev, int max_out,
                      void* stream);

/* Non-causal attention, head_dim 64 (Whisper encoder).  q,k: [B*H, Spad, 64] bf16 with rows >= S
 * zero; vt: [B*H, 64, Spad] bf16; o: [B*S, H*64] bf16.  Softmax scale 1/8 (= 64^-0.25 on q and k). */
int ccx_enc_attention(ccx_ctx* ctx, const void* q_dev, const void* k_dev, const void* vt_dev, void* o_dev,
                      int B, int H, int S, int Spad, void* stream);

/* ---- Whisper (replaces self.whisper_model, reference back/api.py:665-703; calls at
 *      back/api.py:1286-1292, 1432-1438, 1474-1480) ---------------------------------------------- */

typedef struct {
  int n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} ccx_whisper_dims;

/* Token-id constants of the tokenizer the checkpoint was trained with (openai-whisper
 * tokenizer.py; for *.en models sot=50257, eot=50256, ... see clearconverse_amd/tokenizer.py). */
typedef struct {
  int eot, sot, sot_prev, no_speech, no_timestamps, timestamp_begin, blank; /* blank = id of " " */
  int max_initial_timestamp_index;  /* 50 = 1.0 s; < 0 disables the rule */
  int n_suppress;
  const int* suppress;              /* host array, copied */
} ccx_decode_rules;

int ccx_whisper_create(ccx_ctx* ctx, const ccx_whisper_dims* dims, int max_batch, ccx_whisper** out);
void ccx_whisper_destroy(ccx_whisper* w);
/* Register one tensor by its openai-whisper state_dict name (the key layout of the reference's
 * fine-tune overlay, back/api.py:671-692), plus "mel_filters" [n_mels, 201].  data may be a host
 * or a device pointer.  Unknown names are an error. */
int ccx_whisper_set_tensor(ccx_whisper* w, const char* name, const void* data, int dtype, int ndim,
                           const int64_t* shape);
/* Longest clip (seconds) ccx_whisper_logmel must accept (default 30); call before finalize.  The log-mel of the
 * whole clip is normalised with its global maximum, exactly like whisper.audio.log_mel_spectrogram. */
int ccx_whisper_set_max_audio(ccx_whisper* w, double seconds);
/* Optional, before finalize: take the log-mel / encoder workspaces of `donor` (finalized, same dimensions, at least this
 * instance's capacity) instead of allocating them.  They are only live between ccx_whisper_logmel / set_mel and the end of
 * ccx_whisper_encode.  Users of one group are ordered by the library: every logmel / set_mel waits (event) for the end of the
 * group's previous encode, on whatever streams they run -- so an instance may encode while ANOTHER instance of the group decodes
 * (the software-pipelined batch driver), but the host must issue logmel / set_mel .. encode of one instance as an UNINTERRUPTED pair:
 * the event orders a logmel behind the group's previous encode only, and between an instance's logmel and its encode the workspaces
 * hold its staged windows.  Enforced: a logmel / set_mel of another instance of the group in between fails with CCX_ERR_ARG.
 * A donor destroyed while takers are alive is freed when its last taker is destroyed. */
int ccx_whisper_share_encoder_scratch(ccx_whisper* w, ccx_whisper* donor);
/* Checks every tensor is present, builds the fused/bf16 device layouts, uploads. */
int ccx_whisper_finalize(ccx_whisper* w);
int ccx_whisper_set_rules(ccx_whisper* w, const ccx_decode_rules* rules);

/* Log-mel of B clips (whisper.audio.log_mel_spectrogram + pad_or_trim to 3000 frames).
 * audio_dev: [B, stride] f32; n_samples / seek_frames: host int arrays (seek may be NULL = 0).
 * Fills the model's conv-stem input; if mel_out_dev != NULL also writes [B, n_mels, 3000] f32. */
int ccx_whisper_logmel(ccx_whisper* w, const float* audio_dev, int64_t stride, const int* n_samples,
                       const int* seek_frames, int B, float* mel_out_dev, void* stream);
/* Alternative input: take a ready [B, n_mels, 3000] f32 mel (BASELINE config 2 "mel [8,80,3000]"). */
int ccx_whisper_set_mel(ccx_whisper* w, const float* mel_dev, int B, void* stream);
/* AudioEncoder.forward for the B staged windows.  The encoder output (bf16) stays with the instance: decodes of more than 80 sequences
 * read it directly in their cross attention (csrc/cross_x.hip), decodes of <= 80 sequences project the per-layer cross-attention K / V
 * of their sequences out of it when they start (with CCX_CROSS_X=0 at ccx_whisper_finalize this call projects K / V for all B, as in
 * openai-whisper's kv_cache hooks).
 * xa_out_dev (optional): [B, n_audio_ctx, n_audio_state] f32 copy of the encoder output. */
int ccx_whisper_encode(ccx_whisper* w, int B, float* xa_out_dev, void* stream);

/* Teacher-forced decoder pass for parity tests: tokens [B, T] (host int32) -> logits
 * [B, T, n_vocab] f32 on device.  Uses the same step kernels as greedy decoding. */
int ccx_whisper_decoder_logits(ccx_whisper* w, const int32_t* tokens, int B, int T, float* logits_dev,
                               void* stream);

/* Greedy (temperature 0) DecodingTask.run for the B encoded windows.
 * prompt_ids: host [B, max_prompt] initial tokens (sot_prev + prompt + sot), prompt_lens: [B].
 * Outputs (host): tokens [B, sample_len] sampled ids (eot-padded), n_tokens [B] count before eot,
 * sum_logprob [B], no_speech_prob [B]. */
int ccx_whisper_decode_greedy(ccx_whisper* w, const int32_t* prompt_ids, const int32_t* prompt_lens,
                              int max_prompt, int B, int sample_len, int32_t* tokens_out,
                              int32_t* n_tokens_out, float* sum_logprob_out, float* no_speech_prob_out,
                              void* stream);

/* DecodingTask.run with GreedyDecoder.update's temperature branch: temperature 0 = argmax (identical to
 * ccx_whisper_decode_greedy); temperature > 0 = one sample per step from Categorical(logits / temperature) over the
 * filtered logits -- what the reference gets from transcribe(..., temperature=self.config.temperature) with
 * Config.temperature = 0.1 (back/api.py:128, 1286-1292).  The draw is Gumbel-max over Philox4x32-10 noise keyed by
 * (seed, sequence row, step, vocabulary id): reproducible, independent of batch composition, but not bit-equal to
 * torch's sampler (distributional parity).  sum_logprob uses the unscaled log-softmax, as upstream. */
int ccx_whisper_decode(ccx_whisper* w, const int32_t* prompt_ids, const int32_t* prompt_lens, int max_prompt, int B,
                       int sample_len, float temperature, uint64_t seed, int32_t* tokens_out, int32_t* n_tokens_out,
                       float* sum_logprob_out, float* no_speech_prob_out, void* stream);

/* Which cross-attention formulation the last ccx_whisper_decode of this instance ran (measurement / test records; the reference has one
 * formulation, MultiHeadAttention.forward(x, xa) behind back/api.py:1286-1292): 0 = "kv16" (per-layer K / V caches, split-KV kernels,
 * <= 16 sequences), 1 = "kv_stream" (per-layer K / V caches, dec_cross_stream_kernel, 17 - 80 sequences), 2 = "xa_stream" (one pass over
 * the encoder output per layer, csrc/cross_x.hip, more than 80 sequences); -1 before the first decode. */
int ccx_whisper_last_cross_path(ccx_whisper* w);

/* Batch-driver helper with no counterpart in the reference (it decodes one window at a time, back/api.py:1286): picks, once, the
 * internal streams on which the lanes of a large decode batch will run beside `stream` (HIP maps streams onto a few hardware
 * queues; the choice is made by a short timing probe, which must not be disturbed by other work on the GPU).  Call it on an
 * otherwise idle device before decodes are overlapped with other streams; ccx_whisper_decode does it lazily otherwise. */
int ccx_whisper_prepare_lanes(ccx_whisper* w, void* stream);
/* Measurement helper (no counterpart in the reference): while `path` is non-NULL every hipGraph-captured decode step carries
 * one-thread stamp kernels (100 MHz s_memrealtime) around the cross attention of every layer (level 1) or after every kernel of
 * the chain (level 2), per decode lane; each ccx_whisper_decode appends its trace to `path` ("decode B <n> lanes <l>" + one line
 * per lane; tools/decode_stamps.py, bench.py `roofline.frac_in_situ`).  NULL switches it off.  Either call drops the captured
 * step graphs.  Equivalent to creating the model with CCX_DEC_STAMPS=<path> in the environment. */
int ccx_whisper_trace_lanes(ccx_whisper* w, const char* path, int level);

/* ---- RE-SepFormer separator (replaces self.separator, reference back/api.py:713-717; call at
 *      back/api.py:1077 `separated = self.separator.separate_batch(subsegment)`) -------------------- */

typedef struct ccx_sepformer ccx_sepformer;
typedef struct {
  int n_filters, kernel, stride, d_model, n_head, d_ffn, n_layers, n_blocks, segment, n_spk;
} ccx_sepformer_dims;

/* max_tokens: capacity in encoder frames summed over the utterances of one call (each padded to
 * whole `segment`-frame chunks); max_utts: utterances per call. */
int ccx_sepformer_create(ccx_ctx* ctx, const ccx_sepformer_dims* dims, int max_tokens, int max_utts,
                         ccx_sepformer** out);
void ccx_sepformer_destroy(ccx_sepformer* s);
/* Tensors by SpeechBrain checkpoint key, prefixed with the module name of the reference's overlay
 * files (back/api.py:729): "encoder.conv1d.weight", "decoder.weight", "masknet.model....". f32. */
int ccx_sepformer_set_tensor(ccx_sepformer* s, const char* name, const float* data, int64_t numel);
int ccx_sepformer_finalize(ccx_sepformer* s);
/* separate_batch for B independent utterances: mix_dev [B, stride] f32, n_samples host [B] ->
 * out_dev [B, stride, 2] f32 (rows past n_samples[b] are zero). */
int ccx_sepformer_separate(ccx_sepformer* s, const float* mix_dev, int64_t stride, const int* n_samples, int B,
                           float* out_dev, void* stream);

/* ---- pyannote-style speaker networks (replace self.embedding_model = Inference("pyannote/embedding",
 *      window="whole"), reference back/api.py:776-780, called at back/api.py:869; and the segmentation
 *      network inside self.vad_pipeline / self.diarization, reference back/api.py:782-792, called at
 *      back/api.py:1311, 1056, 1124) ------------------------------------------------------------------ */

typedef struct ccx_speaker ccx_speaker;
/* kind 0: XVectorSincNet embedder (512-d).  kind 1: PyanNet segmentation (n_classes per frame; powerset
 * != 0 -> log-softmax, else sigmoid).  max_samples: total samples of all crops of one call. */
int ccx_speaker_create(ccx_ctx* ctx, int kind, int n_classes, int powerset, int max_crops, int64_t max_samples,
                       ccx_speaker** out);
void ccx_speaker_destroy(ccx_speaker* s);
/* Tensors by pyannote checkpoint key (f32): "sincnet.wav_norm1d.weight", "sincnet.conv1d.0.filters"
 * ([80,251] band-pass filters expanded from low_hz_/band_hz_ by the host), "sincnet.conv1d.1.weight", ...,
 * "tdnns.N.0.weight", "tdnns.N.2.running_mean", "embedding.weight" | "lstm.weight_ih_l0_reverse",
 * "linear.0.weight", "classifier.weight". */
int ccx_speaker_set_tensor(ccx_speaker* s, const char* name, const float* data, int64_t numel);
int ccx_speaker_finalize(ccx_speaker* s);
/* n crops stored in wav_dev at sample offsets[i], n_samples[i] long (host arrays) -> out_dev [n, 512] f32.
 * weights_dev (optional, NULL = plain mean/std pooling): per-crop frame weights at any resolution
 * (w_lens[i] values at w_offsets[i]), nearest-interpolated to the pooling frames (pyannote StatsPool). */
int ccx_speaker_embed(ccx_speaker* s, const float* wav_dev, const int64_t* offsets, const int* n_samples, int n,
                      const float* weights_dev, const int64_t* w_offsets, const int* w_lens, float* out_dev, void* stream);
/* per-frame class scores of n crops, concatenated in crop order: out_dev [sum frames, n_classes] f32;
 * frames_out[i] (host) = frames of crop i. */
int ccx_speaker_segment(ccx_speaker* s, const float* wav_dev, const int64_t* offsets, const int* n_samples, int n,
                        float* out_dev, int64_t out_capacity_rows, int* frames_out, void* stream);

/* ---- WeSpeaker ResNet-34 speaker embedder: the embedding model inside self.diarization =
 *      Pipeline.from_pretrained("pyannote/speaker-diarization-3.1"), reference back/api.py:788-792, called at
 *      back/api.py:1056-1060 and 1124-1128 (one embedding per 10 s chunk and local speaker) ---------------- */
typedef struct ccx_resnet ccx_resnet;
/* max_samples: longest chunk of one call; max_masks: most (chunk, speaker) masks of one call */
int ccx_resnet_create(ccx_ctx* ctx, int max_chunks, int64_t max_samples, int max_masks, ccx_resnet** out);
void ccx_resnet_destroy(ccx_resnet* r);
/* Tensors by checkpoint key (f32): "resnet.conv1.weight", "resnet.bn1.running_mean", "resnet.layer2.0.conv1.weight",
 * "resnet.layer2.0.shortcut.0.weight", "resnet.layer2.0.shortcut.1.weight", ..., "resnet.seg_1.weight". */
int ccx_resnet_set_tensor(ccx_resnet* r, const char* name, const float* data, int64_t numel);
int ccx_resnet_finalize(ccx_resnet* r);
/* wav_dev [n_chunks, stride] f32, every chunk n_samples long.  weights_dev NULL: one embedding per chunk, out_dev
 * [n_chunks, 256] f32.  Else weights_dev [n_masks, n_w] f32 frame weights (any resolution, nearest-interpolated to
 * the pooling frames) and mask_chunk [n_masks] (host: the chunk each mask pools over): out_dev [n_masks, 256].
 * The convolutional trunk runs once per chunk. */
int ccx_resnet_embed(ccx_resnet* r, const float* wav_dev, int64_t stride, int n_samples, int n_chunks, const float* weights_dev,
                     int n_w, const int* mask_chunk, int n_masks, float* out_dev, void* stream);

/* ---- stationary spectral-gate denoiser (replaces nr.reduce_noise(y=, sr=, stationary=True,
 *      prop_decrease=), reference back/api.py:349 and 832-833) ---------------------------------------- */
typedef struct ccx_specgate ccx_specgate;
int ccx_specgate_create(ccx_ctx* ctx, int64_t max_samples, int max_clips, int sample_rate, ccx_specgate** out);
void ccx_specgate_destroy(ccx_specgate* g);
/* y_dev [B, stride] f32, n_samples host [B] -> out_dev [B, stride] f32 (samples past n_samples[b] are zero) */
int ccx_specgate_reduce(ccx_specgate* g, const float* y_dev, int64_t stride, const int* n_samples, int B,
                        float prop_decrease, float* out_dev, void* stream);
/* noisereduce's `clip_noise_stationary` (default on): the noise statistics of ccx_specgate_reduce_long come from the first 600000
 * samples of the signal only (y_noise = y clipped to chunk_size); 0 = from the whole signal.  Parity unpinned (no fixture, the
 * package is not importable): [UPSTREAM-RECALL] says the clip is applied when the signal stands in for the noise clip as well. */
int ccx_specgate_set_clip_noise(ccx_specgate* g, int on);
/* One signal of ANY length (device pointers): noisereduce's chunked path for inputs beyond its chunk_size of 600000 samples --
 * threshold from the noise clip (see above), then each 600000-sample chunk gated with 30000 samples of real context on either side
 * (the reference passes whole files to nr.reduce_noise, back/api.py:832-833).  Capacity: n / 256 + 1 <= frame rows of the
 * workspace = (max_samples + 60000) / 256 + 2 per clip x max_clips.  y_dev and out_dev must not overlap. */
int ccx_specgate_reduce_long(ccx_specgate* g, const float* y_dev, int64_t n, float prop_decrease, float* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCX_H */
