// logmel.h -- launcher for the log-mel front end (see logmel.hip).
#pragma once
#include "ccx_common.h"

struct LogmelTables {
  const float* dft_cos;   // [400][208] cos(2*pi*n*k/400) * hann[n]
  const float* dft_sin;   // [400][208]
  const float* mel_fb;    // [80][208]
  const int* mel_range;   // [80][2]
};

// audio: [B][audio_stride] f32 device; n_samples_dev/seek_dev/seg_len_dev: [B] int device (seek and
// seg_len may be null = 0 / 3000).  Window frames t >= seg_len[b] are written as zeros.
// raw: [B][80][Fraw] scratch of which frames [0, Fcomp) are computed (the rest is all-zero padding = -10), gmax_bits: [B] scratch.  mel_out [B][80][3000] f32 and
// im2col [B*3000][256] bf16 are optional outputs.
int ccx_launch_logmel(ccx_ctx* ctx, const LogmelTables& tb, const float* audio, long audio_stride,
                      const int* n_samples_dev, const int* seek_dev, const int* seg_len_dev, int B, int Fraw, int Fcomp, float* raw,
                      unsigned int* gmax_bits, float* mel_out, bf16_t* im2col, hipStream_t stream);
