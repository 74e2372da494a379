// attention.h -- launchers for the attention kernels (see attention.hip).
#pragma once
#include "ccx_common.h"

// Encoder self-attention, head_dim 64, non-causal.
// Q,K: [B*H, Spad, 64] bf16 (rows >= S zero); Vt: [B*H, 64, Spad] bf16; O: [B*S, H*64] bf16.
int ccx_launch_enc_attention(ccx_ctx* ctx, const bf16_t* Q, const bf16_t* K, const bf16_t* Vt, bf16_t* O,
                             int B, int n_head, int S, int Spad, hipStream_t stream);
