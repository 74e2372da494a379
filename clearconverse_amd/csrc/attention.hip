// attention.hip -- attention kernels for the Whisper encoder (flash-style, MFMA) and the
// decoder (single-query, HBM-bound KV streaming with split-KV partials).
//
// Encoder kernel (non-causal, head_dim 64): one wave owns 32 queries; a 256-thread block owns
// 128.  It computes S^T = K * Q^T with v_mfma_f32_32x32x16_bf16 so that each lane holds a
// full column of scores for ONE query (softmax max/sum are lane-local plus one cross-half
// exchange), then O^T = V^T * P^T re-using the S^T accumulator registers directly as the
// B operand (no LDS round trip for P): see cdna_hip_programming.md section 3 "An accumulator
// tile as the next MFMA's operand".  The row permutation that makes the P registers a
// natural-k-order fragment (swap bits 2,3 of the MFMA row index) is applied when the K
// fragment is read from LDS, so it is free.  V arrives pre-transposed ([B,H,64,Spad]) from the
// QKV GEMM epilogue, so both LDS operands are plain ds_read_b128 row reads.
// K / V^T tiles stream HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), double buffered,
// XOR-swizzled on the source address (key (row>>1)&7, conflict-free per tools/lds_bank_sim.py).
#include <type_traits>
#include "attention.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define ATT_STAGE 16384  // K tile 8 KB + V^T tile 8 KB

__global__ __launch_bounds__(256, 2) void enc_attention_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kc, const bf16_t* __restrict__ Vt,
    bf16_t* __restrict__ O, int S, int Spad, int n_head, float scale_log2e, int n_qt) {
  __shared__ __attribute__((aligned(16))) char smem[2 * ATT_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware bijective remap (blocks b, b + 8 share an XCD and its L2; the same map as gemm_bf16.hip): every XCD gets a
  // contiguous run of logical tiles, so the n_qt query tiles of one (batch, head) run back to back on ONE XCD and its K / V^T
  // (384 KB at 1500 keys) is fetched into one L2 once instead of into most of the eight (FETCH 4.6x the algorithmic bytes before).
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
  const int bh = wg / n_qt;
  const int q0 = (wg - bh * n_qt) * 128 + wave * 32;
  const int l31 = lane & 31, hh = lane >> 5;

  const bf16_t* Qb = Q + (long)bh * Spad * 64;
  const bf16_t* Kb = Kc + (long)bh * Spad * 64;
  const bf16_t* Vb = Vt + (long)bh * 64 * Spad;

  // Q^T fragments (B operand): lane holds Q[q0 + l31][16*ds + 8*hh .. +8]
  bf16x8 qf[4];
  {
    int qr = q0 + l31; qr = qr < Spad ? qr : Spad - 1;
#pragma unroll
    for (int ds = 0; ds < 4; ds++) qf[ds] = *(const bf16x8*)(Qb + (long)qr * 64 + 16 * ds + 8 * hh);
  }

  // DMA source pointers: wave handles instructions 2*wave, 2*wave+1 of the K tile and of the V^T tile
  const bf16_t* srcK[2];
  const bf16_t* srcV[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int inst = wave * 2 + i;
    const int r = inst * 8 + (lane >> 3);       // LDS row (key for K tile, d for V^T tile)
    const int c = (lane & 7) ^ ((r >> 1) & 7);  // logical 16-byte chunk
    srcK[i] = Kb + (long)r * 64 + c * 8;        // + t*64*64
    srcV[i] = Vb + (long)r * Spad + c * 8;      // + t*64
  }
  auto stage = [&](int t, int buf) {
    char* sK = smem + buf * ATT_STAGE;
    char* sV = sK + 8192;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int inst = wave * 2 + i;
      __builtin_amdgcn_global_load_lds((gptr_t)(srcK[i] + (long)t * 64 * 64), (lptr_t)(sK + inst * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(srcV[i] + (long)t * 64), (lptr_t)(sV + inst * 1024), 16, 0, 0);
    }
  };

  // LDS read offsets.  K rows are permuted: MFMA row i <-> key i with bits 2,3 swapped.
  const int prow = (l31 & ~0xC) | (((l31 >> 2) & 1) << 3) | (((l31 >> 3) & 1) << 2);
  int offK[4], offV[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    offK[s] = prow * 128 + (((2 * s + hh) ^ ((prow >> 1) & 7)) << 4);  // + kt*32*128
    offV[s] = l31 * 128 + (((2 * s + hh) ^ ((l31 >> 1) & 7)) << 4);    // + dt*32*128
  }

  f32x16 oT[2];
#pragma unroll
  for (int i = 0; i < 16; i++) { oT[0][i] = 0.f; oT[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  const int nT = (S + 63) >> 6;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // One K/V tile.  TAIL (only the last tile can be ragged) is a compile-time tag: the key-range masking costs two VALU
  // slots per score when it sits in the common path.
  auto tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int buf = t & 1;
    if (t + 1 < nT) stage(t + 1, buf ^ 1);
    const char* sK = smem + buf * ATT_STAGE;
    const char* sV = sK + 8192;

    // ---- S^T = K * Q^T : two 32-key sub-tiles (the first MFMA of a chain takes the constant 0 as its accumulator) ----
    f32x16 sT[2];
#pragma unroll
    for (int kt = 0; kt < 2; kt++) {
#pragma unroll
      for (int ds = 0; ds < 4; ds++) {
        const bf16x8 kf = *(const bf16x8*)(sK + kt * 32 * 128 + offK[ds]);
        sT[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], ds == 0 ? zero16 : sT[kt], 0, 0, 0);
      }
    }
    // reg r of sub-tile kt <-> key t*64 + kt*32 + 16*(r>>3) + 8*hh + (r&7)
    // softmax in the exp2 domain with the scale folded into ONE fma per score: p = exp2(s * c - m * c), c = scale * log2(e).
    // (The kernel is VALU-bound at d = 64: every issue slot saved per score is MFMA time regained.)  m_run is kept scaled.
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        if (TAIL) {
          const int key = t * 64 + kt * 32 + 16 * (r >> 3) + 8 * hh + (r & 7);
          if (key >= S) sT[kt][r] = -INFINITY;
        }
        mx = fmaxf(mx, sT[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * scale_log2e);      // scale > 0: the maximum commutes with it
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    // (the row sums through the matrix core instead -- a constant "ones" row tile as a third V^T tile, 4 more MFMAs per key tile,
    // 32 adds per lane fewer -- were measured: 20.0 -> 22.5 ms per step; an MFMA holds the issue port longer than the adds it saves)
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(sT[kt][r], scale_log2e, -m_new));   // -inf stays -inf -> 0
        sT[kt][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    // the running maximum settles after the first tiles: skip the 32 rescaling multiplies when no lane of the wave moved
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) { oT[0][i] *= alpha; oT[1][i] *= alpha; }
    }

    // ---- P^T fragments straight from the accumulators (B operand, 16 keys per k-step) ----
    bf16x8 pf[4];
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int s = 0; s < 2; s++) {
        union { bf16x8 v; uint32_t u[4]; } cv;
#pragma unroll
        for (int j = 0; j < 4; j++) cv.u[j] = pack_bf16x2(sT[kt][8 * s + 2 * j], sT[kt][8 * s + 2 * j + 1]);
        pf[2 * kt + s] = cv.v;
      }
    // ---- O^T += V^T * P^T ----
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const bf16x8 vf = *(const bf16x8*)(sV + dt * 32 * 128 + offV[ks]);
        oT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], oT[dt], 0, 0, 0);
      }
    __syncthreads();
  };
  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < nT - 1; t++) tile(t, std::false_type{});
  if ((S & 63) != 0) tile(nT - 1, std::true_type{});
  else tile(nT - 1, std::false_type{});

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < S) {
    const int b = bh / n_head, h = bh - b * n_head;
    bf16_t* dst = O + ((long)(b * S + q) * n_head + h) * 64;
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        // regs 4g..4g+3 <-> d = dt*32 + 8g + 4*hh + (0..3)
        uint2 v;
        v.x = pack_bf16x2(oT[dt][4 * g + 0] * inv, oT[dt][4 * g + 1] * inv);
        v.y = pack_bf16x2(oT[dt][4 * g + 2] * inv, oT[dt][4 * g + 3] * inv);
        *(uint2*)(dst + dt * 32 + 8 * g + 4 * hh) = v;
      }
  }
}

int ccx_launch_enc_attention(ccx_ctx* ctx, const bf16_t* Q, const bf16_t* K, const bf16_t* Vt, bf16_t* O,
                             int B, int n_head, int S, int Spad, hipStream_t stream) {
  CCX_REQUIRE(ctx, B > 0 && n_head > 0 && S > 0, "enc_attention: empty problem");
  CCX_REQUIRE(ctx, Spad % 64 == 0 && Spad >= S, "enc_attention: Spad=%d must be a multiple of 64 and >= S=%d", Spad, S);
  const float scale_log2e = 0.125f * 1.4426950408889634f;  // head_dim 64: (64^-0.25)^2 = 1/8
  const int n_qt = ccx_cdiv(S, 128);
  dim3 grid(n_qt * B * n_head);
  {
    const double bh = (double)B * n_head;
    ccx_prof_scope ps(ctx, stream, "enc_attention_kernel", 4.0 * bh * S * (double)S * 64, 2.0 * bh * S * 64 * 4);
    hipLaunchKernelGGL(enc_attention_kernel, grid, dim3(256), 0, stream, Q, K, Vt, O, S, Spad, n_head, scale_log2e, n_qt);
  }
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}
