// gemm_bf16.hip -- hand-written CDNA4 bf16 GEMM, C = A * W^T, fp32 accumulate, fused epilogues.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile as 4x4
// v_mfma_f32_16x16x32_bf16 accumulators.  Operand tiles go HBM -> LDS with
// global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip), double buffered.  The LDS image
// is lane-linear (DMA constraint), so the XOR bank swizzle is applied to the per-lane
// SOURCE address and again on the ds_read_b128 side (cdna_hip_programming.md rule 21).
// Swizzle keys were chosen with tools/lds_bank_sim.py: both operand reads are
// conflict-free.
//
// The MFMA is issued with operands swapped (W fragment as "A", activation fragment as "B"),
// so the accumulator holds C^T: each lane owns ONE output row, and a row permutation applied when
// reading W from LDS decides WHICH columns.  The map is chosen per output type so that one store
// INSTRUCTION writes 64 contiguous bytes per row (tools/microbench_store.hip: a tile store whose
// lanes each own 64 contiguous bytes -- 16-byte pieces at a 64-byte pitch per instruction -- reaches
// 3.4 TB/s, the same bytes with the four lanes of a row side by side 5.2 TB/s):
//   CM 1 (fp32 out): lane (l15, h) owns columns 16 j + 4 h + r        -> one float4 per accumulator j,
//   CM 2 (bf16 out): lane owns columns 32 (j >> 1) + 8 h + 4 (j & 1) + r -> one uint4 (8 bf16) per accumulator pair.
// For V^T destinations (EPI_HEADS, v_transposed) the un-swapped order is used, so each lane owns 4
// consecutive ROWS of one column instead.
#include <map>
#include <string>
#include "gemm_bf16.h"

#ifndef CCX_GEMM_SETPRIO
#define CCX_GEMM_SETPRIO 1
#endif
// diagnostic builds of the phased kernel (tools/README.md): -DCCX_ABL_NO_MFMA=1 / _NO_DMA (in-loop prefetches) / _NO_EPI /
// _STORE_LOCAL (every tile stores to the first 256 output rows: the epilogue's instructions without its HBM traffic)
#ifndef CCX_ABL_NO_MFMA
#define CCX_ABL_NO_MFMA 0
#endif
#ifndef CCX_ABL_NO_DMA
#define CCX_ABL_NO_DMA 0
#endif
#ifndef CCX_ABL_NO_EPI
#define CCX_ABL_NO_EPI 0
#endif
// Experiment switches of the tile epilogue (compile time; tools/README.md): non-temporal stores of the output tile (CCX_EPI_NT_STORE) and
// non-temporal loads of the residual rows (CCX_EPI_NT_LOAD) -- every output / residual byte of these launches is touched once.
#ifndef CCX_EPI_NT_STORE
#define CCX_EPI_NT_STORE 0
#endif
#ifndef CCX_EPI_NT_LOAD
#define CCX_EPI_NT_LOAD 0
#endif
// fp32-residual epilogue of FULL 256 x 256 tiles through the LDS (see resid_via_lds below); 0 = the register-staged groups
#ifndef CCX_EPI_RESID_LDS
#define CCX_EPI_RESID_LDS 1
#endif
#ifndef CCX_ABL_STORE_LOCAL
#define CCX_ABL_STORE_LOCAL 0
#endif
#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES 32768  // A 16 KB + B 16 KB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int keyA(int r) { return (r >> 1) & 7; }
// W rows: row (0..63 inside a wave column) read by the lanes with l15 = i for accumulator j, and the swizzle key of a row.
// Both keys are conflict-free for the 16 rows one ds_read_b128 touches (tools/lds_bank_sim.py) and do not depend on j.
template <int CM> __device__ __forceinline__ int w_row(int j, int i) {
  return CM == 1 ? 16 * j + i : 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
}
template <int CM> __device__ __forceinline__ int keyW(int r) {
  return CM == 1 ? (r >> 1) & 7 : ((r >> 1) & 1) | (((r >> 3) & 1) << 1) | (((r >> 4) & 1) << 2);
}
// first of the 4 consecutive columns (inside the wave's 64) that accumulator j holds in a lane of quarter h
template <int CM> __device__ __forceinline__ int col4(int j, int h) {
  return CM == 1 ? 16 * j + 4 * h : 32 * (j >> 1) + 8 * h + 4 * (j & 1);
}
typedef unsigned int ccx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void epi_store16(void* dst, uint4 v) {
  if (CCX_EPI_NT_STORE) __builtin_nontemporal_store((ccx_u32x4){v.x, v.y, v.z, v.w}, (ccx_u32x4*)dst);
  else *(uint4*)dst = v;
}
__device__ __forceinline__ void epi_store16f(void* dst, float4 v) {
  if (CCX_EPI_NT_STORE) __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, (f32x4*)dst);
  else *(float4*)dst = v;
}
__device__ __forceinline__ float4 epi_load16f(const float* src) {
  if (CCX_EPI_NT_LOAD) { const f32x4 t = __builtin_nontemporal_load((const f32x4*)src); return make_float4(t[0], t[1], t[2], t[3]); }
  return *(const float4*)src;
}

constexpr int colmap_of(int epi) { return (epi == EPI_F32 || epi == EPI_F32_RESID || epi == EPI_F32_GELU_POS) ? 1 : 2; }

// Block geometry: WM x WN waves, each wave owns (MT*16) x 64 outputs as MT x 4 accumulators.
//   <2,2,4>: 128x128 tile, 4 waves, 64 KB LDS (2 stages) -> 2 blocks per CU        (64 flop per LDS-DMA byte)
//   <2,4,8>: 256x256 tile, 8 waves, 128 KB LDS (2 stages) -> 1 block per CU        (128 flop per LDS-DMA byte)
// The encoder GEMMs (K = 768) are bound by the L2 -> LDS operand stream, not by HBM or MFMA issue, so the
// large tile is used whenever it still yields enough tiles to fill the 256 CUs.
template <bool SWAP, int CM, int WM, int WN, int MT>
__device__ __forceinline__ void gemm_mainloop(const GemmParams& p, char* smem, int m0, int n0, f32x4 (&acc)[MT][4]) {
  constexpr int NW = WM * WN;
  constexpr int A_ROWS = WM * MT * 16, B_ROWS = WN * 64;
  constexpr int A_BYTES = A_ROWS * 128, B_BYTES = B_ROWS * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NA = A_ROWS / 8 / NW, NB = B_ROWS / 8 / NW;   // DMA instructions (8 rows x 128 B) per wave per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;

  // ---- per-lane DMA source rows (loop invariant) ----
  const bf16_t* srcA[NA];
  const bf16_t* srcB[NB];
#pragma unroll
  for (int i = 0; i < NA; i++) {
    const int r = (wave * NA + i) * 8 + (lane >> 3);
    int gm = m0 + r; gm = gm < p.M ? gm : p.M - 1;
    srcA[i] = p.A + (long)gm * p.lda + (((lane & 7) ^ keyA(r)) << 3);
  }
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int r = (wave * NB + i) * 8 + (lane >> 3);
    int gn = n0 + r; gn = gn < p.N ? gn : p.N - 1;
    srcB[i] = p.W + (long)gn * p.ldw + (((lane & 7) ^ keyW<CM>(r)) << 3);
  }
  const int kt_per_tap = p.K / BK;
  auto stage = [&](int t, int buf) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_BYTES;
    const int tap = t / kt_per_tap;                               // scalar: which shifted view of A
    const long ka = (long)(t - tap * kt_per_tap) * BK + (long)tap * p.a_tap_stride;
    const int kw = t * BK;                                        // W rows hold the taps back to back
#pragma unroll
    for (int i = 0; i < NA; i++)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + ka), (lptr_t)(sA + (wave * NA + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; i++)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcB[i] + kw), (lptr_t)(sB + (wave * NB + i) * 1024), 16, 0, 0);
  };

  // ---- per-lane LDS read offsets ----
  const int l15 = lane & 15, h = lane >> 4;
  const int ka = keyA(l15);                       // rows wr*MT*16 + mt*16 + l15
  const int kb = keyW<CM>(w_row<CM>(0, l15));     // rows wc*64 + w_row(j, l15): the key is the same for every j
  int offA[2], offB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    offA[ks] = (wr * MT * 16 + l15) * 128 + (((4 * ks + h) ^ ka) << 4);
    offB[ks] = (wc * 64 + w_row<CM>(0, l15)) * 128 + (((4 * ks + h) ^ kb) << 4);
  }

  const int nt = kt_per_tap * (p.ntaps > 1 ? p.ntaps : 1);
  stage(0, 0);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and makes it visible to all waves
  for (int t = 0; t < nt; t++) {
    const int buf = t & 1;
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
    const char* sA = smem + buf * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 b[4];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = *(const bf16x8*)(sB + offB[ks] + w_row<CM>(j, 0) * 128);
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        const bf16x8 a = *(const bf16x8*)(sA + offA[ks] + mt * 16 * 128);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (SWAP)
            acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a, acc[mt][j], 0, 0, 0);
          else
            acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[mt][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ long remap_row(const GemmParams& p, int m, bool& valid) {
  if (p.rpb_in <= 0) { valid = true; return m; }
  int g = m / p.rpb_in;
  const int i = m - g * p.rpb_in;
  valid = i < p.rpb_valid;
  if (p.img_rows_in > 0) {
    const int img = g / p.img_rows_in, r = g - img * p.img_rows_in;
    valid = valid && r < p.img_rows_valid;
    g = img * p.img_rows_out + r;
  }
  return (long)g * p.rpb_out + i + p.roff;
}

// One output tile: main loop (a functor (p, smem, m0, n0, acc, swap_tag)) and epilogue.  `acc` as left by a SWAP main loop --
// lane (l15, h): output row l15 of each row tile, accumulator j = the 4 columns col4(j, h) .. +3 of the wave's 64 -- or by the
// un-swapped one for V^T destinations.
template <int EPI, int WM, int WN, int MT, typename MainLoop>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, char* smem, int m0, int n0, MainLoop mainloop) {
  constexpr int CM = colmap_of(EPI);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform on purpose: column block, head and the q / k / v
  const int wr = wave / WN, wc = wave % WN;                            // destination pointer stay in SGPRs (no pointer select through memory)
  const int l15 = lane & 15, h = lane >> 4;
  const int cb = n0 + wc * 64;                    // first column of the wave's 64

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (EPI == EPI_HEADS) {
    const int blk = n0 / p.d_model + p.first_block;  // 0=q 1=k 2=v  (d_model % 128 == 0)
    if (blk == 2 && p.v_transposed) {
      mainloop(p, smem, m0, n0, acc, std::false_type{});
      // lane: column n = cb + w_row(j, l15) ; rows m0 + wr*MT*16 + mt*16 + 4h + reg
      // (the four bias values are loaded AND consumed before the first store: a load waited for inside the row branches costs a
      // vmcnt(0) per branch, i.e. every store waits for the one before it)
      float bvj[4];
#pragma unroll
      for (int j = 0; j < 4; j++) bvj[j] = p.bias ? p.bias[cb + w_row<CM>(j, l15)] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) asm volatile("" ::"v"(bvj[j]));
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int n = cb + w_row<CM>(j, l15);
        const int nn = n % p.d_model, hh = nn >> 6, d = nn & 63;
        const float bv = bvj[j];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const int m = m0 + wr * MT * 16 + mt * 16 + 4 * h;
          if (m < p.M) {
            const int b = m / p.S, s = m - b * p.S;
            bf16_t* dst = p.hv + ((long)(b * p.n_head + hh) * 64 + d) * p.Spad + s;
            uint2 v;
            v.x = pack_bf16x2(acc[mt][j][0] + bv, acc[mt][j][1] + bv);
            v.y = pack_bf16x2(acc[mt][j][2] + bv, acc[mt][j][3] + bv);
            *(uint2*)dst = v;
          }
        }
      }
      return;
    }
  }

  mainloop(p, smem, m0, n0, acc, std::true_type{});

  // columns past N rounded up to 16 are not written (the destination has that many, see gemm_bf16.h); 4- and 8-wide groups
  // start on multiples of 4 / 8, so a group lies wholly on one side
  const int Nw = (p.N + 15) & ~15;
  if (EPI != EPI_HEADS && cb >= Nw) return;       // the wave's 64 columns lie wholly past N (narrow layers)
  int c4[4];
#pragma unroll
  for (int j = 0; j < 4; j++) c4[j] = cb + col4<CM>(j, h);
  float bias[16];
#pragma unroll
  for (int i = 0; i < 16; i++) bias[i] = 0.f;
  if (p.bias) {
    const bool al = ((uintptr_t)p.bias & 15) == 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (al && c4[j] + 4 <= p.N) {
        const float4 b4 = *(const float4*)(p.bias + c4[j]);
        bias[4 * j] = b4.x; bias[4 * j + 1] = b4.y; bias[4 * j + 2] = b4.z; bias[4 * j + 3] = b4.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) bias[4 * j + r] = (c4[j] + r < p.N) ? p.bias[c4[j] + r] : 0.f;
      }
    }
  }
  // Consume the bias HERE, in straight-line code: hipcc otherwise waits for it with a `vmcnt(0)` at the head of every row
  // tile's block (it cannot prove across the branches that the loads have been waited for), and since stores count in vmcnt
  // each row tile then waited for the previous one's stores to land -- eight dependent store round trips per output tile.
#pragma unroll
  for (int i = 0; i < 16; i++) asm volatile("" ::"v"(bias[i]));

  // Rows are finished in groups of GM row tiles, software-pipelined: the residual rows of group g+1 are requested BEFORE group
  // g is finished and stored.  Stores count in vmcnt like loads and retire in order, so a residual load issued after a store
  // can only be waited for together with that store; requested first, the wait is a counted one that leaves the stores in
  // flight (the fp32-residual epilogue of a 256 x 256 tile used to be eight, then two, dependent load -> add -> store rounds).
  // FULL tiles (no row remap, wholly inside M x N) run the same code without per-row branches, which is what lets hipcc count.
  constexpr bool RES_F32 = (EPI == EPI_F32_RESID || EPI == EPI_F32_GELU_POS || EPI == EPI_BF16_LRELU_AFFINE);
  constexpr bool RES_BF16 = (EPI == EPI_BF16_ADD_RELU);
  constexpr bool OUT_BF16 = CM == 2;
  constexpr int GM = EPI == EPI_BF16_LRELU_AFFINE ? 1 : (RES_F32 || RES_BF16) ? 2 : 4;   // (the affine epilogue is short of registers)
  constexpr int NG = MT / GM;
  struct Group {
    long orow[GM];
    bool ok[GM];
    float4 rf[RES_F32 ? GM : 1][4];
    uint4 rb[RES_BF16 ? GM : 1][2];
  };
  auto rows = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    auto request = [&](int mt0, Group& G) {
      if (EPI == EPI_HEADS) return;
#pragma unroll
      for (int g = 0; g < GM; g++) {
        const int m = m0 + wr * MT * 16 + (mt0 + g) * 16 + l15;
        bool valid = FULL;
        long orow = m;
        if (!FULL) { orow = 0; if (m < p.M) orow = remap_row(p, m, valid); }
        G.ok[g] = valid; G.orow[g] = orow;
        // Residual rows are requested WITHOUT a branch around the loads (rows and column groups that will not be stored read row 0 /
        // column 0 of the residual instead and their sums are dropped): hipcc waits with vmcnt(0) for a load that was issued
        // under a lane mask as soon as its use sits in another masked block, and with stores in between every store then
        // waited for the one before it (the ResNet convolutions run this path for every tile: their rows are remapped).
        if constexpr (RES_F32) {
          const bool have = EPI != EPI_BF16_LRELU_AFFINE || p.resid != nullptr;     // block-uniform
          const long rrow = !valid ? 0 : (EPI != EPI_BF16_LRELU_AFFINE && p.resid_mod > 0) ? (orow % p.resid_mod) : orow;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int cj = (FULL || c4[j] < Nw) ? c4[j] : 0;
            if (have) G.rf[g][j] = epi_load16f(p.resid + rrow * p.ldr + cj);
            else G.rf[g][j] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
        if constexpr (RES_BF16) {
          const long rrow = valid ? orow : 0;
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const int ci = (FULL || c4[2 * i] < Nw) ? c4[2 * i] : 0;
            if (p.resid_bf16) G.rb[g][i] = *(const uint4*)(p.resid_bf16 + rrow * p.ldrb + ci);
            else G.rb[g][i] = make_uint4(0, 0, 0, 0);
          }
        }
      }
    };
    auto finish = [&](int mt0, const Group& G) {
      // the group's residual is waited for HERE, once and outside the row branches (see request())
#pragma unroll
      for (int g = 0; g < GM; g++) {
        if constexpr (RES_F32) {
#pragma unroll
          for (int j = 0; j < 4; j++) asm volatile("" ::"v"(G.rf[g][j].x), "v"(G.rf[g][j].y), "v"(G.rf[g][j].z), "v"(G.rf[g][j].w));
        }
        if constexpr (RES_BF16) {
#pragma unroll
          for (int i = 0; i < 2; i++) asm volatile("" ::"v"(G.rb[g][i].x), "v"(G.rb[g][i].y), "v"(G.rb[g][i].z), "v"(G.rb[g][i].w));
        }
      }
#pragma unroll
      for (int g = 0; g < GM; g++) {
        const int mt = mt0 + g;
        const int m = m0 + wr * MT * 16 + mt * 16 + l15;
        if (!FULL && m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) v[4 * j + r] = acc[mt][j][r] + bias[4 * j + r];

        if (EPI == EPI_HEADS) {
          // the wave's 64 columns are one head of one of q / k / v (d_model % 128 == 0, cb % 64 == 0)
          const int blk = cb / p.d_model + p.first_block;
          const int hh = (cb % p.d_model) >> 6;
          const int b = m / p.S, s = m - b * p.S;
          bf16_t* base = blk == 0 ? p.hq : (blk == 1 ? p.hk : p.hv);
          bf16_t* dst = base + ((long)(b * p.n_head + hh) * p.Spad + s) * 64;
#pragma unroll
          for (int i = 0; i < 2; i++) {
            uint4 o;
            o.x = pack_bf16x2(v[8 * i], v[8 * i + 1]);     o.y = pack_bf16x2(v[8 * i + 2], v[8 * i + 3]);
            o.z = pack_bf16x2(v[8 * i + 4], v[8 * i + 5]); o.w = pack_bf16x2(v[8 * i + 6], v[8 * i + 7]);
            epi_store16(dst + (c4[2 * i] - cb), o);
          }
          continue;
        }

        if (!FULL && !G.ok[g]) continue;
        const long orow = CCX_ABL_STORE_LOCAL ? (G.orow[g] & 255) : G.orow[g];

        if constexpr (EPI == EPI_BF16_LRELU_AFFINE) {
#pragma unroll
          for (int j = 0; j < 4; j++) {     // zeros when there is no residual
            v[4 * j + 0] += G.rf[g][j].x; v[4 * j + 1] += G.rf[g][j].y; v[4 * j + 2] += G.rf[g][j].z; v[4 * j + 3] += G.rf[g][j].w;
          }
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int c = c4[j] + r;
              float t = v[4 * j + r] >= 0.f ? v[4 * j + r] : p.slope * v[4 * j + r];
              const float sc = (p.scale && c < p.N) ? p.scale[c] : 1.f;
              const float sh = (p.shift && c < p.N) ? p.shift[c] : 0.f;
              v[4 * j + r] = t * sc + sh;
            }
        }
        if constexpr (EPI == EPI_BF16_ADD_RELU) {
          const uint32_t rw[8] = {G.rb[g][0].x, G.rb[g][0].y, G.rb[g][0].z, G.rb[g][0].w, G.rb[g][1].x, G.rb[g][1].y, G.rb[g][1].z, G.rb[g][1].w};
#pragma unroll
          for (int i = 0; i < 8; i++) {     // zeros when there is no residual
            v[2 * i] += __uint_as_float(rw[i] << 16);
            v[2 * i + 1] += __uint_as_float(rw[i] & 0xffff0000u);
          }
#pragma unroll
          for (int i = 0; i < 16; i++) v[i] = fmaxf(v[i], 0.f);
        }
        if constexpr (OUT_BF16) {
          if (EPI == EPI_BF16_GELU) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = gelu_erf(v[i]);
          }
          if (EPI == EPI_BF16_RELU) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = fmaxf(v[i], 0.f);
          }
          bf16_t* dst = (bf16_t*)p.out + orow * p.ldo;
#pragma unroll
          for (int i = 0; i < 2; i++) {
            if (!FULL && c4[2 * i] >= Nw) continue;
            uint4 o;
            o.x = pack_bf16x2(v[8 * i], v[8 * i + 1]);     o.y = pack_bf16x2(v[8 * i + 2], v[8 * i + 3]);
            o.z = pack_bf16x2(v[8 * i + 4], v[8 * i + 5]); o.w = pack_bf16x2(v[8 * i + 6], v[8 * i + 7]);
            epi_store16(dst + c4[2 * i], o);
          }
        } else {
          if (EPI == EPI_F32_GELU_POS) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = gelu_erf(v[i]);
          }
          if constexpr (EPI == EPI_F32_RESID || EPI == EPI_F32_GELU_POS) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              v[4 * j + 0] += G.rf[g][j].x; v[4 * j + 1] += G.rf[g][j].y; v[4 * j + 2] += G.rf[g][j].z; v[4 * j + 3] += G.rf[g][j].w;
            }
          }
          float* dst = (float*)p.out + orow * p.ldo;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (!FULL && c4[j] >= Nw) continue;
            epi_store16f(dst + c4[j], make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]));
          }
        }
      }
    };
    Group G[2];
    request(0, G[0]);
#pragma unroll
    for (int g0 = 0; g0 < NG; g0++) {
      if (g0 + 1 < NG) request((g0 + 1) * GM, G[(g0 + 1) & 1]);
      finish(g0 * GM, G[g0 & 1]);
    }
  };
  if (CCX_ABL_NO_EPI) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int j = 0; j < 4; j++) asm volatile("" ::"v"(acc[mt][j]));
    return;
  }
  const bool full = p.rpb_in <= 0 && m0 + WM * MT * 16 <= p.M && n0 + WN * 64 <= p.N;
  if constexpr (CCX_EPI_RESID_LDS && EPI == EPI_F32_RESID && WM * WN == 8 && MT == 8) {
    // fp32 residual of a FULL 256 x 256 tile THROUGH THE LDS (round 3).  The register-staged path above keeps two row tiles of
    // residual (8 float4) in flight per lane = 64 KB per CU and needs four dependent HBM round trips per tile, each ~2.5 us under
    // load: 512 KB of read-modify-write at ~22 GB/s per CU, latency-bound (Little), not bandwidth-bound.  The operand image is dead
    // once the main loop is over, so the residual rows travel by LDS-DMA instead -- no registers while in flight: a wave fetches the
    // 64 rows x 64 columns (16 KB, 16 DMA instructions of 4 rows x 256 B) of one HALF of its 128 x 64 block into its own 16 KB of
    // LDS, 128 KB in flight per CU, two round trips, and the second half is requested before the first half's sums are stored.
    // Image: row r = 256 B = 16 chunks of 16 B; chunk c of the LDS row holds global chunk c ^ (r & 15) (swizzle on the DMA source),
    // the lane of row l15 that wants chunk 4 j + h reads (4 j + h) ^ l15: conflict-free for ds_read_b128.  Same arithmetic, same
    // order ((acc + bias) + residual): bit-identical to the register path.
    if (full && p.resid != nullptr && p.resid_mod == 0) {
      __syncthreads();                       // every wave is done with the operand image (all DMA was waited for in the last phase)
      char* my = smem + wave * 16384;
      const float* rbase = p.resid + (long)(m0 + wr * MT * 16) * p.ldr + cb;
      auto fetch_half = [&](int hh) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int rl = 4 * i + (lane >> 4);                 // row inside the half (0..63)
          const int ch = (lane & 15) ^ (rl & 15);             // global chunk that lands in LDS chunk (lane & 15)
          __builtin_amdgcn_global_load_lds((gptr_t)(rbase + (long)(hh * 64 + rl) * p.ldr + 4 * ch), (lptr_t)(my + i * 1024), 16, 0, 0);
        }
      };
      // two row tiles (32 rows) at a time: 8 ds_read_b128 -> 32 registers of residual -> 8 stores
      auto read_pair = [&](int qq, float4 (&rf)[2][4]) {
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            f32x4 t;
            asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(uintptr_t)(lptr_t)(my + ((2 * qq + q) * 16 + l15) * 256 + (((4 * j + h) ^ l15) << 4))));
            rf[q][j] = make_float4(t[0], t[1], t[2], t[3]);
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      };
      auto finish_pair = [&](int hh, int qq, const float4 (&rf)[2][4]) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int mt = hh * 4 + 2 * qq + q;
          const long orow = m0 + wr * MT * 16 + mt * 16 + l15;
          float* dst = (float*)p.out + orow * p.ldo;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            float4 o;
            o.x = (acc[mt][j][0] + bias[4 * j + 0]) + rf[q][j].x; o.y = (acc[mt][j][1] + bias[4 * j + 1]) + rf[q][j].y;
            o.z = (acc[mt][j][2] + bias[4 * j + 2]) + rf[q][j].z; o.w = (acc[mt][j][3] + bias[4 * j + 3]) + rf[q][j].w;
            epi_store16f(dst + c4[j], o);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      float4 rf[2][4];
      fetch_half(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the wave's own 16 DMA instructions (nobody else reads this region)
      read_pair(0, rf);
      finish_pair(0, 0, rf);                                 // 8 stores
      read_pair(1, rf);                                      // every read of half 0 has completed (lgkmcnt(0)) ...
      fetch_half(1);                                         // ... so its region may be overwritten
      __builtin_amdgcn_sched_barrier(0);                     // the vmcnt(8) below counts on this issue order: 8 stores, 16 DMA, 8 stores
      finish_pair(0, 1, rf);                                 // 8 stores behind the DMA
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // in-order retirement: the DMA (and the 8 older stores) are done
      read_pair(0, rf);
      finish_pair(1, 0, rf);
      read_pair(1, rf);
      finish_pair(1, 1, rf);
      return;
    }
  }
  if (full) rows(std::true_type{});
  else rows(std::false_type{});
}

// XCD-aware bijective remap (blocks b, b+8 share an XCD/L2): give each XCD a contiguous
// run of tiles so the A row-panel and the (small) W are re-read from that XCD's L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
  return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
}

template <int EPI, int WM, int WN, int MT>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN) >= 4 ? (WM * WN) / 4 * (MT == 4 ? 2 : 1) : 4) void gemm_bf16_nt_kernel(GemmParams p) {
  constexpr int TBM = WM * MT * 16, TBN = WN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (p.N + TBN - 1) / TBN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  gemm_tile<EPI, WM, WN, MT>(p, smem, tm * TBM, tn * TBN, [](const GemmParams& pp, char* sm, int m0, int n0, f32x4 (&acc)[MT][4], auto swap) {
    gemm_mainloop<decltype(swap)::value, colmap_of(EPI), WM, WN, MT>(pp, sm, m0, n0, acc);
  });
}

// ------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, phased main loop (cdna_hip_programming.md "The 256^2 8-phase template"): same LDS image, swizzle keys,
// fragment maps and epilogues as the kernel above; what changes is WHEN things are issued.
//   * a K tile (64 KB: A 256 rows, W 256 rows) is staged as four 16 KB HALF-TILES: A-half a = the a-th 64 rows of each wave
//     row's 128, W-half b = rows 32 b .. 32 b + 31 of every wave column's 64 (= accumulators j in {2b, 2b+1});
//   * a K tile is consumed in four PHASES of 16 MFMAs per wave: (A0,W0) (A0,W1) (A1,W1) (A1,W0), each phase = [LDS reads of the
//     operand half it is the first to need + ONE half-tile prefetch] barrier [MFMAs] barrier;
//   * the prefetch runs 3 half-tiles ahead and stays in flight ACROSS the barriers: one counted `s_waitcnt vmcnt(6)` per K tile
//     (phase 4), never 0 inside the loop;
//   * the two wave rows (one wave of each per SIMD) run ONE BARRIER APART, so that a SIMD's matrix core works for one wave
//     while the other wave issues its reads and prefetches.
// Hazards, with barrier b_n the n-th barrier of wave row 0 (row 1 passes it as its (n+1)-th call), phase g = [L(g)] b_2g [M(g)]
// b_2g+1 for row 0 and [L(g)] b_2g+1 [M(g)] b_2g+2 for row 1:
//   RAW  a half-tile waited for (every wave: its own two DMA instructions) in L(g) is complete and visible after b_2g+1;
//        both rows read it in L(g+1) or later.  Tile t+1 is waited for in phase 4 of tile t and first read in phase 1 of t+1.
//   WAR  the reads of phase g are complete (lgkmcnt(0) before the MFMAs) before b_2g+1 in row 0 and before b_2g+2 in row 1;
//        row 0 restages in L(g') after b_2g'-1, so a slot BOTH rows read (W halves) is restaged two phases after its last read;
//        a slot only the staging row reads (A halves: a wave stages rows of its own wave row) one phase after.
//   Stage schedule (slot <- half-tile): phase 1 of tile t: A1(t+1); phase 2: A0(t+2); phase 3: W0(t+2); phase 4: W1(t+2).
//   Last reads: A0, W0 phase 1; W1 phase 2; A1 phase 3.
template <bool SWAP, int CM>
__device__ __forceinline__ void gemm_mainloop_phased(const GemmParams& p, char* smem, int m0, int n0, f32x4 (&acc)[8][4]) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- per-lane DMA sources: [half][instruction]; an instruction fills 8 LDS rows (1 KB) ----
  const bf16_t* srcA[2][2];
  const bf16_t* srcB[2][2];
  int dstA[2][2], dstB[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int r0 = wr * 128 + a * 64 + wc * 16 + i * 8;
      const int r = r0 + (lane >> 3);
      int gm = m0 + r; gm = gm < p.M ? gm : p.M - 1;
      srcA[a][i] = p.A + (long)gm * p.lda + (((lane & 7) ^ keyA(r)) << 3);
      dstA[a][i] = r0 * 128;
      const int x = 2 * wave + i;                       // W-half a = rows 32 a .. 32 a + 31 of every wave column's 64
      const int g8 = 8 * (x >> 2) + 4 * a + (x & 3);    // 8-row group
      const int rb = g8 * 8 + (lane >> 3);
      int gn = n0 + rb; gn = gn < p.N ? gn : p.N - 1;
      srcB[a][i] = p.W + (long)gn * p.ldw + (((lane & 7) ^ keyW<CM>(rb)) << 3);
      dstB[a][i] = 32768 + g8 * 1024;
    }
  const int kt_per_tap = p.K / BK;
  const int nt = kt_per_tap * (p.ntaps > 1 ? p.ntaps : 1);
  // element offset of K tile t in a row of A: taps are shifted views of A (a_tap_stride apart), kt_per_tap tiles each
  auto a_koff = [&](int t) -> long {
    const int tap = t / kt_per_tap;
    return (long)(t - tap * kt_per_tap) * BK + (long)tap * p.a_tap_stride;
  };
  auto stage_a = [&](int t, long ka, int a) {
    char* buf = smem + (t & 1) * 65536;
    __builtin_amdgcn_global_load_lds((gptr_t)(srcA[a][0] + ka), (lptr_t)(buf + dstA[a][0]), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(srcA[a][1] + ka), (lptr_t)(buf + dstA[a][1]), 16, 0, 0);
  };
  auto stage_b = [&](int t, int b) {
    char* buf = smem + (t & 1) * 65536;
    const int kw = t * BK;
    __builtin_amdgcn_global_load_lds((gptr_t)(srcB[b][0] + kw), (lptr_t)(buf + dstB[b][0]), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(srcB[b][1] + kw), (lptr_t)(buf + dstB[b][1]), 16, 0, 0);
  };

  // ---- per-lane LDS read offsets (as in gemm_mainloop) ----
  const int l15 = lane & 15, h = lane >> 4;
  const int ka_ = keyA(l15);
  const int kb_ = keyW<CM>(w_row<CM>(0, l15));
  int offA[2], offB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    offA[ks] = (wr * 128 + l15) * 128 + (((4 * ks + h) ^ ka_) << 4);
    offB[ks] = 32768 + (wc * 64 + w_row<CM>(0, l15)) * 128 + (((4 * ks + h) ^ kb_) << 4);
  }

  // ---- prologue: tile 0 and three half-tiles of tile 1 in flight; tile 0 landed and visible to everybody ----
  long ka1 = nt > 1 ? a_koff(1) : 0;           // of tile t+1 and t+2 inside the loop: advanced without a division there
  long ka2 = nt > 2 ? a_koff(2) : 0;
  int kk2 = nt > 2 ? 2 % kt_per_tap : 0;
  stage_a(0, 0, 0); stage_b(0, 0); stage_b(0, 1); stage_a(0, 0, 1);
  if (nt > 1) {
    stage_a(1, ka1, 0); stage_b(1, 0); stage_b(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind from here on

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define CCX_PHASE_MFMA(MT0, J0, FB)                                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                                        \
  __builtin_amdgcn_s_barrier();                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
  __builtin_amdgcn_sched_barrier(0);                                                                        \
  if (CCX_GEMM_SETPRIO) __builtin_amdgcn_s_setprio(1);                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 2; ks++)                                                          \
  _Pragma("unroll") for (int m = 0; m < 4; m++)                                                             \
  _Pragma("unroll") for (int jj = 0; jj < 2; jj++) {                                                        \
    if (CCX_ABL_NO_MFMA) { asm volatile("" ::"v"(FB[jj][ks]), "v"(fa[m][ks])); }                           \
    else if (SWAP) acc[MT0 + m][J0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[jj][ks], fa[m][ks], acc[MT0 + m][J0 + jj], 0, 0, 0); \
    else      acc[MT0 + m][J0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][ks], FB[jj][ks], acc[MT0 + m][J0 + jj], 0, 0, 0); \
  }                                                                                                         \
  if (CCX_GEMM_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                        \
  __builtin_amdgcn_s_barrier();

  for (int t = 0; t < nt; t++) {
    const char* buf = smem + (t & 1) * 65536;
    // phase 1: W0 and A0 fragments; prefetch A1(t+1)
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int jj = 0; jj < 2; jj++) fb0[jj][ks] = *(const bf16x8*)(buf + offB[ks] + w_row<CM>(jj, 0) * 128);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int m = 0; m < 4; m++) fa[m][ks] = *(const bf16x8*)(buf + offA[ks] + m * 16 * 128);
    if (!CCX_ABL_NO_DMA && t + 1 < nt) stage_a(t + 1, ka1, 1);
    CCX_PHASE_MFMA(0, 0, fb0)
    // phase 2: W1 fragments; prefetch A0(t+2)
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int jj = 0; jj < 2; jj++) fb1[jj][ks] = *(const bf16x8*)(buf + offB[ks] + w_row<CM>(2 + jj, 0) * 128);
    if (!CCX_ABL_NO_DMA && t + 2 < nt) stage_a(t + 2, ka2, 0);
    CCX_PHASE_MFMA(0, 2, fb1)
    // phase 3: A1 fragments; prefetch W0(t+2)
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int m = 0; m < 4; m++) fa[m][ks] = *(const bf16x8*)(buf + offA[ks] + (4 + m) * 16 * 128);
    if (!CCX_ABL_NO_DMA && t + 2 < nt) stage_b(t + 2, 0);
    CCX_PHASE_MFMA(4, 2, fb1)
    // phase 4: W0 fragments are still in registers; prefetch W1(t+2); tile t+1 must have landed before phase 1 of t+1
    if (!CCX_ABL_NO_DMA && t + 2 < nt) {
      stage_b(t + 2, 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CCX_PHASE_MFMA(4, 0, fb0)
    ka1 = ka2;
    kk2++;
    if (kk2 == kt_per_tap) { kk2 = 0; ka2 += BK + p.a_tap_stride - (long)kt_per_tap * BK; }
    else ka2 += BK;
  }
#undef CCX_PHASE_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();   // pairs with wave row 1's last barrier
}

template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_bf16_phased_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (p.N + 255) / 256;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  gemm_tile<EPI, 2, 4, 8>(p, smem, tm * 256, tn * 256, [](const GemmParams& pp, char* sm, int m0, int n0, f32x4 (&acc)[8][4], auto swap) {
    gemm_mainloop_phased<decltype(swap)::value, colmap_of(EPI)>(pp, sm, m0, n0, acc);
  });
}

template <int EPI, int WM, int WN, int MT>
static int launch_epi_geo(ccx_ctx* ctx, const GemmParams& p, hipStream_t stream) {
  constexpr int TBM = WM * MT * 16, TBN = WN * 64;
  constexpr int LDS = 2 * (TBM + TBN) * 128;
  const int tiles = ccx_cdiv(p.M, TBM) * ccx_cdiv(p.N, TBN);
  static ccx_lds_optin optin;
  CCX_HIP(ctx, optin.ensure(ctx->device, (const void*)gemm_bf16_nt_kernel<EPI, WM, WN, MT>, LDS));
  {
    // algorithmic work: 2*M*N*K flops; bytes = A + W read once + output written once
    const double obytes = (EPI == EPI_F32 || EPI == EPI_F32_RESID || EPI == EPI_F32_GELU_POS) ? 4.0 : 2.0;  // output element size
    const double kt = (double)p.K * (p.ntaps > 1 ? p.ntaps : 1);
    // CCX_PROF_SHAPES=1: one label per (epilogue, tile, M, N, K, taps) for shape-level timing tables
    static const bool by_shape = getenv("CCX_PROF_SHAPES") != nullptr;
    const char* label = "gemm_bf16_nt_kernel";
    if (by_shape && ctx->prof_on) {
      static std::mutex mu;
      static std::map<std::string, std::string> names;
      char buf[160];
      snprintf(buf, sizeof(buf), "gemm<epi%d,%dx%d> M=%d N=%d K=%d taps=%d", EPI, TBM, TBN, p.M, p.N, p.K, p.ntaps > 1 ? p.ntaps : 1);
      std::lock_guard<std::mutex> lk(mu);
      label = names.emplace(buf, buf).first->second.c_str();
    }
    ccx_prof_scope ps(ctx, stream, label, 2.0 * p.M * (double)p.N * kt,
                      2.0 * ((double)p.M * p.K + (double)p.N * kt) + obytes * p.M * (double)p.N);
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI, WM, WN, MT>), dim3(tiles), dim3(WM * WN * 64), LDS, stream, p);
  }
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

template <int EPI>
static int launch_phased(ccx_ctx* ctx, const GemmParams& p, hipStream_t stream) {
  constexpr int LDS = 131072;
  const int tiles = ccx_cdiv(p.M, 256) * ccx_cdiv(p.N, 256);
  static ccx_lds_optin optin;
  CCX_HIP(ctx, optin.ensure(ctx->device, (const void*)gemm_bf16_phased_kernel<EPI>, LDS));
  {
    const double obytes = (EPI == EPI_F32 || EPI == EPI_F32_RESID || EPI == EPI_F32_GELU_POS) ? 4.0 : 2.0;
    const double kt = (double)p.K * (p.ntaps > 1 ? p.ntaps : 1);
    static const bool by_shape = getenv("CCX_PROF_SHAPES") != nullptr;
    const char* label = "gemm_bf16_nt_kernel";
    if (by_shape && ctx->prof_on) {
      static std::mutex mu;
      static std::map<std::string, std::string> names;
      char buf[160];
      snprintf(buf, sizeof(buf), "gemm<epi%d,256x256> M=%d N=%d K=%d taps=%d", EPI, p.M, p.N, p.K, p.ntaps > 1 ? p.ntaps : 1);
      std::lock_guard<std::mutex> lk(mu);
      label = names.emplace(buf, buf).first->second.c_str();
    }
    ccx_prof_scope ps(ctx, stream, label, 2.0 * p.M * (double)p.N * kt,
                      2.0 * ((double)p.M * p.K + (double)p.N * kt) + obytes * p.M * (double)p.N);
    hipLaunchKernelGGL((gemm_bf16_phased_kernel<EPI>), dim3(tiles), dim3(512), LDS, stream, p);
  }
  CCX_CHECK_LAUNCH(ctx);
  return CCX_OK;
}

template <int EPI>
static int launch_epi(ccx_ctx* ctx, const GemmParams& p, hipStream_t stream) {
  // 256x256 tiles once there are enough of them to fill the 256 CUs and N is wide enough not to waste half a tile;
  // the 128x128 kernel (2 blocks per CU) otherwise.  CCX_GEMM_TILE=128|256 forces one for A/B measurements.
  static const int forced = [] { const char* e = getenv("CCX_GEMM_TILE"); return e ? atoi(e) : 0; }();
  const long big_tiles = (long)ccx_cdiv(p.M, 256) * ccx_cdiv(p.N, 256);
  bool fits = p.N >= 256 && (p.N % 256 == 0 || p.N >= 1024);
  if (EPI == EPI_HEADS) fits = fits && p.N % 256 == 0 && p.d_model % 256 == 0;
  const bool big = fits && (forced == 256 || (forced != 128 && big_tiles >= 224));
  if (big) {
    static const bool phased = [] { const char* e = getenv("CCX_GEMM_PHASED"); return e ? atoi(e) != 0 : true; }();
    if (phased) return launch_phased<EPI>(ctx, p, stream);
    return launch_epi_geo<EPI, 2, 4, 8>(ctx, p, stream);
  }
  // narrow layers (ResNet 32/64 channels, SincNet 60): 256 x 64 tiles, 4 waves of 64 rows (80 KB LDS, 2 blocks per CU).
  // Measured over the ResNet-34 convolutions: 128 x 64 / 2 waves 59.6 ms per step, 128 x 64 / 4 waves 51.6, 256 x 64 49.9.
  if (EPI != EPI_HEADS && p.N <= 64 && forced == 0) return launch_epi_geo<EPI, 4, 1, 4>(ctx, p, stream);
  return launch_epi_geo<EPI, 2, 2, 4>(ctx, p, stream);
}

int ccx_launch_gemm(ccx_ctx* ctx, int epi, const GemmParams& p, hipStream_t stream) {
  CCX_REQUIRE(ctx, p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  CCX_REQUIRE(ctx, p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
  CCX_REQUIRE(ctx, p.lda % 8 == 0 && p.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements");
  CCX_REQUIRE(ctx, p.ntaps <= 1 || (p.a_tap_stride % 8 == 0 && p.ldw >= (long)p.ntaps * p.K), "gemm: bad tap layout");
  CCX_REQUIRE(ctx, ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0, "gemm: A/W must be 16-byte aligned");
  if (epi == EPI_HEADS) {
    CCX_REQUIRE(ctx, p.d_model % 128 == 0 && p.S > 0 && p.S % 4 == 0 && p.Spad >= p.S && p.Spad % 8 == 0,
                "gemm heads: bad d_model/S/Spad");
    CCX_REQUIRE(ctx, p.N % 128 == 0, "gemm heads: N must be a multiple of 128");
  } else {
    CCX_REQUIRE(ctx, p.out != nullptr && p.ldo % 8 == 0, "gemm: out null or ldo not a multiple of 8");
    // each lane stores 16 consecutive columns; groups wholly past N are skipped
    CCX_REQUIRE(ctx, p.ldo >= (long)ccx_cdiv(p.N, 16) * 16, "gemm: ldo=%ld must cover N rounded up to 16", p.ldo);
  }
  switch (epi) {
    case EPI_BF16: return launch_epi<EPI_BF16>(ctx, p, stream);
    case EPI_BF16_GELU: return launch_epi<EPI_BF16_GELU>(ctx, p, stream);
    case EPI_BF16_RELU: return launch_epi<EPI_BF16_RELU>(ctx, p, stream);
    case EPI_F32_RESID: return launch_epi<EPI_F32_RESID>(ctx, p, stream);
    case EPI_F32: return launch_epi<EPI_F32>(ctx, p, stream);
    case EPI_HEADS: return launch_epi<EPI_HEADS>(ctx, p, stream);
    case EPI_F32_GELU_POS: return launch_epi<EPI_F32_GELU_POS>(ctx, p, stream);
    case EPI_BF16_LRELU_AFFINE: return launch_epi<EPI_BF16_LRELU_AFFINE>(ctx, p, stream);
    case EPI_BF16_ADD_RELU: return launch_epi<EPI_BF16_ADD_RELU>(ctx, p, stream);
  }
  return ccx_fail(ctx, CCX_ERR_ARG, "gemm: unknown epilogue %d", epi);
}
