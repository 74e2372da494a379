// dec_ln.h -- the LayerNorm pieces every decoder kernel shares (decoder.hip, cross_x.hip)
#pragma once
#include "ccx_common.h"

// ------------------------------------------------------------------------------------------
// LayerNorm of one residual row, in pieces shared by EVERY decoder kernel that normalises a row (the LN prologue of the skinny
// linear, the stand-alone resolve + LN, the fused cross-attention query).  Every multiply-add is an explicit fmaf: hipcc has no
// contraction freedom left, so all of them round alike and a sequence's numbers do not depend on which kernel normalised its row.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ln_add_pend(float4 a, float wgt, const float4 q) {
  a.x = fmaf(wgt, q.x, a.x); a.y = fmaf(wgt, q.y, a.y); a.z = fmaf(wgt, q.z, a.z); a.w = fmaf(wgt, q.w, a.w);
  return a;
}
__device__ __forceinline__ float ln_sum4(const float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float ln_sq4(const float4 v, float mean) {
  const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
  return fmaf(d, d, fmaf(c, c, fmaf(b, b, a * a)));
}
__device__ __forceinline__ uint2 ln_pack4(const float4 v, float mean, float rstd, const float4 g, const float4 bb) {
  uint2 o;
  o.x = pack_bf16x2(fmaf((v.x - mean) * rstd, g.x, bb.x), fmaf((v.y - mean) * rstd, g.y, bb.y));
  o.y = pack_bf16x2(fmaf((v.z - mean) * rstd, g.z, bb.z), fmaf((v.w - mean) * rstd, g.w, bb.w));
  return o;
}

