// fp32-accurate products on the bf16 matrix pipe ("x3"): shared by the fused edge-layer kernel and the edge MLP kernels.
//
// Every fp32 operand is split EXACTLY into three bf16 terms by truncation (x = h + m + l, 8 mantissa bits each,
// |x - h - m - l| <= 2^-24 |x|) and a product is evaluated as the six leading partial products
// hl' + lh' + mm' + mh' + hm' + hh' on v_mfma_f32_16x16x32_bf16 (bf16 x bf16 is exact in the fp32 accumulator): the
// dropped terms are below 2^-24 relative.  Operand form: lane (key / output row l & 15, group g = l >> 4) supplies the 8
// k-slots [C16 block 0 r = 0..3 | C16 block 1 r = 0..3] of a 32-channel chunk.
#pragma once
#include "mfma_chain.h"

namespace scgg {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// exact three-way bf16 split of 8 fp32 values (two C16 blocks of one lane) into MFMA operand form
__device__ __forceinline__ void split3(const f32x4& b0, const f32x4& b1, u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float x0 = d < 2 ? b0[2 * d] : b1[2 * d - 4], x1 = d < 2 ? b0[2 * d + 1] : b1[2 * d - 3];
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    hi[d] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                 // [bf16(x1) : bf16(x0)], truncating
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    mid[d] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    lo[d] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
}

// the inverse: h + m + l is the value split3 was given, bit for bit (the split is exact, and each partial sum is
// representable: m + l is the remainder of x after its 8 leading bits, h + that is x)
__device__ __forceinline__ void unsplit3(const u32x4& hi, const u32x4& mid, const u32x4& lo, f32x4& b0, f32x4& b1) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float x0 = __uint_as_float(hi[d] << 16) + (__uint_as_float(mid[d] << 16) + __uint_as_float(lo[d] << 16));
    const float x1 = __uint_as_float(hi[d] & 0xffff0000u) + (__uint_as_float(mid[d] & 0xffff0000u) + __uint_as_float(lo[d] & 0xffff0000u));
    if (d < 2) { b0[2 * d] = x0; b0[2 * d + 1] = x1; }
    else { b1[2 * d - 4] = x0; b1[2 * d - 3] = x1; }
  }
}

struct Frag3 { u32x4 h, m, l; };   // one weight block: the three bf16 terms of this lane's 8 k-slots

__device__ __forceinline__ void load_frag3(Frag3& f, const f32x4* w, int lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(w);
  f.h = p[lane];
  f.m = p[64 + lane];
  f.l = p[128 + lane];
}

// acc += W . x with fp32 accuracy: the six leading partial products, smallest first
__device__ __forceinline__ void mfma_x3(f32x4& acc, const Frag3& w, const u32x4& xh, const u32x4& xm, const u32x4& xl) {
#ifdef SCGG_ABL_NO_MFMA
  acc[0] += __uint_as_float(w.h[0] ^ xh[0] ^ w.m[1] ^ xm[1] ^ w.l[2] ^ xl[2]);
  return;
#endif
#define SCGG_BF(v) __builtin_bit_cast(bf16x8, v)
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(w.l), SCGG_BF(xh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(w.h), SCGG_BF(xl), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(w.m), SCGG_BF(xm), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(w.m), SCGG_BF(xh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(w.h), SCGG_BF(xm), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(w.h), SCGG_BF(xh), acc, 0, 0, 0);
#undef SCGG_BF
}

// NB weight blocks starting at W (192 float4 slots each); block t+1 is read from LDS under the MFMAs of block t.  P0 says
// which register set holds block 0 on entry (0: fa, 1: fb); on exit the set (P0 + NB) & 1 holds the block at `after`.
template <int NB, int P0, class Body>
__device__ __forceinline__ void block_pipeline(const f32x4* W, const f32x4* after, Frag3& fa, Frag3& fb, int lane, Body body) {
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    Frag3& cur = ((t + P0) & 1) ? fb : fa;
    Frag3& nxt = ((t + P0) & 1) ? fa : fb;
    if (t + 1 < NB) load_frag3(nxt, W + (t + 1) * 192, lane);
    else if (after != nullptr) load_frag3(nxt, after, lane);
    __builtin_amdgcn_sched_barrier(0);
    body(t, cur);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same with the weight blocks read TWO blocks ahead (three register sets): for a wave that runs its matrix work
// alone on its SIMD (edge_unit_layer_x3s_kernel) one block of cover (96 pipe cycles) is less than an LDS round trip under
// load, and every block then starts with a stall.  On entry nothing is loaded.
template <int NB, class Body>
__device__ __forceinline__ void block_pipeline_deep(const f32x4* W, int lane, Body body) {
  Frag3 f[3];
  load_frag3(f[0], W, lane);
  if (NB > 1) load_frag3(f[1], W + 192, lane);
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    if (t + 2 < NB) load_frag3(f[(t + 2) % 3], W + (t + 2) * 192, lane);
    __builtin_amdgcn_sched_barrier(0);
    body(t, f[t % 3]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace scgg
