// Tile machinery of the bf16x3 node-path kernels (node_layer.hip, sit_block.hip): a tile of 16 NRG rows lives in LDS as
// three bf16 term planes [rows][K + 8]; a wave owns a pair of 16-channel output blocks at a time and streams their weight
// blocks (image of launch_pack_linear_x3: [16-channel block][32-channel K chunk][term][64 lanes] x 16 B) from L2 through a
// register ring, across Linear boundaries.  See node_layer.hip for the scheme.
#pragma once
#include "mfma_chain.h"

namespace scgg {

typedef unsigned nu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned nu32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 nbf16x8 __attribute__((ext_vector_type(8)));

// sum over the 64 lanes, in every lane: four DPP steps inside the 16-lane rows, then the two row exchanges of gfx950
// (v_permlane16_swap, v_permlane32_swap).  The six ds_bpermute round trips of a __shfl_xor butterfly were the latency chain of
// the LayerNorm phases (two reductions per row, six rows per wave).
__device__ __forceinline__ float wave_sum64(float v) {
  v += SCGG_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
  v += SCGG_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
  v += SCGG_DPP(v, 0x141);   // row_half_mirror
  v += SCGG_DPP(v, 0x140);   // row_mirror
  return other_half_sum(swap_rows16_sum(v));
}

// exact three-way bf16 split of 4 consecutive channels -> 8 bytes per term
__device__ __forceinline__ void split3_4(const f32x4& x, nu32x2& hi, nu32x2& mid, nu32x2& lo) {
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const float x0 = x[2 * d], x1 = x[2 * d + 1];
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    hi[d] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    mid[d] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    lo[d] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
}
// write 4 consecutive channels of tile row `r` starting at channel c into the three planes
template <int KS, int PLANE>
__device__ __forceinline__ void put_split4(__bf16* planes, int r, int c, const f32x4& x) {
  nu32x2 h, m, l;
  split3_4(x, h, m, l);
  __bf16* d = planes + r * KS + c;
  *reinterpret_cast<nu32x2*>(d) = h;
  *reinterpret_cast<nu32x2*>(d + PLANE) = m;
  *reinterpret_cast<nu32x2*>(d + 2 * PLANE) = l;
}

struct NFrag3 { nu32x4 h, m, l; };
constexpr int NWD = 2;                                       // K chunks of weights in flight per wave
template <int D>
struct NWRingT {
  NFrag3 f[D][2];                                            // [chunk slot][output block of the pair]
  const nu32x4* next;                                        // block pair whose first D chunks the ring holds (null: none)
};
using NWRing = NWRingT<NWD>;

// slot pointer of 16-channel output block ob of an x3 image with KC 32-channel chunks per block
__device__ __forceinline__ const nu32x4* w3_block(const f32x4* w3, int ob, int KC, int lane) {
  return reinterpret_cast<const nu32x4*>(w3) + (size_t)ob * KC * 192 + lane;
}
template <int KC>
__device__ __forceinline__ void w3_load(NFrag3 (&f)[2], const nu32x4* w, int kc) {
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    f[blk].h = w[(blk * KC + kc) * 192];
    f[blk].m = w[(blk * KC + kc) * 192 + 64];
    f[blk].l = w[(blk * KC + kc) * 192 + 128];
  }
}
// start the stream of a block pair: its first NWD chunks (L2 latency ~1 us: issued ahead of whatever precedes the Linear)
template <int KC, int D>
__device__ __forceinline__ void w3_prime(NWRingT<D>& r, const nu32x4* w) {
#pragma unroll
  for (int d = 0; d < D; ++d) w3_load<KC>(r.f[d], w, d);
  r.next = w;
}

// acc[blk][rg] (C layout of the 16x16 MFMA: tile row 16 rg + (lane & 15) on the lane, output channels 4 (lane >> 4) + r
// of block blk in the registers) += W . X^T for the pair of 16-channel output blocks at w (consecutive in the image).
// The ring holds the pair's first NWD chunks on entry when r.next == w (else they are fetched here); on exit it holds
// those of `wnext` (KCN chunks per block; when not null).
template <int KC, int KCN, int KS, int PLANE, int NRG, int D>
__device__ __forceinline__ void tile_linear_x3(f32x4 (&acc)[2][NRG], NWRingT<D>& r, const nu32x4* w, const nu32x4* wnext,
                                               const __bf16* planes, int p, int g) {
  static_assert(KC >= D && KCN >= D, "K too small for the weight ring");
  static_assert(KC % D == 0, "chunk j of the next pair has to land in ring slot j % D");
  const __bf16* x = planes + p * KS + 8 * g;
  NFrag3 xf[2][NRG];
  auto xload = [&](NFrag3 (&f)[NRG], int kc) {
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg) {
      const __bf16* q = x + rg * 16 * KS + 32 * kc;
      f[rg].h = *reinterpret_cast<const nu32x4*>(q);
      f[rg].m = *reinterpret_cast<const nu32x4*>(q + PLANE);
      f[rg].l = *reinterpret_cast<const nu32x4*>(q + 2 * PLANE);
    }
  };
  if (r.next != w) w3_prime<KC>(r, w);                       // wave-uniform
  xload(xf[0], 0);
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    if (kc + 1 < KC) xload(xf[(kc + 1) & 1], kc + 1);
    __builtin_amdgcn_sched_barrier(0);
#define SCGG_NB(v) __builtin_bit_cast(nbf16x8, v)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const NFrag3& a = r.f[kc % D][blk];
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) {
        const NFrag3& b = xf[kc & 1][rg];
        f32x4 c = acc[blk][rg];
#ifdef SCGG_ABL_TILE_NO_MFMA   // timing experiment: the operands stay live, the matrix pipe stays idle
        c[0] += __uint_as_float((a.l[0] ^ b.h[0]) & 0x3f800000u);
        c[1] += __uint_as_float((a.m[1] ^ b.m[1]) & 0x3f800000u);
        c[2] += __uint_as_float((a.h[2] ^ b.l[2]) & 0x3f800000u);
        acc[blk][rg] = c;
        continue;
#endif
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_NB(a.l), SCGG_NB(b.h), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_NB(a.h), SCGG_NB(b.l), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_NB(a.m), SCGG_NB(b.m), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_NB(a.m), SCGG_NB(b.h), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_NB(a.h), SCGG_NB(b.m), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_NB(a.h), SCGG_NB(b.h), c, 0, 0, 0);
        acc[blk][rg] = c;
      }
    }
#undef SCGG_NB
    __builtin_amdgcn_sched_barrier(0);
#ifndef SCGG_ABL_TILE_NO_W     // timing experiment: no weight stream beyond the first D chunks of a pair
    if (kc + D < KC) w3_load<KC>(r.f[kc % D], w, kc + D);
    else if (wnext != nullptr) w3_load<KCN>(r.f[kc % D], wnext, kc + D - KC);
#endif
  }
  r.next = wnext;
}

}  // namespace scgg
