// C16-layout device helpers shared by the 16-key-unit edge kernels (edge_unit_layer.hip, block3d.hip): key on lane & 15,
// channel group g = lane >> 4, a 16-channel block of activations is one f32x4 whose register r holds channel 4g + r.
#pragma once
#include "mfma_chain.h"

namespace scgg {

// ---- 16-lane row reductions (DPP only; a DPP row is exactly one channel group)
__device__ __forceinline__ float row_sum16(float v) {
  v += SCGG_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
  v += SCGG_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
  v += SCGG_DPP(v, 0x141);   // row_half_mirror
  v += SCGG_DPP(v, 0x140);   // row_mirror
  return v;
}
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, SCGG_DPP(v, 0xB1));
  v = fmaxf(v, SCGG_DPP(v, 0x4E));
  v = fmaxf(v, SCGG_DPP(v, 0x141));
  v = fmaxf(v, SCGG_DPP(v, 0x140));
  return v;
}
// the same maximum as one fused v_max_f32_dpp per step (hipcc expands fmaxf(x, dpp(x)) into mov_dpp + two canonicalising
// maxes); the leading s_nop covers the VALU-write -> DPP-read hazard, which the compiler cannot see inside an asm
#define SCGG_DPP_MAX_STEP(v, ctrl_text) \
  asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " ctrl_text " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "v"(v))
__device__ __forceinline__ float row_max16_fused(float v) {
  SCGG_DPP_MAX_STEP(v, "quad_perm:[1,0,3,2]");
  SCGG_DPP_MAX_STEP(v, "quad_perm:[2,3,0,1]");
  SCGG_DPP_MAX_STEP(v, "row_half_mirror");
  SCGG_DPP_MAX_STEP(v, "row_mirror");
  return v;
}
// N independent 16-lane sums, step-major: consecutive DPP instructions never read the register the previous one wrote,
// so no hazard nops are needed between them (a dependent DPP chain pays 2 wait states per step)
template <int N>
__device__ __forceinline__ void row_sum16_many(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += SCGG_DPP(v[i], 0xB1);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += SCGG_DPP(v[i], 0x4E);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += SCGG_DPP(v[i], 0x141);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += SCGG_DPP(v[i], 0x140);
}
// 16 independent 16-lane sums as a REDUCE-SCATTER: lane p of every row ends up with sum over the row's lanes of v[p].
// Four halving stages (partners: row_mirror, row_half_mirror, quad xor 2, quad xor 1): a lane keeps the half of its
// values that its lane bit selects and adds the partner's copy of them, so the work is 8 + 4 + 2 + 1 sums instead of the
// 16 x 4 of sixteen butterflies.  The first two stages select by DPP bank mask (a bank = 4 lanes: bit 3 / bit 2 of the lane
// pick whole banks), two v_add_f32_dpp per result; the last two select by v_cndmask.  33 VALU instead of 64-80.
__device__ __forceinline__ float row_reduce_scatter16(const float (&v)[16], int p) {
  float a[8], b[4];
  // stage 1: lanes 0-7 (banks 0,1) keep v[0..7], lanes 8-15 keep v[8..15]; partner = 15 - p
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %8, %8 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %0, %16, %16 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %1, %9, %9 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %1, %17, %17 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %10, %10 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %2, %18, %18 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %3, %11, %11 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %3, %19, %19 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %4, %12, %12 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %4, %20, %20 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %5, %13, %13 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %5, %21, %21 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %6, %14, %14 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %6, %22, %22 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %7, %15, %15 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %7, %23, %23 row_mirror row_mask:0xf bank_mask:0xc"
      : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
        "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
  // stage 2: bit 2 of the lane = banks 1 and 3 keep the upper four; partner = 7 - (p & 7) within the half row
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %1, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %2, %6, %6 row_half_mirror row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %2, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %3, %7, %7 row_half_mirror row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %3, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa"
      : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
  // stage 3: bit 1 (partner p ^ 2), stage 4: bit 0 (partner p ^ 1)
  const bool h2 = (p & 2) != 0, h1 = (p & 1) != 0;
  float c[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float lo = b[i] + SCGG_DPP(b[i], 0x4E), hi = b[i + 2] + SCGG_DPP(b[i + 2], 0x4E);   // quad_perm [2,3,0,1]
    c[i] = h2 ? hi : lo;
  }
  const float lo = c[0] + SCGG_DPP(c[0], 0xB1), hi = c[1] + SCGG_DPP(c[1], 0xB1);             // quad_perm [1,0,3,2]
  return h1 ? hi : lo;
}

// across the four channel groups (same key): lane ^ 16 then lane ^ 32
__device__ __forceinline__ float groups_sum(float v) {
  v = swap_rows16_sum(v);
  return other_half_sum(v);
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 splat4(float v) { f32x4 x = {v, v, v, v}; return x; }

// two independent accumulators, same input block, MFMAs interleaved: v_mfma_f32_16x16x4_f32 issues every 32 cycles but
// a dependent accumulation needs 40 (MI355X_MICROARCH.md cycle constants), so a single chain would idle the pipe 20 %
__device__ __forceinline__ void mfma4x2(f32x4& acc0, f32x4& acc1, const f32x4& f0, const f32x4& f1, const f32x4& x) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0[r], x[r], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1[r], x[r], acc1, 0, 0, 0);
  }
}

// LayerNorm over IBE*16 channels in C16 layout (eps = 1e-5, biased variance, two-pass)
template <int IBE>
__device__ __forceinline__ void layernorm_c16(f32x4 (&x)[IBE], const float* __restrict__ gamma,
                                              const float* __restrict__ beta, int g) {
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < IBE; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += x[b][r];
  const float mean = groups_sum(s) * (1.0f / (16 * IBE));
  float q = 0.f;
#pragma unroll
  for (int b = 0; b < IBE; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = x[b][r] - mean;
      q += d * d;
    }
  const float var = groups_sum(q) * (1.0f / (16 * IBE));
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int b = 0; b < IBE; ++b) {
    const f32x4 gm = ld4(gamma + 16 * b + 4 * g);
    const f32x4 bt = ld4(beta + 16 * b + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[b][r] = (x[b][r] - mean) * rstd * gm[r] + bt[r];
  }
}

}  // namespace scgg
