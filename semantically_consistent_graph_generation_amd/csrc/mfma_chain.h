// Device helpers shared by the edge-path kernels (gfx950 / CDNA4 only).
//
// Data layout ("C layout"): all per-pair activations live in registers with the PAIR on the lane and the
// CHANNEL in the register index, exactly as v_mfma_f32_32x32x2_f32 leaves its result:
//   lane l = (column p = l & 31, half h = l >> 5);  a 32-channel block is one f32x16;
//   register r of the block holds channel  ch(r,h) = (r & 3) + 8*(r >> 2) + 4*h.
// A Linear layer  out[N] = W[N][K] . in[K]  is evaluated transposed, out^T = W . in^T, with W as the MFMA
// A operand (output channel on the lane) and the activation block as the B operand.  Because a 32x32 f32
// accumulator has its column on the lane and its rows in the registers, the result of one Linear is already
// the B operand of the next (k order permuted consistently in the packed weights), so a whole
// Linear->FiLM->Linear->LN->FFN chain never leaves the register file (cdna_hip_programming.md section 3,
// "An accumulator tile as the next MFMA's operand", here in its f32 / K=2 form).
//
// Packed weight layout (built once by pack_linear_kernel): for W[N][K] (torch layout), NB = ceil(N/32),
// KB = ceil(K/32), float4 index ((rb*KB + kb)*4 + g)*64 + lane holds, for e = 0..3,
//   W[32*rb + (lane & 31)][32*kb + 8*g + 4*(lane >> 5) + e]      (zero outside N x K)
// so that one wave-wide float4 load is 1 KiB contiguous and feeds 4 MFMAs (k-steps 4g..4g+3 of block kb).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scgg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int ch_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 x;
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = v;
  return x;
}

// ---- loads of per-channel vectors into C layout: vec points at the first channel of a 32-channel block
__device__ __forceinline__ f32x16 load_vec_block(const float* __restrict__ vec, int h) {
  f32x16 x;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 t = *reinterpret_cast<const f32x4*>(vec + 8 * g + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[4 * g + e] = t[e];
  }
  return x;
}

__device__ __forceinline__ void store_vec_block(float* __restrict__ dst, int h, const f32x16& x) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = x[4 * g + e];
    *reinterpret_cast<f32x4*>(dst + 8 * g + 4 * h) = t;
  }
}

// ---- acc += Wp[rb][kb] . x   (16 MFMAs, 4 float4 weight loads)
template <int KB>
__device__ __forceinline__ void mfma_block(f32x16& acc, const f32x4* __restrict__ wp, int rb, int kb,
                                           const f32x16& x, int lane) {
  const f32x4* w = wp + ((rb * KB + kb) * 4) * 64 + lane;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 a = w[g * 64];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], x[4 * g + e], acc, 0, 0, 0);
  }
}

// ---- fragment helpers for software-pipelined MFMA chains.  hipcc otherwise schedules  load -> vmcnt(0) -> 4 MFMAs ->
// load ...  (one fragment register quad, every MFMA group waits a full L2 / LDS round trip); loading the NEXT block's
// four fragments before the current block's 16 MFMAs, with sched_barrier(0) pinning that order, keeps one block of
// weights in flight behind 1024 cycles of matrix work.
__device__ __forceinline__ void load_frag(f32x4 (&f)[4], const f32x4* __restrict__ w, int lane) {
#pragma unroll
  for (int g = 0; g < 4; ++g) f[g] = w[g * 64 + lane];
}

__device__ __forceinline__ void mfma16(f32x16& acc, const f32x4 (&f)[4], const f32x16& x) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f[g][e], x[4 * g + e], acc, 0, 0, 0);
}

// ---- acc += Wp[rb][:] . x[:]  over all KB input blocks (pipelined: block kb+1 is loaded under block kb's MFMAs)
template <int KB>
__device__ __forceinline__ void linear_rb(f32x16& acc, const f32x4* __restrict__ wp, int rb,
                                          const f32x16 (&x)[KB], int lane) {
  const f32x4* w = wp + (rb * KB * 4) * 64;
  f32x4 cur[4], nxt[4];
  load_frag(cur, w, lane);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) load_frag(nxt, w + (kb + 1) * 256, lane);
    __builtin_amdgcn_sched_barrier(0);
    mfma16(acc, cur, x[kb]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) cur[g] = nxt[g];
  }
}

// ---- cross-lane helpers.  All reductions stay inside one 32-lane half (bit 5 of the lane id is never crossed).
// (mov_dpp: "old" operand undefined and bound_ctrl set, so no zero-initialising v_mov is emitted and the DPP combiner can
// fold the move into its consumer; every control used here reads a valid lane, the old value is never observed)
#define SCGG_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (ctrl), 0xF, 0xF, true))

__device__ __forceinline__ float swap_rows16_sum(float v) {
  // v_permlane16_swap: odd rows of vdst <-> even rows of src.  With vdst = src = v the two results are
  // [r0 r0 r2 r2] and [r1 r1 r3 r3]; their sum is the pairwise row sum in every lane.
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_rows16_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float half_sum(float v) {   // sum over the 32 lanes of this half, result in all of them
#ifdef SCGG_REDUCE_SHFL
#pragma unroll
  for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m);
  return v;
#else
  v += SCGG_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
  v += SCGG_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
  v += SCGG_DPP(v, 0x141);   // row_half_mirror
  v += SCGG_DPP(v, 0x140);   // row_mirror
  return swap_rows16_sum(v);
#endif
}

__device__ __forceinline__ float half_max(float v) {
#ifdef SCGG_REDUCE_SHFL
#pragma unroll
  for (int m = 1; m < 32; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
#else
  v = fmaxf(v, SCGG_DPP(v, 0xB1));
  v = fmaxf(v, SCGG_DPP(v, 0x4E));
  v = fmaxf(v, SCGG_DPP(v, 0x141));
  v = fmaxf(v, SCGG_DPP(v, 0x140));
  return swap_rows16_max(v);
#endif
}

__device__ __forceinline__ float other_half_sum(float v) {  // v(lane) + v(lane ^ 32)
#ifdef SCGG_REDUCE_SHFL
  return v + __shfl_xor(v, 32);
#else
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}

// ---- LayerNorm over NB*32 channels held in C layout (eps = 1e-5, biased variance, two-pass)
template <int NB>
__device__ __forceinline__ void layernorm_c(f32x16 (&x)[NB], const float* __restrict__ gamma,
                                            const float* __restrict__ beta, int h) {
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x[b][r];
  const float mean = other_half_sum(s) * (1.0f / (32 * NB));
  float q = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = x[b][r] - mean;
      q += d * d;
    }
  const float var = other_half_sum(q) * (1.0f / (32 * NB));
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const f32x16 gm = load_vec_block(gamma + 32 * b, h);
    const f32x16 bt = load_vec_block(beta + 32 * b, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) x[b][r] = (x[b][r] - mean) * rstd * gm[r] + bt[r];
  }
}

// ---- symmetrising load of a hidden-E tile:  0.5*(E[b,i,j,:] + E[b,j,i,:]) for valid off-diagonal pairs, else 0.
// Hidden E tensors are stored UNsymmetrised by their producer (DESIGN.md "Symmetrisation"); every consumer
// applies  E <- 0.5 (E + E^T) * m_i m_j (1 - I)  while loading (transformer_model.py:99,110-112; utils.py:154).
template <int NB>
__device__ __forceinline__ void load_sym_tile(f32x16 (&x)[NB], const float* __restrict__ E, long long off_ij,
                                              long long off_ji, bool live, int h) {
  if (live) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const f32x16 a = load_vec_block(E + off_ij + 32 * b, h);
      const f32x16 c = load_vec_block(E + off_ji + 32 * b, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) x[b][r] = 0.5f * (a[r] + c[r]);
    }
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) x[b] = splat16(0.f);
  }
}

}  // namespace scgg
