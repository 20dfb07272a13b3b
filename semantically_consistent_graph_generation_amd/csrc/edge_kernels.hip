// Edge-path kernels of the GraphTransformer denoiser for gfx950 (MI355X), fp32 on v_mfma_f32_32x32x2_f32.
//
//   edge_mlp_in_kernel   mlp_in_E                        transformer_model.py:290-297, :358
//   (the fused layer kernel itself lives in edge_layer.hip)
//   edge_mlp_out_kernel  mlp_out_E + residual + diagonal mask + symmetrise + mask
//                                                        transformer_model.py:372-376, :387-389
//
// One wave owns one query row (b, i) and walks its keys j in chunks of 32 (pair on the lane, channel in the
// register: see mfma_chain.h).  Everything between the load of E[b,i,j,:] and the store of the layer's output
// for that pair happens in registers; the only cross-lane work is the per-channel softmax over j and the
// two LayerNorm reductions.
#include "mfma_chain.h"
#include "kernels.h"

namespace scgg {

// =====================================================================================================
// mlp_in_E : relu(W2 relu(W1 e + b1) + b2), written unsymmetrised for valid off-diagonal pairs, 0 elsewhere.
// =====================================================================================================
template <int KIB, int HB, int NEB>   // input blocks (ceil(dE_in/32)), hidden blocks, output blocks
__global__ __launch_bounds__(256, 2) void edge_mlp_in_kernel(EdgeMlpInArgs a) {
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B * a.n) return;
  const int b = row / a.n, i = row - b * a.n;
  const bool vi = a.mask[row] != 0;
  const int nchunks = (a.n + 31) >> 5;
  for (int jc = 0; jc < nchunks; ++jc) {
    const int j = jc * 32 + p;
    const bool inb = j < a.n;
    const bool live = vi && inb && a.mask[b * a.n + (inb ? j : 0)] != 0 && j != i;
    const long long pair = (long long)row * a.n + (inb ? j : 0);
    f32x16 out[NEB];
    if (vi) {   // wave-uniform
      f32x16 x[KIB];
      const float* src = a.E_in + pair * a.dE_in;
#pragma unroll
      for (int kb = 0; kb < KIB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = 32 * kb + ch_of(r, h);
          x[kb][r] = (live && c < a.dE_in) ? src[c] : 0.f;
        }
      f32x16 hid[HB];
#pragma unroll
      for (int rb = 0; rb < HB; ++rb) {
        hid[rb] = load_vec_block(a.b1 + 32 * rb, h);
        linear_rb<KIB>(hid[rb], a.W1, rb, x, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[rb][r] = fmaxf(hid[rb][r], 0.f);
      }
#pragma unroll
      for (int rb = 0; rb < NEB; ++rb) {
        out[rb] = load_vec_block(a.b2 + 32 * rb, h);
        linear_rb<HB>(out[rb], a.W2, rb, hid, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) out[rb][r] = live ? fmaxf(out[rb][r], 0.f) : 0.f;
      }
    } else {
#pragma unroll
      for (int rb = 0; rb < NEB; ++rb) out[rb] = splat16(0.f);
    }
    if (inb) {
      float* dst = a.E_out + pair * (32 * NEB);
#pragma unroll
      for (int rb = 0; rb < NEB; ++rb) store_vec_block(dst + 32 * rb, h, out[rb]);
    }
  }
}

// =====================================================================================================
// mlp_out_E + residual + diag mask + symmetrise + mask  ->  logits[B,n,n,C]
// =====================================================================================================
template <int NEB, int HB>
__global__ __launch_bounds__(256, 2) void edge_mlp_out_kernel(EdgeMlpOutArgs a) {
  constexpr int DE = 32 * NEB;
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B * a.n) return;
  const int n = a.n;
  const int b = row / n, i = row - b * n;
  const bool vi = a.mask[row] != 0;
  const int nchunks = (n + 31) >> 5;
  for (int jc = 0; jc < nchunks; ++jc) {
    const int j = jc * 32 + p;
    const bool inb = j < n;
    const int jj = inb ? j : 0;
    const bool live = vi && inb && a.mask[b * n + jj] != 0 && j != i;
    const long long pij = (long long)row * n + jj;
    const long long pji = ((long long)b * n + jj) * n + i;
    f32x16 o = splat16(0.f);
    if (vi) {   // wave-uniform
      f32x16 e[NEB];
      load_sym_tile<NEB>(e, a.E_hid, pij * DE, pji * DE, live, h);
      f32x16 hid[HB];
#pragma unroll
      for (int rb = 0; rb < HB; ++rb) {
        hid[rb] = load_vec_block(a.b1 + 32 * rb, h);
        linear_rb<NEB>(hid[rb], a.W1, rb, e, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[rb][r] = fmaxf(hid[rb][r], 0.f);
      }
      o = load_vec_block(a.b2, h);
      linear_rb<HB>(o, a.W2, 0, hid, lane);
      // (mlp + E_in[ij]) and (mlp + E_in[ji]) averaged: the hidden E is symmetric after the sym-load, so the
      // MLP term is bit-identical for (i,j) and (j,i); only the residual needs both orientations.
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = ch_of(r, h);
        float val = 0.f;
        if (live && c < a.C) {
          const float rij = a.E_in[pij * a.dE_in + c];
          const float rji = a.E_in[pji * a.dE_in + c];
          val = 0.5f * ((o[r] + rij) + (o[r] + rji));
        }
        o[r] = val;
      }
    }
    if (inb) {
      float* dst = a.logits + pij * a.C;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = ch_of(r, h);
        if (c < a.C) dst[c] = o[r];
      }
    }
  }
}

// =====================================================================================================
// launchers
// =====================================================================================================
static inline int row_blocks(int B, int n) { return (B * n + 3) / 4; }

int launch_edge_mlp_in(const EdgeMlpInArgs& a, int dE_in, int hid, int de, hipStream_t st) {
  const int kib = (dE_in + 31) / 32;
  dim3 g(row_blocks(a.B, a.n)), blk(256);
#define CASE(KIB, HB, NEB)                                                       \
  if (kib == KIB && hid == 32 * HB && de == 32 * NEB) {                          \
    hipLaunchKernelGGL((edge_mlp_in_kernel<KIB, HB, NEB>), g, blk, 0, st, a);    \
    return 0;                                                                    \
  }
  CASE(1, 4, 2) CASE(2, 4, 2) CASE(1, 1, 1)
#undef CASE
  return -1;
}

int launch_edge_mlp_out(const EdgeMlpOutArgs& a, int de, int hid, hipStream_t st) {
  dim3 g(row_blocks(a.B, a.n)), blk(256);
#define CASE(NEB, HB)                                                            \
  if (de == 32 * NEB && hid == 32 * HB) {                                        \
    hipLaunchKernelGGL((edge_mlp_out_kernel<NEB, HB>), g, blk, 0, st, a);        \
    return 0;                                                                    \
  }
  CASE(2, 4) CASE(1, 1)
#undef CASE
  return -1;
}

}  // namespace scgg
