// Position-aware NodeEdgeBlock + XEy E path of Graph3DTransformer, gfx950, fp32 on
// v_mfma_f32_16x16x4_f32, 16-key units in C16 layout (c16.h).
//
//   NodeEdgeBlock.forward          src/ConStruct/models/pos_transformer_model.py:106-209
//   XEyTransformerLayer E path     src/ConStruct/models/pos_transformer_model.py:278-290
//   EtoX / Etoy pooling            src/ConStruct/models/layers.py:55-71, :24-43
//   PlaceHolder.mask on pos        src/ConStruct/utils.py:160-164
//
// Unlike the 2D block (edge_unit_layer.hip) every Linear of this block is de x de or de x ffE, so the whole layer's edge
// weights fit LDS at once in fp32 (4 de^2 + 2 de ffE + 2*16 de + 16 dx floats = 148 KB at the shipped vessel dims): a
// persistent workgroup stages them once and its waves then stream units with no barrier and no weight traffic at all.
//
// Per unit (valid query row i, 16 keys j), all in registers:
//   x      = 0.5 (E_ij + E_ji) off the diagonal, 0 on it                      (sym-load of the producer's tile)
//   dist1  = relu(w_d |p_i - p_j| + w_c cos(p_i, p_j) + w_1 |p_i| + w_2 |p_j| + b)               :122-132
//   Y      = in_E(x) . x_e_mul1_j . x_e_mul2_i;  Y = Y + dist_add(dist1) + Y dist_mul(dist1)     :137-147
//   E1     = Y + y_e_add + Y y_e_mul;  newE = e_out(E1)                                          :150-155
//   a^T    = K_j . (W_a (.) Q_i) / sqrt(de), times (1 + e_att_mul(E1)) (1 + pos_att_mul(dist1))  :159-171
//            evaluated TRANSPOSED (keys as MFMA rows): lane (head, key group) holds 4 keys, which is the B operand of
//            the weighted-V product  wV^T[d][h] = sum_key V[key][d] p[key][h]  without any lane movement
//   E_out  = LN2(E' + linE2 relu linE1 E'),  E' = LN1(x + newE)                                  :278-288
// and per unit the partial results the row / graph merges need: softmax (max, sum, sum p V) per head, and per channel
// (mean, M2, min, max) of Y and of dist1 over the unit's valid keys (exact two-pass inside the unit, Chan merge later).
#include <stdlib.h>

#include <type_traits>

#include "pool_stats.h"
#include "mfma_chain.h"
#include "c16.h"
#include "kernels.h"
#include "x3_mfma.h"

namespace scgg {

template <int DEB, int FFB, int DXB>
struct Geom3D {
  static constexpr int DE = 16 * DEB, FFE = 16 * FFB, DX = 16 * DXB;
  static constexpr int SQ = DEB * DEB * 64;                       // float4s of a de x de fragment set
  static constexpr int W_IN = 0, W_DADD = SQ, W_DMUL = 2 * SQ, W_EOUT = 3 * SQ;
  static constexpr int W_FF1 = 4 * SQ;                            // (hidden block, input block)
  static constexpr int W_FF2 = W_FF1 + FFB * DEB * 64;            // (output block, hidden block)
  static constexpr int W_EATT = W_FF2 + DEB * FFB * 64;           // 16 rows (heads, zero padded) x de
  static constexpr int W_PATT = W_EATT + DEB * 64;
  static constexpr int W_A = W_PATT + DEB * 64;                   // 16 rows (heads) x dx
  static constexpr int VEC = W_A + DXB * 64;
  enum { B_IN = 0, B_DADD, B_DMUL, B_EOUT, B_FF2, LN1G, LN1B, LN2G, LN2B, WD0, WD1, BD, WN1, WN2, NVEC_DE };
  static constexpr int V_FF1 = NVEC_DE * DE;
  static constexpr int V_EATT = V_FF1 + FFE;
  static constexpr int V_PATT = V_EATT + 16;
  static constexpr int V_FLOATS = V_PATT + 16;
  static constexpr int IMG_F4 = VEC + (V_FLOATS + 3) / 4;
  static_assert(DEB % 2 == 0 && FFB % 4 == 0, "de: multiple of 32, dim_ffE: multiple of 64");
};


// ---- default form: the same layer as TWO kernels on the bf16 matrix pipe with fp32 accuracy (x3_mfma.h).  As three bf16
// terms the layer's weights are 1.5 x their fp32 size and no longer fit LDS together, but the block part (in_E,
// dist_add_e | dist_mul_e, e_out: 96 KB, plus the fp32 attention heads) and the FFN part (linE1, linE2: 96 KB) each do:
//   block3d_edge_kernel<.., X3 = true>   everything up to E'' = x + newE (stored), attention and statistics partials
//   block3d_ffn_kernel                   E_out = LN2(E' + linE2 relu linE1 E'), E' = LN1(E''), in place on the stored tile
// Both keep their weights resident (no phases, no barriers in the unit loop); the price is one extra write + read of the
// tile (512 B per pair), which a compute-bound kernel hides.  Weight blocks: (16 output rows, 32-channel K chunk) x three
// terms x 64 lanes x 16 B (192 float4 slots), ordered as the pipelines walk them.
template <int DEB, int DXB>
struct Geom3DA {
  static constexpr int DE = 16 * DEB, DX = 16 * DXB, Q = DEB / 2;
  static constexpr int NBLK = DEB * Q;                            // blocks of a de x de matrix
  static constexpr int W_IN = 0;                                  // (ob, q)
  static constexpr int W_DIST = NBLK * 192;                       // (ob, {dist_add_e, dist_mul_e}, q)
  static constexpr int W_EOUT = 3 * NBLK * 192;                   // (ob, q)
  static constexpr int W_EATT = 4 * NBLK * 192;                   // e_att_mul: Q blocks (16 rows = heads, zero padded)
  static constexpr int W_PATT = W_EATT + Q * 192;                 // pos_att_mul
  static constexpr int W_A = W_PATT + Q * 192;                    // fp32 fragments of `a` (16 heads x dx), as in Geom3D
  static constexpr int VEC = W_A + DXB * 64;
  enum { B_IN = 0, B_DADD, B_DMUL, B_EOUT, WD0, WD1, BD, WN1, WN2, NVEC_DE };
  static constexpr int V_EATT = NVEC_DE * DE;
  static constexpr int V_PATT = V_EATT + 16;
  static constexpr int V_FLOATS = V_PATT + 16;
  static constexpr int IMG_F4 = VEC + (V_FLOATS + 3) / 4;
};
template <int DEB, int FFB>
struct Geom3DF {
  static constexpr int DE = 16 * DEB, FFE = 16 * FFB, Q1 = DEB / 2, Q2 = FFB / 2;
  static constexpr int W_1 = 0;                                   // linE1 (hidden block, q)
  static constexpr int W_2 = FFB * Q1 * 192;                      // linE2 (hidden chunk q2, output block): a chunk is split once
  static constexpr int VEC = W_2 + DEB * Q2 * 192;
  enum { B_FF2 = 0, LN1G, LN1B, LN2G, LN2B, NVEC_DE };
  static constexpr int V_FF1 = NVEC_DE * DE;
  static constexpr int V_FLOATS = V_FF1 + FFE;
  static constexpr int IMG_F4 = VEC + (V_FLOATS + 3) / 4;
  static_assert((FFB * Q1) % 2 == 0 && (DEB * Q2) % 2 == 0, "even block counts");
};

// one 16-byte slot of a three-term bf16 weight block: rows 16 ob + (l & 15), channels 32 q + {4g..4g+3, 16+4g..16+4g+3}
__device__ __forceinline__ f32x4 x3_slot(const float* W, int N, int K, int ob, int q, int term, int l) {
  const int g = l >> 4, row = 16 * ob + (l & 15);
  unsigned hw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int col = 32 * q + (i < 4 ? 4 * g + i : 16 + 4 * g + i - 4);
    float x = (row < N && col < K) ? W[(size_t)row * K + col] : 0.f;
    for (int k = 0; k < term; ++k) x = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    hw[i] = __float_as_uint(x) >> 16;
  }
  f32x4 out;
#pragma unroll
  for (int d = 0; d < 4; ++d) out[d] = __uint_as_float(hw[2 * d] | (hw[2 * d + 1] << 16));
  return out;
}

template <int DEB, int FFB, int DXB>
__global__ __launch_bounds__(256) void block3d_pack_x3_kernel(Block3DPackArgs a) {
  using GA = Geom3DA<DEB, DXB>;
  using GF = Geom3DF<DEB, FFB>;
  constexpr int DE = GA::DE, DX = GA::DX, FFE = GF::FFE, Q = GA::Q;
  for (int idx0 = blockIdx.x * 256 + threadIdx.x; idx0 < GA::IMG_F4 + GF::IMG_F4; idx0 += gridDim.x * 256) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (idx0 < GA::IMG_F4) {
      const int idx = idx0;
      if (idx < GA::W_EATT) {
        const int blk = idx / 192, term = (idx % 192) >> 6, l = idx & 63;
        if (blk < GA::NBLK) v = x3_slot(a.W_in, DE, DE, blk / Q, blk % Q, term, l);
        else if (blk < 3 * GA::NBLK) {
          const int t = blk - GA::NBLK;
          v = x3_slot(((t / Q) & 1) ? a.W_dmul : a.W_dadd, DE, DE, t / (2 * Q), t % Q, term, l);
        } else {
          const int t = blk - 3 * GA::NBLK;
          v = x3_slot(a.W_eout, DE, DE, t / Q, t % Q, term, l);
        }
      } else if (idx < GA::W_A) {
        const int i2 = idx - GA::W_EATT, blk = i2 / 192, term = (i2 % 192) >> 6, l = i2 & 63;
        v = x3_slot(blk < Q ? a.W_eatt : a.W_patt, a.n_head, DE, 0, blk % Q, term, l);
      } else if (idx < GA::VEC) {
        const int t = (idx - GA::W_A) >> 6, l = idx & 63, row = l & 15;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = 16 * t + 4 * (l >> 4) + e;
          if (row < a.n_head && col < DX) v[e] = a.W_a[(size_t)row * DX + col];
        }
      } else {
        const int f0 = (idx - GA::VEC) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = f0 + e;
          float x = 0.f;
          if (f < GA::V_EATT) {
            const int which = f / DE, c = f - which * DE;
            switch (which) {
              case GA::B_IN: x = a.b_in[c]; break;
              case GA::B_DADD: x = a.b_dadd[c]; break;
              case GA::B_DMUL: x = a.b_dmul[c]; break;
              case GA::B_EOUT: x = a.b_eout[c]; break;
              case GA::WD0: x = a.W_dist[2 * c]; break;
              case GA::WD1: x = a.W_dist[2 * c + 1]; break;
              case GA::BD: x = (a.b_dist[c] + a.b_n1[c]) + a.b_n2[c]; break;
              case GA::WN1: x = a.W_n1[c]; break;
              case GA::WN2: x = a.W_n2[c]; break;
            }
          } else if (f < GA::V_PATT) x = (f - GA::V_EATT) < a.n_head ? a.b_eatt[f - GA::V_EATT] : 0.f;
          else if (f < GA::V_FLOATS) x = (f - GA::V_PATT) < a.n_head ? a.b_patt[f - GA::V_PATT] : 0.f;
          v[e] = x;
        }
      }
    } else {
      const int idx = idx0 - GA::IMG_F4;
      if (idx < GF::VEC) {
        const int blk = idx / 192, term = (idx % 192) >> 6, l = idx & 63;
        if (blk < FFB * GF::Q1) v = x3_slot(a.W_ff1, FFE, DE, blk / GF::Q1, blk % GF::Q1, term, l);
        else {
          const int t = blk - FFB * GF::Q1;
          v = x3_slot(a.W_ff2, DE, FFE, t % DEB, t / DEB, term, l);
        }
      } else {
        const int f0 = (idx - GF::VEC) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = f0 + e;
          float x = 0.f;
          if (f < GF::V_FF1) {
            const int which = f / DE, c = f - which * DE;
            switch (which) {
              case GF::B_FF2: x = a.b_ff2[c]; break;
              case GF::LN1G: x = a.ln1g[c]; break;
              case GF::LN1B: x = a.ln1b[c]; break;
              case GF::LN2G: x = a.ln2g[c]; break;
              case GF::LN2B: x = a.ln2b[c]; break;
            }
          } else if (f < GF::V_FLOATS) x = a.b_ff1[f - GF::V_FF1];
          v[e] = x;
        }
      }
    }
    a.img[idx0] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// LDS image.  Fragment (ob, ib) of W[N][K], lane l: W[16 ob + (l & 15)][16 ib + 4 (l >> 4) + 0..3], zero outside.
// ---------------------------------------------------------------------------------------------------------
template <int DEB, int FFB, int DXB>
__global__ __launch_bounds__(256) void block3d_pack_kernel(Block3DPackArgs a) {
  using G = Geom3D<DEB, FFB, DXB>;
  constexpr int DE = G::DE, FFE = G::FFE, DX = G::DX;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < G::IMG_F4; idx += gridDim.x * 256) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (idx < G::VEC) {
      const float* W;
      int N, K, KBLK, off;
      if (idx < G::W_FF1) {
        const int m = idx / G::SQ;
        W = m == 0 ? a.W_in : m == 1 ? a.W_dadd : m == 2 ? a.W_dmul : a.W_eout;
        N = DE; K = DE; KBLK = DEB; off = m * G::SQ;
      } else if (idx < G::W_FF2) { W = a.W_ff1; N = FFE; K = DE; KBLK = DEB; off = G::W_FF1; }
      else if (idx < G::W_EATT) { W = a.W_ff2; N = DE; K = FFE; KBLK = FFB; off = G::W_FF2; }
      else if (idx < G::W_PATT) { W = a.W_eatt; N = a.n_head; K = DE; KBLK = DEB; off = G::W_EATT; }
      else if (idx < G::W_A) { W = a.W_patt; N = a.n_head; K = DE; KBLK = DEB; off = G::W_PATT; }
      else { W = a.W_a; N = a.n_head; K = DX; KBLK = DXB; off = G::W_A; }
      const int t = (idx - off) >> 6, l = idx & 63;
      const int ob = t / KBLK, ib = t - ob * KBLK;
      const int row = 16 * ob + (l & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 16 * ib + 4 * (l >> 4) + e;
        if (row < N && col < K) v[e] = W[(size_t)row * K + col];
      }
    } else {
      const int f0 = (idx - G::VEC) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = f0 + e;
        float x = 0.f;
        if (f < G::V_FF1) {
          const int which = f / DE, c = f - which * DE;
          switch (which) {
            case G::B_IN: x = a.b_in[c]; break;
            case G::B_DADD: x = a.b_dadd[c]; break;
            case G::B_DMUL: x = a.b_dmul[c]; break;
            case G::B_EOUT: x = a.b_eout[c]; break;
            case G::B_FF2: x = a.b_ff2[c]; break;
            case G::LN1G: x = a.ln1g[c]; break;
            case G::LN1B: x = a.ln1b[c]; break;
            case G::LN2G: x = a.ln2g[c]; break;
            case G::LN2B: x = a.ln2b[c]; break;
            case G::WD0: x = a.W_dist[2 * c]; break;          // lin_dist1.weight [de, 2]: distance column
            case G::WD1: x = a.W_dist[2 * c + 1]; break;      //                           cosine column
            case G::BD: x = (a.b_dist[c] + a.b_n1[c]) + a.b_n2[c]; break;
            case G::WN1: x = a.W_n1[c]; break;                // lin_norm_pos1.weight [de, 1] (follows the query row i)
            case G::WN2: x = a.W_n2[c]; break;                // lin_norm_pos2 (follows the key j)
          }
        } else if (f < G::V_EATT) x = a.b_ff1[f - G::V_FF1];
        else if (f < G::V_PATT) x = (f - G::V_EATT) < a.n_head ? a.b_eatt[f - G::V_EATT] : 0.f;
        else if (f < G::V_FLOATS) x = (f - G::V_PATT) < a.n_head ? a.b_patt[f - G::V_PATT] : 0.f;
        v[e] = x;
      }
    }
    a.img[idx] = v;
  }
}

#define SCGG_DPP_MIN_STEP(v, ctrl_text) \
  asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 " ctrl_text " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "v"(v))
__device__ __forceinline__ float row_min16_fused(float v) {   // as row_max16_fused (c16.h): one v_min_f32_dpp per step
  SCGG_DPP_MIN_STEP(v, "quad_perm:[1,0,3,2]");
  SCGG_DPP_MIN_STEP(v, "quad_perm:[2,3,0,1]");
  SCGG_DPP_MIN_STEP(v, "row_half_mirror");
  SCGG_DPP_MIN_STEP(v, "row_mirror");
  return v;
}
__device__ __forceinline__ float groups_max(float v) {          // across the four lane rows (same lane & 15)
  v = swap_rows16_max(v);
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// acc += x . W^T with the ACTIVATION as the A operand (keys as rows) and the weight block as B: the same six partial
// products as mfma_x3, roles swapped
__device__ __forceinline__ void mfma_x3_t(f32x4& acc, const u32x4& xh, const u32x4& xm, const u32x4& xl, const Frag3& w) {
#define SCGG_BF(v) __builtin_bit_cast(bf16x8, v)
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(xh), SCGG_BF(w.l), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(xl), SCGG_BF(w.h), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(xm), SCGG_BF(w.m), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(xh), SCGG_BF(w.m), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(xm), SCGG_BF(w.h), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(xh), SCGG_BF(w.h), acc, 0, 0, 0);
#undef SCGG_BF
}
// 16-lane minima of mn[0..3] and maxima of mx[0..3], step-major inside one asm block per step: an instruction never reads
// a register written by one of the seven before it, so one s_nop per step (for whatever the compiler placed in front)
// replaces the one per instruction of row_min16_fused / row_max16_fused
#define SCGG_MINMAX_STEP(ctrl)                                                                                  \
  asm("s_nop 1\n\t"                                                                                             \
      "v_min_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
      "v_min_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
      "v_min_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
      "v_min_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
      "v_max_f32_dpp %4, %4, %4 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
      "v_max_f32_dpp %5, %5, %5 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
      "v_max_f32_dpp %6, %6, %6 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
      "v_max_f32_dpp %7, %7, %7 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1"                                 \
      : "+v"(mn[0]), "+v"(mn[1]), "+v"(mn[2]), "+v"(mn[3]), "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]), "+v"(mx[3]))
__device__ __forceinline__ void row_minmax16_x4(f32x4& mn, f32x4& mx) {
  SCGG_MINMAX_STEP("quad_perm:[1,0,3,2]");
  SCGG_MINMAX_STEP("quad_perm:[2,3,0,1]");
  SCGG_MINMAX_STEP("row_half_mirror");
  SCGG_MINMAX_STEP("row_mirror");
}
// two accumulators, two fragments, two inputs
__device__ __forceinline__ void mfma4x2d(f32x4& acc0, f32x4& acc1, const f32x4& f0, const f32x4& f1, const f32x4& x0,
                                         const f32x4& x1) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0[r], x0[r], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1[r], x1[r], acc1, 0, 0, 0);
  }
}

template <int DEB, int FFB, int DXB, int NW, bool X3>
__global__ __launch_bounds__(NW * 64) void block3d_edge_kernel(Block3DArgs a) {
  using G = typename std::conditional<X3, Geom3DA<DEB, DXB>, Geom3D<DEB, FFB, DXB>>::type;
  constexpr int DE = G::DE, DX = G::DX;
  constexpr int Q = DEB / 2;                                           // 32-channel K chunks of the E width (x3 form)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* img = reinterpret_cast<f32x4*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  for (int idx = tid; idx < G::IMG_F4; idx += NW * 64) img[idx] = a.img[idx];
  __syncthreads();
  const float* vec = reinterpret_cast<const float*>(img + G::VEC);
  const int n = a.n, nch = (n + 15) >> 4;
  const int count = *a.unit_count;
  const int AP = a.n_head * DX + 32;

  for (int uidx = blockIdx.x * NW + wave; uidx < count; uidx += gridDim.x * NW) {
    const int u = a.units[uidx];
    const int row = u / nch, jc = u - row * nch;
    const int b = row / n, i = row - b * n;
    const int j = jc * 16 + p;
    const bool inb = j < n;
    const int jj = inb ? j : 0;
    const long long rj = (long long)b * n + jj;
    const bool vj = inb && a.mask[rj] != 0;                            // the query row is valid by construction
    const bool offd = vj && j != i;
    const unsigned keymask = (unsigned)(__ballot(vj) & 0xffffull);     // lanes 0..15 carry keys 0..15 of the unit
    const long long pij = ((long long)row * n + jj) * DE, pji = (rj * n + i) * DE;
    const float* nl_i = a.nodelin + (long long)row * a.ldn;
    const float* nl_j = a.nodelin + rj * a.ldn;

    f32x4 x[DEB];
#pragma unroll
    for (int ib = 0; ib < DEB; ++ib) {
      x[ib] = splat4(0.f);
      if (offd) x[ib] = 0.5f * (ld4(a.E_in + pij + 16 * ib + 4 * g) + ld4(a.E_in + pji + 16 * ib + 4 * g));
    }
    // ---- distance features
    f32x4 d1[DEB];
    {
      const f32x4 pi = ld4(a.pos4 + (long long)row * 4), pj = ld4(a.pos4 + rj * 4);
      const float ex = pi[0] - pj[0], ey = pi[1] - pj[1], ez = pi[2] - pj[2];
      const float dist = sqrtf(ex * ex + ey * ey + ez * ez);
      const float di = pi[3] + 1e-7f, dj = pj[3] + 1e-7f;
      const float cs = (pi[0] / di) * (pj[0] / dj) + (pi[1] / di) * (pj[1] / dj) + (pi[2] / di) * (pj[2] / dj);
#pragma unroll
      for (int ib = 0; ib < DEB; ++ib) {
        const int c = 16 * ib + 4 * g;
        const f32x4 w0 = ld4(vec + G::WD0 * DE + c), w1 = ld4(vec + G::WD1 * DE + c), bd = ld4(vec + G::BD * DE + c);
        const f32x4 n1 = ld4(vec + G::WN1 * DE + c), n2 = ld4(vec + G::WN2 * DE + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = dist * w0[r] + cs * w1[r] + bd[r] + n1[r] * pi[3] + n2[r] * pj[3];
          d1[ib][r] = vj ? fmaxf(t, 0.f) : 0.f;
        }
      }
    }
    // ---- Y = in_E(x) . x_e_mul1_j . x_e_mul2_i, then the distance FiLM
    f32x4 Y[DEB];
    Frag3 fa, fb;
    u32x4 eh[Q], em[Q], el[Q];     // x3 form: bf16 terms of E1 (operands of e_out and of the attention multiplier e_att_mul)
    if constexpr (X3) {
      load_frag3(fa, img + Geom3DA<DEB, DXB>::W_IN, lane);
      u32x4 xh[Q], xm[Q], xl[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) split3(x[2 * q], x[2 * q + 1], xh[q], xm[q], xl[q]);
#pragma unroll
      for (int ob = 0; ob < DEB; ++ob) Y[ob] = ld4(vec + G::B_IN * DE + 16 * ob + 4 * g);
      block_pipeline<DEB * Q, 0>(img + Geom3DA<DEB, DXB>::W_IN, img + Geom3DA<DEB, DXB>::W_DIST, fa, fb, lane,
                                 [&](int t, const Frag3& w) { mfma_x3(Y[t / Q], w, xh[t % Q], xm[t % Q], xl[t % Q]); });
#pragma unroll
      for (int ob = 0; ob < DEB; ++ob) {
        const f32x4 m1 = ld4(nl_j + 3 * DX + 16 * ob + 4 * g), m2 = ld4(nl_i + 3 * DX + DE + 16 * ob + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[ob][r] = vj ? Y[ob][r] * m1[r] * m2[r] : 0.f;
      }
      u32x4 dh[Q], dm3[Q], dl[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) split3(d1[2 * q], d1[2 * q + 1], dh[q], dm3[q], dl[q]);
      f32x4 acc2[2];
      block_pipeline<2 * DEB * Q, 0>(img + Geom3DA<DEB, DXB>::W_DIST, img + Geom3DA<DEB, DXB>::W_EOUT, fa, fb, lane,
                                     [&](int t, const Frag3& w) {
        const int ob = t / (2 * Q), m = (t / Q) & 1, q = t % Q;
        if (q == 0) acc2[m] = ld4(vec + (m ? G::B_DMUL : G::B_DADD) * DE + 16 * ob + 4 * g);
        mfma_x3(acc2[m], w, dh[q], dm3[q], dl[q]);
        if (m == 1 && q == Q - 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Y[ob][r] = vj ? (Y[ob][r] + acc2[0][r]) + Y[ob][r] * acc2[1][r] : 0.f;
        }
      });
    } else {
#pragma unroll
      for (int ob = 0; ob < DEB; ob += 2) {
        f32x4 a0 = ld4(vec + G::B_IN * DE + 16 * ob + 4 * g), a1 = ld4(vec + G::B_IN * DE + 16 * (ob + 1) + 4 * g);
#pragma unroll
        for (int ib = 0; ib < DEB; ++ib)
          mfma4x2(a0, a1, img[Geom3D<DEB, FFB, DXB>::W_IN + (ob * DEB + ib) * 64 + lane],
                  img[Geom3D<DEB, FFB, DXB>::W_IN + ((ob + 1) * DEB + ib) * 64 + lane], x[ib]);
        const f32x4 m10 = ld4(nl_j + 3 * DX + 16 * ob + 4 * g), m11 = ld4(nl_j + 3 * DX + 16 * (ob + 1) + 4 * g);
        const f32x4 m20 = ld4(nl_i + 3 * DX + DE + 16 * ob + 4 * g), m21 = ld4(nl_i + 3 * DX + DE + 16 * (ob + 1) + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Y[ob][r] = vj ? a0[r] * m10[r] * m20[r] : 0.f;
          Y[ob + 1][r] = vj ? a1[r] * m11[r] * m21[r] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ob = 0; ob < DEB; ++ob) {
        f32x4 da = ld4(vec + G::B_DADD * DE + 16 * ob + 4 * g), dm = ld4(vec + G::B_DMUL * DE + 16 * ob + 4 * g);
#pragma unroll
        for (int ib = 0; ib < DEB; ++ib)
          mfma4x2(da, dm, img[Geom3D<DEB, FFB, DXB>::W_DADD + (ob * DEB + ib) * 64 + lane],
                  img[Geom3D<DEB, FFB, DXB>::W_DMUL + (ob * DEB + ib) * 64 + lane], d1[ib]);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[ob][r] = vj ? (Y[ob][r] + da[r]) + Y[ob][r] * dm[r] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- pooled statistics of Y and dist1 over this unit's valid keys
    if (a.need_attn & 2) {
      const float inv = 1.0f / (float)__popc(keymask);
      float* sp = a.stat_part + (long long)u * (8 * DE);
      // (the statistics of dist1 depend on the positions only: block3d_dist_stats_kernel makes them for all layers up front)
      constexpr int t = 0;
#pragma unroll
        for (int ob = 0; ob < DEB; ++ob) {
          const f32x4 val = Y[ob];
          float s[4], q[4];
          f32x4 mean, m2, mn, mx;
#pragma unroll
          for (int r = 0; r < 4; ++r) s[r] = vj ? val[r] : 0.f;
          row_sum16_many<4>(s);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            mean[r] = s[r] * inv;
            const float dv = vj ? val[r] - mean[r] : 0.f;
            q[r] = dv * dv;
          }
          row_sum16_many<4>(q);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            m2[r] = q[r];
            mn[r] = vj ? val[r] : INFINITY;
            mx[r] = vj ? val[r] : -INFINITY;
          }
          row_minmax16_x4(mn, mx);
          if (p == 0) {
            float* d = sp + t * 4 * DE + 16 * ob + 4 * g;
            *reinterpret_cast<f32x4*>(d) = mean;
            *reinterpret_cast<f32x4*>(d + DE) = m2;
            *reinterpret_cast<f32x4*>(d + 2 * DE) = mn;
            *reinterpret_cast<f32x4*>(d + 3 * DE) = mx;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- y FiLM (E1 replaces Y), newE = e_out(E1)
    {
      const float* fl = a.film + (long long)b * a.ldf;
#pragma unroll
      for (int ob = 0; ob < DEB; ++ob) {
        const f32x4 ya = ld4(fl + 16 * ob + 4 * g), ym = ld4(fl + DE + 16 * ob + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[ob][r] = vj ? (Y[ob][r] + ya[r]) + Y[ob][r] * ym[r] : 0.f;
      }
    }
    f32x4 nE[DEB];
    if constexpr (X3) {
#pragma unroll
      for (int q = 0; q < Q; ++q) split3(Y[2 * q], Y[2 * q + 1], eh[q], em[q], el[q]);
      // accumulators start at bias + x: the sum leaves as E'' = x + newE.  x is read again here (an L2 hit) rather than
      // held in 16 registers across the distance pipeline and the statistics
#pragma unroll
      for (int ob = 0; ob < DEB; ++ob) {
        nE[ob] = ld4(vec + G::B_EOUT * DE + 16 * ob + 4 * g);
        if (offd) nE[ob] += 0.5f * (ld4(a.E_in + pij + 16 * ob + 4 * g) + ld4(a.E_in + pji + 16 * ob + 4 * g));
      }
      block_pipeline<DEB * Q, 0>(img + Geom3DA<DEB, DXB>::W_EOUT, nullptr, fa, fb, lane,
                                 [&](int t, const Frag3& w) { mfma_x3(nE[t / Q], w, eh[t % Q], em[t % Q], el[t % Q]); });
#pragma unroll
      for (int ob = 0; ob < DEB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) nE[ob][r] = vj ? nE[ob][r] : 0.f;
    } else {
#pragma unroll
      for (int ob = 0; ob < DEB; ob += 2) {
        f32x4 a0 = ld4(vec + G::B_EOUT * DE + 16 * ob + 4 * g), a1 = ld4(vec + G::B_EOUT * DE + 16 * (ob + 1) + 4 * g);
#pragma unroll
        for (int ib = 0; ib < DEB; ++ib)
          mfma4x2(a0, a1, img[Geom3D<DEB, FFB, DXB>::W_EOUT + (ob * DEB + ib) * 64 + lane],
                  img[Geom3D<DEB, FFB, DXB>::W_EOUT + ((ob + 1) * DEB + ib) * 64 + lane], Y[ib]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          nE[ob][r] = vj ? a0[r] : 0.f;
          nE[ob + 1][r] = vj ? a1[r] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- attention, transposed: lane (head p, key group g) register r <-> key 4 g + r
    if (a.need_attn & 1) {
      f32x4 la0 = splat4(0.f), la1 = splat4(0.f);
#pragma unroll
      for (int db = 0; db < DXB; db += 2) {
        f32x4 k0 = splat4(0.f), k1 = splat4(0.f);
        if (vj) {
          k0 = ld4(nl_j + DX + 16 * db + 4 * g);
          k1 = ld4(nl_j + DX + 16 * (db + 1) + 4 * g);
        }
        const f32x4 q0 = ld4(nl_i + 16 * db + 4 * g) * a.inv_sqrt_de, q1 = ld4(nl_i + 16 * (db + 1) + 4 * g) * a.inv_sqrt_de;
        const f32x4 w0 = img[G::W_A + db * 64 + lane] * q0, w1 = img[G::W_A + (db + 1) * 64 + lane] * q1;
        mfma4x2d(la0, la1, k0, k1, w0, w1);                               // A = K tile (keys as rows), B = W_a (.) Q_i
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x4 fe = splat4(vec[G::V_EATT + p]), fp = splat4(vec[G::V_PATT + p]);
      if constexpr (X3) {      // the split operand of e_out serves again, now as the A side; dist1 is split once more (holding its
                               // 6 Q registers across e_out spills)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          Frag3 we, wp;
          u32x4 dh, dm3, dl;
          split3(d1[2 * q], d1[2 * q + 1], dh, dm3, dl);
          load_frag3(we, img + Geom3DA<DEB, DXB>::W_EATT + q * 192, lane);
          load_frag3(wp, img + Geom3DA<DEB, DXB>::W_PATT + q * 192, lane);
          mfma_x3_t(fe, eh[q], em[q], el[q], we);
          mfma_x3_t(fp, dh, dm3, dl, wp);
        }
      } else {
#pragma unroll
        for (int ib = 0; ib < DEB; ++ib)
          mfma4x2d(fe, fp, Y[ib], d1[ib], img[Geom3D<DEB, FFB, DXB>::W_EATT + ib * 64 + lane],
                   img[Geom3D<DEB, FFB, DXB>::W_PATT + ib * 64 + lane]);
      }
      __builtin_amdgcn_sched_barrier(0);
      float lg[4], mloc = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool kv = (keymask >> (4 * g + r)) & 1u;
        float t = kv ? la0[r] + la1[r] : 0.f;
        t = t + fe[r] * t;
        t = t + fp[r] * t;
        lg[r] = kv ? t : -INFINITY;
        mloc = fmaxf(mloc, lg[r]);
      }
      const float M = groups_max(mloc);                                   // finite: a unit holds at least one valid key
      f32x4 pe;
      float sl = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pe[r] = lg[r] == -INFINITY ? 0.f : __expf(lg[r] - M);
        sl += pe[r];
      }
      const float S = groups_sum(sl);
      float* ap = a.attn_part + (long long)u * AP;
      // weighted V: out[h][d] = sum_key p[key][h] V[key][d].  A = p^T (the registers as they are: lane (head, key group),
      // register r = key 4g + r), B = V with the channel on the lane: lane n reads the float4 V[key][64 c + 4n .. 4n+3] and
      // element e goes to MFMA (c, e), whose column n is therefore channel 64 c + 4n + e -- coalesced 256-byte reads per
      // key, and the four results of a lane are again one float4 of the output row of head 4 g' + r'.
      const long long vbase = (long long)b * n;
      const float* vrow[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)      // masked keys carry p = 0: point them at the query row, whose V is always written
        vrow[r] = a.nodelin + (vbase + (((keymask >> (4 * g + r)) & 1u) ? jc * 16 + 4 * g + r : i)) * a.ldn + 2 * DX + 4 * p;
#pragma unroll
      for (int c = 0; c < (DX + 63) / 64; ++c) {
        const bool on = 64 * c + 4 * p < DX;
        f32x4 vq[4], acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vq[r] = on ? ld4(vrow[r] + 64 * c) : splat4(0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = splat4(0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(pe[r], vq[r][e], acc[e], 0, 0, 0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (on && 4 * g + rr < a.n_head) {
            const f32x4 o = {acc[0][rr], acc[1][rr], acc[2][rr], acc[3][rr]};
            *reinterpret_cast<f32x4*>(ap + (4 * g + rr) * DX + 64 * c + 4 * p) = o;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (g == 0 && p < a.n_head) {
        ap[a.n_head * DX + p] = M;
        ap[a.n_head * DX + 16 + p] = S;
      }
    }
    // ---- x3 form: E'' = x + newE (already summed in nE) goes to memory, block3d_ffn_kernel finishes the layer in place
#pragma unroll
    for (int ib = 0; ib < DEB; ++ib) x[ib] = X3 ? nE[ib] : x[ib] + nE[ib];
    if constexpr (!X3) {
      using GG = Geom3D<DEB, FFB, DXB>;
      // ---- E' = LN1(x + newE); E_out = LN2(E' + linE2 relu linE1 E')
      layernorm_c16<DEB>(x, vec + GG::LN1G * DE, vec + GG::LN1B * DE, g);
      f32x4 o2[DEB];
#pragma unroll
      for (int ob = 0; ob < DEB; ++ob) o2[ob] = ld4(vec + GG::B_FF2 * DE + 16 * ob + 4 * g);
#pragma unroll
      for (int f = 0; f < FFB / 4; ++f) {
        f32x4 hid[4];
#pragma unroll
        for (int hb = 0; hb < 4; hb += 2) {
          const int h0 = 4 * f + hb;
          f32x4 a0 = ld4(vec + GG::V_FF1 + 16 * h0 + 4 * g), a1 = ld4(vec + GG::V_FF1 + 16 * (h0 + 1) + 4 * g);
#pragma unroll
          for (int ib = 0; ib < DEB; ++ib)
            mfma4x2(a0, a1, img[GG::W_FF1 + (h0 * DEB + ib) * 64 + lane], img[GG::W_FF1 + ((h0 + 1) * DEB + ib) * 64 + lane], x[ib]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            hid[hb][r] = fmaxf(a0[r], 0.f);
            hid[hb + 1][r] = fmaxf(a1[r], 0.f);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ob = 0; ob < DEB; ob += 2) {
#pragma unroll
          for (int hb = 0; hb < 4; ++hb)
            mfma4x2(o2[ob], o2[ob + 1], img[GG::W_FF2 + (ob * FFB + 4 * f + hb) * 64 + lane],
                    img[GG::W_FF2 + ((ob + 1) * FFB + 4 * f + hb) * 64 + lane], hid[hb]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int ib = 0; ib < DEB; ++ib) x[ib] += o2[ib];
      layernorm_c16<DEB>(x, vec + GG::LN2G * DE, vec + GG::LN2B * DE, g);
    }
    if (inb) {
      float* dst = a.E_out + pij + 4 * g;
#pragma unroll
      for (int ib = 0; ib < DEB; ++ib) *reinterpret_cast<f32x4*>(dst + 16 * ib) = vj ? x[ib] : splat4(0.f);
    }
  }
}

// E_out = LN2(E' + linE2 relu linE1 E'), E' = LN1(E''), in place on the tiles block3d_edge_kernel<.., true> stored
// (pos_transformer_model.py:278-288): the edge_mlp_unit_x3 scheme (both weight matrices resident as three-term images,
// no barrier in the unit loop) with the two LayerNorms and the residual around it.
template <int DEB, int FFB, int NW>
__global__ __launch_bounds__(NW * 64) void block3d_ffn_kernel(Block3DArgs a) {
  using G = Geom3DF<DEB, FFB>;
  constexpr int DE = G::DE, Q1 = G::Q1, Q2 = G::Q2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* img = reinterpret_cast<f32x4*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  for (int idx = tid; idx < G::IMG_F4; idx += NW * 64) img[idx] = a.img[idx];
  __syncthreads();
  const float* vec = reinterpret_cast<const float*>(img + G::VEC);
  const f32x4* W1 = img + G::W_1;
  const f32x4* W2 = img + G::W_2;
  const int n = a.n, nch = (n + 15) >> 4;
  const int count = *a.unit_count;
  for (int uidx = blockIdx.x * NW + wave; uidx < count; uidx += gridDim.x * NW) {
    const int u = a.units[uidx];
    const int row = u / nch, jc = u - row * nch;
    const int b = row / n;
    const int j = jc * 16 + p;
    const bool inb = j < n;
    const int jj = inb ? j : 0;
    const bool vj = inb && a.mask[(long long)b * n + jj] != 0;
    float* tile = a.E_out + ((long long)row * n + jj) * DE + 4 * g;
    Frag3 fa, fb;
    load_frag3(fa, W1, lane);
    f32x4 x[DEB];
#pragma unroll
    for (int ib = 0; ib < DEB; ++ib) x[ib] = vj ? ld4(tile + 16 * ib) : splat4(0.f);
    layernorm_c16<DEB>(x, vec + G::LN1G * DE, vec + G::LN1B * DE, g);
    u32x4 xh[Q1], xm[Q1], xl[Q1];
#pragma unroll
    for (int q = 0; q < Q1; ++q) split3(x[2 * q], x[2 * q + 1], xh[q], xm[q], xl[q]);
    f32x4 hid[FFB];
#pragma unroll
    for (int hb = 0; hb < FFB; ++hb) hid[hb] = ld4(vec + G::V_FF1 + 16 * hb + 4 * g);
    block_pipeline<FFB * Q1, 0>(W1, W2, fa, fb, lane,
                                [&](int t, const Frag3& w) { mfma_x3(hid[t / Q1], w, xh[t % Q1], xm[t % Q1], xl[t % Q1]); });
#pragma unroll
    for (int hb = 0; hb < FFB; ++hb)
#pragma unroll
      for (int r = 0; r < 4; ++r) hid[hb][r] = fmaxf(hid[hb][r], 0.f);
    f32x4 o[DEB];
#pragma unroll
    for (int ob = 0; ob < DEB; ++ob) o[ob] = ld4(vec + G::B_FF2 * DE + 16 * ob + 4 * g);
    {
      u32x4 hh, hm, hl;
      block_pipeline<DEB * Q2, 0>(W2, nullptr, fa, fb, lane, [&](int t, const Frag3& w) {
        if (t % DEB == 0) split3(hid[2 * (t / DEB)], hid[2 * (t / DEB) + 1], hh, hm, hl);
        mfma_x3(o[t % DEB], w, hh, hm, hl);
      });
    }
#pragma unroll
    for (int ib = 0; ib < DEB; ++ib) x[ib] += o[ib];
    layernorm_c16<DEB>(x, vec + G::LN2G * DE, vec + G::LN2B * DE, g);
    if (vj) {
#pragma unroll
      for (int ib = 0; ib < DEB; ++ib) *reinterpret_cast<f32x4*>(tile + 16 * ib) = x[ib];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Row merge: one wave per valid row.  Softmax partials of the row's units -> normalised weighted V per head
// ([h][dx], the input of `out`, :179-180); statistic partials -> the row's (mean, M2, min, max) of Y and dist1 over its
// valid keys (for the graph merge) and the EtoX inputs [mean|min|max|var] of E1 = Y + y_e_add + Y y_e_mul (exact
// per-channel affine map of the statistics of Y) and of dist1 (:183-187, layers.py:60-69).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void block3d_row_merge_kernel(RowMerge3DArgs a) {
  // per wave: softmax weight of (head, chunk) = exp(m - M_head) / S_head, formed once by H * nch lanes; the channel loop then
  // reads one value per chunk and its loads are independent of one another (they were a chain of round trips: three loads
  // per chunk and channel behind a maximum that was re-read per channel)
  __shared__ float wsm[4][8 * 32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ridx = blockIdx.x * 4 + wave;
  if (ridx >= *a.row_count) return;
  const int row = a.rows[ridx];
  const int n = a.n, nch = (n + 15) >> 4, b = row / n;
  const int H = a.n_head, DX = a.dx, DE = a.de, AP = H * DX + 32;
  const int* __restrict__ cc = a.chunk_cnt + b * nch;
  const float* __restrict__ ap0 = a.attn_part + (long long)row * nch * AP;
  float* __restrict__ wv = a.wvcat + (long long)row * (H * DX);
  float* ws = wsm[wave];
  for (int hc0 = 0; hc0 < H * nch; hc0 += 64) {            // (head, chunk) pairs, head-major; H <= 8, nch <= 32
    const int hc = hc0 + lane, h = hc / nch;
    if (hc < H * nch) {
      float M = -INFINITY;
      for (int c = 0; c < nch; ++c)
        if (cc[c] > 0) M = fmaxf(M, ap0[(long long)c * AP + H * DX + h]);
      float S = 0.f;
      for (int c = 0; c < nch; ++c)
        if (cc[c] > 0) S += ap0[(long long)c * AP + H * DX + 16 + h] * __expf(ap0[(long long)c * AP + H * DX + h] - M);
      const int c = hc - h * nch;
      ws[hc] = cc[c] > 0 ? __expf(ap0[(long long)c * AP + H * DX + h] - M) / S : 0.f;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): the LDS writes above, before other lanes read them
  __builtin_amdgcn_wave_barrier();
#pragma unroll 4
  for (int t = lane; t < H * DX; t += 64) {
    const int h = t / DX;
    float acc = 0.f;
    for (int c = 0; c < nch; ++c) {
      const float w = ws[h * nch + c];
      if (w != 0.f) acc += ap0[(long long)c * AP + t] * w;
    }
    wv[t] = acc;
  }
  const float* __restrict__ sp0 = a.stat_part + (long long)row * nch * (8 * DE);
  const float* __restrict__ fl = a.film + (long long)b * a.ldf;
  for (int t = lane; t < DE; t += 64) {
    const int which = 0, ch = t;
    float N = 0.f, mean = 0.f, m2 = 0.f, mn = INFINITY, mx = -INFINITY;
    for (int c = 0; c < nch; ++c)
      if (cc[c] > 0) {
        const float* sp = sp0 + (long long)c * (8 * DE) + which * 4 * DE + ch;
        const float k = (float)cc[c], dl = sp[0] - mean, nn = N + k;
        mean += dl * (k / nn);
        m2 += sp[DE] + dl * dl * (N * k / nn);
        mn = fminf(mn, sp[2 * DE]);
        mx = fmaxf(mx, sp[3 * DE]);
        N = nn;
      }
    float* ag = a.rowagg + (long long)row * (8 * DE) + which * 4 * DE + ch;
    ag[0] = mean; ag[DE] = m2; ag[2 * DE] = mn; ag[3 * DE] = mx;
    float* rs = a.rowstat + (long long)row * (8 * DE) + which * 4 * DE + ch;
    if (which == 0) {
      const float ya = fl[ch], ym = fl[DE + ch], s = 1.0f + ym;
      const float lo = (mn + ya) + mn * ym, hi = (mx + ya) + mx * ym;
      rs[0] = (mean + ya) + mean * ym;
      rs[DE] = fminf(lo, hi);
      rs[2 * DE] = fmaxf(lo, hi);
      rs[3 * DE] = (m2 / N) * s * s;
    } else {
      rs[0] = mean; rs[DE] = mn; rs[2 * DE] = mx; rs[3 * DE] = m2 / N;
    }
  }
}

// newX = wV (y_x_mul + 1) + y_x_add with wV = out(.) FiLMed by the two EtoX vectors (:180-192).  Padded rows: wV is masked
// to zero there (:180), so newX = y_x_add, as in the reference; out / ex / px are not read on them (the row-list Linears
// leave those rows unwritten).
__global__ __launch_bounds__(256) void block3d_newx_kernel(const float* __restrict__ outv, const float* __restrict__ ex,
                                                           const float* __restrict__ px, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ film, int ldf, int film_off,
                                                           float* __restrict__ newX, long long total, int n, int dx) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % dx);
  const long long row = idx / dx;
  const float* fl = film + (row / n) * ldf + film_off;
  float w = 0.f;
  if (mask[row]) {
    w = outv[idx];
    w = w + ex[idx] * w;
    w = w + px[idx] * w;
  }
  newX[idx] = w * (fl[dx + c] + 1.0f) + fl[c];
}

// ---------------------------------------------------------------------------------------------------------
// Graph merge: one workgroup per graph.  Etoy inputs of Y and dist1 from the row aggregates (every valid row holds the
// same number of keys), Xtoy inputs of the UNMASKED newX (layers.py:15: the mean's numerator runs over every row of the
// padded batch, the divisor, minimum, maximum and variance over the valid ones) -> zcat[b] columns [dy, dy + 4dx + 8de).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void block3d_graph_merge_kernel(GraphMerge3DArgs a) {   // arithmetic: pool_stats.h (shared with the y kernel)
  const int b = blockIdx.x, n = a.n;
  const uint8_t* m = a.mask + (long long)b * n;
  __shared__ int s_nv;
  if (threadIdx.x == 0) {
    int c = 0;
    for (int i = 0; i < n; ++i) c += m[i] != 0;
    s_nv = c;
  }
  __syncthreads();
  if (s_nv == 0) return;
  pool_graph3d(a.rowagg + (long long)b * n * 8 * a.de, a.rowaggD + (long long)b * n * 4 * a.de, a.newX + (long long)b * n * a.dx, m, n,
               s_nv, a.dx, a.de, a.zcat + (long long)b * a.ldz + a.dy, a.etoy + (long long)b * (8 * a.de), threadIdx.x, 256);
}

// ---------------------------------------------------------------------------------------------------------
// Positions every layer sees (utils.py:160-164 applied at :461 and after every layer, :301): P <- m (P - mean_n P),
// starting from P = m pos.  pos4[l][b*n+i] = (P_l, |P_l|).  One wave per graph; valid nodes of one graph <= 64 * 64.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void block3d_pos_kernel(const float* __restrict__ pos, const uint8_t* __restrict__ mask, int n,
                                                         int n_layers, float* __restrict__ pos4, long long layer_stride) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* P = pos + (long long)b * n * 3;
  const uint8_t* m = mask + (long long)b * n;
  float* out = pos4 + (long long)b * n * 4;
  // state lives in the output buffer of the previous layer (layer 0 reads the input)
  for (int l = 0; l < n_layers; ++l) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    const float* prev = l == 0 ? nullptr : out + (l - 1) * layer_stride;
    for (int i = lane; i < n; i += 64)
      if (m[i]) {
        sx += prev ? prev[i * 4] : P[i * 3];
        sy += prev ? prev[i * 4 + 1] : P[i * 3 + 1];
        sz += prev ? prev[i * 4 + 2] : P[i * 3 + 2];
      }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      sx += __shfl_xor(sx, o);
      sy += __shfl_xor(sy, o);
      sz += __shfl_xor(sz, o);
    }
    const float mx = sx / (float)n, my = sy / (float)n, mz = sz / (float)n;
    float* cur = out + l * layer_stride;
    for (int i = lane; i < n; i += 64) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m[i]) {
        v[0] = (prev ? prev[i * 4] : P[i * 3]) - mx;
        v[1] = (prev ? prev[i * 4 + 1] : P[i * 3 + 1]) - my;
        v[2] = (prev ? prev[i * 4 + 2] : P[i * 3 + 2]) - mz;
        v[3] = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      }
      *reinterpret_cast<f32x4*>(cur + i * 4) = v;
    }
    __syncthreads();   // one wave: orders this layer's stores before the next layer's loads
  }
}

// ---------------------------------------------------------------------------------------------------------
// Statistics of dist1 (pos_transformer_model.py:122-132) over the keys of every valid row, for ALL layers in one launch:
// dist1 depends on the positions and on five weight vectors per layer only, so the EtoX input of pos_x_mul (:186) and the
// row aggregates behind the Etoy input of dist_y (:203) need not come out of the edge kernel.  One wave per (row, layer),
// lane = channel; two passes over the keys (mean, then squared deviations), as the reference's own formula.
//   dvec[l]     = [w_dist | w_cos | b | w_norm_i | w_norm_j][de]
//   rowstatD[l][row] = [mean | min | max | var][de],  rowaggD[l][row] = [mean | M2 | min | max][de]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void block3d_dist_stats_kernel(const float* __restrict__ pos4, const uint8_t* __restrict__ mask,
                                                                 const float* __restrict__ dvec, const int* __restrict__ rows,
                                                                 const int* __restrict__ row_count, float* __restrict__ rowstatD,
                                                                 float* __restrict__ rowaggD, int M, int n, int de) {
  // per wave: (distance, cosine, key norm) of the row's valid keys, compacted, 64 keys at a time
  __shared__ float geo[4][64 * 3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l = blockIdx.y;
  const int ridx = blockIdx.x * 4 + wave;
  if (ridx >= *row_count) return;                    // wave-uniform; no workgroup barrier below
  const long long row = rows[ridx];
  const int b = (int)(row / n);
  const float* P = pos4 + (long long)l * M * 4;
  const uint8_t* mb = mask + (long long)b * n;
  const f32x4 pi = *reinterpret_cast<const f32x4*>(P + row * 4);
  const float di = pi[3] + 1e-7f, uix = pi[0] / di, uiy = pi[1] / di, uiz = pi[2] / di;
  const float* dv = dvec + (long long)l * 5 * de;
  float* gw = geo[wave];
  // one channel per lane and pass over the channels (de <= 64: one pass); statistics in one sweep over the keys with the
  // first value as the shift (sum and sum of squares of t - k: nothing is lost to cancellation)
  for (int c0 = 0; c0 < de; c0 += 64) {
    const int c = c0 + lane;
    const bool con = c < de;
    const float w0 = con ? dv[c] : 0.f, w1 = con ? dv[de + c] : 0.f, bd = con ? dv[2 * de + c] : 0.f;
    const float n1 = con ? dv[3 * de + c] : 0.f, n2 = con ? dv[4 * de + c] : 0.f;
    const float base = bd + n1 * pi[3];
    float k = 0.f, s1 = 0.f, s2 = 0.f, mn = INFINITY, mx = -INFINITY, cnt = 0.f;
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int j = j0 + lane;
      const bool vj = j < n && mb[j] != 0;
      const unsigned long long bal = __ballot(vj);
      const int m = __popcll(bal);
      if (vj) {
        const f32x4 pj = *reinterpret_cast<const f32x4*>(P + ((long long)b * n + j) * 4);
        const float ex = pi[0] - pj[0], ey = pi[1] - pj[1], ez = pi[2] - pj[2];
        const float dj = pj[3] + 1e-7f;
        const int slot = __popcll(bal & ((1ull << lane) - 1ull));
        gw[3 * slot] = sqrtf(ex * ex + ey * ey + ez * ez);
        gw[3 * slot + 1] = uix * (pj[0] / dj) + uiy * (pj[1] / dj) + uiz * (pj[2] / dj);
        gw[3 * slot + 2] = pj[3];
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's LDS writes are visible to its other lanes
      for (int q = 0; q < m; ++q) {
        const float t = fmaxf(gw[3 * q] * w0 + gw[3 * q + 1] * w1 + base + n2 * gw[3 * q + 2], 0.f);
        if (cnt == 0.f) k = t;
        const float d = t - k;
        s1 += d; s2 += d * d;
        mn = fminf(mn, t); mx = fmaxf(mx, t);
        cnt += 1.f;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (con) {
      const float mean = k + s1 / cnt, m2 = fmaxf(s2 - s1 * s1 / cnt, 0.f);
      float* rs = rowstatD + ((long long)l * M + row) * (4 * de) + c;
      rs[0] = mean; rs[de] = mn; rs[2 * de] = mx; rs[3 * de] = m2 / cnt;
      float* ag = rowaggD + ((long long)l * M + row) * (4 * de) + c;
      ag[0] = mean; ag[de] = m2; ag[2 * de] = mn; ag[3 * de] = mx;
    }
  }
}

// valid keys per (graph, 16-key chunk)
__global__ void block3d_chunk_count_kernel(const uint8_t* mask, int B, int n, int nch, int* cnt) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * nch) return;
  const int b = idx / nch, c = idx - b * nch;
  int k = 0;
  for (int j = 16 * c; j < min(n, 16 * c + 16); ++j) k += mask[(long long)b * n + j] != 0;
  cnt[idx] = k;
}

// ---------------------------------------------------------------------------------------------------------
#define SCGG_B3D_CASES CASE(4, 8, 8) CASE(4, 8, 16) CASE(2, 4, 2)   /* (de/16, dim_ffE/16, dx/16) */

// SCGG_B3D_MODE=f32 selects the one-kernel fp32-MFMA form (development A/B switch); default: the bf16x3 pair
static bool b3d_x3() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("SCGG_B3D_MODE");
    mode = (e != nullptr && e[0] == 'f') ? 0 : 1;
  }
  return mode == 1;
}

long long block3d_img_f4(int dx, int de, int ffe) {
#define CASE(DEB, FFB, DXB)                                                                            \
  if (de == 16 * DEB && ffe == 16 * FFB && dx == 16 * DXB)                                             \
    return b3d_x3() ? (long long)Geom3DA<DEB, DXB>::IMG_F4 + Geom3DF<DEB, FFB>::IMG_F4 : (long long)Geom3D<DEB, FFB, DXB>::IMG_F4;
  SCGG_B3D_CASES
#undef CASE
  return -1;
}

int launch_block3d_pack(const Block3DPackArgs& a, int dx, int de, int ffe, hipStream_t st) {
  if (a.n_head < 1 || a.n_head > 16) return -1;
#define CASE(DEB, FFB, DXB)                                                                           \
  if (de == 16 * DEB && ffe == 16 * FFB && dx == 16 * DXB) {                                          \
    if (b3d_x3()) hipLaunchKernelGGL((block3d_pack_x3_kernel<DEB, FFB, DXB>), dim3(64), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((block3d_pack_kernel<DEB, FFB, DXB>), dim3(64), dim3(256), 0, st, a);     \
    return 0;                                                                                         \
  }
  SCGG_B3D_CASES
#undef CASE
  return -1;
}

template <int DEB, int FFB, int NW>
static int launch_b3d_ffn(const Block3DArgs& a, int n_cu, hipStream_t st) {
  static LdsAttrOnce attr;
  void (*kf)(Block3DArgs) = block3d_ffn_kernel<DEB, FFB, NW>;
  const size_t lds = (size_t)Geom3DF<DEB, FFB>::IMG_F4 * 16;
  if (set_max_dynamic_lds(attr, reinterpret_cast<const void*>(kf), lds)) return -2;
  hipLaunchKernelGGL(kf, dim3(n_cu), dim3(NW * 64), lds, st, a);
  return 0;
}
template <int DEB, int FFB, int DXB, int NW, bool X3>
static int launch_b3d_edge(const Block3DArgs& a, int n_cu, hipStream_t st) {
  using GA = typename std::conditional<X3, Geom3DA<DEB, DXB>, Geom3D<DEB, FFB, DXB>>::type;
  static LdsAttrOnce attr;
  void (*ka)(Block3DArgs) = block3d_edge_kernel<DEB, FFB, DXB, NW, X3>;
  const size_t lds = (size_t)GA::IMG_F4 * 16;
  if (set_max_dynamic_lds(attr, reinterpret_cast<const void*>(ka), lds)) return -2;
  hipLaunchKernelGGL(ka, dim3(n_cu), dim3(NW * 64), lds, st, a);
  return 0;
}
// Waves per workgroup (= per CU: the LDS images leave room for one workgroup).  x3 pair: 8 for the block kernel (it needs
// ~200 VGPRs; at 12 waves = 168 it spills 69 and is 14 % slower), 12 for the FFN kernel (8: 2.69 ms per forward, 12: 2.65, 16: 2.68); the fp32
// kernel 12.  SCGG_B3D_NW / SCGG_B3D_FFN_NW = 8 | 12 | 16 for A/B runs.
template <int DEB, int FFB, int DXB>
static int launch_b3d(const Block3DArgs& a, hipStream_t st) {
  static int nw = 0, nwf = 0;
  if (nw == 0) {
    const char *e = getenv("SCGG_B3D_NW"), *f = getenv("SCGG_B3D_FFN_NW");
    nw = e ? atoi(e) : (b3d_x3() ? 8 : 12);
    nwf = f ? atoi(f) : 12;
  }
  const int n_cu = device_cu_count();
  if (n_cu <= 0) return -2;
  if (!b3d_x3()) {
    if (nw == 16) return launch_b3d_edge<DEB, FFB, DXB, 16, false>(a, n_cu, st);
    if (nw == 12) return launch_b3d_edge<DEB, FFB, DXB, 12, false>(a, n_cu, st);
    return launch_b3d_edge<DEB, FFB, DXB, 8, false>(a, n_cu, st);
  }
  int rc;
  if (nw == 16) rc = launch_b3d_edge<DEB, FFB, DXB, 16, true>(a, n_cu, st);
  else if (nw == 12) rc = launch_b3d_edge<DEB, FFB, DXB, 12, true>(a, n_cu, st);
  else rc = launch_b3d_edge<DEB, FFB, DXB, 8, true>(a, n_cu, st);
  if (rc) return rc;
  Block3DArgs f = a;
  f.img = a.img + Geom3DA<DEB, DXB>::IMG_F4;
  if (nwf == 8) return launch_b3d_ffn<DEB, FFB, 8>(f, n_cu, st);
  if (nwf == 12) return launch_b3d_ffn<DEB, FFB, 12>(f, n_cu, st);
  return launch_b3d_ffn<DEB, FFB, 16>(f, n_cu, st);
}

int launch_block3d_edge(const Block3DArgs& a, int dx, int de, int ffe, hipStream_t st) {
  if (a.n_head < 1 || a.n_head > 16 || a.ldn % 4 != 0 || a.ldf % 4 != 0) return -1;
#define CASE(DEB, FFB, DXB) \
  if (de == 16 * DEB && ffe == 16 * FFB && dx == 16 * DXB) return launch_b3d<DEB, FFB, DXB>(a, st);
  SCGG_B3D_CASES
#undef CASE
  return -1;
}

int launch_block3d_row_merge(const RowMerge3DArgs& a, int M, hipStream_t st) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(block3d_row_merge_kernel, dim3((M + 3) / 4), dim3(256), 0, st, a);
  return 0;
}

int launch_block3d_newx(const float* outv, const float* ex, const float* px, const uint8_t* mask, const float* film, int ldf,
                        int film_off, float* newX, int M, int n, int dx, hipStream_t st) {
  const long long total = (long long)M * dx;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(block3d_newx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, outv, ex, px, mask, film,
                     ldf, film_off, newX, total, n, dx);
  return 0;
}

int launch_block3d_graph_merge(const GraphMerge3DArgs& a, int B, hipStream_t st) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(block3d_graph_merge_kernel, dim3(B), dim3(256), 0, st, a);
  return 0;
}

int launch_block3d_pos(const float* pos, const uint8_t* mask, int B, int n, int n_layers, float* pos4, hipStream_t st) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(block3d_pos_kernel, dim3(B), dim3(64), 0, st, pos, mask, n, n_layers, pos4, (long long)B * n * 4);
  return 0;
}

int launch_block3d_dist_stats(const float* pos4, const uint8_t* mask, const float* dvec, const int* rows, const int* row_count,
                              float* rowstatD, float* rowaggD, int B, int n, int de, int n_layers, hipStream_t st) {
  const int M = B * n;
  if (M <= 0 || n_layers <= 0) return 0;
  hipLaunchKernelGGL(block3d_dist_stats_kernel, dim3((M + 3) / 4, n_layers), dim3(256), 0, st, pos4, mask, dvec, rows, row_count,
                     rowstatD, rowaggD, M, n, de);
  return 0;
}

int launch_block3d_chunk_count(const uint8_t* mask, int B, int n, int* cnt, hipStream_t st) {
  const int nch = (n + 15) / 16, tot = B * nch;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(block3d_chunk_count_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, mask, B, n, nch, cnt);
  return 0;
}

}  // namespace scgg
