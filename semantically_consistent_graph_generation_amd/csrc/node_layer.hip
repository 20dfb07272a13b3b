// Fused XEy layer, node part (gfx950, fp32 MFMA): everything between two edge-layer launches in ONE kernel.
//
//   weighted_V merge + y-FiLM  transformer_model.py:207-214, :230-232   (the per-unit partials of edge_unit_layer.hip)
//   x_out                      transformer_model.py:234-235
//   X = normX1(X + newX)       transformer_model.py:88-89
//   X = normX2(X + linX2(relu(linX1(X))))   :90-92 (+ node mask, :110)
//   q | k | v of the NEXT layer transformer_model.py:176-184 (masked rows are never read: the edge kernel and this one
//                              walk the valid-row list)
//
// One workgroup of 8 waves owns a tile of 32 valid rows.  The tile lives in LDS (two [32][D+4] buffers alternating as operand and result; the residual stream
// stays in registers); a Linear is evaluated transposed on v_mfma_f32_32x32x2_f32 with the packed weights of
// pack_linear_kernel as the A operand (each wave owns output row blocks w, w+8, ...: no weight is read twice inside a
// workgroup) and the tile rows read from LDS as the B operand (one ds_read_b128 per 4 MFMAs, the k order of the
// packed layout).  LayerNorms run one wave per 4 rows.  Replaces six launches per layer (attn_finish, x_out, LN,
// linX1, linX2, LN) and the separate q|k|v launch, and the activations make no HBM round trip between them.
#include "mfma_chain.h"
#include "kernels.h"
#include "x3_tile.h"
#include <stdlib.h>

namespace scgg {
#ifdef SCGG_STAMP_NODE   // development aid: s_memtime at the phase boundaries of node_layer_x3_kernel (scripts/probe/node_stamps.py)
__device__ unsigned long long scgg_node_stamps[4][2][16];
#define SCGG_NSTAMP(k) do { if (lane == 0 && (wave == 0 || wave == 7) && (blockIdx.x == 0 || blockIdx.x == 57 || blockIdx.x == 130 || blockIdx.x == 200)) \
  { unsigned long long* q_ = &scgg_node_stamps[blockIdx.x == 0 ? 0 : blockIdx.x == 57 ? 1 : blockIdx.x == 130 ? 2 : 3][wave == 7][k]; \
    if (*q_ == 0) *q_ = __builtin_readcyclecounter(); } } while (0)   /* first launch after load only: layer 0, with the q|k|v tail */
#else
#define SCGG_NSTAMP(k) do {} while (0)
#endif


template <int DXB, int FXB>
struct NodeGeom {
  static constexpr int DX = 32 * DXB, FX = 32 * FXB;
  static constexpr int LD = (DX > FX ? DX : FX) + 4;       // LDS row stride in floats (16-byte aligned rows)
  static constexpr size_t LDS_BYTES = (size_t)2 * 32 * LD * 4;   // two workgroups per CU
};

// acc (C layout: row of the tile on lane & 31, 32 output channels of block rb in the registers) += W[rb] . X^T
// weights: global, software pipelined one k-block ahead; activations: LDS row `xrow` (+ 4h folded in by the caller)
template <int KB>
__device__ __forceinline__ void tile_linear(f32x16& acc, const f32x4* __restrict__ wp, int rb, const float* xrow, int lane) {
  const f32x4* w = wp + (size_t)(rb * KB * 4) * 64;
  f32x4 cur[4], nxt[4];
  load_frag(cur, w, lane);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) load_frag(nxt, w + (kb + 1) * 256, lane);
    f32x4 x[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) x[g] = *reinterpret_cast<const f32x4*>(xrow + 32 * kb + 8 * g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[g][e], x[g][e], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) cur[g] = nxt[g];
  }
}


// rows 4w..4w+3 of the tile: res <- LN(res + lin) * gamma + beta (eps 1e-5, biased variance); the residual stream stays in
// this wave's registers (lane = 4 channels), the normalised rows also go to LDS `dst` (next operand) and optionally global
template <int DX, int LD>
__device__ __forceinline__ void tile_add_layernorm(f32x4 (&res)[4], const float* lin, float* dst,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   int wave, int lane, float* gout, const long long* grow, const bool* gok) {
  const int c = lane * 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = wave * 4 + t;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < DX) v = res[t] + *reinterpret_cast<const f32x4*>(lin + r * LD + c);
    const float mean = wave_sum64(v[0] + v[1] + v[2] + v[3]) * (1.0f / DX);
    f32x4 d = v - mean;
    if (c >= DX) d = f32x4{0.f, 0.f, 0.f, 0.f};
    const float var = wave_sum64(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / DX);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (c < DX) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
      const f32x4 o = d * rstd * gm + bt;
      res[t] = o;
      *reinterpret_cast<f32x4*>(dst + r * LD + c) = o;
      if (gout != nullptr && gok[t]) *reinterpret_cast<f32x4*>(gout + grow[t] * DX + c) = o;
    }
  }
}

template <int DXB, int FXB>
__global__ __launch_bounds__(512) void node_layer_kernel(NodeLayerArgs a) {
  using G = NodeGeom<DXB, FXB>;
  constexpr int DX = G::DX, LD = G::LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bufA = reinterpret_cast<float*>(smem);        // the two tile buffers alternate as operand / result
  float* bufB = bufA + 32 * LD;
  f32x4 res[4];                                        // residual stream X of this wave's 4 rows (lane = 4 channels)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 31, h = lane >> 5;
  const int count = *a.row_count;
  const int tile = blockIdx.x;
  if (tile * 32 >= count) return;

  // ---- prologue: this wave's 4 rows: newX_pre from the attention partials of the row's units, residual X
  const int n = a.n, nch = (n + 15) >> 4;
  long long grow[4];
  bool gok[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ridx = tile * 32 + wave * 4 + t;
    gok[t] = ridx < count;
    const long long row = a.rows[gok[t] ? ridx : count - 1];
    grow[t] = row;
    const int b = (int)(row / n);
    const uint8_t* mb = a.mask + (long long)b * n;
    const float* film = a.film + (long long)b * (4 * DX);
    const int c = lane * 4;
    if (c < DX) {
      f32x4 M = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, S = {0.f, 0.f, 0.f, 0.f}, A = {0.f, 0.f, 0.f, 0.f};
      for (int jc = 0; jc < nch; ++jc) {
        bool any = false;
        for (int j = jc * 16; j < min(n, jc * 16 + 16); ++j) any |= mb[j] != 0;      // wave-uniform
        if (!any) continue;
        const float* pt = a.attn_part + (row * nch + jc) * (3 * DX) + c;
        const f32x4 m = *reinterpret_cast<const f32x4*>(pt), s = *reinterpret_cast<const f32x4*>(pt + DX),
                    ac = *reinterpret_cast<const f32x4*>(pt + 2 * DX);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float mn = fmaxf(M[e], m[e]);
          const float fo = M[e] == -INFINITY ? 0.f : __expf(M[e] - mn);
          const float fn = m[e] == -INFINITY ? 0.f : __expf(m[e] - mn);
          S[e] = S[e] * fo + s[e] * fn;
          A[e] = A[e] * fo + ac[e] * fn;
          M[e] = mn;
        }
      }
      const f32x4 fa = *reinterpret_cast<const f32x4*>(film + 2 * DX + c), fm = *reinterpret_cast<const f32x4*>(film + 3 * DX + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fa[e] + fm[e] * (A[e] / S[e]);
      *reinterpret_cast<f32x4*>(bufA + (wave * 4 + t) * LD + c) = o;
      res[t] = *reinterpret_cast<const f32x4*>(a.X + row * DX + c);
    }
  }
  __syncthreads();

  // ---- x_out: bufA -> bufB
  for (int rb = wave; rb < DXB; rb += 8) {
    f32x16 acc = load_vec_block(a.b_xout + 32 * rb, h);
    tile_linear<DXB>(acc, a.W_xout, rb, bufA + p * LD + 4 * h, lane);
    store_vec_block(bufB + p * LD + 32 * rb, h, acc);
  }
  __syncthreads();
  // ---- X = normX1(X + newX): registers, and bufA as the next operand
  tile_add_layernorm<DX, LD>(res, bufB, bufA, a.ln1_g, a.ln1_b, wave, lane, nullptr, grow, gok);
  __syncthreads();
  // ---- linX1 + relu: bufA -> bufB
  for (int rb = wave; rb < FXB; rb += 8) {
    f32x16 acc = load_vec_block(a.b_lin1 + 32 * rb, h);
    tile_linear<DXB>(acc, a.W_lin1, rb, bufA + p * LD + 4 * h, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
    store_vec_block(bufB + p * LD + 32 * rb, h, acc);
  }
  __syncthreads();
  // ---- linX2: bufB -> bufA
  for (int rb = wave; rb < DXB; rb += 8) {
    f32x16 acc = load_vec_block(a.b_lin2 + 32 * rb, h);
    tile_linear<FXB>(acc, a.W_lin2, rb, bufB + p * LD + 4 * h, lane);
    store_vec_block(bufA + p * LD + 32 * rb, h, acc);
  }
  __syncthreads();
  // ---- X = normX2(X + ff): global and bufB (every row of the tile is a valid node, so the node mask is 1)
  tile_add_layernorm<DX, LD>(res, bufA, bufB, a.ln2_g, a.ln2_b, wave, lane, a.X_out, grow, gok);
  if (a.W_qkv == nullptr) return;
  __syncthreads();
  // ---- q | k | v of the next layer: bufB -> global
  {
    const int ridx = tile * 32 + p;
    const bool ok = ridx < count;
    const long long row = a.rows[ok ? ridx : count - 1];
    for (int rb = wave; rb < 3 * DXB; rb += 8) {
      f32x16 acc = load_vec_block(a.b_qkv + 32 * rb, h);
      tile_linear<DXB>(acc, a.W_qkv, rb, bufB + p * LD + 4 * h, lane);
      if (ok) store_vec_block(a.qkv + row * (3 * DX) + 32 * rb, h, acc);
    }
  }
}

// =========================================================================================================
// The same layer on the bf16 matrix pipe with fp32 accuracy ("x3", see edge_unit_layer.hip): every Linear operand is
// split exactly into three bf16 terms (weights at pack time: launch_pack_linear_x3; activations when they are written
// to LDS, once per element) and a product is the six leading partial products on v_mfma_f32_16x16x32_bf16 = 6/16 of
// the fp32 MFMA time.
//
// Tile = 48 valid rows (three 16-row groups): the 10 917 rows of the headline workload are 228 tiles, ONE round on 256
// CUs (32-row tiles were 342 = two rounds, and a tile's time is mostly its chain of barriers and latencies, not its
// MFMAs).  The tile's operand lives in LDS as three bf16 planes [48][K + 8]; results that feed a LayerNorm go through
// an fp32 buffer.  Two LDS regions alternate: planes(R1) -x_out-> fp32(R2) -LN1-> planes(R1) -linX1-> planes(R2)
// -linX2-> fp32(R1) -LN2-> planes(R2) -qkv-> global.  A wave owns two 16-channel output blocks at a time; their weight
// blocks (3 x 16 B per lane per 32-channel K chunk) stream from L2 two chunks ahead, across Linear boundaries.
// =========================================================================================================

template <int DXB, int FXB>
struct NodeGeomX3 {
  static constexpr int DX = 32 * DXB, FX = 32 * FXB;
  static constexpr int TR = 48, NRG = TR / 16;              // tile rows, 16-row groups
  static constexpr int KMAX = DX > FX ? DX : FX;
  static constexpr int KS = KMAX + 8;                       // bf16 row stride of a plane: 4 banks (mod 32) per row
  static constexpr int PLANE = TR * KS;                     // bf16 elements per term plane
  static constexpr int LD = KMAX + 4;                       // fp32 result buffer stride
  static constexpr size_t SPLIT_BYTES = (size_t)3 * PLANE * 2, F_BYTES = (size_t)TR * LD * 4;
  static constexpr size_t REGION = SPLIT_BYTES > F_BYTES ? SPLIT_BYTES : F_BYTES;
  static constexpr size_t LDS_BYTES = 2 * REGION;
};

// rows RPW w .. RPW w + RPW - 1 of the tile: res <- LN(res + lin) * gamma + beta (eps 1e-5, biased variance); the residual
// stream stays in this wave's registers (lane = 4 channels), the normalised rows are written as split planes (the next
// Linear's operand) and optionally to global
template <int DX, int LD, int KS, int PLANE, int RPW>
__device__ __forceinline__ void tile_add_layernorm_x3(f32x4 (&res)[RPW], const float* lin, __bf16* planes,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int wave, int lane, float* gout, const long long* grow, const bool* gok) {
  const int c = lane * 4;
#pragma unroll
  for (int t = 0; t < RPW; ++t) {
    const int r = wave * RPW + t;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < DX) v = res[t] + *reinterpret_cast<const f32x4*>(lin + r * LD + c);
    const float mean = wave_sum64(v[0] + v[1] + v[2] + v[3]) * (1.0f / DX);
    f32x4 d = v - mean;
    if (c >= DX) d = f32x4{0.f, 0.f, 0.f, 0.f};
    const float var = wave_sum64(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / DX);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (c < DX) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
      const f32x4 o = d * rstd * gm + bt;
      res[t] = o;
      put_split4<KS, PLANE>(planes, r, c, o);
      if (gout != nullptr && gok[t]) *reinterpret_cast<f32x4*>(gout + grow[t] * DX + c) = o;
    }
  }
}

template <int DXB, int FXB>
__global__ __launch_bounds__(512) void node_layer_x3_kernel(NodeLayerArgs a) {
  using G = NodeGeomX3<DXB, FXB>;
  constexpr int DX = G::DX, FX = G::FX, LD = G::LD, KS = G::KS, PLANE = G::PLANE, TR = G::TR, NRG = G::NRG;
  constexpr int RPW = TR / 8;                                   // rows per wave in the row-wise parts
  constexpr int KCX = DX / 32, KCF = FX / 32;                   // 32-channel K chunks
  constexpr int NPX = DX / 32, NPF = FX / 32;                   // pairs of 16-channel output blocks of a Linear
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R1 = smem;
  char* R2 = smem + G::REGION;
  f32x4 res[RPW];                                               // residual stream X of this wave's rows (lane = 4 channels)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  SCGG_NSTAMP(0);
  const int count = *a.row_count;
  const int tile = blockIdx.x;
  if (tile * TR >= count) return;
  // this wave's block pairs of the four weight images, in order of use; each Linear leaves the ring primed for the next
  const bool has_qkv = a.W3_qkv != nullptr;
  const nu32x4* w_xout = wave < NPX ? w3_block(a.W3_xout, 2 * wave, KCX, lane) : nullptr;
  const nu32x4* w_lin1 = wave < NPF ? w3_block(a.W3_lin1, 2 * wave, KCX, lane) : nullptr;
  const nu32x4* w_lin2 = wave < NPX ? w3_block(a.W3_lin2, 2 * wave, KCF, lane) : nullptr;
  const int tail_pairs = a.tail_pairs > 0 ? a.tail_pairs : 3 * NPX;
  const int tail_ld = a.tail_pairs > 0 ? a.tail_ld : 3 * DX;
  const nu32x4* w_qkv0 = has_qkv && wave < tail_pairs ? w3_block(a.W3_qkv, 2 * wave, KCX, lane) : nullptr;
  NWRing ring;
  ring.next = nullptr;
  if (w_xout != nullptr) w3_prime<KCX>(ring, w_xout);

  SCGG_NSTAMP(1);
  // ---- prologue: this wave's rows: newX_pre from the attention partials of the row's units, residual X
  const int n = a.n, nch = (n + 15) >> 4;
  long long grow[RPW];
  bool gok[RPW];
  __bf16* S1 = reinterpret_cast<__bf16*>(R1);
  const int c = lane * 4;
#pragma unroll
  for (int t = 0; t < RPW; ++t) {
    const int ridx = tile * TR + wave * RPW + t;
    gok[t] = ridx < count;
    grow[t] = a.rows[gok[t] ? ridx : count - 1];
  }
  if (a.newX_in != nullptr) {
#pragma unroll
    for (int t = 0; t < RPW; ++t)
      if (c < DX) {
        put_split4<KS, PLANE>(S1, wave * RPW + t, c, *reinterpret_cast<const f32x4*>(a.newX_in + grow[t] * DX + c));
        res[t] = *reinterpret_cast<const f32x4*>(a.X + grow[t] * DX + c);
      }
  } else {
    // merge of the attention partials of a row's units: the rows of this wave advance together, one key chunk at a time,
    // so the partials of all RPW rows are in flight at once (row by row it was a chain of RPW * nch round trips)
    unsigned bits[RPW];                                         // chunks of the row's graph that hold a valid key
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const uint8_t* mb = a.mask + (grow[t] / n) * n;
      unsigned bt = 0;
      if (a.chunk_bits != nullptr) bits[t] = a.chunk_bits[grow[t] / n];   // (scanning the mask here: 32 dependent byte loads per row)
      else for (int jc = 0; jc < nch; ++jc) {
        bool any = false;
        for (int j = jc * 16; j < min(n, jc * 16 + 16); ++j) any |= mb[j] != 0;      // wave-uniform
        bt |= any ? (1u << (jc & 31)) : 0u;
        if ((jc & 31) == 31 || jc == nch - 1) {                 // n <= 512: one word
          bits[t] = bt;
        }
      }
      res[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < DX) res[t] = *reinterpret_cast<const f32x4*>(a.X + grow[t] * DX + c);
    }
    constexpr int RH = RPW / 2;                                 // rows per batch (register budget)
    if (nch <= 2) {
      // up to 32 nodes: the partials of a batch's rows for BOTH chunks and their FiLM vectors are requested together: one round
      // trip per batch (96 registers; all six rows at once spill)
#pragma unroll
      for (int t0 = 0; t0 < RPW; t0 += RH) {
        f32x4 m[RH][2], s2[RH][2], ac[RH][2], fa[RH], fm[RH];
#pragma unroll
        for (int t = 0; t < RH; ++t) {
#pragma unroll
          for (int jc = 0; jc < 2; ++jc) {
            m[t][jc] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; s2[t][jc] = f32x4{0.f, 0.f, 0.f, 0.f}; ac[t][jc] = s2[t][jc];
            if (c < DX && jc < nch && ((bits[t0 + t] >> jc) & 1u)) {
              const float* pt = a.attn_part + (grow[t0 + t] * nch + jc) * (3 * DX) + c;
              m[t][jc] = *reinterpret_cast<const f32x4*>(pt); s2[t][jc] = *reinterpret_cast<const f32x4*>(pt + DX);
              ac[t][jc] = *reinterpret_cast<const f32x4*>(pt + 2 * DX);
            }
          }
          fa[t] = fm[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (c < DX) {
            const float* film = a.film + (grow[t0 + t] / n) * (4 * DX);
            fa[t] = *reinterpret_cast<const f32x4*>(film + 2 * DX + c); fm[t] = *reinterpret_cast<const f32x4*>(film + 3 * DX + c);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < RH; ++t) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {       // the same operations, in the same order, as the chunk loop below
            float Mx = -INFINITY, Sx = 0.f, Ax = 0.f;
#pragma unroll
            for (int jc = 0; jc < 2; ++jc) {
              const float mn = fmaxf(Mx, m[t][jc][e]);
              const float fo = Mx == -INFINITY ? 0.f : __expf(Mx - mn);
              const float fn = m[t][jc][e] == -INFINITY ? 0.f : __expf(m[t][jc][e] - mn);
              Sx = Sx * fo + s2[t][jc][e] * fn;
              Ax = Ax * fo + ac[t][jc][e] * fn;
              Mx = mn;
            }
            o[e] = fa[t][e] + fm[t][e] * (Ax / Sx);
          }
          if (c < DX) put_split4<KS, PLANE>(S1, wave * RPW + t0 + t, c, o);
        }
      }
    } else
#pragma unroll
    for (int t0 = 0; t0 < RPW; t0 += RH) {
      f32x4 Mx[RH], Sx[RH], Ax[RH];
#pragma unroll
      for (int t = 0; t < RH; ++t) { Mx[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; Sx[t] = f32x4{0.f, 0.f, 0.f, 0.f}; Ax[t] = Sx[t]; }
      for (int jc = 0; jc < nch; ++jc) {
        f32x4 m[RH], s2[RH], ac[RH];
#pragma unroll
        for (int t = 0; t < RH; ++t) {
          m[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; s2[t] = f32x4{0.f, 0.f, 0.f, 0.f}; ac[t] = s2[t];
          if (c < DX && ((bits[t0 + t] >> jc) & 1u)) {
            const float* pt = a.attn_part + (grow[t0 + t] * nch + jc) * (3 * DX) + c;
            m[t] = *reinterpret_cast<const f32x4*>(pt); s2[t] = *reinterpret_cast<const f32x4*>(pt + DX);
            ac[t] = *reinterpret_cast<const f32x4*>(pt + 2 * DX);
          }
        }
#pragma unroll
        for (int t = 0; t < RH; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float mn = fmaxf(Mx[t][e], m[t][e]);
            const float fo = Mx[t][e] == -INFINITY ? 0.f : __expf(Mx[t][e] - mn);
            const float fn = m[t][e] == -INFINITY ? 0.f : __expf(m[t][e] - mn);
            Sx[t][e] = Sx[t][e] * fo + s2[t][e] * fn;
            Ax[t][e] = Ax[t][e] * fo + ac[t][e] * fn;
            Mx[t][e] = mn;
          }
      }
#pragma unroll
      for (int t = 0; t < RH; ++t)
        if (c < DX) {
          const float* film = a.film + (grow[t0 + t] / n) * (4 * DX);
          const f32x4 fa = *reinterpret_cast<const f32x4*>(film + 2 * DX + c), fm = *reinterpret_cast<const f32x4*>(film + 3 * DX + c);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fa[e] + fm[e] * (Ax[t][e] / Sx[t][e]);
          put_split4<KS, PLANE>(S1, wave * RPW + t0 + t, c, o);
        }
    }
  }
  __syncthreads();

  auto bias_acc = [&](f32x4 (&acc)[2][NRG], const float* bias, int ob0) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * (ob0 + blk) + 4 * g);
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) acc[blk][rg] = bv;
    }
  };
  SCGG_NSTAMP(2);
  // ---- x_out: planes(R1) -> fp32(R2)
  if (wave < NPX) {
    float* F = reinterpret_cast<float*>(R2);
    f32x4 acc[2][NRG];
    bias_acc(acc, a.b_xout, 2 * wave);
    tile_linear_x3<KCX, KCX, KS, PLANE, NRG>(acc, ring, w_xout, w_lin1, S1, p, g);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg)
        *reinterpret_cast<f32x4*>(F + (16 * rg + p) * LD + 16 * (2 * wave + blk) + 4 * g) = acc[blk][rg];
  }
  __syncthreads();
  SCGG_NSTAMP(3);
  // ---- X = normX1(X + newX): registers, and planes(R1) as the next operand
  tile_add_layernorm_x3<DX, LD, KS, PLANE, RPW>(res, reinterpret_cast<const float*>(R2), S1, a.ln1_g, a.ln1_b, wave, lane,
                                                nullptr, grow, gok);
  __syncthreads();
  SCGG_NSTAMP(4);
  // ---- linX1 + relu: planes(R1) -> planes(R2)
  __bf16* S2 = reinterpret_cast<__bf16*>(R2);
  if (wave < NPF) {
    f32x4 acc[2][NRG];
    bias_acc(acc, a.b_lin1, 2 * wave);
    tile_linear_x3<KCX, KCF, KS, PLANE, NRG>(acc, ring, w_lin1, w_lin2, S1, p, g);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = fmaxf(acc[blk][rg][e], 0.f);
        put_split4<KS, PLANE>(S2, 16 * rg + p, 16 * (2 * wave + blk) + 4 * g, t);
      }
  }
  __syncthreads();
  SCGG_NSTAMP(5);
  // ---- linX2: planes(R2) -> fp32(R1)
  if (wave < NPX) {
    float* F = reinterpret_cast<float*>(R1);
    f32x4 acc[2][NRG];
    bias_acc(acc, a.b_lin2, 2 * wave);
    tile_linear_x3<KCF, KCX, KS, PLANE, NRG>(acc, ring, w_lin2, w_qkv0, S2, p, g);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg)
        *reinterpret_cast<f32x4*>(F + (16 * rg + p) * LD + 16 * (2 * wave + blk) + 4 * g) = acc[blk][rg];
  }
  __syncthreads();
  SCGG_NSTAMP(6);
  // ---- X = normX2(X + ff): global and planes(R2) (every row of the tile is a valid node, so the node mask is 1)
  tile_add_layernorm_x3<DX, LD, KS, PLANE, RPW>(res, reinterpret_cast<const float*>(R1), S2, a.ln2_g, a.ln2_b, wave, lane,
                                                a.X_out, grow, gok);
  SCGG_NSTAMP(8);
  if (!has_qkv) return;
  __syncthreads();
  SCGG_NSTAMP(7);
  // ---- trailing Linear on the new X (q | k | v of the next layer, or mlp_out_X.0 + ReLU after the last): planes(R2) -> global
  {
    long long qrow[NRG];
    bool qok[NRG];
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg) {
      const int ridx = tile * TR + 16 * rg + p;
      qok[rg] = ridx < count;
      qrow[rg] = a.rows[qok[rg] ? ridx : count - 1];
    }
    for (int pr = wave; pr < tail_pairs; pr += 8) {
      f32x4 acc[2][NRG];
      bias_acc(acc, a.b_qkv, 2 * pr);
      const nu32x4* wn = pr + 8 < tail_pairs ? w3_block(a.W3_qkv, 2 * (pr + 8), KCX, lane) : nullptr;
      tile_linear_x3<KCX, KCX, KS, PLANE, NRG>(acc, ring, w3_block(a.W3_qkv, 2 * pr, KCX, lane), wn, S2, p, g);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg)
          if (qok[rg]) {
            f32x4 o = acc[blk][rg];
            if (a.tail_relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            *reinterpret_cast<f32x4*>(a.qkv + qrow[rg] * tail_ld + 16 * (2 * pr + blk) + 4 * g) = o;
          }
    }
  }
  SCGG_NSTAMP(9);
}

// ---------------------------------------------------------------------------------------------------------
// Row-list Linear on the same scheme (mlp_in_X.3 and layer 0's q|k|v, the two Linears in front of the first edge kernel):
// C[row, :N] = act(A[row, :K] . W^T + bias) for the valid rows.  48-row tiles; the operand rows are split into planes once,
// each wave walks output block pairs w, w + 8, ...
// ---------------------------------------------------------------------------------------------------------
template <int KC, int NRG, int D>
__device__ __forceinline__ void node_linear_x3_body(const NodeLinearArgs& a, const f32x4* W3) {
  constexpr int K = 32 * KC, TR = 16 * NRG, KS = K + 8, PLANE = TR * KS, RPW = TR / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* S = reinterpret_cast<__bf16*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int count = *a.row_count;
  const int tile = blockIdx.x;
  if (tile * TR >= count) return;
  const int npairs = (a.N + 31) / 32;
  NWRingT<D> ring;                                     // D chunks of this wave's weight stream in flight: the stream is what the kernel waits for
  ring.next = nullptr;
  if (wave < npairs) w3_prime<KC>(ring, w3_block(W3, 2 * wave, KC, lane));
  if (K <= 256 && a.ln_h != nullptr) {                  // wave-uniform; one float4 per lane and row
    constexpr int RH = RPW / 2;                         // rows per batch: every load of a batch is issued before its arithmetic
    const int c = lane * 4;
    const bool in = c < K;
#pragma unroll
    for (int t0 = 0; t0 < RPW; t0 += RH) {
      f32x4 v[RH], o[RH], gt[RH], sc[RH], sh[RH];
      long long row[RH];
#pragma unroll
      for (int t = 0; t < RH; ++t) {
        const int ridx = tile * TR + wave * RPW + t0 + t;
        row[t] = a.rows[ridx < count ? ridx : count - 1];
      }
#pragma unroll
      for (int t = 0; t < RH; ++t) {
        const size_t tt = (size_t)(a.ln_t ? a.ln_t[row[t] / a.ln_n] : a.ln_tc) * a.ln_ld;
        v[t] = o[t] = gt[t] = sc[t] = sh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (in) {
          if (a.ln_x != nullptr) {               // first block: the residual row is the embedding of the raw input row
            v[t] = *reinterpret_cast<const f32x4*>(a.ln_be + c);
            for (int j = 0; j < a.ln_cin; ++j) {
              const float xj = a.ln_x[row[t] * a.ln_cin + j];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[t][e] += xj * a.ln_We[(c + e) * a.ln_cin + j];
            }
          } else v[t] = *reinterpret_cast<const f32x4*>(a.ln_h + row[t] * K + c);
          if (a.ln_o != nullptr) {
            o[t] = *reinterpret_cast<const f32x4*>(a.ln_o + row[t] * K + c);
            gt[t] = *reinterpret_cast<const f32x4*>(a.ln_gate + tt + c);
          }
          sc[t] = *reinterpret_cast<const f32x4*>(a.ln_scale + tt + c);
          sh[t] = *reinterpret_cast<const f32x4*>(a.ln_shift + tt + c);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < RH; ++t) {
        const int ridx = tile * TR + wave * RPW + t0 + t;
        v[t] += gt[t] * o[t];
        if (in && (a.ln_o != nullptr || a.ln_x != nullptr) && ridx < count)      // the filler rows of the last tile repeat a row
          *reinterpret_cast<f32x4*>(a.ln_h + row[t] * K + c) = v[t];
        const float mean = wave_sum64((v[t][0] + v[t][1]) + (v[t][2] + v[t][3])) * (1.0f / K);
        f32x4 d = v[t] - mean;
        if (!in) d = f32x4{0.f, 0.f, 0.f, 0.f};
        const float rstd = 1.0f / sqrtf(wave_sum64((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / K) + a.ln_eps);
        if (in) put_split4<KS, PLANE>(S, wave * RPW + t0 + t, c, d * rstd * (sc[t] + 1.0f) + sh[t]);
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int ridx = tile * TR + wave * RPW + t;
      const long long row = a.rows[ridx < count ? ridx : count - 1];
      for (int c = lane * 4; c < K; c += 256)
        put_split4<KS, PLANE>(S, wave * RPW + t, c, *reinterpret_cast<const f32x4*>(a.A + row * a.lda + c));
    }
  }
  long long qrow[NRG];
  bool qok[NRG];
#pragma unroll
  for (int rg = 0; rg < NRG; ++rg) {
    const int ridx = tile * TR + 16 * rg + p;
    qok[rg] = ridx < count;
    qrow[rg] = a.rows[qok[rg] ? ridx : count - 1];
  }
  __syncthreads();
  for (int pr = wave; pr < npairs; pr += 8) {
    f32x4 acc[2][NRG];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + 16 * (2 * pr + blk) + 4 * g);
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) acc[blk][rg] = bv;
    }
    const nu32x4* wn = pr + 8 < npairs ? w3_block(W3, 2 * (pr + 8), KC, lane) : nullptr;
    tile_linear_x3<KC, KC, KS, PLANE, NRG>(acc, ring, w3_block(W3, 2 * pr, KC, lane), wn, S, p, g);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int c0 = 16 * (2 * pr + blk) + 4 * g;
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) {
        f32x4 o = acc[blk][rg];
        if (a.relu == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        } else if (a.relu == 2) {                     // tanh-approximated GELU (the SiT MLP)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = 0.5f * o[e] * (1.0f + tanhf(0.7978845608028654f * (o[e] + 0.044715f * o[e] * o[e] * o[e])));
        }
        if (qok[rg] && c0 < a.N) {
          float* dst = a.C + qrow[rg] * a.ldc + c0;
          if (a.relu == 3) o += *reinterpret_cast<const f32x4*>(dst);      // accumulate: one K slice of a longer product
          *reinterpret_cast<f32x4*>(dst) = o;
        }
      }
    }
  }
}

template <int KC, int NRG, int D>
__global__ __launch_bounds__(512) void node_linear_x3_kernel(NodeLinearArgs a, const f32x4* W3) {
  node_linear_x3_body<KC, NRG, D>(a, W3);
}
// two Linears of the same K on the same row list in one launch (blockIdx.y picks one): the e_x_mul / pos_x_mul pair of the
// position-aware block
template <int KC, int NRG, int D>
__global__ __launch_bounds__(512) void node_linear_x3_pair_kernel(NodeLinearArgs a0, const f32x4* W0, NodeLinearArgs a1, const f32x4* W1) {
  if (blockIdx.y == 0) node_linear_x3_body<KC, NRG, D>(a0, W0);
  else node_linear_x3_body<KC, NRG, D>(a1, W1);
}

template <int KC, int NRG = 3, int D = NWD>
static int launch_linear_x3_pair(const NodeLinearArgs& a0, const f32x4* W0, const NodeLinearArgs& a1, const f32x4* W1, int M, hipStream_t st) {
  constexpr int TR = 16 * NRG;
  constexpr size_t lds = (size_t)3 * TR * (32 * KC + 8) * 2;
  static LdsAttrOnce attr;
  auto kern = node_linear_x3_pair_kernel<KC, NRG, D>;
  if (set_max_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds)) return -2;
  hipLaunchKernelGGL(kern, dim3((M + TR - 1) / TR, 2), dim3(512), lds, st, a0, W0, a1, W1);
  return 0;
}

template <int KC, int NRG = 3, int D = NWD>
static int launch_linear_x3(const NodeLinearArgs& a, const f32x4* W3, int M, hipStream_t st) {
  constexpr int TR = 16 * NRG;
  constexpr size_t lds = (size_t)3 * TR * (32 * KC + 8) * 2;
  static LdsAttrOnce attr;
  auto kern = node_linear_x3_kernel<KC, NRG, D>;
  if (set_max_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds)) return -2;
  hipLaunchKernelGGL(kern, dim3((M + TR - 1) / TR), dim3(512), lds, st, a, W3);
  return 0;
}

// SCGG_NODE_MODE=f32 selects the plain fp32-MFMA kernel (development A/B switch); default: the x3 kernel
static bool node_x3() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("SCGG_NODE_MODE");
    mode = (e != nullptr && e[0] == 'f') ? 0 : 1;
  }
  return mode == 1;
}

template <int DXB, int FXB>
static int launch_one(const NodeLayerArgs& a, int M, hipStream_t st) {
  static LdsAttrOnce attr_x3, attr_f32;
  const bool x3 = node_x3() && a.W3_xout && a.W3_lin1 && a.W3_lin2;
  if (a.newX_in != nullptr && !x3) return -1;      // the fp32 kernel has no such mode
  auto kern = x3 ? node_layer_x3_kernel<DXB, FXB> : node_layer_kernel<DXB, FXB>;
  const size_t lds = x3 ? NodeGeomX3<DXB, FXB>::LDS_BYTES : NodeGeom<DXB, FXB>::LDS_BYTES;
  if (set_max_dynamic_lds(x3 ? attr_x3 : attr_f32, reinterpret_cast<const void*>(kern), lds)) return -2;
  const int rows_per_tile = x3 ? NodeGeomX3<DXB, FXB>::TR : 32;
  hipLaunchKernelGGL(kern, dim3((M + rows_per_tile - 1) / rows_per_tile), dim3(512), lds, st, a);
  return 0;
}

bool node_layer_x3_enabled() { return node_x3(); }

#ifdef SCGG_STAMP_NODE
extern "C" int scgg_debug_node_stamps_reset(void) {
  static unsigned long long zero[4 * 2 * 16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(scgg_node_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
extern "C" int scgg_debug_node_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(scgg_node_stamps), sizeof(unsigned long long) * 4 * 2 * 16) == hipSuccess ? 0 : -1;
}
#endif

bool node_layer_supported(int dx, int ffx) { return (dx == 256 && ffx == 256) || (dx == 128 && ffx == 128); }

// row-list Linear through the x3 kernel when the shape allows (K = 128, 192, 256 or 512, N a multiple of 4, padded bias), else -1
int launch_node_linear_x3(const NodeLinearArgs& a, const f32x4_t* W3, int M, hipStream_t st) {
  if (!node_x3() || W3 == nullptr || a.lda % 4 != 0 || a.ldc % 4 != 0 || a.N % 4 != 0) return -1;
  if (a.ln_h != nullptr && a.K > 256) return -1;
  if (a.K == 512) return launch_linear_x3<16>(a, W3, M, st);
  if (a.K == 256) return launch_linear_x3<8>(a, W3, M, st);
  if (a.K == 192) return launch_linear_x3<6>(a, W3, M, st);
  if (a.K == 128) return launch_linear_x3<4>(a, W3, M, st);
  return -1;
}

// two row-list Linears of the same K in one launch; -1: not covered (launch them one by one)
int launch_node_linear_x3_pair(const NodeLinearArgs& a0, const f32x4_t* W0, const NodeLinearArgs& a1, const f32x4_t* W1, int M,
                               hipStream_t st) {
  if (!node_x3() || W0 == nullptr || W1 == nullptr || a0.K != a1.K || a0.ln_h != nullptr || a1.ln_h != nullptr) return -1;
  for (const NodeLinearArgs* a : {&a0, &a1})
    if (a->lda % 4 != 0 || a->ldc % 4 != 0 || a->N % 4 != 0) return -1;
  if (a0.K == 256) return launch_linear_x3_pair<8>(a0, W0, a1, W1, M, st);
  if (a0.K == 128) return launch_linear_x3_pair<4>(a0, W0, a1, W1, M, st);
  return -1;
}

int launch_node_layer(const NodeLayerArgs& a, int dx, int ffx, int M, hipStream_t st) {
  if (dx == 256 && ffx == 256) return launch_one<8, 8>(a, M, st);
  if (dx == 128 && ffx == 128) return launch_one<4, 4>(a, M, st);
  return -1;
}

}  // namespace scgg


// ---------------------------------------------------------------------------------------------------------
// C ABI: a Linear over a list of rows on the bf16x3 scheme (fp32 results), for callers outside the engines (stress.py: the
// q | k | v and x_out Linears of the bf16 NodeEdgeBlock, which went through rocBLAS until round 3).
// image = [fp32 fragments NB KB 256 float4 | three-term bf16 image NB (K/16) 192 float4 | bias padded to 32 NB floats]
// ---------------------------------------------------------------------------------------------------------
namespace {
struct RowLinearImage { size_t w, w3, b, bytes; int NB, KB; };
RowLinearImage row_linear_image(int N, int K) {
  RowLinearImage im;
  im.NB = (N + 31) / 32; im.KB = K / 32;
  im.w = 0;
  im.w3 = (size_t)im.NB * im.KB * 256 * 16;
  im.b = im.w3 + (size_t)im.NB * (K / 16) * 192 * 16;
  im.bytes = im.b + (size_t)im.NB * 32 * 4;
  return im;
}
bool row_linear_ok(int N, int K) { return N > 0 && (K == 32 || K == 64 || K == 128 || K == 256); }
}  // namespace

extern "C" size_t scgg_linear_rows_image_bytes(int32_t N, int32_t K) { return row_linear_ok(N, K) ? row_linear_image(N, K).bytes : 0; }

extern "C" int scgg_linear_rows_pack(const float* W, const float* b, int32_t N, int32_t K, void* image, void* stream) {
  if (!W || !b || !image) return scgg::fail("linear_rows_pack: null argument");
  if (!row_linear_ok(N, K)) return scgg::fail("linear_rows_pack: K=%d is not one of 32 / 64 / 128 / 256", K);
  if (reinterpret_cast<uintptr_t>(image) % 16 != 0) return scgg::fail("linear_rows_pack: image must be 16-byte aligned");
  const RowLinearImage im = row_linear_image(N, K);
  char* base = static_cast<char*>(image);
  hipStream_t st = static_cast<hipStream_t>(stream);
  scgg::PackArgs pa{W, reinterpret_cast<scgg::f32x4_t*>(base + im.w), N, K, im.NB, im.KB};
  scgg::PackArgs p3{W, reinterpret_cast<scgg::f32x4_t*>(base + im.w3), N, K, im.NB, im.KB};
  if (scgg::launch_pack_linear(pa, st) || scgg::launch_pack_linear_x3(p3, st) ||
      scgg::launch_pad_vec(b, reinterpret_cast<float*>(base + im.b), N, im.NB * 32, 0.f, st))
    return scgg::fail("linear_rows_pack: launch failed");
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : scgg::fail("linear_rows_pack: %s", hipGetErrorString(e));
}

extern "C" int scgg_linear_rows(const float* A, int32_t lda, const void* image, int32_t N, int32_t K, float* C, int32_t ldc,
                                const int32_t* rows, const int32_t* row_count, int32_t max_rows, int32_t relu, void* stream) {
  if (!A || !image || !C || !rows || !row_count) return scgg::fail("linear_rows: null argument");
  if (!row_linear_ok(N, K) || lda < K || ldc < N || max_rows <= 0 || relu < 0 || relu > 2) return scgg::fail("linear_rows: bad shape");
  const RowLinearImage im = row_linear_image(N, K);
  const char* base = static_cast<const char*>(image);
  const int rbw = im.NB >= 24 ? 8 : (im.NB >= 4 ? 2 : 1);
  scgg::NodeLinearArgs a{A, reinterpret_cast<const scgg::f32x4_t*>(base + im.w), reinterpret_cast<const float*>(base + im.b), C, rows,
                         row_count, N, K, im.NB, lda, ldc, rbw, relu};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (scgg::launch_node_linear_x3(a, reinterpret_cast<const scgg::f32x4_t*>(base + im.w3), max_rows, st) != 0) {
    if (relu > 1 || scgg::launch_node_linear(a, max_rows, st)) return scgg::fail("linear_rows: shape N=%d K=%d not covered", N, K);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : scgg::fail("linear_rows: %s", hipGetErrorString(e));
}
