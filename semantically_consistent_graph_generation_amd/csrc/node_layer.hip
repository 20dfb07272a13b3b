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

namespace scgg {

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

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
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
  tile_add_layernorm<DX, LD>(res, bufA, bufB, a.ln2_g, a.ln2_b, wave, lane, a.X, grow, gok);
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

template <int DXB, int FXB>
static int launch_one(const NodeLayerArgs& a, int M, hipStream_t st) {
  using G = NodeGeom<DXB, FXB>;
  static bool attr_set = false;
  auto kern = node_layer_kernel<DXB, FXB>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)G::LDS_BYTES) != hipSuccess)
      return -2;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((M + 31) / 32), dim3(512), G::LDS_BYTES, st, a);
  return 0;
}

bool node_layer_supported(int dx, int ffx) { return (dx == 256 && ffx == 256) || (dx == 128 && ffx == 128); }

int launch_node_layer(const NodeLayerArgs& a, int dx, int ffx, int M, hipStream_t st) {
  if (dx == 256 && ffx == 256) return launch_one<8, 8>(a, M, st);
  if (dx == 128 && ffx == 128) return launch_one<4, 4>(a, M, st);
  return -1;
}

}  // namespace scgg
