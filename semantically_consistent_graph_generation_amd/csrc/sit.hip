// Node-coordinate DDPM of the reference: the SiT-T denoiser and its ancestral sampler, gfx950.
//
//   SiT.forward, SiTBlock (adaLN-zero), FinalLayer, TimestepEmbedder, AttentionWithPads
//                              src/bootstrap/node_coordinate_diffusion/sit_model.py:41-77, :80-103, :112-154, :228-247
//   NodeCoordsDiffusionModel   src/bootstrap/node_coordinate_diffusion/cow_diff_nodes_with_3d_checks.py:22-81
//   called once per sampled batch from sample_limit_dist (src/ConStruct/diffusion/noise_model.py:323-325)
//
// What shapes the design: 999 ancestral steps of a SMALL transformer (6 blocks, hidden 192, 6 heads) on B graphs of n <= a
// few dozen tokens, and every graph of a step sees the same timestep.
//   * everything that depends on t only -- the timestep embedding, its MLP, the SiLU and the adaLN Linear of every block
//     and of the final layer -- is computed for ALL timesteps once per weight set (finalize: GEMMs with M = noise_steps)
//     into tables; a step reads its six vectors per block from them.
//   * a step works on the valid rows only (row list from the mask).  Block 0's entry is a row-list bf16x3 Linear for q|k|v with
//     [pos_embedder + LayerNorm + modulate] as its prologue (sit_lnmod_kernel when that Linear's shape is not built); attention is one wave per (graph, head) on the
//     matrix pipe for n <= 64 (transposed scores, so the normalised P^T registers are the operand of P.V), else with the
//     head's keys / values of a 64-key chunk in LDS and an online softmax (padded keys are skipped, which is exact: the
//     reference gives them -1e9 and a valid query always has itself as a key); everything from an attention to the next
//     one -- proj, gated residual, LayerNorm-modulate, fc1 + GELU, fc2, gated residual, the next block's LayerNorm-modulate
//     and q|k|v -- is ONE launch (sit_block.hip); [last gated residual + final LayerNorm / modulate + the 192 -> 3 Linear +
//     the DDPM update x_{t-1} = ... + sqrt(beta) z] is one kernel.  14 launches per step, no synchronisation, no
//     allocation.  Widths the fused kernel is not built for run as separate row-list Linears (with the LayerNorm-modulate
//     as their prologue and GELU / accumulate epilogues) or, below those, through the generic fp32-MFMA GEMM.
//   * the normal draws are Philox4x32-10 + Box-Muller keyed by (seed, graph id, node, step), or a caller-provided buffer
//     (tests: the oracle takes the same draws).  The reference draws torch.randn from the global generator.
// Padded tokens: the reference lets them evolve too (they attend uniformly to every slot); nothing downstream reads them
// (sample_limit_dist masks the positions), so the step kernels work on the valid rows and leave the others at zero.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/scgg.h"
#include "c16.h"
#include "kernels.h"
#include "rng.h"

namespace scgg {
namespace {

// emb[t][0:half] = cos(t f_k), emb[t][half:] = sin(t f_k), f_k = exp(-ln(10000) k / half)     (sit_model.py:53-72)
__global__ void sit_temb_kernel(float* emb, int T, int dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, half = dim / 2;
  if (idx >= T * half) return;
  const int t = idx / half, k = idx - t * half;
  const float f = expf(-logf(10000.0f) * (float)k / (float)half);
  const float a = (float)t * f;
  emb[(size_t)t * dim + k] = cosf(a);
  emb[(size_t)t * dim + half + k] = sinf(a);
}
__global__ void sit_silu_kernel(const float* in, float* out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = in[i]; out[i] = v / (1.0f + expf(-v)); }
}
// tanh-approximated GELU in place (nn.GELU(approximate="tanh"), sit_model.py:122)
__global__ void sit_gelu_kernel(float* x, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  x[i] = 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
}

struct LnModArgs {
  float* h;              // [M,H] residual stream, updated in place when o != nullptr
  const float* o;        // [M,H] output of the previous sub-layer (or null)
  const float* tab;      // modulation table of this block: [T][ld]
  int ld, off_gate, off_shift, off_scale;   // columns of the gate of the previous sub-layer / this sub-layer's shift, scale
  const float* tab_gate; // table the gate comes from (the previous block's for a block's first sub-layer), same ld
  float* u;              // [M,H] = LN(h) (1 + scale) + shift
  const int* rows;
  const int* row_count;
  const int32_t* t;      // [B] timestep per graph, or null: t_const for all
  int t_const, n;
  const float* x;        // block 0 only: [M,cin] sample; h = pos_embedder(x) is formed (and stored) here    (sit_model.py:182)
  const float* We;       // [H,cin]
  const float* be;       // [H]
  int cin;
};

// 16 lanes per row, F4 float4 per lane (H = 64 F4)
template <int F4>
__global__ __launch_bounds__(256) void sit_lnmod_kernel(LnModArgs a) {
  constexpr int H = 64 * F4;
  const int ridx = blockIdx.x * 16 + (threadIdx.x >> 4), p = threadIdx.x & 15;
  const int count = *a.row_count;
  const bool on = ridx < count;
  const long long row = a.rows[on ? ridx : 0];
  const int t = a.t ? a.t[row / a.n] : a.t_const;
  f32x4 v[F4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < F4; ++k) {
    const int c = 64 * k + 4 * p;
    if (a.x != nullptr) {
      v[k] = ld4(a.be + c);
      for (int j = 0; j < a.cin; ++j) {
        const float xj = a.x[row * a.cin + j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k][e] += xj * a.We[(c + e) * a.cin + j];
      }
    } else v[k] = ld4(a.h + row * H + c);
    if (a.o != nullptr) v[k] += ld4(a.tab_gate + (size_t)t * a.ld + a.off_gate + c) * ld4(a.o + row * H + c);
    s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  const float mean = row_sum16(s) * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < F4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(row_sum16(q) * (1.0f / H) + 1e-6f);
  if (!on) return;
#pragma unroll
  for (int k = 0; k < F4; ++k) {
    const int c = 64 * k + 4 * p;
    if (a.o != nullptr || a.x != nullptr) *reinterpret_cast<f32x4*>(a.h + row * H + c) = v[k];
    const f32x4 sh = ld4(a.tab + (size_t)t * a.ld + a.off_shift + c), sc = ld4(a.tab + (size_t)t * a.ld + a.off_scale + c);
    f32x4 u;
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = (v[k][e] - mean) * rstd * (1.0f + sc[e]) + sh[e];
    *reinterpret_cast<f32x4*>(a.u + row * H + c) = u;
  }
}

// One wave per (graph, head).  The head's K / V rows of up to 64 valid keys are staged into LDS by the whole wave (HD/4
// lanes per row, float4 each); lane = (query token, half of the head's channels): both halves of a query form the full
// score from broadcast K reads and accumulate their own half of the weighted V sum, online softmax over the key chunks.
// qkv row = [q | k | v][heads][HD] (sit_model.py:87-88); out[row][head HD + d].
template <int HD>
__global__ __launch_bounds__(64) void sit_attn_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                      float* __restrict__ out, int n, int H, float scale) {
  constexpr int LD = HD + 4, Q4 = HD / 4, HV = HD / 2, LPR = HD / 4, RPP = 64 / LPR;
  __shared__ __align__(16) float ks[64 * LD], vs[64 * LD];
  __shared__ int slot_of[64];
  const int b = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  const int qi = lane & 31, half = lane >> 5;
  const uint8_t* mb = mask + (long long)b * n;
  const float* base = qkv + (long long)b * n * 3 * H + head * HD;
  for (int i0 = 0; i0 < n; i0 += 32) {
    const int i = i0 + qi;
    const bool qon = i < n && mb[i] != 0;
    f32x4 q[Q4];
    float acc[HV];
#pragma unroll
    for (int k = 0; k < Q4; ++k) q[k] = qon ? ld4(base + (long long)i * 3 * H + 4 * k) * scale : splat4(0.f);
#pragma unroll
    for (int d = 0; d < HV; ++d) acc[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int j = j0 + lane;
      const bool kon = j < n && mb[j] != 0;
      const unsigned long long bal = __ballot(kon);
      const int cnt = __popcll(bal);
      __syncthreads();
      slot_of[lane] = kon ? __popcll(bal & ((1ull << lane) - 1ull)) : -1;
      __syncthreads();
      const int rows_here = min(64, n - j0);
      for (int r0 = 0; r0 < rows_here; r0 += RPP) {
        const int r = r0 + lane / LPR, c4 = (lane % LPR) * 4;
        const int sl = r < rows_here ? slot_of[r] : -1;
        if (sl >= 0) {
          const float* kr = base + (long long)(j0 + r) * 3 * H + H + c4;
          *reinterpret_cast<f32x4*>(&ks[sl * LD + c4]) = ld4(kr);
          *reinterpret_cast<f32x4*>(&vs[sl * LD + c4]) = ld4(kr + H);
        }
      }
      __syncthreads();
      for (int c = 0; c < cnt; ++c) {
        f32x4 s4 = splat4(0.f);
#pragma unroll
        for (int k = 0; k < Q4; ++k) s4 += q[k] * *reinterpret_cast<const f32x4*>(&ks[c * LD + 4 * k]);
        const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        const float mn = fmaxf(m, s), fo = __expf(m - mn), fn = __expf(s - mn);
        l = l * fo + fn;
#pragma unroll
        for (int k = 0; k < HV / 4; ++k) {
          const f32x4 vv = *reinterpret_cast<const f32x4*>(&vs[c * LD + half * HV + 4 * k]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[4 * k + e] = acc[4 * k + e] * fo + fn * vv[e];
        }
        m = mn;
      }
    }
    if (qon) {
      float* o = out + ((long long)b * n + i) * H + head * HD + half * HV;
      const float inv = 1.0f / l;
#pragma unroll
      for (int k = 0; k < HV / 4; ++k) {
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = acc[4 * k + e] * inv;
        *reinterpret_cast<f32x4*>(o + 4 * k) = ov;
      }
    }
  }
}

// The same attention on the matrix pipe for n <= 64 (v_mfma_f32_16x16x4_f32, fp32 operands): one wave per (graph, head), no
// LDS.  Scores are formed transposed, S^T[key][query] = K . Q^T (A = K tile, B = Q tile; a lane holds HD/4 consecutive
// channels of its key / query row, two float4 loads), so a lane ends up with query (lane & 15) and keys 4 (lane >> 4) + r
// of every 16-key tile: the softmax over keys is in-lane plus two lane exchanges, and the normalised P^T registers are, as
// they stand, the A operand of P . V (k index <-> key 4 g + r); V is read as the B operand (lane = channel, 4 keys).
// Masked keys: score -inf, V taken as 0 (their q|k|v rows are never written).
template <int HD>
__global__ __launch_bounds__(256) void sit_attn_mfma_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                            float* __restrict__ out, int n, int H, int n_heads, int total, float scale) {
  constexpr int KST = HD / 4, DT = HD / 16, MAXT = 4;
  const int widx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (widx >= total) return;                                   // wave-uniform
  const int b = widx / n_heads, head = widx % n_heads, lane = threadIdx.x & 63, x = lane & 15, g = lane >> 4;
  const int nt = (n + 15) >> 4;
  const uint8_t* mb = mask + (long long)b * n;
  const float* base = qkv + (long long)b * n * 3 * H + head * HD;
  float kf[MAXT][KST], vf[MAXT][DT][4];
  bool kv[MAXT][4];                                            // validity of key 16 kt + 4 g + r
#pragma unroll
  for (int kt = 0; kt < MAXT; ++kt) {
    if (kt >= nt) continue;
    const int key = 16 * kt + x;
    const bool ok = key < n && mb[key] != 0;
#pragma unroll
    for (int q4 = 0; q4 < KST / 4; ++q4) {
      f32x4 t = splat4(0.f);
      if (ok) t = ld4(base + (long long)key * 3 * H + H + KST * g + 4 * q4);
#pragma unroll
      for (int e = 0; e < 4; ++e) kf[kt][4 * q4 + e] = t[e];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kr = 16 * kt + 4 * g + r;
      kv[kt][r] = kr < n && mb[kr] != 0;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) vf[kt][dt][r] = kv[kt][r] ? base[(long long)kr * 3 * H + 2 * H + 16 * dt + x] : 0.f;
    }
  }
#pragma unroll
  for (int qt = 0; qt < MAXT; ++qt) {
    if (qt >= nt) continue;
    const int qi = 16 * qt + x;
    const bool qok = qi < n && mb[qi] != 0;
    float qf[KST];
#pragma unroll
    for (int q4 = 0; q4 < KST / 4; ++q4) {
      f32x4 t = splat4(0.f);
      if (qok) t = ld4(base + (long long)qi * 3 * H + KST * g + 4 * q4) * scale;
#pragma unroll
      for (int e = 0; e < 4; ++e) qf[4 * q4 + e] = t[e];
    }
    f32x4 sT[MAXT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
      if (kt >= nt) continue;
      f32x4 acc = splat4(0.f);
#pragma unroll
      for (int s = 0; s < KST; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][s], qf[s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[r] = kv[kt][r] ? acc[r] : -INFINITY;
        m = fmaxf(m, acc[r]);
      }
      sT[kt] = acc;
    }
    m = swap_rows16_max(m);                                    // lane ^ 16 (v_permlane16_swap)
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
      if (kt >= nt) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sT[kt][r] = __expf(sT[kt][r] - m);                     // masked keys: exp(-inf) = 0
        l += sT[kt][r];
      }
    }
    l = other_half_sum(swap_rows16_sum(l));                    // lane ^ 16, lane ^ 32 (v_permlane16_swap / 32_swap)
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 acc = splat4(0.f);
#pragma unroll
      for (int kt = 0; kt < MAXT; ++kt) {
        if (kt >= nt) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[kt][r] * inv, vf[kt][dt][r], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)                              // acc[r]: query 16 qt + 4 g + r, channel 16 dt + x
        if (kv[qt][r]) out[((long long)b * n + 16 * qt + 4 * g + r) * H + head * HD + 16 * dt + x] = acc[r];
    }
  }
}

struct UpdateArgs {
  const float* h;        // [M,H]
  const float* o;        // [M,H] last block's MLP output
  const float* tab_gate; // last block's table (gate_mlp at off_gate)
  int ld_gate, off_gate;
  const float* tabF;     // final layer's table [T][2H]: shift | scale
  const float* Wf;       // [3,H]
  const float* bf;       // [3]
  const float* x;        // [M,3] current sample
  float* x_out;          // [M,3]
  float* eps_out;        // [M,3] predicted noise (optional: the forward-only entry point)
  const float* noise;    // [M,3] normal draws of this step, or null: Philox
  const int* rows;
  const int* row_count;
  const int32_t* t;
  int t_const, n;
  float c_x, c_eps, c_noise;   // x_{t-1} = c_x (x - c_eps eps) + c_noise z       (cow_diff_nodes_with_3d_checks.py:76-79)
  int do_update;
  uint32_t k0, k1, step, graph_id_base;
};

// Three standard normals of (node, global graph id, step word): Philox4x32-10 + Box-Muller.  Keyed by the GLOBAL graph id, so
// a graph's draws do not depend on which shard or batch slot it sits in.
__device__ __forceinline__ void philox_normal3(uint32_t node, uint32_t graph, uint32_t step, uint32_t k0, uint32_t k1, float (&z)[3]) {
  uint32_t r[4];
  philox4x32_10(node, graph, step, 0x51u, k0, k1, r);
  const float r0 = sqrtf(-2.0f * logf(1.0f - u01(r[0]))), r1 = sqrtf(-2.0f * logf(1.0f - u01(r[2])));
  const float a0 = 6.283185307179586f * u01(r[1]), a1 = 6.283185307179586f * u01(r[3]);
  z[0] = r0 * cosf(a0); z[1] = r0 * sinf(a0); z[2] = r1 * cosf(a1);
}

__global__ __launch_bounds__(256) void philox_normal_rows_kernel(float* out, int rows, int n, uint32_t k0, uint32_t k1, uint32_t step,
                                                                 uint32_t graph_id_base) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  float z[3];
  philox_normal3((uint32_t)(row % n), graph_id_base + (uint32_t)(row / n), step, k0, k1, z);
  out[(size_t)row * 3] = z[0]; out[(size_t)row * 3 + 1] = z[1]; out[(size_t)row * 3 + 2] = z[2];
}

template <int F4>
__global__ __launch_bounds__(256) void sit_update_kernel(UpdateArgs a) {
  constexpr int H = 64 * F4;
  const int ridx = blockIdx.x * 16 + (threadIdx.x >> 4), p = threadIdx.x & 15;
  const int count = *a.row_count;
  const bool on = ridx < count;
  const long long row = a.rows[on ? ridx : 0];
  const int b = (int)(row / a.n), i = (int)(row - (long long)b * a.n);
  const int t = a.t ? a.t[b] : a.t_const;
  f32x4 v[F4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < F4; ++k) {
    const int c = 64 * k + 4 * p;
    v[k] = ld4(a.h + row * H + c) + ld4(a.tab_gate + (size_t)t * a.ld_gate + a.off_gate + c) * ld4(a.o + row * H + c);
    s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  const float mean = row_sum16(s) * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < F4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(row_sum16(q) * (1.0f / H) + 1e-6f);
  float e3[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < F4; ++k) {
    const int c = 64 * k + 4 * p;
    const f32x4 sh = ld4(a.tabF + (size_t)t * (2 * H) + c), sc = ld4(a.tabF + (size_t)t * (2 * H) + H + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u = (v[k][e] - mean) * rstd * (1.0f + sc[e]) + sh[e];
#pragma unroll
      for (int o = 0; o < 3; ++o) e3[o] += u * a.Wf[o * H + c + e];
    }
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) e3[o] = row_sum16(e3[o]) + a.bf[o];
  if (!on || p != 0) return;
  if (a.eps_out != nullptr) { a.eps_out[row * 3] = e3[0]; a.eps_out[row * 3 + 1] = e3[1]; a.eps_out[row * 3 + 2] = e3[2]; }
  if (!a.do_update) return;
  float z[3] = {0.f, 0.f, 0.f};
  if (a.c_noise != 0.f) {
    if (a.noise != nullptr) { z[0] = a.noise[row * 3]; z[1] = a.noise[row * 3 + 1]; z[2] = a.noise[row * 3 + 2]; }
    else {
      philox_normal3((uint32_t)i, a.graph_id_base + (uint32_t)b, a.step, a.k0, a.k1, z);
    }
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) a.x_out[row * 3 + o] = a.c_x * (a.x[row * 3 + o] - a.c_eps * e3[o]) + a.c_noise * z[o];
}

struct ParamS { const float* p; int64_t numel; };
struct Bump2 {
  char* base; size_t off = 0;
  explicit Bump2(char* b) : base(b) {}
  template <typename T> T* take(size_t count) {
    off = (off + 255) / 256 * 256;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

}  // namespace
}  // namespace scgg

using namespace scgg;

struct scgg_sit {
  scgg_sit_config c;
  std::map<std::string, ParamS> params;
  bool finalized = false;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  std::vector<float*> tab;     // per block [T][6H]
  float* tabF = nullptr;       // [T][2H]
  float *emb = nullptr, *t1 = nullptr, *cemb = nullptr, *csilu = nullptr;
  std::vector<float> beta, alpha, alpha_hat;   // host copies of the schedule
  struct Lin { f32x4_t* W3 = nullptr; float* b = nullptr; int N = 0, K = 0; };   // row-list bf16x3 image + padded bias
  struct Blk { Lin qkv, proj, fc1; std::vector<Lin> fc2; float* Wslice = nullptr; float* zero_b = nullptr; };
  std::vector<Blk> lin;
  int fc2_slice = 256;
};

namespace {

const ParamS* findS(const scgg_sit* h, const std::string& name, int64_t numel) {
  auto it = h->params.find(name);
  if (it == h->params.end()) { fail("missing parameter '%s'", name.c_str()); return nullptr; }
  if (it->second.numel != numel) {
    fail("parameter '%s' has %lld elements, expected %lld", name.c_str(), (long long)it->second.numel, (long long)numel);
    return nullptr;
  }
  return &it->second;
}
int gemmS(scgg_sit* h, const float* A, int lda, const std::string& name, float* C, int ldc, int M, int N, int K, hipStream_t st) {
  const ParamS *W = findS(h, name + ".weight", (int64_t)N * K), *b = findS(h, name + ".bias", N);
  if (!W || !b) return -1;
  GemmArgs g{A, W->p, b->p, C, nullptr, M, N, K, lda, K, ldc, 0, 0, 0, 0, 0};
  return launch_gemm(g, st);
}
int checkS(const char* what) {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("%s: %s", what, hipGetErrorString(e));
}

constexpr int TEMB = 256;   // frequency_embedding_size (sit_model.py:45)

int buildS(scgg_sit* h, bool fill, hipStream_t st) {
  const scgg_sit_config& c = h->c;
  const int T = c.noise_steps, H = c.hidden;
  Bump2 ar(fill ? h->arena : nullptr);
  h->emb = ar.take<float>((size_t)T * TEMB);
  h->t1 = ar.take<float>((size_t)T * H);
  h->cemb = ar.take<float>((size_t)T * H);
  h->csilu = ar.take<float>((size_t)T * H);
  h->tab.resize(c.n_blocks);
  for (int i = 0; i < c.n_blocks; ++i) h->tab[i] = ar.take<float>((size_t)T * 6 * H);
  h->tabF = ar.take<float>((size_t)T * 2 * H);
  // row-list bf16x3 images of the four Linears of every block (K = hidden must be one of the built widths; fc2 in K slices
  // of 256 accumulated into the output: its whole K does not fit the kernel's LDS)
  const int sw = 256;                                             // K slice of fc2
  const bool rowlin = (H == 128 || H == 192 || H == 256) && c.mlp_hidden % sw == 0;
  h->fc2_slice = sw;
  h->lin.assign(c.n_blocks, scgg_sit::Blk());
  auto take_lin = [&](scgg_sit::Lin& L, int N, int K) {
    const int NB = (N + 31) / 32;
    L.N = N; L.K = K;
    L.W3 = ar.take<f32x4_t>((size_t)NB * (K / 16) * 192);
    L.b = ar.take<float>((size_t)NB * 32);
  };
  if (rowlin)
    for (int i = 0; i < c.n_blocks; ++i) {
      scgg_sit::Blk& Bk = h->lin[i];
      take_lin(Bk.qkv, 3 * H, H); take_lin(Bk.proj, H, H); take_lin(Bk.fc1, c.mlp_hidden, H);
      Bk.fc2.assign(c.mlp_hidden / sw, scgg_sit::Lin());
      for (auto& L : Bk.fc2) take_lin(L, H, sw);
      Bk.Wslice = ar.take<float>((size_t)H * sw);
      Bk.zero_b = ar.take<float>((size_t)(H + 31) / 32 * 32);
    }
  h->arena_bytes = (ar.off + 255) / 256 * 256;
  if (!fill) return 0;
  if (rowlin) {
    auto fill_lin = [&](scgg_sit::Lin& L, const float* W, const float* b) {
      PackArgs p3{W, L.W3, L.N, L.K, (L.N + 31) / 32, L.K / 32};
      launch_pack_linear_x3(p3, st);
      launch_pad_vec(b, L.b, L.N, (L.N + 31) / 32 * 32, 0.f, st);
    };
    for (int i = 0; i < c.n_blocks; ++i) {
      const std::string p = "blocks." + std::to_string(i) + ".";
      scgg_sit::Blk& Bk = h->lin[i];
      const ParamS *wq = findS(h, p + "attn.qkv.weight", (int64_t)3 * H * H), *bq = findS(h, p + "attn.qkv.bias", 3 * H),
                   *wp = findS(h, p + "attn.proj.weight", (int64_t)H * H), *bp = findS(h, p + "attn.proj.bias", H),
                   *w1 = findS(h, p + "mlp.fc1.weight", (int64_t)c.mlp_hidden * H), *b1 = findS(h, p + "mlp.fc1.bias", c.mlp_hidden),
                   *w2 = findS(h, p + "mlp.fc2.weight", (int64_t)H * c.mlp_hidden), *b2 = findS(h, p + "mlp.fc2.bias", H);
      if (!wq || !bq || !wp || !bp || !w1 || !b1 || !w2 || !b2) return -1;
      fill_lin(Bk.qkv, wq->p, bq->p); fill_lin(Bk.proj, wp->p, bp->p); fill_lin(Bk.fc1, w1->p, b1->p);
      if (hipMemsetAsync(Bk.zero_b, 0, (size_t)(H + 31) / 32 * 32 * sizeof(float), st) != hipSuccess) return fail("memset failed");
      for (size_t s2 = 0; s2 < Bk.fc2.size(); ++s2) {          // slice s: columns [sw s, sw s + sw) of fc2.weight [H, mlp_hidden]
        launch_copy_cols(w2->p + sw * s2, c.mlp_hidden, Bk.Wslice, sw, H, sw, st);
        fill_lin(Bk.fc2[s2], Bk.Wslice, s2 == 0 ? b2->p : Bk.zero_b);
      }
    }
  }
  hipLaunchKernelGGL(sit_temb_kernel, dim3((T * (TEMB / 2) + 255) / 256), dim3(256), 0, st, h->emb, T, TEMB);
  if (gemmS(h, h->emb, TEMB, "t_embedder.mlp.0", h->t1, H, T, H, TEMB, st)) return -1;
  hipLaunchKernelGGL(sit_silu_kernel, dim3((T * H + 255) / 256), dim3(256), 0, st, h->t1, h->t1, (long long)T * H);
  if (gemmS(h, h->t1, H, "t_embedder.mlp.2", h->cemb, H, T, H, H, st)) return -1;
  hipLaunchKernelGGL(sit_silu_kernel, dim3((T * H + 255) / 256), dim3(256), 0, st, h->cemb, h->csilu, (long long)T * H);
  for (int i = 0; i < c.n_blocks; ++i)
    if (gemmS(h, h->csilu, H, "blocks." + std::to_string(i) + ".adaLN_modulation.1", h->tab[i], 6 * H, T, 6 * H, H, st)) return -1;
  if (gemmS(h, h->csilu, H, "final_layer.adaLN_modulation.1", h->tabF, 2 * H, T, 2 * H, H, st)) return -1;
  // every parameter the steps will ask for must be there now
  for (int i = 0; i < c.n_blocks; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    if (!findS(h, p + "attn.qkv.weight", (int64_t)3 * H * H) || !findS(h, p + "attn.qkv.bias", 3 * H) ||
        !findS(h, p + "attn.proj.weight", (int64_t)H * H) || !findS(h, p + "attn.proj.bias", H) ||
        !findS(h, p + "mlp.fc1.weight", (int64_t)c.mlp_hidden * H) || !findS(h, p + "mlp.fc1.bias", c.mlp_hidden) ||
        !findS(h, p + "mlp.fc2.weight", (int64_t)H * c.mlp_hidden) || !findS(h, p + "mlp.fc2.bias", H)) return -1;
  }
  if (!findS(h, "pos_embedder.weight", (int64_t)H * c.in_channels) || !findS(h, "pos_embedder.bias", H) ||
      !findS(h, "final_layer.linear.weight", (int64_t)c.in_channels * H) || !findS(h, "final_layer.linear.bias", c.in_channels))
    return -1;
  return checkS("sit tables");
}

struct WorkS {
  float *h, *u, *qkv, *att, *o, *f, *xa, *xb;
  int *rows, *row_count, *units, *unit_count, *goff;
  size_t bytes;
};
WorkS carveS(const scgg_sit_config& c, int B, int n, char* base) {
  Bump2 ar(base);
  WorkS w;
  const size_t M = (size_t)B * n, nch = (size_t)(n + 15) / 16;
  w.h = ar.take<float>(M * c.hidden);
  w.u = ar.take<float>(M * c.hidden);
  w.qkv = ar.take<float>(M * 3 * c.hidden);
  w.att = ar.take<float>(M * c.hidden);
  w.o = ar.take<float>(M * c.hidden);
  w.f = ar.take<float>(M * c.mlp_hidden);
  w.xa = ar.take<float>(M * 3);
  w.xb = ar.take<float>(M * 3);
  w.rows = ar.take<int>(M);
  w.row_count = ar.take<int>(64);
  w.units = ar.take<int>(M * nch);
  w.unit_count = ar.take<int>(64);
  w.goff = ar.take<int>((size_t)2 * B);
  w.bytes = (ar.off + 255) / 256 * 256;
  return w;
}

#define TRYS(x) do { if ((x) != 0) return -1; } while (0)

// one denoiser evaluation on x (and, with do_update, the DDPM step into x_out)
int stepS(scgg_sit* h, const WorkS& w, const float* x, const uint8_t* mask, int B, int n, const int32_t* t_ptr, int t_const,
          float* eps_out, float* x_out, int do_update, float c_x, float c_eps, float c_noise, const float* noise, uint64_t seed,
          uint32_t step, uint32_t graph_id_base, hipStream_t st) {
  const scgg_sit_config& c = h->c;
  const int M = B * n, H = c.hidden, F4 = H / 64;
  const ParamS *We = findS(h, "pos_embedder.weight", (int64_t)H * c.in_channels), *be = findS(h, "pos_embedder.bias", H);
  if (!We || !be) return -1;
  auto lnmod = [&](const float* o, const float* tab_gate, int off_gate, const float* tab, int off_shift, int off_scale) -> int {
    LnModArgs a{w.h, o, tab, 6 * H, off_gate, off_shift, off_scale, tab_gate, w.u, w.rows, w.row_count, t_ptr, t_const, n,
                o == nullptr ? x : nullptr, We->p, be->p, c.in_channels};
    const dim3 grid((M + 15) / 16), blk(256);
    if (F4 == 3) hipLaunchKernelGGL(sit_lnmod_kernel<3>, grid, blk, 0, st, a);
    else if (F4 == 1) hipLaunchKernelGGL(sit_lnmod_kernel<1>, grid, blk, 0, st, a);
    else if (F4 == 2) hipLaunchKernelGGL(sit_lnmod_kernel<2>, grid, blk, 0, st, a);
    else if (F4 == 6) hipLaunchKernelGGL(sit_lnmod_kernel<6>, grid, blk, 0, st, a);
    else return fail("sit: hidden %d not built (64, 128, 192, 384)", H);
    return 0;
  };
  const int hd = H / c.n_heads;
  // SCGG_SIT_FUSED=0 keeps the separate kernels (development A/B switch)
  static const bool fused_on = [] { const char* e = getenv("SCGG_SIT_FUSED"); return !(e != nullptr && e[0] == '0'); }();
  const bool fused = fused_on && node_layer_x3_enabled() && H == 192 && !h->lin.empty() && h->lin[0].qkv.W3 != nullptr &&
                     h->fc2_slice == 256;
  for (int i = 0; i < c.n_blocks; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    const scgg_sit::Blk& Bk = h->lin[i];
    // row-list bf16x3 Linear on the valid rows; -1: shape not built or the x3 node kernels are switched off -> generic GEMM
    auto lin = [&](const scgg_sit::Lin& L, const float* A, int lda, float* Cc, int ldc, int mode) -> int {
      if (L.W3 == nullptr) return -1;
      NodeLinearArgs a{A, nullptr, L.b, Cc, w.rows, w.row_count, L.N, L.K, (L.N + 31) / 32, lda, ldc, 1, mode};
      return launch_node_linear_x3(a, L.W3, M, st);
    };
    // the same with the sub-layer's entry ([gate (.) o] + h -> h, LN, modulate) formed in the Linear's prologue instead of
    // by sit_lnmod_kernel
    auto lin_ln = [&](const scgg_sit::Lin& L, const float* o, const float* tab_gate, int off_gate, const float* tab, int off_shift,
                      int off_scale, float* Cc, int ldc, int mode) -> int {
      if (L.W3 == nullptr || H > 256) return -1;
      NodeLinearArgs a{w.u, nullptr, L.b, Cc, w.rows, w.row_count, L.N, L.K, (L.N + 31) / 32, H, ldc, 1, mode};
      a.ln_h = w.h; a.ln_o = o; a.ln_gate = o ? tab_gate + off_gate : nullptr; a.ln_shift = tab + off_shift; a.ln_scale = tab + off_scale;
      a.ln_t = t_ptr; a.ln_ld = 6 * H; a.ln_n = n; a.ln_tc = t_const; a.ln_eps = 1e-6f;
      return launch_node_linear_x3(a, L.W3, M, st);
    };
    // [gate_mlp of block i-1 (.) its MLP output] + LN + modulate(shift_msa, scale_msa), then q|k|v      (sit_model.py:129-132)
    // block 0: pos_embedder + LN + modulate as the prologue of its q|k|v Linear (x -> h is stored on the way)
    auto lin_embed = [&]() -> int {
      const scgg_sit::Lin& L = Bk.qkv;
      if (i != 0 || L.W3 == nullptr || H > 256) return -1;
      NodeLinearArgs a{w.u, nullptr, L.b, w.qkv, w.rows, w.row_count, L.N, L.K, (L.N + 31) / 32, H, 3 * H, 1, 0};
      a.ln_h = w.h; a.ln_shift = h->tab[0]; a.ln_scale = h->tab[0] + H; a.ln_t = t_ptr; a.ln_ld = 6 * H; a.ln_n = n; a.ln_tc = t_const;
      a.ln_eps = 1e-6f; a.ln_x = x; a.ln_We = We->p; a.ln_be = be->p; a.ln_cin = c.in_channels;
      return launch_node_linear_x3(a, L.W3, M, st);
    };
    if (fused && i > 0) {
      // q|k|v of this block came out of the previous block's fused launch
    } else if (lin_embed() == 0) {
    } else if (i == 0 || lin_ln(Bk.qkv, w.o, h->tab[i - 1], 5 * H, h->tab[i], 0, H, w.qkv, 3 * H, 0) != 0) {
      TRYS(lnmod(i == 0 ? nullptr : w.o, i == 0 ? nullptr : h->tab[i - 1], 5 * H, h->tab[i], 0, H));
      if (lin(Bk.qkv, w.u, H, w.qkv, 3 * H, 0) != 0) TRYS(gemmS(h, w.u, H, p + "attn.qkv", w.qkv, 3 * H, M, 3 * H, H, st));
    }
    // SCGG_SIT_ATTN=valu keeps the LDS / vector-unit kernel for every size (development A/B switch)
    static const bool attn_mfma = [] { const char* e = getenv("SCGG_SIT_ATTN"); return !(e != nullptr && e[0] == 'v'); }();
    if (hd == 32 && n <= 64 && attn_mfma) {
      const int total = B * c.n_heads;
      hipLaunchKernelGGL(sit_attn_mfma_kernel<32>, dim3((total + 3) / 4), dim3(256), 0, st, w.qkv, mask, w.att, n, H, c.n_heads, total,
                         1.0f / sqrtf((float)hd));
    } else if (hd == 32) hipLaunchKernelGGL(sit_attn_kernel<32>, dim3(B, c.n_heads), dim3(64), 0, st, w.qkv, mask, w.att, n, H, 1.0f / sqrtf((float)hd));
    else if (hd == 16) hipLaunchKernelGGL(sit_attn_kernel<16>, dim3(B, c.n_heads), dim3(64), 0, st, w.qkv, mask, w.att, n, H, 1.0f / sqrtf((float)hd));
    else if (hd == 64) hipLaunchKernelGGL(sit_attn_kernel<64>, dim3(B, c.n_heads), dim3(64), 0, st, w.qkv, mask, w.att, n, H, 1.0f / sqrtf((float)hd));
    else return fail("sit: head dim %d not built (16, 32, 64)", hd);
    if (fused) {                                                 // everything up to the next attention in one launch
      const bool last = i + 1 == c.n_blocks;
      SitBlockArgs fa{w.att, w.h, w.o, Bk.proj.W3, Bk.fc1.W3, Bk.fc2[0].W3, last ? nullptr : h->lin[i + 1].qkv.W3, Bk.proj.b, Bk.fc1.b,
                      Bk.fc2[0].b, last ? nullptr : h->lin[i + 1].qkv.b, h->tab[i], last ? nullptr : h->tab[i + 1], w.qkv, w.rows,
                      w.row_count, t_ptr, t_const, n, (int)Bk.fc2.size(),
                      Bk.fc2.size() > 1 ? (long long)(Bk.fc2[1].W3 - Bk.fc2[0].W3) : 0};
      if (launch_sit_block(fa, H, M, st) != 0) return fail("sit: fused block launch failed");
      continue;
    }
    if (lin(Bk.proj, w.att, H, w.o, H, 0) != 0) TRYS(gemmS(h, w.att, H, p + "attn.proj", w.o, H, M, H, H, st));
    // gate_msa (.) attention + h, LN, modulate(shift_mlp, scale_mlp), then fc1 + GELU
    if (lin_ln(Bk.fc1, w.o, h->tab[i], 2 * H, h->tab[i], 3 * H, 4 * H, w.f, c.mlp_hidden, 2) != 0) {
      TRYS(lnmod(w.o, h->tab[i], 2 * H, h->tab[i], 3 * H, 4 * H));
      if (lin(Bk.fc1, w.u, H, w.f, c.mlp_hidden, 2) != 0) {
        TRYS(gemmS(h, w.u, H, p + "mlp.fc1", w.f, c.mlp_hidden, M, c.mlp_hidden, H, st));
        const long long nf = (long long)M * c.mlp_hidden;
        hipLaunchKernelGGL(sit_gelu_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, w.f, nf);
      }
    }
    bool sliced = !Bk.fc2.empty();
    for (size_t s2 = 0; sliced && s2 < Bk.fc2.size(); ++s2)
      if (lin(Bk.fc2[s2], w.f + h->fc2_slice * s2, c.mlp_hidden, w.o, H, s2 == 0 ? 0 : 3) != 0) {
        if (s2 != 0) return fail("sit: fc2 slice launch failed");
        sliced = false;
      }
    if (!sliced) TRYS(gemmS(h, w.f, c.mlp_hidden, p + "mlp.fc2", w.o, H, M, H, c.mlp_hidden, st));
  }
  const ParamS *Wf = findS(h, "final_layer.linear.weight", (int64_t)c.in_channels * H), *bf = findS(h, "final_layer.linear.bias", c.in_channels);
  if (!Wf || !bf) return -1;
  UpdateArgs a{w.h, w.o, h->tab[c.n_blocks - 1], 6 * H, 5 * H, h->tabF, Wf->p, bf->p, x, x_out, eps_out, noise, w.rows, w.row_count,
               t_ptr, t_const, n, c_x, c_eps, c_noise, do_update, (uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32), step,
               graph_id_base};
  const dim3 grid((M + 15) / 16), blk(256);
  if (F4 == 3) hipLaunchKernelGGL(sit_update_kernel<3>, grid, blk, 0, st, a);
  else if (F4 == 1) hipLaunchKernelGGL(sit_update_kernel<1>, grid, blk, 0, st, a);
  else if (F4 == 2) hipLaunchKernelGGL(sit_update_kernel<2>, grid, blk, 0, st, a);
  else hipLaunchKernelGGL(sit_update_kernel<6>, grid, blk, 0, st, a);
  return 0;
}

int prepS(scgg_sit* h, WorkS& w, const uint8_t* mask, int B, int n, void* workspace, size_t ws_bytes, hipStream_t st) {
  if (!h || !h->finalized) return fail("sit not finalized");
  if (B <= 0 || n <= 0 || !mask) return fail("sit: bad arguments");
  if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return fail("workspace must be 256-byte aligned");
  w = carveS(h->c, B, n, static_cast<char*>(workspace));
  if (ws_bytes < w.bytes) return fail("workspace too small: %zu < %zu", ws_bytes, w.bytes);
  // padded rows of every activation stay zero (the GEMMs run over all rows)
  if (hipMemsetAsync(workspace, 0, w.bytes, st) != hipSuccess) return fail("memset failed");
  return launch_build_lists(mask, B, n, w.goff, w.rows, w.row_count, w.units, w.unit_count, st);
}

}  // namespace

extern "C" scgg_sit* scgg_sit_create(const scgg_sit_config* cfg) {
  if (!cfg) { fail("null config"); return nullptr; }
  const scgg_sit_config& c = *cfg;
  if (c.in_channels != 3) { fail("sit: in_channels must be 3 (node coordinates)"); return nullptr; }
  if (c.hidden % 64 != 0 || (c.hidden != 64 && c.hidden != 128 && c.hidden != 192 && c.hidden != 384)) { fail("sit: hidden %d not built", c.hidden); return nullptr; }
  if (c.n_heads < 1 || c.hidden % c.n_heads != 0) { fail("sit: bad n_heads"); return nullptr; }
  const int hd = c.hidden / c.n_heads;
  if (hd != 16 && hd != 32 && hd != 64) { fail("sit: head dim %d not built", hd); return nullptr; }
  if (c.n_blocks < 1 || c.mlp_hidden < 1 || c.noise_steps < 2 || !(c.beta_start > 0.f) || !(c.beta_end < 1.f)) { fail("sit: bad config"); return nullptr; }
  scgg_sit* h = new scgg_sit();
  h->c = c;
  // torch.linspace(beta_start, beta_end, steps) in fp32, alpha = 1 - beta, alpha_hat = cumprod (fp32, as the reference)
  h->beta.resize(c.noise_steps); h->alpha.resize(c.noise_steps); h->alpha_hat.resize(c.noise_steps);
  const float stepv = (c.beta_end - c.beta_start) / (float)(c.noise_steps - 1);
  float prod = 1.0f;
  for (int i = 0; i < c.noise_steps; ++i) {
    // torch.linspace fills the first half from the start and the second half from the end
    h->beta[i] = i < c.noise_steps / 2 ? c.beta_start + stepv * (float)i : c.beta_end - stepv * (float)(c.noise_steps - 1 - i);
    h->alpha[i] = 1.0f - h->beta[i];
    prod *= h->alpha[i];
    h->alpha_hat[i] = prod;
  }
  return h;
}
extern "C" void scgg_sit_destroy(scgg_sit* h) {
  if (!h) return;
  if (h->arena) (void)hipFree(h->arena);
  delete h;
}
extern "C" int scgg_sit_set_param(scgg_sit* h, const char* name, const float* data, int64_t numel) {
  if (!h || !name || !data) return fail("set_param: null argument");
  h->params[name] = ParamS{data, numel};
  return 0;
}
extern "C" int scgg_sit_finalize(scgg_sit* h, void* stream) {
  if (!h) return fail("finalize: null handle");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (h->arena == nullptr) {
    if (buildS(h, false, st)) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&h->arena), h->arena_bytes) != hipSuccess) return fail("arena allocation failed");
  }
  if (buildS(h, true, st)) return -1;
  h->finalized = true;
  return 0;
}
extern "C" size_t scgg_sit_workspace_bytes(const scgg_sit* h, int32_t B, int32_t n) {
  if (!h || B <= 0 || n <= 0) return 0;
  return carveS(h->c, B, n, nullptr).bytes;
}
extern "C" int scgg_sit_forward(scgg_sit* h, const float* x, const int32_t* t, const uint8_t* node_mask, int32_t B, int32_t n,
                                float* eps_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !t || !eps_out) return fail("sit_forward: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  WorkS w;
  if (prepS(h, w, node_mask, B, n, workspace, workspace_bytes, st)) return -1;
  if (hipMemsetAsync(eps_out, 0, (size_t)B * n * 3 * sizeof(float), st) != hipSuccess) return fail("memset failed");
  if (stepS(h, w, x, node_mask, B, n, t, 0, eps_out, nullptr, 0, 0.f, 0.f, 0.f, nullptr, 0, 0, 0, st)) return -1;
  return checkS("sit forward");
}
extern "C" int scgg_sit_sample(scgg_sit* h, const float* x_T, const uint8_t* node_mask, int32_t B, int32_t n, int32_t steps,
                               const float* noise, uint64_t seed, uint32_t graph_id_base, float* x_out, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!x_T || !x_out) return fail("sit_sample: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  WorkS w;
  if (prepS(h, w, node_mask, B, n, workspace, workspace_bytes, st)) return -1;
  const int T = h->c.noise_steps;
  const size_t xbytes = (size_t)B * n * 3 * sizeof(float);
  if (hipMemcpyAsync(w.xa, x_T, xbytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("copy failed");
  float *cur = w.xa, *nxt = w.xb;
  int k = 0;
  for (int i = T - 1; i >= 1; --i, ++k) {                       // reversed(range(1, noise_steps))
    if (steps >= 0 && k >= steps) break;
    const float al = h->alpha[i], ah = h->alpha_hat[i], be = h->beta[i];
    const float c_x = 1.0f / sqrtf(al), c_eps = (1.0f - al) / sqrtf(1.0f - ah), c_noise = i > 1 ? sqrtf(be) : 0.f;
    if (stepS(h, w, cur, node_mask, B, n, nullptr, i, nullptr, nxt, 1, c_x, c_eps, c_noise,
              noise ? noise + (size_t)k * B * n * 3 : nullptr, seed, (uint32_t)i, graph_id_base, st)) return -1;
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  if (hipMemcpyAsync(x_out, cur, xbytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("copy failed");
  launch_zero_masked_rows(x_out, node_mask, B * n, 3, st);      // padded slots: zero (the caller masks the positions anyway)
  return checkS("sit sample");
}

extern "C" int scgg_philox_normal_rows(float* out, int32_t B, int32_t n, uint64_t seed, uint32_t graph_id_base, uint32_t step,
                                       void* stream) {
  if (!out || B <= 0 || n <= 0) return fail("philox_normal_rows: bad arguments");
  const int rows = B * n;
  hipLaunchKernelGGL(philox_normal_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), out, rows, n,
                     (uint32_t)seed, (uint32_t)(seed >> 32), step, graph_id_base);
  return checkS("philox normal rows");
}
