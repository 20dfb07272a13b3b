// Training step, global-feature (y) path of one XEyTransformerLayer as three kernels (gfx950, fp32).  SURVEY.md row a11.
//
//   y update of NodeEdgeBlock      src/ConStruct/models/transformer_model.py:229-246 (y_y + x_y + e_y, y_out)
//   y path of XEyTransformerLayer  transformer_model.py:100-105 (dropout sites y1, y2-inner, y2-outer; norm_y1 / norm_y2)
//
// Under autograd this path is eleven Linears on [B, dy] vectors plus LayerNorms, ReLUs and dropouts: ~60 launches forward and
// ~60 backward per layer of a few microseconds each, i.e. almost half of the launches of a training step for a negligible
// share of its arithmetic.  Here: one forward kernel, one backward kernel for the data gradients (it recomputes the forward
// chain, so only the inputs are kept) and one kernel for all weight gradients of the path.
//
//   z = y_y(y) + x_y(xstats) + e_y(estats);  ny = y_out.2(relu(y_out.0(z)));  y1 = LN1(y + drop6(ny))
//   f = drop7(relu(lin_y1(y1)));  y2 = LN2(y1 + drop8(lin_y2(f)))
//
// One workgroup (256 threads) per graph, vectors in LDS; a Linear is one wave per output (64-lane dot product over the input,
// coalesced weight rows), a transposed Linear (data gradient) one thread per input element over slices of the outputs.
// Dropout: the stream of scgg_dropout (train_kernels.hip): element (b, c) of a [B, width] tensor has flat index b width + c.
// Bound: latency (B workgroups, ~1 MB of weights each from L2); it replaces launches, not arithmetic.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/scgg.h"
#include "kernels.h"
#include "rng.h"

namespace scgg {
namespace {

typedef scgg_train_y_args YArgs;
constexpr int YT = 256, YWAVES = 4;

__device__ __forceinline__ float ywave_sum(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// out[o] (+)= b[o] + sum_k W[o][k] in[k];  in / out in LDS; contains no barrier (the caller synchronises).
// LPO lanes per output (2 .. 16, so that a lane has ~8 float4 loads of its weight row, all independent and in flight together)
// and 256 / LPO outputs per pass of the workgroup: a 64-wide Linear is one pass, the 512-wide one four.  (One wave per output
// with a scalar k loop was one exposed round trip per output: 0.75 ms per launch.)
template <bool ACC, bool RELU>
__device__ __forceinline__ void ydense(const float* __restrict__ W, const float* __restrict__ b, int O, int K, const float* in, float* out,
                                       int tid) {
  const bool v4 = (K & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(in)) & 15) == 0;
  const int K4 = K >> 2;
  int LPO = 16;
  if (v4) while (LPO > 2 && LPO * 8 > K4) LPO >>= 1;
  const int sub = tid & (LPO - 1), grp = tid / LPO, per_pass = YT / LPO;
  for (int ob = 0; ob < O; ob += per_pass) {
    const int o = ob + grp;
    float s = 0.f;
    if (o < O) {
      const float* w = W + (size_t)o * K;
      if (v4) {
        const f32x4_t* w4 = reinterpret_cast<const f32x4_t*>(w);
        const f32x4_t* i4 = reinterpret_cast<const f32x4_t*>(in);
        int k = sub;
        for (; k + 7 * LPO < K4; k += 8 * LPO) {
          f32x4_t a[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = w4[k + u * LPO];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const f32x4_t t = a[u] * i4[k + u * LPO];
            s += (t[0] + t[1]) + (t[2] + t[3]);
          }
        }
        for (; k < K4; k += LPO) {
          const f32x4_t t = w4[k] * i4[k];
          s += (t[0] + t[1]) + (t[2] + t[3]);
        }
      } else {
        for (int k = sub; k < K; k += LPO) s += w[k] * in[k];
      }
    }
    for (int m = 1; m < LPO; m <<= 1) s += __shfl_xor(s, m);
    if (sub == 0 && o < O) {
      float v = s + b[o];
      if (ACC) v += out[o];
      if (RELU) v = fmaxf(v, 0.f);
      out[o] = v;
    }
  }
}

// out[k] = sum_o W[o][k] g[o]  (ACC: added to out);  g / out in LDS, part: LDS scratch of 256 floats.  Contains barriers.
template <bool ACC>
__device__ __forceinline__ void ydense_t(const float* __restrict__ W, int O, int K, const float* g, float* out, float* part, int tid) {
  int KL = 1;
  while (KL < K && KL < YT) KL <<= 1;                   // threads along k: a power of two >= min(K, 256)
  const int OS = YT / KL, kk = tid % KL, sl = tid / KL;
  const int o0 = (int)((long long)O * sl / OS), o1 = (int)((long long)O * (sl + 1) / OS);
  for (int kb = 0; kb < K; kb += KL) {
    const int k = kb + kk;
    float s = 0.f;
    if (k < K) {
      int o = o0;
      for (; o + 32 <= o1; o += 32) {
        float w[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) w[u] = W[(size_t)(o + u) * K + k];
#pragma unroll
        for (int u = 0; u < 32; ++u) s += w[u] * g[o + u];
      }
      for (; o + 8 <= o1; o += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = W[(size_t)(o + u) * K + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += w[u] * g[o + u];
      }
      for (; o < o1; ++o) s += W[(size_t)o * K + k] * g[o];
    }
    part[tid] = s;
    __syncthreads();
    if (sl == 0 && k < K) {
      float t = s;
      for (int q = 1; q < OS; ++q) t += part[q * KL + kk];
      out[k] = ACC ? out[k] + t : t;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float ykeep(uint32_t k0, uint32_t k1, uint32_t site, unsigned long long idx, float p) {
  uint32_t r[4];
  const unsigned long long c = idx >> 2;
  philox4x32_10((uint32_t)(c & 0xffffffffull), (uint32_t)(c >> 32), site, 0xD0u, k0, k1, r);
  return u01(r[idx & 3]) >= p ? 1.f : 0.f;
}

// r <- LN(r) over D <= 256 elements by wave 0: xh (normalised, for the backward) and out = xh gamma + beta; returns rstd
__device__ __forceinline__ void yln(const float* r, int D, const float* __restrict__ gamma, const float* __restrict__ beta, float* xh,
                                    float* out, float* rstd_out, int tid) {
  if (tid >= 64) return;
  float v[4], s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = tid + 64 * i; v[i] = c < D ? r[c] : 0.f; s += v[i]; }
  const float mean = ywave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = tid + 64 * i; const float d = c < D ? v[i] - mean : 0.f; q += d * d; }
  const float rstd = 1.0f / sqrtf(ywave_sum(q) / (float)D + 1e-5f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 64 * i;
    if (c < D) { const float h = (v[i] - mean) * rstd; xh[c] = h; out[c] = h * gamma[c] + beta[c]; }
  }
  if (tid == 0) *rstd_out = rstd;
}

struct YLds {
  float *cat, *z, *h0, *ny, *xh1, *y1, *f, *fm, *t2, *xh2, *y2, *part, *sc;
};
__device__ __forceinline__ YLds ylds(float* sm, const YArgs& a) {
  YLds l;
  const int kcat = a.dy + a.dx4 + a.de4;
  l.cat = sm; l.z = l.cat + kcat; l.h0 = l.z + a.dy; l.ny = l.h0 + a.dy; l.xh1 = l.ny + a.dy; l.y1 = l.xh1 + a.dy;
  l.f = l.y1 + a.dy; l.fm = l.f + a.ffy; l.t2 = l.fm + a.ffy; l.xh2 = l.t2 + a.dy; l.y2 = l.xh2 + a.dy; l.part = l.y2 + a.dy;
  l.sc = l.part + YT;                                   // [0] rstd1, [1] rstd2
  return l;
}
size_t ylds_bytes(const YArgs& a) { return (size_t)(a.dy + a.dx4 + a.de4 + 9 * a.dy + 2 * a.ffy + YT + 8) * 4; }

// the forward chain of graph b into LDS (cat, z, h0, ny, xh1, y1, f = dropped hidden, fm = its multiplier, xh2, y2)
__device__ __forceinline__ void y_chain(const YArgs& a, const YLds& l, int b, uint32_t k0, uint32_t k1, int tid) {
  const int dy = a.dy, kx = a.dx4, ke = a.de4, ffy = a.ffy;
  const float denom = 1.0f - a.p;
  const bool drop = a.p > 0.f;
  for (int c = tid; c < dy + kx + ke; c += YT)
    l.cat[c] = c < dy ? a.hy[(size_t)b * dy + c] : (c < dy + kx ? a.xstats[(size_t)b * kx + c - dy] : a.estats[(size_t)b * ke + c - dy - kx]);
  __syncthreads();
  ydense<false, false>(a.W_yy, a.b_yy, dy, dy, l.cat, l.z, tid);
  __syncthreads();
  ydense<true, false>(a.W_xy, a.b_xy, dy, kx, l.cat + dy, l.z, tid);
  __syncthreads();
  ydense<true, false>(a.W_ey, a.b_ey, dy, ke, l.cat + dy + kx, l.z, tid);
  __syncthreads();
  ydense<false, true>(a.W_o0, a.b_o0, dy, dy, l.z, l.h0, tid);
  __syncthreads();
  ydense<false, false>(a.W_o2, a.b_o2, dy, dy, l.h0, l.ny, tid);
  __syncthreads();
  for (int c = tid; c < dy; c += YT) {
    const bool keep = !drop || ykeep(k0, k1, a.site1, (unsigned long long)b * dy + c, a.p) != 0.f;
    l.t2[c] = l.cat[c] + (keep ? (drop ? l.ny[c] / denom : l.ny[c]) : 0.f);   // r1 = y + drop(ny); a division, as scgg_dropout
  }
  __syncthreads();
  yln(l.t2, dy, a.g1, a.be1, l.xh1, l.y1, l.sc, tid);
  __syncthreads();
  ydense<false, true>(a.W_1, a.b_1, ffy, dy, l.y1, l.f, tid);
  __syncthreads();
  for (int c = tid; c < ffy; c += YT) {
    const bool keep = !drop || ykeep(k0, k1, a.site2, (unsigned long long)b * ffy + c, a.p) != 0.f;
    const float v = l.f[c];
    l.fm[c] = (v > 0.f && keep) ? 1.f : 0.f;            // does the hidden unit pass a gradient: relu' and dropout
    l.f[c] = keep ? (drop ? v / denom : v) : 0.f;
  }
  __syncthreads();
  ydense<false, false>(a.W_2, a.b_2, dy, ffy, l.f, l.t2, tid);
  __syncthreads();
  for (int c = tid; c < dy; c += YT) {
    const bool keep = !drop || ykeep(k0, k1, a.site3, (unsigned long long)b * dy + c, a.p) != 0.f;
    l.t2[c] = l.y1[c] + (keep ? (drop ? l.t2[c] / denom : l.t2[c]) : 0.f);     // r2 = y1 + drop(lin_y2(f))
  }
  __syncthreads();
  yln(l.t2, dy, a.g2, a.be2, l.xh2, l.y2, l.sc + 1, tid);
  __syncthreads();
}

__global__ __launch_bounds__(YT) void train_y_fwd_kernel(YArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ysm2[];
  const YLds l = ylds(ysm2, a);
  const int b = blockIdx.x, tid = threadIdx.x;
  uint32_t k0 = a.k0, k1 = a.k1;
  if (a.key_dev != nullptr) { k0 = a.key_dev[0]; k1 = a.key_dev[1]; }
  y_chain(a, l, b, k0, k1, tid);
  for (int c = tid; c < a.dy; c += YT) a.out[(size_t)b * a.dy + c] = l.y2[c];
  if (a.fsave != nullptr) {                             // what the backward would otherwise recompute: z | h0 | xh1 | y1 | xh2 | f | fm | rstd1, rstd2
    float* fs = a.fsave + (size_t)b * (5 * a.dy + 2 * a.ffy + 4);
    for (int c = tid; c < a.dy; c += YT) {
      fs[c] = l.z[c]; fs[a.dy + c] = l.h0[c]; fs[2 * a.dy + c] = l.xh1[c]; fs[3 * a.dy + c] = l.y1[c]; fs[4 * a.dy + c] = l.xh2[c];
    }
    for (int c = tid; c < a.ffy; c += YT) { fs[5 * a.dy + c] = l.f[c]; fs[5 * a.dy + a.ffy + c] = l.fm[c]; }
    if (tid < 2) fs[5 * a.dy + 2 * a.ffy + tid] = l.sc[tid];
  }
}

// LayerNorm backward over D <= 256 by wave 0: dr = rstd (dxh - mean(dxh) - xh mean(dxh xh)), dxh = dy gamma
__device__ __forceinline__ void yln_bwd(const float* dy_, const float* xh, float rstd, int D, const float* __restrict__ gamma, float* dr,
                                        int tid) {
  if (tid >= 64) return;
  float dx[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 64 * i;
    dx[i] = c < D ? dy_[c] * gamma[c] : 0.f;
    s1 += dx[i];
    s2 += c < D ? dx[i] * xh[c] : 0.f;
  }
  const float m1 = ywave_sum(s1) / (float)D, m2 = ywave_sum(s2) / (float)D;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 64 * i;
    if (c < D) dr[c] = rstd * (dx[i] - m1 - xh[c] * m2);
  }
}

// Data gradients of graph b.  save [B, SW]: what the weight-gradient kernel needs, per graph:
//   G side: dz | gh0 | gny | gf (ffy) | g2 | dy2 (.) xh2 | dy2 | dy1 (.) xh1 | dy1      X side: z | h0 | y1 | fdrop (ffy)
// (hy, xstats, estats are read from the inputs).  Offsets: scgg_train_y_save_width and the launcher below.
__global__ __launch_bounds__(YT) void train_y_bwd_kernel(YArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ysm2[];
  const YLds l = ylds(ysm2, a);
  const int b = blockIdx.x, tid = threadIdx.x, dy = a.dy, ffy = a.ffy;
  uint32_t k0 = a.k0, k1 = a.k1;
  if (a.key_dev != nullptr) { k0 = a.key_dev[0]; k1 = a.key_dev[1]; }
  if (a.fsave != nullptr) {                             // the forward's intermediates as it left them
    const float* fs = a.fsave + (size_t)b * (5 * dy + 2 * ffy + 4);
    for (int c = tid; c < dy; c += YT) {
      l.z[c] = fs[c]; l.h0[c] = fs[dy + c]; l.xh1[c] = fs[2 * dy + c]; l.y1[c] = fs[3 * dy + c]; l.xh2[c] = fs[4 * dy + c];
    }
    for (int c = tid; c < ffy; c += YT) { l.f[c] = fs[5 * dy + c]; l.fm[c] = fs[5 * dy + ffy + c]; }
    if (tid < 2) l.sc[tid] = fs[5 * dy + 2 * ffy + tid];
    __syncthreads();
  } else {
    y_chain(a, l, b, k0, k1, tid);
  }
  const float denom = 1.0f - a.p;
  const bool drop = a.p > 0.f;
  float* sv = a.save + (size_t)b * (12 * dy + 2 * ffy);
  float *s_dz = sv, *s_gh0 = sv + dy, *s_gny = sv + 2 * dy, *s_gf = sv + 3 * dy, *s_g2 = s_gf + ffy, *s_dg2 = s_g2 + dy, *s_db2 = s_dg2 + dy,
        *s_dg1 = s_db2 + dy, *s_db1 = s_dg1 + dy, *s_z = s_db1 + dy, *s_h0 = s_z + dy, *s_y1 = s_h0 + dy, *s_f = s_y1 + dy;
  // LDS reuse for gradients: g (dy) in ny's place once ny is consumed, etc.  Keep it simple: separate small arrays at the end of y2.
  float* g = l.y2;        // dL/dy2, then reused
  float* dr = l.t2;       // dL/dr2, dL/dr1
  float* dy1 = l.ny;      // accumulates dL/dy1 (ny itself is no longer needed: its products were taken in the chain)
  for (int c = tid; c < dy; c += YT) {
    const float go = a.dout[(size_t)b * dy + c];
    g[c] = go;
    s_dg2[c] = go * l.xh2[c];
    s_db2[c] = go;
    s_z[c] = l.z[c]; s_h0[c] = l.h0[c]; s_y1[c] = l.y1[c];
  }
  for (int c = tid; c < ffy; c += YT) s_f[c] = l.f[c];
  __syncthreads();
  yln_bwd(g, l.xh2, l.sc[1], dy, a.g2, dr, tid);                       // dr2
  __syncthreads();
  for (int c = tid; c < dy; c += YT) {
    const bool keep = !drop || ykeep(k0, k1, a.site3, (unsigned long long)b * dy + c, a.p) != 0.f;
    dy1[c] = dr[c];                                                    // residual branch
    g[c] = keep ? (drop ? dr[c] / denom : dr[c]) : 0.f;                // g2 = dL/d lin_y2 out
    s_g2[c] = g[c];
  }
  __syncthreads();
  ydense_t<false>(a.W_2, dy, ffy, g, l.f, l.part, tid);                // dL/df (dropped hidden) into f's place
  for (int c = tid; c < ffy; c += YT) {
    const float v = l.fm[c] != 0.f ? (drop ? l.f[c] / denom : l.f[c]) : 0.f;   // through dropout and relu
    l.f[c] = v;
    s_gf[c] = v;
  }
  __syncthreads();
  ydense_t<true>(a.W_1, ffy, dy, l.f, dy1, l.part, tid);               // dy1 += W_1^T gf
  for (int c = tid; c < dy; c += YT) { s_dg1[c] = dy1[c] * l.xh1[c]; s_db1[c] = dy1[c]; }
  __syncthreads();
  yln_bwd(dy1, l.xh1, l.sc[0], dy, a.g1, dr, tid);                     // dr1
  __syncthreads();
  float* dhy = l.y1;                                                   // y1 is consumed
  for (int c = tid; c < dy; c += YT) {
    const bool keep = !drop || ykeep(k0, k1, a.site1, (unsigned long long)b * dy + c, a.p) != 0.f;
    dhy[c] = dr[c];                                                    // residual branch
    g[c] = keep ? (drop ? dr[c] / denom : dr[c]) : 0.f;                // gny
    s_gny[c] = g[c];
  }
  __syncthreads();
  ydense_t<false>(a.W_o2, dy, dy, g, dr, l.part, tid);                 // dL/dh0
  for (int c = tid; c < dy; c += YT) {
    const float v = l.h0[c] > 0.f ? dr[c] : 0.f;
    dr[c] = v;
    s_gh0[c] = v;
  }
  __syncthreads();
  ydense_t<false>(a.W_o0, dy, dy, dr, g, l.part, tid);                 // dz
  for (int c = tid; c < dy; c += YT) s_dz[c] = g[c];
  __syncthreads();
  ydense_t<true>(a.W_yy, dy, dy, g, dhy, l.part, tid);
  for (int c = tid; c < dy; c += YT) a.dhy[(size_t)b * dy + c] = dhy[c];
  ydense_t<false>(a.W_xy, dy, a.dx4, g, l.cat, l.part, tid);           // cat is free: d xstats
  for (int c = tid; c < a.dx4; c += YT) a.dxstats[(size_t)b * a.dx4 + c] = l.cat[c];
  __syncthreads();
  ydense_t<false>(a.W_ey, dy, a.de4, g, l.cat, l.part, tid);
  for (int c = tid; c < a.de4; c += YT) a.destats[(size_t)b * a.de4 + c] = l.cat[c];
}

// All weight gradients of the path in one launch: item t: dW[o][k] = sum_b G[b][o] X[b][k] (K > 0) and db[o] = sum_b G[b][o].
struct YWList { const float* G[12]; const float* X[12]; float* dW[12]; float* db[12]; int ldg[12], ldx[12], O[12], K[12]; int count, B; };
__global__ __launch_bounds__(256) void train_y_wgrad_kernel(YWList l) {
  const int t = blockIdx.y;
  const int O = l.O[t], K = l.K[t];
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long nW = (long long)O * K;
  if (idx >= nW + O) return;
  const float* G = l.G[t];
  const int ldg = l.ldg[t];
  if (idx < nW) {
    const int o = (int)(idx / K), k = (int)(idx - (long long)o * K);
    const float* X = l.X[t];
    const int ldx = l.ldx[t];
    float s = 0.f;
    for (int b = 0; b < l.B; ++b) s += G[(size_t)b * ldg + o] * X[(size_t)b * ldx + k];
    l.dW[t][idx] = s;
  } else if (l.db[t] != nullptr) {
    const int o = (int)(idx - nW);
    float s = 0.f;
    for (int b = 0; b < l.B; ++b) s += G[(size_t)b * ldg + o];
    l.db[t][o] = s;
  }
}

int ycheck(const YArgs* a, bool bwd) {
  if (!a) return fail("train_y: null arguments");
  if (a->B <= 0 || a->dy <= 0 || a->dy > 256 || a->dx4 <= 0 || a->de4 <= 0 || a->ffy <= 0) return fail("train_y: bad shape (dy <= 256)");
  if (!(a->p >= 0.f) || !(a->p < 1.f)) return fail("train_y: bad dropout p");
  if (!a->hy || !a->xstats || !a->estats || !a->W_yy || !a->b_yy || !a->W_xy || !a->b_xy || !a->W_ey || !a->b_ey || !a->W_o0 || !a->b_o0 ||
      !a->W_o2 || !a->b_o2 || !a->g1 || !a->be1 || !a->W_1 || !a->b_1 || !a->W_2 || !a->b_2 || !a->g2 || !a->be2)
    return fail("train_y: null input");
  if (!bwd && !a->out) return fail("train_y forward: null output");
  if (bwd && (!a->dout || !a->dhy || !a->dxstats || !a->destats || !a->save)) return fail("train_y backward: null argument");
  if (ylds_bytes(*a) > 150 * 1024) return fail("train_y: widths beyond the LDS budget");
  return 0;
}

template <typename Kern>
int ylaunch(Kern kern, const YArgs* a, hipStream_t st, LdsAttrOnce& attr) {
  const size_t lds = ylds_bytes(*a);
  if (set_max_dynamic_lds(attr, reinterpret_cast<const void*>(kern), 150 * 1024)) return fail("train_y: LDS attribute");   // the cap of ycheck
  hipLaunchKernelGGL(kern, dim3((unsigned)a->B), dim3(YT), lds, st, *a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_y: %s", hipGetErrorString(e));
}

}  // namespace
}  // namespace scgg

using namespace scgg;

extern "C" int64_t scgg_train_y_save_width(int32_t dy, int32_t ffy) { return 12ll * dy + 2ll * ffy; }
extern "C" int64_t scgg_train_y_keep_width(int32_t dy, int32_t ffy) { return 5ll * dy + 2ll * ffy + 4; }

extern "C" int scgg_train_y_forward(const scgg_train_y_args* a, void* stream) {
  if (ycheck(a, false)) return -1;
  static LdsAttrOnce attr;
  return ylaunch(train_y_fwd_kernel, a, static_cast<hipStream_t>(stream), attr);
}

extern "C" int scgg_train_y_backward(const scgg_train_y_args* a, void* stream) {
  if (ycheck(a, true)) return -1;
  static LdsAttrOnce attr;
  return ylaunch(train_y_bwd_kernel, a, static_cast<hipStream_t>(stream), attr);
}

// grads: dW_yy, db_yy, dW_xy, db_xy, dW_ey, db_ey, dW_o0, db_o0, dW_o2, db_o2, dg1, dbe1, dW_1, db_1, dW_2, db_2, dg2, dbe2 (the order of
// the parameters in scgg_train_y_args), from the inputs of the call and the `save` buffer the backward kernel filled
extern "C" int scgg_train_y_wgrad(const scgg_train_y_args* a, float* const* grads, void* stream) {
  if (ycheck(a, true) || !grads) return fail("train_y_wgrad: bad arguments");
  for (int i = 0; i < 18; ++i)
    if (!grads[i]) return fail("train_y_wgrad: null gradient %d", i);
  const int dy = a->dy, ffy = a->ffy, SW = 12 * dy + 2 * ffy;
  const float* sv = a->save;
  const float *s_dz = sv, *s_gh0 = sv + dy, *s_gny = sv + 2 * dy, *s_gf = sv + 3 * dy, *s_g2 = s_gf + ffy, *s_dg2 = s_g2 + dy,
              *s_db2 = s_dg2 + dy, *s_dg1 = s_db2 + dy, *s_db1 = s_dg1 + dy, *s_z = s_db1 + dy, *s_h0 = s_z + dy, *s_y1 = s_h0 + dy,
              *s_f = s_y1 + dy;
  YWList l;
  int t = 0;
  auto item = [&](const float* G, int O, const float* X, int ldx, int K, float* dW, float* db) {
    l.G[t] = G; l.ldg[t] = SW; l.O[t] = O; l.X[t] = X; l.ldx[t] = ldx; l.K[t] = K; l.dW[t] = dW; l.db[t] = db; ++t;
  };
  item(s_dz, dy, a->hy, dy, dy, grads[0], grads[1]);
  item(s_dz, dy, a->xstats, a->dx4, a->dx4, grads[2], grads[3]);
  item(s_dz, dy, a->estats, a->de4, a->de4, grads[4], grads[5]);
  item(s_gh0, dy, s_z, SW, dy, grads[6], grads[7]);
  item(s_gny, dy, s_h0, SW, dy, grads[8], grads[9]);
  item(s_dg1, dy, nullptr, 0, 0, nullptr, grads[10]);
  item(s_db1, dy, nullptr, 0, 0, nullptr, grads[11]);
  item(s_gf, ffy, s_y1, SW, dy, grads[12], grads[13]);
  item(s_g2, dy, s_f, SW, ffy, grads[14], grads[15]);
  item(s_dg2, dy, nullptr, 0, 0, nullptr, grads[16]);
  item(s_db2, dy, nullptr, 0, 0, nullptr, grads[17]);
  l.count = t; l.B = a->B;
  for (int i = t; i < 12; ++i) { l.G[i] = nullptr; l.X[i] = nullptr; l.dW[i] = nullptr; l.db[i] = nullptr; l.ldg[i] = l.ldx[i] = l.O[i] = l.K[i] = 0; }
  long long most = 0;
  for (int i = 0; i < t; ++i) { const long long e = (long long)l.O[i] * l.K[i] + l.O[i]; most = e > most ? e : most; }
  hipLaunchKernelGGL(train_y_wgrad_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)t), dim3(256), 0, static_cast<hipStream_t>(stream), l);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_y_wgrad: %s", hipGetErrorString(e));
}
