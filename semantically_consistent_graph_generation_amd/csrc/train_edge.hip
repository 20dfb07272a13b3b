// Training step, edge path of one XEy layer as fused kernels with a hand-written backward (gfx950, fp32 on
// v_mfma_f32_16x16x4_f32).  SURVEY.md row a11.
//
//   NodeEdgeBlock (E part + attention)  src/ConStruct/models/transformer_model.py:165-227
//   XEyTransformerLayer E path          transformer_model.py:94-98 (dropout sites E1, E2-inner, E2-outer)
//   the step they are part of           src/ConStruct/diffusion_model_discrete_3d.py:215-239
//
// What the reference's autograd keeps per layer -- about ten [B,n,n,dx] tensors -- is never materialised here: the forward
// keeps the layer's INPUTS (E, Q, K, V, the two FiLM vectors) plus per (row, channel) the log-sum-exp of the attention
// softmax and the attention output, and the backward kernel recomputes the chain per 16-key unit in registers.
//
// Layout ("C16", as the inference kernels): work unit = (query row (b,i), chunk of 16 keys); one wave per unit; key j on
// lane & 15, channel group g = lane >> 4; a 16-channel block of activations is one f32x4 whose register r holds channel
// 4g + r -- how v_mfma_f32_16x16x4_f32 leaves its result and takes its B operand.  A Linear is out^T = W . in^T with W as
// the A operand; the data gradient of a Linear is the same product with W^T, so the backward needs fragment images of W
// and of W^T (train_edge_pack_kernel: float4 ((rb CB + cb) 64 + l) = M[16 rb + (l & 15)][16 cb + 4 (l >> 4) + 0..3]).
// Weight fragments are read from global memory (the images of a layer, 160-256 KB, stay in the L2s).
//
// Forward per valid pair (i, j), vectors over channels (e = E[b,i,j,:]):
//   a1 = W_mul e + b_mul + 1      a2 = W_add e + b_add      y0 = (Q_i / sqrt(df)) (.) K_j      Y = y0 (.) a1 + a2
//   z = yea + yem1 (.) Y          newE = W_out z + b_out     attention: p = exp(Y - lse_i), wV_i = sum_j p (.) V_j
//   r1 = e + drop1(newE)   h1 = LN1(r1)   f = drop2(relu(W_1 h1 + b_1))   r2 = h1 + drop3(W_2 f + b_2)   out = LN2(r2)
// Backward: see train_edge_bwd_kernel.  Weight gradients are sums over all pairs of outer products g (x) x, i.e. GEMMs
// with the PAIR as the reduction index; in C16 layout the pair sits on the MFMA's free index, so the backward kernel writes
// the factors ("G tensors", [pair slots, width]) and train_wgrad_kernel forms G^T X and the bias sums from them with the
// slot as the MFMA's K index (split over slabs of slots; HBM-bound).
//
// Dropout: the mask of site s is Philox4x32-10 of (flat element index >> 2, site) with the element's word (index & 3),
// the definition of train_kernels.hip / oracle/training.py:dropout_keep; a lane's f32x4 is one Philox call.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/scgg.h"
#include "c16.h"
#include "kernels.h"
#include "rng.h"

namespace scgg {
namespace {

typedef scgg_train_edge_args TArgs;

__device__ __forceinline__ void mfma4(f32x4& acc, const f32x4& w, const f32x4& x) {
#pragma unroll
  for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], x[r], acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 frag(const float* img, int idx, int lane) {
  return reinterpret_cast<const f32x4*>(img)[(size_t)idx * 64 + lane];
}

// dropout of four consecutive elements starting at flat index idx (a multiple of 4): keep iff u >= p, scale 1 / (1 - p)
__device__ __forceinline__ void drop4(f32x4& v, unsigned long long idx, uint32_t site, float p, float denom, uint32_t k0, uint32_t k1) {
  const unsigned long long c = idx >> 2;
  uint32_t r[4];
  philox4x32_10((uint32_t)(c & 0xffffffffull), (uint32_t)(c >> 32), site, 0xD0u, k0, k1, r);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = u01(r[e]) >= p ? v[e] / denom : 0.f;
}

// LayerNorm forward over IBE*16 channels in C16 layout, keeping what the backward needs: xh = (x - mean) rstd, out = xh g + b
template <int IBE>
__device__ __forceinline__ float ln_fwd(const f32x4 (&x)[IBE], f32x4 (&xh)[IBE], f32x4 (&out)[IBE], const float* gamma,
                                        const float* beta, int g) {
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
    for (int r = 0; r < 4; ++r) { const float d = x[b][r] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(groups_sum(q) * (1.0f / (16 * IBE)) + 1e-5f);
#pragma unroll
  for (int b = 0; b < IBE; ++b) {
    const f32x4 gm = ld4(gamma + 16 * b + 4 * g), bt = ld4(beta + 16 * b + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) { xh[b][r] = (x[b][r] - mean) * rstd; out[b][r] = xh[b][r] * gm[r] + bt[r]; }
  }
  return rstd;
}
// LayerNorm backward: dy -> dx = rstd (dxh - mean(dxh) - xh mean(dxh xh)), dxh = dy g
template <int IBE>
__device__ __forceinline__ void ln_bwd(const f32x4 (&dy)[IBE], const f32x4 (&xh)[IBE], float rstd, const float* gamma, int g,
                                       f32x4 (&dx)[IBE]) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int b = 0; b < IBE; ++b) {
    const f32x4 gm = ld4(gamma + 16 * b + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) { dx[b][r] = dy[b][r] * gm[r]; s1 += dx[b][r]; s2 += dx[b][r] * xh[b][r]; }
  }
  const float m1 = groups_sum(s1) * (1.0f / (16 * IBE)), m2 = groups_sum(s2) * (1.0f / (16 * IBE));
#pragma unroll
  for (int b = 0; b < IBE; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) dx[b][r] = rstd * (dx[b][r] - m1 - xh[b][r] * m2);
}

struct UnitId {
  long long unit;
  int row, b, i, j, jj;
  bool inb, vi, vj, any;
};
__device__ __forceinline__ UnitId unit_id(const TArgs& a, int lane) {
  UnitId u;
  u.unit = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  u.row = (int)(u.unit / a.nch);
  const int jc = (int)(u.unit - (long long)u.row * a.nch);
  u.b = u.row / a.n;
  u.i = u.row - u.b * a.n;
  u.j = jc * 16 + (lane & 15);
  u.inb = u.j < a.n;
  u.jj = u.inb ? u.j : 0;
  u.vi = a.mask[u.row] != 0;
  u.vj = u.vi && u.inb && a.mask[u.b * a.n + u.jj] != 0;
  u.any = __any(u.vj);
  return u;
}

// the first half of the chain: newE = W_out (yea + yem1 (.) Y) + b_out over all dx blocks (accO), optionally the softmax
// partials of the unit (FWD) -- shared by the forward kernel and the backward's recomputation
template <int IBE, bool FWD>
__device__ __forceinline__ void attention_half(const TArgs& a, const UnitId& u, const f32x4 (&e)[IBE], f32x4 (&accO)[IBE], int lane) {
  const int p = lane & 15, g = lane >> 4, DX = a.dx, XB = DX / 16;
  const float* qrow = a.Q + (long long)u.row * DX;
  const float* krow = a.K + ((long long)u.b * a.n + u.jj) * DX;
  const float* vrow = a.V + ((long long)u.b * a.n + u.jj) * DX;
  const float* yea = a.yea + (long long)u.b * DX;
  const float* yem = a.yem1 + (long long)u.b * DX;
  float* part = FWD ? a.part + u.unit * 3 * DX : nullptr;
#pragma unroll
  for (int ob = 0; ob < IBE; ++ob) accO[ob] = ld4(a.b_out + 16 * ob + 4 * g);
#pragma unroll 1
  for (int xb = 0; xb < XB; ++xb) {
    const int co = 16 * xb + 4 * g;
    f32x4 a1 = ld4(a.b_mul + co) + splat4(1.0f), a2 = ld4(a.b_add + co);
#pragma unroll
    for (int ib = 0; ib < IBE; ++ib) {
      mfma4(a1, frag(a.F_mul, xb * IBE + ib, lane), e[ib]);
      mfma4(a2, frag(a.F_add, xb * IBE + ib, lane), e[ib]);
    }
    const f32x4 q = ld4(qrow + co) * a.inv_sqrt_df;
    const f32x4 k = u.vj ? ld4(krow + co) : splat4(0.f);
    const f32x4 ya = ld4(yea + co), ym = ld4(yem + co);
    f32x4 yy, z;
#pragma unroll
    for (int r = 0; r < 4; ++r) { yy[r] = (q[r] * k[r]) * a1[r] + a2[r]; z[r] = ya[r] + ym[r] * yy[r]; }
#pragma unroll
    for (int ob = 0; ob < IBE; ++ob) mfma4(accO[ob], frag(a.F_out, ob * XB + xb, lane), z);
    if (FWD) {
      const f32x4 v = u.vj ? ld4(vrow + co) : splat4(0.f);
      f32x4 cm, cs, ca;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = u.vj ? yy[r] : -INFINITY;
        cm[r] = row_max16(y);
        const float pe = u.vj ? __expf(y - cm[r]) : 0.f;
        cs[r] = row_sum16(pe);
        ca[r] = row_sum16(pe * v[r]);
      }
      if (p == 0) {
        *reinterpret_cast<f32x4*>(part + co) = cm;
        *reinterpret_cast<f32x4*>(part + DX + co) = cs;
        *reinterpret_cast<f32x4*>(part + 2 * DX + co) = ca;
      }
    }
  }
}

// the second half: r1 = e + drop1(newE), LN1, FFN with its two dropouts, r2, LN2.  Leaves everything the backward needs.
template <int IBE, int IBF>
struct Tail {
  f32x4 xh1[IBE], h1[IBE], xh2[IBE], out[IBE], hid[IBF];
  float rstd1, rstd2;
};
template <int IBE, int IBF>
__device__ __forceinline__ void tail_fwd(const TArgs& a, const UnitId& u, const f32x4 (&e)[IBE], const f32x4 (&newE)[IBE],
                                         Tail<IBE, IBF>& t, int lane) {
  constexpr int DE = 16 * IBE, FFE = 16 * IBF;
  const int g = lane >> 4;
  const bool drop = a.p > 0.f;
  const float denom = 1.0f - a.p;
  const unsigned long long pair = ((unsigned long long)a.b0 * a.n + u.row) * a.n + u.jj;   // GLOBAL pair index: the dropout stream
  f32x4 r1[IBE];
#pragma unroll
  for (int eb = 0; eb < IBE; ++eb) {
    f32x4 d = newE[eb];
    if (drop) drop4(d, pair * DE + 16 * eb + 4 * g, a.site1, a.p, denom, a.k0, a.k1);
    r1[eb] = e[eb] + d;
  }
  t.rstd1 = ln_fwd<IBE>(r1, t.xh1, t.h1, a.g1, a.b1, g);
  f32x4 o[IBE];
#pragma unroll
  for (int ob = 0; ob < IBE; ++ob) o[ob] = ld4(a.b_ff2 + 16 * ob + 4 * g);
#pragma unroll
  for (int fb = 0; fb < IBF; ++fb) {
    f32x4 h = ld4(a.b_ff1 + 16 * fb + 4 * g);
#pragma unroll
    for (int ib = 0; ib < IBE; ++ib) mfma4(h, frag(a.F_ff1, fb * IBE + ib, lane), t.h1[ib]);
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r], 0.f);
    if (drop) drop4(h, pair * FFE + 16 * fb + 4 * g, a.site2, a.p, denom, a.k0, a.k1);
    t.hid[fb] = h;
#pragma unroll
    for (int ob = 0; ob < IBE; ++ob) mfma4(o[ob], frag(a.F_ff2, ob * IBF + fb, lane), h);
  }
  f32x4 r2[IBE];
#pragma unroll
  for (int eb = 0; eb < IBE; ++eb) {
    if (drop) drop4(o[eb], pair * DE + 16 * eb + 4 * g, a.site3, a.p, denom, a.k0, a.k1);
    r2[eb] = t.h1[eb] + o[eb];
  }
  t.rstd2 = ln_fwd<IBE>(r2, t.xh2, t.out, a.g2, a.b2, g);
}

template <int IBE>
__device__ __forceinline__ void load_tile(const TArgs& a, const UnitId& u, f32x4 (&e)[IBE], int lane) {
  const float* ep = a.E + ((long long)u.row * a.n + u.jj) * (16 * IBE) + 4 * (lane >> 4);
#pragma unroll
  for (int eb = 0; eb < IBE; ++eb) e[eb] = u.vj ? ld4(ep + 16 * eb) : splat4(0.f);
}

// ---------------------------------------------------------------------------------------------------------
// Forward: E_out[b,i,j,:] = LN2(...) on the valid pairs (the caller zero-fills E_out; symmetrisation and the diagonal /
// padding mask are the caller's, transformer_model.py:99,110-112) and the unit's softmax partials [max | sum p | sum p V].
// ---------------------------------------------------------------------------------------------------------
template <int IBE, int IBF>
__global__ __launch_bounds__(256) void train_edge_fwd_kernel(TArgs a_in) {
  TArgs a = a_in;
  if (a.key_dev != nullptr) { a.k0 = a.key_dev[0]; a.k1 = a.key_dev[1]; }   // dropout key read at run time (graph replay)
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const long long total = (long long)a.B * a.n * a.nch;
  if ((long long)blockIdx.x * 4 + (threadIdx.x >> 6) >= total) return;
  const UnitId u = unit_id(a, lane);
  if (!u.any) {                                   // neutral partials: the merge walks every chunk of a valid row
    float* part = a.part + u.unit * 3 * a.dx;
    for (int c = lane; c < a.dx; c += 64) { part[c] = -INFINITY; part[a.dx + c] = 0.f; part[2 * a.dx + c] = 0.f; }
    return;
  }
  f32x4 e[IBE], newE[IBE];
  load_tile<IBE>(a, u, e, lane);
  attention_half<IBE, true>(a, u, e, newE, lane);
  Tail<IBE, IBF> t;
  tail_fwd<IBE, IBF>(a, u, e, newE, t, lane);
  if (u.vj) {
    float* dst = a.E_out + ((long long)u.row * a.n + u.jj) * (16 * IBE) + 4 * g;
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) *reinterpret_cast<f32x4*>(dst + 16 * eb) = t.out[eb];
  }
}

// merge of the units of a row: wV[row, c] = sum_j softmax_j(Y)[c] V_j[c], lse[row, c] = log sum_j exp(Y[c])
__global__ __launch_bounds__(256) void train_edge_merge_kernel(TArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int DX = a.dx;
  if (idx >= (long long)a.B * a.n * DX) return;
  const int row = (int)(idx / DX), c = (int)(idx - (long long)row * DX);
  float wv = 0.f, lse = 0.f;
  if (a.mask[row] != 0) {
    const float* part = a.part + (long long)row * a.nch * 3 * DX + c;
    float M = -INFINITY;
    for (int jc = 0; jc < a.nch; ++jc) M = fmaxf(M, part[(long long)jc * 3 * DX]);
    float S = 0.f, A = 0.f;
    for (int jc = 0; jc < a.nch; ++jc) {
      const float* pj = part + (long long)jc * 3 * DX;
      const float w = pj[0] == -INFINITY ? 0.f : __expf(pj[0] - M);
      S += pj[DX] * w;
      A += pj[2 * DX] * w;
    }
    wv = A / S;
    lse = M + logf(S);
  }
  a.wV_out[idx] = wv;
  a.lse_out[idx] = lse;
}

// ---------------------------------------------------------------------------------------------------------
// Backward.  Given dOut = dL/dE_out (read on the valid pairs only) and dwV = dL/dwV:
//   LN2:  dr2 = LN2'(dOut)             do = drop3'(dr2)          df = (W_2^T do) (.) drop2' (.) [f > 0]
//   dh1 = dr2 + W_1^T df               LN1: dr1 = LN1'(dh1)      dnewE = drop1'(dr1)        de = dr1
//   per dx block: dz = W_out^T dnewE   dY = yem1 (.) dz + p (.) dwV_i (.) (V_j - wV_i)       (softmax backward in closed form:
//                                      sum_j p dwV V_j = dwV wV, so no second pass over the row)
//                 da1 = dY (.) y0      da2 = dY      de += W_mul^T da1 + W_add^T da2
//                 dQ_i += sum_j dY a1 K_j / sqrt(df)      dK_j += sum_i dY a1 Q_i / sqrt(df)      dV_j += sum_i p dwV_i
// Outputs, all indexed by the DENSE pair slot s = unit * 16 + (lane & 15) (every slot is written, zeros where the pair is
// not valid, so the host needs no memset and its sums are deterministic):
//   dE[b,i,j,:] (j < n)     G_e [s, de] = e      G_da [s, 2 dx] = da1 | da2      G_z [s, dx]      G_dnewE [s, de]
//   G_h1 [s, de]   G_df [s, ffE]   G_f [s, ffE] (= f after ReLU and dropout)   G_do [s, de]   G_dkv [s, 2 dx] = dK | dV summands
//   upart [unit, 2 dx + 4 de] = dQ row partial (unscaled) | sum dz (.) Y | dgamma2 | dbeta2 | dgamma1 | dbeta1
// ---------------------------------------------------------------------------------------------------------
template <int W16>
__device__ __forceinline__ void zero_rows(float* base, long long slot, int g) {
  float* dst = base + slot * (16 * W16) + 4 * g;
#pragma unroll
  for (int b = 0; b < W16; ++b) *reinterpret_cast<f32x4*>(dst + 16 * b) = splat4(0.f);
}
__device__ __forceinline__ void zero_rows_rt(float* base, long long slot, int width, int g) {
  float* dst = base + slot * width + 4 * g;
  for (int b = 0; b < width / 16; ++b) *reinterpret_cast<f32x4*>(dst + 16 * b) = splat4(0.f);
}

template <int IBE, int IBF>
__global__ __launch_bounds__(256) void train_edge_bwd_kernel(TArgs a_in) {
  TArgs a = a_in;
  if (a.key_dev != nullptr) { a.k0 = a.key_dev[0]; a.k1 = a.key_dev[1]; }   // dropout key read at run time (graph replay)
  constexpr int DE = 16 * IBE, FFE = 16 * IBF;
  const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
  const long long total = (long long)a.B * a.n * a.nch;
  if ((long long)blockIdx.x * 4 + (threadIdx.x >> 6) >= total) return;
  const UnitId u = unit_id(a, lane);
  const int DX = a.dx, XB = DX / 16;
  const long long slot = u.unit * 16 + p;
  float* up = a.upart + u.unit * (2 * DX + 4 * DE);
  float* dEp = a.dE + ((long long)u.row * a.n + u.jj) * DE + 4 * g;
  if (!u.any) {
    if (u.inb)
#pragma unroll
      for (int eb = 0; eb < IBE; ++eb) *reinterpret_cast<f32x4*>(dEp + 16 * eb) = splat4(0.f);
    zero_rows<IBE>(a.G_e, slot, g);
    zero_rows_rt(a.G_da, slot, 2 * DX, g);
    zero_rows_rt(a.G_z, slot, DX, g);
    zero_rows<IBE>(a.G_dnewE, slot, g);
    zero_rows<IBE>(a.G_h1, slot, g);
    zero_rows<IBF>(a.G_df, slot, g);
    zero_rows<IBF>(a.G_f, slot, g);
    zero_rows<IBE>(a.G_do, slot, g);
    zero_rows_rt(a.G_dkv, slot, 2 * DX, g);
    for (int c = lane; c < 2 * DX + 4 * DE; c += 64) up[c] = 0.f;
    return;
  }
  const bool drop = a.p > 0.f;
  const float denom = 1.0f - a.p;
  const unsigned long long pair = ((unsigned long long)a.b0 * a.n + u.row) * a.n + u.jj;   // GLOBAL pair index: the dropout stream

  // ---- recompute the forward
  f32x4 e[IBE], newE[IBE];
  load_tile<IBE>(a, u, e, lane);
  attention_half<IBE, false>(a, u, e, newE, lane);
  Tail<IBE, IBF> t;
  tail_fwd<IBE, IBF>(a, u, e, newE, t, lane);

  // ---- LN2 backward
  f32x4 gy[IBE], dr2[IBE];
  {
    const float* gp = a.dOut + ((long long)u.row * a.n + u.jj) * DE + 4 * g;
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) gy[eb] = u.vj ? ld4(gp + 16 * eb) : splat4(0.f);
  }
  ln_bwd<IBE>(gy, t.xh2, t.rstd2, a.g2, g, dr2);
#pragma unroll
  for (int eb = 0; eb < IBE; ++eb) {
    f32x4 s1, s2;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = row_sum16(gy[eb][r] * t.xh2[eb][r]); s2[r] = row_sum16(gy[eb][r]); }
    if (p == 0) {
      *reinterpret_cast<f32x4*>(up + 2 * DX + 16 * eb + 4 * g) = s1;
      *reinterpret_cast<f32x4*>(up + 2 * DX + DE + 16 * eb + 4 * g) = s2;
    }
  }
  // ---- FFN backward
  f32x4 dov[IBE], dh1[IBE];
#pragma unroll
  for (int eb = 0; eb < IBE; ++eb) {
    dov[eb] = dr2[eb];
    if (drop) drop4(dov[eb], pair * DE + 16 * eb + 4 * g, a.site3, a.p, denom, a.k0, a.k1);
    dh1[eb] = dr2[eb];
    *reinterpret_cast<f32x4*>(a.G_do + slot * DE + 16 * eb + 4 * g) = dov[eb];
    *reinterpret_cast<f32x4*>(a.G_h1 + slot * DE + 16 * eb + 4 * g) = u.vj ? t.h1[eb] : splat4(0.f);
  }
#pragma unroll
  for (int fb = 0; fb < IBF; ++fb) {
    f32x4 df = splat4(0.f);
#pragma unroll
    for (int ob = 0; ob < IBE; ++ob) mfma4(df, frag(a.F_ff2T, fb * IBE + ob, lane), dov[ob]);
#pragma unroll
    for (int r = 0; r < 4; ++r) df[r] = t.hid[fb][r] > 0.f ? (drop ? df[r] / denom : df[r]) : 0.f;
    *reinterpret_cast<f32x4*>(a.G_df + slot * FFE + 16 * fb + 4 * g) = df;
    *reinterpret_cast<f32x4*>(a.G_f + slot * FFE + 16 * fb + 4 * g) = u.vj ? t.hid[fb] : splat4(0.f);
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) mfma4(dh1[eb], frag(a.F_ff1T, eb * IBF + fb, lane), df);
  }
  // ---- LN1 backward
  f32x4 dr1[IBE], dnewE[IBE];
  ln_bwd<IBE>(dh1, t.xh1, t.rstd1, a.g1, g, dr1);
#pragma unroll
  for (int eb = 0; eb < IBE; ++eb) {
    f32x4 s1, s2;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = row_sum16(dh1[eb][r] * t.xh1[eb][r]); s2[r] = row_sum16(dh1[eb][r]); }
    if (p == 0) {
      *reinterpret_cast<f32x4*>(up + 2 * DX + 2 * DE + 16 * eb + 4 * g) = s1;
      *reinterpret_cast<f32x4*>(up + 2 * DX + 3 * DE + 16 * eb + 4 * g) = s2;
    }
    dnewE[eb] = dr1[eb];
    if (drop) drop4(dnewE[eb], pair * DE + 16 * eb + 4 * g, a.site1, a.p, denom, a.k0, a.k1);
    *reinterpret_cast<f32x4*>(a.G_dnewE + slot * DE + 16 * eb + 4 * g) = dnewE[eb];
    *reinterpret_cast<f32x4*>(a.G_e + slot * DE + 16 * eb + 4 * g) = e[eb];
  }
  // ---- the dx blocks again: attention and FiLM backward (dr1 accumulates dE)
  const float* qrow = a.Q + (long long)u.row * DX;
  const float* krow = a.K + ((long long)u.b * a.n + u.jj) * DX;
  const float* vrow = a.V + ((long long)u.b * a.n + u.jj) * DX;
  const float* yea = a.yea + (long long)u.b * DX;
  const float* yem = a.yem1 + (long long)u.b * DX;
  const float* lse = a.lse + (long long)u.row * DX;
  const float* wV = a.wV + (long long)u.row * DX;
  const float* dwV = a.dwV + (long long)u.row * DX;
#pragma unroll 1
  for (int xb = 0; xb < XB; ++xb) {
    const int co = 16 * xb + 4 * g;
    f32x4 a1 = ld4(a.b_mul + co) + splat4(1.0f), a2 = ld4(a.b_add + co);
#pragma unroll
    for (int ib = 0; ib < IBE; ++ib) {
      mfma4(a1, frag(a.F_mul, xb * IBE + ib, lane), e[ib]);
      mfma4(a2, frag(a.F_add, xb * IBE + ib, lane), e[ib]);
    }
    f32x4 dz = splat4(0.f);
#pragma unroll
    for (int ob = 0; ob < IBE; ++ob) mfma4(dz, frag(a.F_outT, xb * IBE + ob, lane), dnewE[ob]);
    const f32x4 q = ld4(qrow + co) * a.inv_sqrt_df;
    const f32x4 k = u.vj ? ld4(krow + co) : splat4(0.f);
    const f32x4 v = u.vj ? ld4(vrow + co) : splat4(0.f);
    const f32x4 ya = ld4(yea + co), ym = ld4(yem + co), ls = ld4(lse + co), wv = ld4(wV + co), dw = ld4(dwV + co);
    f32x4 z, da1, da2, dk, dv, sq, sy;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float y0 = q[r] * k[r];
      const float yy = y0 * a1[r] + a2[r];
      z[r] = u.vj ? ya[r] + ym[r] * yy : 0.f;
      const float pe = u.vj ? __expf(yy - ls[r]) : 0.f;
      const float dY = u.vj ? ym[r] * dz[r] + pe * dw[r] * (v[r] - wv[r]) : 0.f;
      da1[r] = dY * y0;
      da2[r] = dY;
      dk[r] = dY * a1[r] * q[r];
      dv[r] = pe * dw[r];
      sq[r] = row_sum16(dY * a1[r] * k[r]);
      sy[r] = row_sum16(u.vj ? dz[r] * yy : 0.f);
    }
    *reinterpret_cast<f32x4*>(a.G_z + slot * DX + co) = z;
    *reinterpret_cast<f32x4*>(a.G_da + slot * (2 * DX) + co) = da1;
    *reinterpret_cast<f32x4*>(a.G_da + slot * (2 * DX) + DX + co) = da2;
    *reinterpret_cast<f32x4*>(a.G_dkv + slot * (2 * DX) + co) = dk;
    *reinterpret_cast<f32x4*>(a.G_dkv + slot * (2 * DX) + DX + co) = dv;
    if (p == 0) {
      *reinterpret_cast<f32x4*>(up + co) = sq;
      *reinterpret_cast<f32x4*>(up + DX + co) = sy;
    }
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) {
      mfma4(dr1[eb], frag(a.F_mulT, eb * XB + xb, lane), da1);
      mfma4(dr1[eb], frag(a.F_addT, eb * XB + xb, lane), da2);
    }
  }
  // the pooled statistics of E feed the y update (Etoy): their gradient lands on the same element
  if (a.st_mean != nullptr) {
    const int pidx = u.i * a.n + u.jj;
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) {
      const int c = u.b * DE + 16 * eb + 4 * g;
      const f32x4 mean = ld4(a.st_mean + c), cm = ld4(a.st_cm + c), cv = ld4(a.st_cv + c), gmi = ld4(a.st_gmi + c), gma = ld4(a.st_gma + c);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float add = cm[r] + cv[r] * (e[eb][r] - mean[r]);
        if (a.st_imn[c + r] == pidx) add += gmi[r];
        if (a.st_imx[c + r] == pidx) add += gma[r];
        dr1[eb][r] += add;
      }
    }
  }
  if (u.inb)
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) *reinterpret_cast<f32x4*>(dEp + 16 * eb) = u.vj ? dr1[eb] : splat4(0.f);
}

// fragment image of M = W (transpose = 0) or W^T (1), W [rows, cols] row-major; M is [R, Cc], both multiples of 16
__global__ __launch_bounds__(256) void train_edge_pack_kernel(const float* W, int rows, int cols, int transpose, float* out) {
  const int R = transpose ? cols : rows, Cc = transpose ? rows : cols;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;          // float4 index
  if (idx >= (long long)R * Cc / 4) return;
  const int l = (int)(idx & 63);
  const long long blk = idx >> 6;
  const int CB = Cc / 16, rb = (int)(blk / CB), cb = (int)(blk - (long long)rb * CB);
  const int r = 16 * rb + (l & 15), c0 = 16 * cb + 4 * (l >> 4);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = transpose ? W[(size_t)(c0 + e) * cols + r] : W[(size_t)r * cols + c0 + e];
  reinterpret_cast<f32x4*>(out)[idx] = v;
}

// every weight image of a layer (five matrices, plain and transposed) in ONE launch: grid (blocks of the largest, count)
struct PackList { const float* W[10]; float* out[10]; int rows[10], cols[10], transpose[10]; int count; };
__global__ __launch_bounds__(256) void train_edge_pack_multi_kernel(PackList l) {
  const int t = blockIdx.y;
  const float* W = l.W[t];
  const int rows = l.rows[t], cols = l.cols[t], transpose = l.transpose[t];
  const int R = transpose ? cols : rows, Cc = transpose ? rows : cols;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;          // float4 index
  if (idx >= (long long)R * Cc / 4) return;
  const int ln = (int)(idx & 63);
  const long long blk = idx >> 6;
  const int CB = Cc / 16, rb = (int)(blk / CB), cb = (int)(blk - (long long)rb * CB);
  const int r = 16 * rb + (ln & 15), c0 = 16 * cb + 4 * (ln >> 4);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = transpose ? W[(size_t)(c0 + e) * cols + r] : W[(size_t)r * cols + c0 + e];
  reinterpret_cast<f32x4*>(l.out[t])[idx] = v;
}

// ---------------------------------------------------------------------------------------------------------
// Etoy's pooled statistics of the layer input (layers.py:34-41) for the y update: per graph and channel, over the pairs of
// valid nodes (zero diagonal included): count, mean, M2 = sum (v - mean)^2, min, max and the flat pair index (i n + j) of
// the two extrema (the backward routes the gradient of min / max to that one element, as `.min(dim)[0]` does).  Grid
// (slices, B): workgroup (s, b) takes the rows i = s, s + slices, ...; thread = (4 channels, one of 256 / (de/4) key
// lanes); each thread keeps sum and sum of squares of (v - k), k = its first value (no cancellation), slices of a
// workgroup are merged with the pairwise update (Chan et al.), the workgroups of a graph by the host (tiny tensors).
// out [B, slices, 5, de] floats (cnt replicated | mean | M2 | min | max), idx [B, slices, 2, de] int32.  One read of E.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void train_edge_stats_kernel(const float* __restrict__ E, const uint8_t* __restrict__ mask, int n,
                                                               int DE, float* __restrict__ out, int* __restrict__ idx) {
  __shared__ f32x4 sh[5][256];
  __shared__ int shi[2][256][4];
  const int tid = threadIdx.x, G4 = DE / 4, KL = 256 / G4, c4 = tid % G4, kl = tid / G4;
  const int b = blockIdx.y, slices = gridDim.x;
  const uint8_t* mb = mask + (size_t)b * n;
  f32x4 k = splat4(0.f), s1 = k, q = k, mn = splat4(INFINITY), mx = splat4(-INFINITY);
  int imn[4] = {0, 0, 0, 0}, imx[4] = {0, 0, 0, 0};
  int cnt = 0;
  for (int i = blockIdx.x; i < n; i += slices) {
    if (mb[i] == 0) continue;
    const float* row = E + ((size_t)b * n + i) * n * DE + 4 * c4;
    for (int j = kl; j < n; j += KL) {
      if (mb[j] == 0) continue;
      const f32x4 v = ld4(row + (size_t)j * DE);
      if (cnt == 0) k = v;
      ++cnt;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[e] - k[e];
        s1[e] += d; q[e] += d * d;
        if (v[e] < mn[e]) { mn[e] = v[e]; imn[e] = i * n + j; }
        if (v[e] > mx[e]) { mx[e] = v[e]; imx[e] = i * n + j; }
      }
    }
  }
  // per-thread (cnt, mean, M2)
  f32x4 mean = k, M2 = splat4(0.f);
  if (cnt > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { mean[e] = k[e] + s1[e] / (float)cnt; M2[e] = q[e] - s1[e] * s1[e] / (float)cnt; }
  }
  sh[0][tid] = splat4((float)cnt); sh[1][tid] = mean; sh[2][tid] = M2; sh[3][tid] = mn; sh[4][tid] = mx;
#pragma unroll
  for (int e = 0; e < 4; ++e) { shi[0][tid][e] = imn[e]; shi[1][tid][e] = imx[e]; }
  __syncthreads();
  if (kl == 0) {
    float ca = (float)cnt;
    for (int t = 1; t < KL; ++t) {
      const int o = t * G4 + c4;
      const float cb = sh[0][o][0];
      if (cb == 0.f) continue;
      const f32x4 mb2 = sh[1][o], Mb2 = sh[2][o], nb = sh[3][o], xb = sh[4][o];
      const float tot = ca + cb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = mb2[e] - mean[e];
        M2[e] = ca == 0.f ? Mb2[e] : M2[e] + Mb2[e] + d * d * (ca * cb / tot);
        mean[e] = ca == 0.f ? mb2[e] : mean[e] + d * (cb / tot);
        if (nb[e] < mn[e] || (nb[e] == mn[e] && shi[0][o][e] < imn[e])) { mn[e] = nb[e]; imn[e] = shi[0][o][e]; }
        if (xb[e] > mx[e] || (xb[e] == mx[e] && shi[1][o][e] < imx[e])) { mx[e] = xb[e]; imx[e] = shi[1][o][e]; }
      }
      ca = tot;
    }
    float* o = out + ((size_t)b * slices + blockIdx.x) * 5 * DE + 4 * c4;
    *reinterpret_cast<f32x4*>(o) = splat4(ca);
    *reinterpret_cast<f32x4*>(o + DE) = mean;
    *reinterpret_cast<f32x4*>(o + 2 * DE) = M2;
    *reinterpret_cast<f32x4*>(o + 3 * DE) = mn;
    *reinterpret_cast<f32x4*>(o + 4 * DE) = mx;
    int* oi = idx + ((size_t)b * slices + blockIdx.x) * 2 * DE + 4 * c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { oi[e] = imn[e]; oi[DE + e] = imx[e]; }
  }
}

// The slices of train_edge_stats_kernel merged per (graph, channel) by the pairwise update of mean / M2; ties of an extremum go
// to the smaller pair index, the rule inside a slice.  stats [B, 4 de] = mean | min | max | var, what Etoy hands to the y update,
// and what the backward needs: mean, count, arg-extrema [B, de] each.  One workgroup per graph, thread = (channel, one of QL
// slice lanes): a lane folds the slices sl, sl + QL, ... (a fixed order; as one thread per channel walking all slices it was a
// chain of ~64 dependent round trips, 80 us), lane 0 then folds the QL lane results in lane order.
struct StatAcc { float ca, mean, M2, mn, mx; int imn, imx; };
__device__ __forceinline__ void stat_fold(StatAcc& a, float cb, float mb, float Mb, float nb, float xb, int inb, int ixb) {
  if (cb == 0.f) return;
  const float tot = a.ca + cb, d = mb - a.mean;
  a.M2 = a.ca == 0.f ? Mb : a.M2 + Mb + d * d * (a.ca * cb / tot);
  a.mean = a.ca == 0.f ? mb : a.mean + d * (cb / tot);
  if (nb < a.mn || (nb == a.mn && inb < a.imn)) { a.mn = nb; a.imn = inb; }
  if (xb > a.mx || (xb == a.mx && ixb < a.imx)) { a.mx = xb; a.imx = ixb; }
  a.ca = tot;
}
__global__ __launch_bounds__(1024) void train_edge_stats_finish_kernel(const float* __restrict__ so, const int* __restrict__ si,
                                                                       int slices, int DE, int QL, float* __restrict__ stats,
                                                                       float* __restrict__ mean_o, float* __restrict__ tot_o,
                                                                       int* __restrict__ imn_o, int* __restrict__ imx_o) {
  __shared__ float shf[5][1024];
  __shared__ int shi[2][1024];
  const int b = blockIdx.x, tid = threadIdx.x, c = tid % DE, sl = tid / DE;
  StatAcc a{0.f, 0.f, 0.f, INFINITY, -INFINITY, 0, 0};
  const float* ob = so + (size_t)b * slices * 5 * DE + c;
  const int* ib = si + (size_t)b * slices * 2 * DE + c;
  int s = sl;
  for (; s + 3 * QL < slices; s += 4 * QL) {           // four slices' worth of loads in flight
    float v[4][5];
    int ix[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* o = ob + (size_t)(s + u * QL) * 5 * DE;
      const int* oi = ib + (size_t)(s + u * QL) * 2 * DE;
#pragma unroll
      for (int f = 0; f < 5; ++f) v[u][f] = o[f * DE];
      ix[u][0] = oi[0]; ix[u][1] = oi[DE];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) stat_fold(a, v[u][0], v[u][1], v[u][2], v[u][3], v[u][4], ix[u][0], ix[u][1]);
  }
  for (; s < slices; s += QL) {
    const float* o = ob + (size_t)s * 5 * DE;
    const int* oi = ib + (size_t)s * 2 * DE;
    stat_fold(a, o[0], o[DE], o[2 * DE], o[3 * DE], o[4 * DE], oi[0], oi[DE]);
  }
  shf[0][tid] = a.ca; shf[1][tid] = a.mean; shf[2][tid] = a.M2; shf[3][tid] = a.mn; shf[4][tid] = a.mx;
  shi[0][tid] = a.imn; shi[1][tid] = a.imx;
  __syncthreads();
  if (sl != 0) return;
  for (int q = 1; q < QL; ++q) {
    const int o = q * DE + c;
    stat_fold(a, shf[0][o], shf[1][o], shf[2][o], shf[3][o], shf[4][o], shi[0][o], shi[1][o]);
  }
  const float tot = a.ca < 1.f ? 1.f : a.ca;
  float* st = stats + (size_t)b * 4 * DE + c;
  st[0] = a.mean; st[DE] = a.mn; st[2 * DE] = a.mx; st[3 * DE] = a.M2 / tot;
  const int idx = b * DE + c;
  mean_o[idx] = a.mean; tot_o[idx] = tot; imn_o[idx] = a.imn; imx_o[idx] = a.imx;
}

// ---------------------------------------------------------------------------------------------------------
// Sum over the middle index: out[split][a][c] = sum over the rows r of slice `split` of src[a][r][c]  (src [A,R,C] fp32,
// C % 4 == 0).  The backward's per-key gradients (dK | dV summed over the query rows) and per-graph sums of a factor tensor
// are reductions of this shape over hundreds of MB; torch's generic reduce kernel ran them at ~1 TB/s.  Bound: HBM.
// Workgroup = (tile of CQ column quads, row slice, a); thread = (column quad, one of 256 / CQ row lanes), eight float4 loads
// in flight per thread, the row lanes merged through LDS in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void train_sum_rows_kernel(const float* __restrict__ src, long long R, int C4, int CQ, int rows_per_split,
                                                             float* __restrict__ out, long long ld_split) {
  __shared__ f32x4 sh[256];
  const int tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ, RL = 256 / CQ;
  const int c4 = blockIdx.x * CQ + cq;
  const long long a = blockIdx.z, r0 = (long long)blockIdx.y * rows_per_split;
  const long long r1 = r0 + rows_per_split < R ? r0 + rows_per_split : R;
  f32x4 acc = splat4(0.f);
  if (c4 < C4) {
    const float* p = src + (a * R) * 4ll * C4 + 4ll * c4;
    long long r = r0 + rl;
    for (; r + 7ll * RL < r1; r += 8ll * RL) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4(p + (r + (long long)u * RL) * 4ll * C4);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < r1; r += RL) acc += ld4(p + r * 4ll * C4);
  }
  sh[tid] = acc;
  __syncthreads();
  if (rl == 0 && c4 < C4) {
    for (int t = 1; t < RL; ++t) acc += sh[t * CQ + cq];
    *reinterpret_cast<f32x4*>(out + (long long)blockIdx.y * ld_split + (a * C4 + c4) * 4ll) = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient: C[M,N] = G^T X and c[M] = column sums of G, G [S,M] and X [S,N] row-major factors written by the
// backward kernel, S = pair slots (the reduction index, ~2.6e5 at B = 4, n = 256).  Bound: HBM and the fp32 matrix pipe
// about equally ((M + N) 4 bytes and 2 M N flop per slot: 30-50 flop/B against a ridge of ~25).
// Grid (slabs, groups of TPW <= 4 output tiles of 64x64); a workgroup owns a slab of slots for ALL tiles of its group, so the
// operand shared by those tiles is fetched from HBM once (its other readers hit L1 / L2): wave w works on tile w / tw of the
// group and, when tw = 4 / TPW > 1 waves share a tile, on every tw-th step of the slab; such waves add their tiles through
// LDS in a fixed order.  One partial per workgroup and tile: row `slab index` of a [slabs, ld] buffer, added by the caller in
// a fixed order (deterministic, no atomics).
// Operand trick: lane (m = l & 15, k = l >> 4) loads ONE float4 = G[s0 + k][64 mt + 4m .. 4m+3] (a slot's 64 channels are 256
// contiguous bytes over the 16 lanes) and register j of that float4 is the A operand of MFMA j: its row m then stands for
// channel 4m + j.  X likewise as B.  The 16 MFMAs (j, j') of a step cover the 64 x 64 tile over 4 slots from two 1-KiB loads;
// four steps of operands are in flight ahead of the MFMAs (two waves per SIMD at this register count: the loads in flight,
// not the occupancy, have to cover the HBM latency).  The column sums of G are vector adds on the operand registers.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void train_wgrad_kernel(const float* __restrict__ G, int M, const float* __restrict__ X, int N,
                                                          long long S, int slab, float* __restrict__ partial, long long ld, int TPW) {
  __shared__ f32x4 red[2][17][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, k = lane >> 4;
  const int ntn = (N + 63) / 64, tiles = ((M + 63) / 64) * ntn;
  const int tw = 4 / TPW, tslot = wave / tw, sub = wave - tslot * tw;
  const int tile = blockIdx.y * TPW + tslot;
  const bool live = tile < tiles;                            // wave-uniform; a wave without a tile still meets the barriers
  const int mt = tile / ntn, nt = tile - mt * ntn;
  const long long s_begin = (long long)blockIdx.x * slab;
  const long long s_end = s_begin + slab < S ? s_begin + slab : S;
  const int gc = 64 * mt + 4 * m, xc = 64 * nt + 4 * m;
  const bool g_ok = live && gc < M, x_ok = live && xc < N;   // widths are multiples of 16 >= 16: a float4 is in or out as a whole
  f32x4 acc[4][4], gs = splat4(0.f);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = splat4(0.f);
  const bool bias = live && nt == 0;
  const float* Gp = G + gc;
  const float* Xp = X + xc;
  const int STEP = 4 * tw;
  auto fetch = [&](long long s0, f32x4& gv, f32x4& xv) {
    const long long slot = s0 + k;
    const bool ok = slot < s_end;
    gv = (ok && g_ok) ? ld4(Gp + slot * M) : splat4(0.f);
    xv = (ok && x_ok) ? ld4(Xp + slot * N) : splat4(0.f);
  };
  auto step = [&](const f32x4& gv, const f32x4& xv) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[j], xv[i], acc[j][i], 0, 0, 0);
    gs += gv;
  };
  if (live) {
    long long s0 = s_begin + 4 * sub;
    f32x4 gq[4], xq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) fetch(s0 + (long long)u * STEP, gq[u], xq[u]);
    for (; s0 < s_end; s0 += 4ll * STEP) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        step(gq[u], xq[u]);
        fetch(s0 + (long long)(4 + u) * STEP, gq[u], xq[u]);
      }
    }
  }
  // waves sharing a tile: tw = 4: (w0 += w2, w1 += w3), then w0 += w1; tw = 2: the pair; tw = 1: nothing to add
  for (int half = tw >> 1; half >= 1; half >>= 1) {
    if (sub >= half && sub < 2 * half) {
      f32x4(*r)[64] = red[tslot * half + sub - half];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) r[4 * j + i][lane] = acc[j][i];
      r[16][lane] = gs;
    }
    __syncthreads();
    if (sub < half) {
      f32x4(*r)[64] = red[tslot * half + sub];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] += r[4 * j + i][lane];
      gs += r[16][lane];
    }
    __syncthreads();
  }
  if (sub != 0 || !live) return;
  // acc[j][i][e]: row (4 k + e) of MFMA (j, i) = G channel 64 mt + 4 (4k + e) + j; column m = X channel 64 nt + 4 m + i
  float* out = partial + (size_t)blockIdx.x * (size_t)ld;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 64 * mt + 4 * (4 * k + e) + j;
      if (row >= M) continue;
      if (x_ok) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[j][i][e];
        *reinterpret_cast<f32x4*>(out + (size_t)row * N + xc) = v;
      }
    }
  if (bias) {                                                // column sums: the four slot groups of a step, then one store per channel quad
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gs[e] += __shfl_xor(gs[e], 16);
      gs[e] += __shfl_xor(gs[e], 32);
    }
    if (k == 0 && g_ok) *reinterpret_cast<f32x4*>(out + (size_t)M * N + gc) = gs;
  }
}

int check_args(const TArgs* a, bool bwd) {
  if (!a) return fail("train_edge: null arguments");
  if (a->B <= 0 || a->n <= 0 || a->nch != (a->n + 15) / 16) return fail("train_edge: bad B / n / nch");
  if (a->dx < 16 || a->dx % 16 != 0) return fail("train_edge: dx = %d must be a multiple of 16", a->dx);
  if (!((a->de == 32 && a->ffe == 64) || (a->de == 64 && a->ffe == 128)))
    return fail("train_edge: (de, dim_ffE) = (%d, %d) not built (32/64 and 64/128 are)", a->de, a->ffe);
  if (!(a->p >= 0.f) || !(a->p < 1.f)) return fail("train_edge: bad dropout p");
  if (!a->E || !a->Q || !a->K || !a->V || !a->yea || !a->yem1 || !a->mask || !a->F_mul || !a->F_add || !a->F_out || !a->F_ff1 ||
      !a->F_ff2 || !a->b_mul || !a->b_add || !a->b_out || !a->g1 || !a->b1 || !a->b_ff1 || !a->b_ff2 || !a->g2 || !a->b2)
    return fail("train_edge: null input");
  if (!bwd && (!a->E_out || !a->part || !a->wV_out || !a->lse_out)) return fail("train_edge forward: null output");
  if (bwd && (!a->F_mulT || !a->F_addT || !a->F_outT || !a->F_ff1T || !a->F_ff2T || !a->dOut || !a->dwV || !a->wV || !a->lse || !a->dE ||
              !a->G_e || !a->G_da || !a->G_z || !a->G_dnewE || !a->G_h1 || !a->G_df || !a->G_f || !a->G_do || !a->G_dkv || !a->upart))
    return fail("train_edge backward: null argument");
  if ((long long)a->B * a->n * a->nch / 4 + 1 > 0x7fffffffLL) return fail("train_edge: too many units for one launch");
  return 0;
}

}  // namespace
}  // namespace scgg

using namespace scgg;

extern "C" int scgg_train_edge_pack(const float* W, int32_t rows, int32_t cols, int32_t transpose, float* out, void* stream) {
  if (!W || !out || rows <= 0 || cols <= 0 || rows % 16 != 0 || cols % 16 != 0) return fail("train_edge_pack: bad arguments");
  const long long f4 = (long long)rows * cols / 4;
  hipLaunchKernelGGL(train_edge_pack_kernel, dim3((unsigned)((f4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), W, rows,
                     cols, transpose, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_edge_pack: %s", hipGetErrorString(e));
}

extern "C" int scgg_train_edge_pack_multi(const scgg_train_pack_list* l, void* stream) {
  if (!l || l->count <= 0 || l->count > 10) return fail("train_edge_pack_multi: 1 .. 10 matrices");
  PackList k;
  long long most = 0;
  for (int t = 0; t < l->count; ++t) {
    if (!l->W[t] || !l->out[t] || l->rows[t] <= 0 || l->cols[t] <= 0 || l->rows[t] % 16 != 0 || l->cols[t] % 16 != 0)
      return fail("train_edge_pack_multi: bad matrix %d", t);
    k.W[t] = l->W[t]; k.out[t] = l->out[t]; k.rows[t] = l->rows[t]; k.cols[t] = l->cols[t]; k.transpose[t] = l->transpose[t];
    const long long f4 = (long long)l->rows[t] * l->cols[t] / 4;
    most = f4 > most ? f4 : most;
  }
  for (int t = l->count; t < 10; ++t) { k.W[t] = nullptr; k.out[t] = nullptr; k.rows[t] = k.cols[t] = k.transpose[t] = 0; }
  k.count = l->count;
  hipLaunchKernelGGL(train_edge_pack_multi_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)l->count), dim3(256), 0,
                     static_cast<hipStream_t>(stream), k);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_edge_pack_multi: %s", hipGetErrorString(e));
}

extern "C" int scgg_train_edge_forward(const scgg_train_edge_args* a, void* stream) {
  if (check_args(a, false)) return -1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long units = (long long)a->B * a->n * a->nch;
  const dim3 grid((unsigned)((units + 3) / 4)), blk(256);
  if (a->de == 32) hipLaunchKernelGGL((train_edge_fwd_kernel<2, 4>), grid, blk, 0, st, *a);
  else hipLaunchKernelGGL((train_edge_fwd_kernel<4, 8>), grid, blk, 0, st, *a);
  const long long el = (long long)a->B * a->n * a->dx;
  hipLaunchKernelGGL(train_edge_merge_kernel, dim3((unsigned)((el + 255) / 256)), dim3(256), 0, st, *a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_edge_forward: %s", hipGetErrorString(e));
}

extern "C" int scgg_train_edge_backward(const scgg_train_edge_args* a, void* stream) {
  if (check_args(a, true)) return -1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long units = (long long)a->B * a->n * a->nch;
  const dim3 grid((unsigned)((units + 3) / 4)), blk(256);
  if (a->de == 32) hipLaunchKernelGGL((train_edge_bwd_kernel<2, 4>), grid, blk, 0, st, *a);
  else hipLaunchKernelGGL((train_edge_bwd_kernel<4, 8>), grid, blk, 0, st, *a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_edge_backward: %s", hipGetErrorString(e));
}

extern "C" int64_t scgg_train_wgrad_slabs(int64_t S, int32_t slab) { return (S <= 0 || slab <= 0) ? 0 : (S + slab - 1) / slab; }

extern "C" int scgg_train_wgrad(const float* G, int32_t M, const float* X, int32_t N, int64_t S, int32_t slab, float* partial,
                                int64_t ld_partial, void* stream) {
  if (!G || !X || !partial || S <= 0) return fail("train_wgrad: bad arguments");
  if (ld_partial == 0) ld_partial = (int64_t)M * N + M;
  if (ld_partial < (int64_t)M * N + M || ld_partial % 4 != 0) return fail("train_wgrad: ld_partial must be a multiple of 4, at least M N + M");
  if (M < 16 || M % 16 != 0 || N < 16 || N % 16 != 0 || slab < 16 || slab % 16 != 0) return fail("train_wgrad: M, N, slab must be multiples of 16");
  const long long nsl = (S + slab - 1) / slab;
  const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  const int tpw = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);     // output tiles per workgroup
  const int groups = (tiles + tpw - 1) / tpw;
  if (nsl > 0x7fffffffLL || groups > 65535) return fail("train_wgrad: too many slabs / tiles");
  hipLaunchKernelGGL(train_wgrad_kernel, dim3((unsigned)nsl, (unsigned)groups), dim3(256), 0, static_cast<hipStream_t>(stream), G, (int)M,
                     X, (int)N, (long long)S, (int)slab, partial, (long long)ld_partial, tpw);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_wgrad: %s", hipGetErrorString(e));
}

extern "C" int scgg_train_edge_stats(const float* E, const uint8_t* mask, int32_t B, int32_t n, int32_t de, int32_t slices, float* out,
                                     int32_t* idx, void* stream) {
  if (!E || !mask || !out || !idx || B <= 0 || n <= 0 || slices <= 0) return fail("train_edge_stats: bad arguments");
  if (de % 4 != 0 || 256 % (de / 4) != 0) return fail("train_edge_stats: de = %d not supported", de);
  hipLaunchKernelGGL(train_edge_stats_kernel, dim3(slices, B), dim3(256), 0, static_cast<hipStream_t>(stream), E, mask, n, de, out, idx);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_edge_stats: %s", hipGetErrorString(e));
}

extern "C" int scgg_train_edge_stats_finish(const float* out, const int32_t* idx, int32_t B, int32_t slices, int32_t de, float* stats,
                                            float* mean, float* count, int32_t* imn, int32_t* imx, void* stream) {
  if (!out || !idx || !stats || !mean || !count || !imn || !imx || B <= 0 || slices <= 0 || de <= 0)
    return fail("train_edge_stats_finish: bad arguments");
  if (de > 1024) return fail("train_edge_stats_finish: de = %d beyond 1024", de);
  int ql = 256 / de;                                                     // slice lanes per channel
  if (ql < 1) ql = 1;
  if (ql > slices) ql = slices;
  hipLaunchKernelGGL(train_edge_stats_finish_kernel, dim3((unsigned)B), dim3((unsigned)(de * ql)), 0, static_cast<hipStream_t>(stream), out,
                     idx, slices, de, ql, stats, mean, count, imn, imx);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_edge_stats_finish: %s", hipGetErrorString(e));
}

extern "C" int64_t scgg_train_sum_rows_splits(int64_t A, int64_t R, int32_t C) {
  if (A <= 0 || R <= 0 || C <= 0) return 0;
  const long long tiles = ((C / 4) + 63) / 64 * A;
  long long want = tiles >= 512 ? 1 : 2048 / (tiles > 0 ? tiles : 1);   // ~8 workgroups per CU over the grid; 2+ per CU: no split
  if (want < 1) want = 1;
  const long long by_rows = (R + 63) / 64;                              // at least 64 rows per slice
  return want < by_rows ? want : by_rows;
}

extern "C" int scgg_train_sum_rows(const float* src, int64_t A, int64_t R, int32_t C, int32_t splits, float* out, int64_t ld_split,
                                   void* stream) {
  if (!src || !out || A <= 0 || R <= 0 || C <= 0 || splits <= 0) return fail("train_sum_rows: bad arguments");
  if (ld_split == 0) ld_split = A * C;
  if (ld_split < A * C || ld_split % 4 != 0) return fail("train_sum_rows: ld_split must be a multiple of 4, at least A C");
  if (C % 4 != 0) return fail("train_sum_rows: C must be a multiple of 4");
  if (A > 65535 || splits > 65535) return fail("train_sum_rows: grid too large");
  const int C4 = C / 4;
  int CQ = 64;
  while (CQ > 1 && CQ / 2 >= C4) CQ /= 2;                                // power of two >= min(C4, 64): the row lanes take the rest
  const int rows_per_split = (int)((R + splits - 1) / splits);
  hipLaunchKernelGGL(train_sum_rows_kernel, dim3((unsigned)((C4 + CQ - 1) / CQ), (unsigned)splits, (unsigned)A), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, (long long)R, C4, CQ, rows_per_split, out, (long long)ld_split);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_sum_rows: %s", hipGetErrorString(e));
}
