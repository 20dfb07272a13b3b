// Pooled statistics of one graph by one 256-thread workgroup: Xtoy (layers.py:11-19) and Etoy (layers.py:34-41), shared by
// the stand-alone kernels (node_stats_kernel, edge_pool_stats_kernel) and by y_layer_kernel, which takes the statistics of
// its own graph itself when the forward runs on one stream (two launches and their dependent-launch gaps less per layer).
// The destination `z` may be global memory or LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "kernels.h"

namespace scgg {

constexpr int PS_T = 256;          // threads of the workgroup

struct PoolScratch {               // 15.4 KB of LDS
  f32x4_t red[3][PS_T];
  f32x4_t smean[128];
  short vidx[512];                 // the valid nodes, in order
  int wcount[PS_T / 64];
};

// the valid nodes of the graph into ps.vidx (n <= 512); returns their number.  Contains barriers.
__device__ __forceinline__ int pool_compact_mask(PoolScratch& ps, const uint8_t* m, int n, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  int nv = 0;
  for (int i0 = 0; i0 < n; i0 += PS_T) {
    const int i = i0 + tid;
    const bool v = i < n && m[i] != 0;
    const unsigned long long bal = __ballot(v);
    if (lane == 0) ps.wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = nv, tot = 0;
    for (int w = 0; w < PS_T / 64; ++w) {
      if (w < wave) off += ps.wcount[w];
      tot += ps.wcount[w];
    }
    if (v) ps.vidx[off + __popcll(bal & ((1ull << lane) - 1ull))] = (short)i;
    nv += tot;
    __syncthreads();
  }
  return nv;
}

// Xtoy of one graph: z[0:dx | dx:2dx | 2dx:3dx | 3dx:4dx] = mean | min | max | biased variance over the nv valid nodes of
// X[n,dx].  thread = (4 channels, slice of the rows): float4 loads, the rows of a slice are independent loads in flight
// together.  dx % 4 == 0, dx / 4 divides PS_T.  Contains barriers; z is complete for every thread after the caller's next
// barrier.
__device__ __forceinline__ void pool_xtoy_graph(PoolScratch& ps, const float* Xg, int nv, int dx, float* z, int tid) {
  const int G4 = dx / 4, S = PS_T / G4, c4 = tid % G4, sl = tid / G4;
  const float* X = Xg + 4 * c4;
  f32x4_t s = {0.f, 0.f, 0.f, 0.f}, mn = {INFINITY, INFINITY, INFINITY, INFINITY}, mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  if (sl < S) {
#pragma unroll 4
    for (int k = sl; k < nv; k += S) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(X + (long long)ps.vidx[k] * dx);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += v[e]; mn[e] = fminf(mn[e], v[e]); mx[e] = fmaxf(mx[e], v[e]); }
    }
  }
  ps.red[0][tid] = s; ps.red[1][tid] = mn; ps.red[2][tid] = mx;
  __syncthreads();
  if (sl == 0) {
    for (int t = 1; t < S; ++t) {
      const f32x4_t s2 = ps.red[0][t * G4 + c4], n2 = ps.red[1][t * G4 + c4], x2 = ps.red[2][t * G4 + c4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += s2[e]; mn[e] = fminf(mn[e], n2[e]); mx[e] = fmaxf(mx[e], x2[e]); }
    }
    ps.smean[c4] = s / (float)nv;
    ps.red[1][c4] = mn; ps.red[2][c4] = mx;
  }
  __syncthreads();
  const f32x4_t mean = ps.smean[c4];
  f32x4_t q = {0.f, 0.f, 0.f, 0.f};
  if (sl < S) {
#pragma unroll 4
    for (int k = sl; k < nv; k += S) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(X + (long long)ps.vidx[k] * dx);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[e] - mean[e]; q[e] += d * d; }
    }
  }
  __syncthreads();
  ps.red[0][tid] = q;
  __syncthreads();
  if (sl == 0) {
    for (int t = 1; t < S; ++t) q += ps.red[0][t * G4 + c4];
    const f32x4_t mnv = ps.red[1][c4], mxv = ps.red[2][c4];
    float* zz = z + 4 * c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      zz[e] = mean[e]; zz[dx + e] = mnv[e]; zz[2 * dx + e] = mxv[e]; zz[3 * dx + e] = q[e] / (float)nv;
    }
  }
  __syncthreads();
}

// Etoy of one graph: z = [mean | min | max | biased variance] per channel over the nv x nv pairs of the valid nodes of the
// SYMMETRISED masked tensor (the stored one is not: mfma_chain.h load_sym_tile), zero diagonal included.  thread = (4
// channels c4, slice sl of the unordered pairs); each stored element is read once (pair {i,j} = 0.5 (E_ij + E_ji), weight
// 2); one pass: thread t keeps, for its slice of the pairs, the sum and the sum of squares of (v - k_t), k_t = the first
// value it sees: the differences are of the size of the spread, so M2_t = q - s^2 / n_t loses nothing to cancellation; the
// slices are then merged with the pairwise-variance update (Chan et al.).  The pairs of a slice are taken four at a time:
// their eight loads are in flight together (one at a time, a slice was a chain of ~15 L2 round trips).
// de % 4 == 0, de / 4 divides PS_T.  Contains barriers.
__device__ __forceinline__ void pool_etoy_graph(PoolScratch& ps, const float* Eg, int n, int nv, int DE, float* z, int tid) {
  const int G4 = DE / 4, S = PS_T / G4, c4 = tid % G4, sl = tid / G4;
  const float* E = Eg + 4 * c4;
  const float cnt = (float)nv * (float)nv;
  f32x4_t k = {0.f, 0.f, 0.f, 0.f}, s1 = k, q = k, mn = {INFINITY, INFINITY, INFINITY, INFINITY},
          mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int cnt_t = 0;
  {
    // the unordered pairs ii < jj in row-major order; slice sl takes pairs sl, sl + S, ...: (ii, r) walks that sequence,
    // r = position inside row ii (jj = ii + 1 + r), row ii holds nv - 1 - ii pairs
    int ii = 0, r = sl;
    bool more = true;
    while (more) {
      f32x4_t x[4], y[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        while (ii < nv - 1 && r >= nv - 1 - ii) { r -= nv - 1 - ii; ++ii; }
        ok[u] = ii < nv - 1;
        if (ok[u]) {
          const long long i = ps.vidx[ii], j = ps.vidx[ii + 1 + r];
          x[u] = *reinterpret_cast<const f32x4_t*>(E + (i * n + j) * DE);
          y[u] = *reinterpret_cast<const f32x4_t*>(E + (j * n + i) * DE);
          r += S;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) { more = false; continue; }
        const f32x4_t v = 0.5f * (x[u] + y[u]);
        if (cnt_t == 0) k = v;
        ++cnt_t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[e] - k[e];
          s1[e] += d; q[e] += d * d;
          mn[e] = fminf(mn[e], v[e]); mx[e] = fmaxf(mx[e], v[e]);
        }
      }
    }
  }
  if (nv > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { mn[e] = fminf(mn[e], 0.f); mx[e] = fmaxf(mx[e], 0.f); }   // the diagonal
  }
  const float nt = (float)cnt_t, inv_nt = cnt_t > 0 ? 1.f / nt : 0.f;
  f32x4_t mean_t, m2_t;
#pragma unroll
  for (int e = 0; e < 4; ++e) { mean_t[e] = k[e] + s1[e] * inv_nt; m2_t[e] = q[e] - s1[e] * s1[e] * inv_nt; }
  ps.red[0][tid] = mean_t; ps.red[1][tid] = m2_t;
  __syncthreads();
  f32x4_t mean = mean_t, m2 = m2_t;
  const int np_all = nv * (nv - 1) / 2;
  if (sl == 0) {
    float na = nt;
    for (int t = 1; t < S; ++t) {
      const int c = t < np_all ? (np_all - t + S - 1) / S : 0;        // pairs of slice t: t, t + S, ...
      if (c == 0) break;
      const float nb = (float)c, nab = na + nb;
      const f32x4_t mb = ps.red[0][t * G4 + c4], qb = ps.red[1][t * G4 + c4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = mb[e] - mean[e];
        mean[e] += d * (nb / nab);
        m2[e] += qb[e] + d * d * (na * nb / nab);
      }
      na = nab;
    }
  }
  __syncthreads();
  ps.red[0][tid] = mn; ps.red[1][tid] = mx;
  __syncthreads();
  if (sl == 0) {
    for (int t = 1; t < S; ++t) {
      const f32x4_t n2 = ps.red[0][t * G4 + c4], x2 = ps.red[1][t * G4 + c4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { mn[e] = fminf(mn[e], n2[e]); mx[e] = fmaxf(mx[e], x2[e]); }
    }
    // from the unordered pairs (mean, M2 about it, np_all of them) to all nv x nv entries: every pair counts twice, the
    // nv diagonal entries are zero
    float* zz = z + 4 * c4;
    const float npf = (float)np_all;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float m = 2.f * npf * mean[e] / cnt;
      const float dm = mean[e] - m;
      const float qq = 2.f * (m2[e] + npf * dm * dm) + (float)nv * m * m;
      zz[e] = m; zz[DE + e] = mn[e]; zz[2 * DE + e] = mx[e]; zz[3 * DE + e] = qq / cnt;
    }
  }
  __syncthreads();
}

// The same statistics with SEVERAL workgroups per graph (few large graphs: one workgroup per graph leaves the chip idle --
// four workgroups took 0.76 ms per layer at B = 4, n = 256).  Workgroup w of W takes the slices w S .. w S + S - 1 of the W S
// slices of the unordered pairs, merges them as above and leaves part[5][DE] = (pairs | mean | M2 | min | max) of its pairs;
// pool_etoy_finish merges the W parts in order (deterministic) and goes from the unordered pairs to all nv x nv entries.
__device__ __forceinline__ void pool_etoy_part(PoolScratch& ps, const float* Eg, int n, int nv, int DE, int w, int W, float* part,
                                               int tid) {
  const int G4 = DE / 4, S = PS_T / G4, c4 = tid % G4, sl = tid / G4, ST = S * W, gsl = w * S + sl;
  const float* E = Eg + 4 * c4;
  f32x4_t k = {0.f, 0.f, 0.f, 0.f}, s1 = k, q = k, mn = {INFINITY, INFINITY, INFINITY, INFINITY},
          mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int cnt_t = 0;
  {
    int ii = 0, r = gsl;
    bool more = true;
    while (more) {
      f32x4_t x[4], y[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        while (ii < nv - 1 && r >= nv - 1 - ii) { r -= nv - 1 - ii; ++ii; }
        ok[u] = ii < nv - 1;
        if (ok[u]) {
          const long long i = ps.vidx[ii], j = ps.vidx[ii + 1 + r];
          x[u] = *reinterpret_cast<const f32x4_t*>(E + (i * n + j) * DE);
          y[u] = *reinterpret_cast<const f32x4_t*>(E + (j * n + i) * DE);
          r += ST;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) { more = false; continue; }
        const f32x4_t v = 0.5f * (x[u] + y[u]);
        if (cnt_t == 0) k = v;
        ++cnt_t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[e] - k[e];
          s1[e] += d; q[e] += d * d;
          mn[e] = fminf(mn[e], v[e]); mx[e] = fmaxf(mx[e], v[e]);
        }
      }
    }
  }
  const float nt = (float)cnt_t, inv_nt = cnt_t > 0 ? 1.f / nt : 0.f;
  f32x4_t mean, m2;
#pragma unroll
  for (int e = 0; e < 4; ++e) { mean[e] = k[e] + s1[e] * inv_nt; m2[e] = q[e] - s1[e] * s1[e] * inv_nt; }
  ps.red[0][tid] = mean; ps.red[1][tid] = m2; ps.red[2][tid] = mn;
  __syncthreads();
  const int np_all = nv * (nv - 1) / 2;
  float na = nt;
  if (sl == 0) {
    for (int t = 1; t < S; ++t) {
      const int g = w * S + t;
      const int c = g < np_all ? (np_all - g + ST - 1) / ST : 0;           // pairs of slice g: g, g + ST, ...
      if (c == 0) break;
      const float nb = (float)c, nab = na + nb;
      const f32x4_t mb = ps.red[0][t * G4 + c4], qb = ps.red[1][t * G4 + c4], n2 = ps.red[2][t * G4 + c4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = mb[e] - mean[e];
        mean[e] = na == 0.f ? mb[e] : mean[e] + d * (nb / nab);
        m2[e] = na == 0.f ? qb[e] : m2[e] + qb[e] + d * d * (na * nb / nab);
        mn[e] = fminf(mn[e], n2[e]);
      }
      na = nab;
    }
  }
  __syncthreads();
  ps.red[0][tid] = mx;
  __syncthreads();
  if (sl == 0) {
    for (int t = 1; t < S; ++t) {
      const f32x4_t x2 = ps.red[0][t * G4 + c4];
#pragma unroll
      for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], x2[e]);
    }
    float* o = part + 4 * c4;
    const f32x4_t cntv = {na, na, na, na};
    *reinterpret_cast<f32x4_t*>(o) = cntv;
    *reinterpret_cast<f32x4_t*>(o + DE) = mean;
    *reinterpret_cast<f32x4_t*>(o + 2 * DE) = m2;
    *reinterpret_cast<f32x4_t*>(o + 3 * DE) = mn;
    *reinterpret_cast<f32x4_t*>(o + 4 * DE) = mx;
  }
}

// parts [W][5][DE] of one graph -> z = [mean | min | max | biased variance] over all nv x nv entries; any number of threads
__device__ __forceinline__ void pool_etoy_finish(const float* parts, int W, int nv, int DE, float* z, int tid, int nthreads) {
  const float cnt = (float)nv * (float)nv, npf = (float)(nv * (nv - 1) / 2);
  for (int c = tid; c < DE; c += nthreads) {
    float na = 0.f, mean = 0.f, m2 = 0.f, mn = INFINITY, mx = -INFINITY;
    for (int w = 0; w < W; ++w) {
      const float* p = parts + (size_t)w * 5 * DE + c;
      const float nb = p[0];
      if (nb == 0.f) continue;
      const float nab = na + nb, d = p[DE] - mean;
      m2 = na == 0.f ? p[2 * DE] : m2 + p[2 * DE] + d * d * (na * nb / nab);
      mean = na == 0.f ? p[DE] : mean + d * (nb / nab);
      mn = fminf(mn, p[3 * DE]); mx = fmaxf(mx, p[4 * DE]);
      na = nab;
    }
    if (nv > 0) { mn = fminf(mn, 0.f); mx = fmaxf(mx, 0.f); }            // the diagonal
    const float m = cnt > 0.f ? 2.f * npf * mean / cnt : 0.f;
    const float dm = mean - m;
    const float qq = 2.f * (m2 + npf * dm * dm) + (float)nv * m * m;
    z[c] = m; z[DE + c] = mn; z[2 * DE + c] = mx; z[3 * DE + c] = cnt > 0.f ? qq / cnt : 0.f;
  }
}

// Position-aware block (pos_transformer_model.py Xtoy / Etoy inputs of the y update), one graph: Xtoy inputs of the UNMASKED
// newX (layers.py:15: the mean's numerator runs over every row of the padded batch, the divisor, minimum, maximum and
// variance over the valid ones) -> zx[4 dx]; Etoy inputs of Y and of dist1 from the row aggregates (every valid row holds
// the same number of keys) -> ze[8 de].  An empty graph gets zeros.  Any number of threads; no barrier inside.
__device__ __forceinline__ void pool_graph3d(const float* rowagg, const float* rowaggD, const float* newX, const uint8_t* m,
                                             int n, int nv, int DX, int DE, float* zx, float* ze, int tid, int nthreads) {
  if (nv == 0) {
    for (int t = tid; t < 4 * DX; t += nthreads) zx[t] = 0.f;
    for (int t = tid; t < 8 * DE; t += nthreads) ze[t] = 0.f;
    return;
  }
  // loads are unconditional and unrolled (8 rows in flight per thread): the loops are chains of L2 round trips otherwise
  for (int t = tid; t < DX; t += nthreads) {
    const float* col = newX + t;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) s += col[(long long)i * DX];
    const float mean = s / (float)nv;
    float q = 0.f, mn = INFINITY, mx = -INFINITY;
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
      const float v = col[(long long)i * DX], d = v - mean;
      const bool on = m[i] != 0;
      q += on ? d * d : 0.f;
      mn = fminf(mn, on ? v : INFINITY);
      mx = fmaxf(mx, on ? v : -INFINITY);
    }
    zx[t] = mean; zx[DX + t] = mn; zx[2 * DX + t] = mx; zx[3 * DX + t] = q / (float)nv;
  }
  for (int t = tid; t < 2 * DE; t += nthreads) {
    const int which = t / DE, ch = t - which * DE;
    float N = 0.f, mean = 0.f, m2 = 0.f, mn = INFINITY, mx = -INFINITY;
    const float k = (float)nv;
    const int ld = which ? 4 * DE : 8 * DE;
    const float* col = (which ? rowaggD : rowagg) + ch;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      const float* ag = col + (long long)i * ld;
      const float r_mean = ag[0], r_m2 = ag[DE], r_mn = ag[2 * DE], r_mx = ag[3 * DE];
      if (m[i]) {
        const float dl = r_mean - mean, nn = N + k;
        mean += dl * (k / nn);
        m2 += r_m2 + dl * dl * (N * k / nn);
        mn = fminf(mn, r_mn);
        mx = fmaxf(mx, r_mx);
        N = nn;
      }
    }
    float* zz = ze + which * 4 * DE + ch;
    zz[0] = mean; zz[DE] = mn; zz[2 * DE] = mx; zz[3 * DE] = m2 / N;
  }
}

}  // namespace scgg
