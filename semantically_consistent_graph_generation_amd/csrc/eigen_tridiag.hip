// Spectral features of the padded graph Laplacians for graphs of up to 256 nodes (vessel graphs: BASELINE configs[2]) on
// gfx950: one 1024-thread workgroup per graph, the whole decomposition in LDS.
//
// Reference: EigenFeatures.compute_features  src/ConStruct/diffusion/extra_features.py:87-112
//   (torch.linalg.eigh :96, eigenvalues_features :138-157, eigenvector_features :159-195).
// The reference takes the FULL decomposition from its library (LAPACK / cuSOLVER syevd; rocSOLVER here: 5.3 ms for four
// 256-node graphs) and then uses n eigenvalues and 1 + num_eigenvectors columns of the eigenvector matrix.  This kernel
// computes exactly that much:
//   1. Householder tridiagonalisation of the valid nv x nv block, packed lower triangle in LDS (nv (nv+1) / 2 floats =
//      129 KiB at nv = 256; masked and padding nodes are decoupled 1 x 1 blocks with eigenvalue 2n and never enter),
//      reflectors kept in place of the eliminated columns;
//   2. all nv eigenvalues of the tridiagonal matrix by Sturm-count multisection (4 threads per eigenvalue, the interval
//      shrinks 5x per round);
//   3. the selected eigenvectors (column 0 and columns ncc .. ncc + num_eigenvectors - 1) by inverse iteration on the
//      tridiagonal matrix (LU with partial pivoting, one wave per vector, three iterations from a fixed start vector), made
//      orthonormal inside clusters of close eigenvalues, then multiplied by the reflectors;
//   4. the reference's bookkeeping (:138-195) as in eigen_features.hip.
// The basis inside a degenerate eigenspace is solver-specific in the reference too (SURVEY.md "Hard parts"); here it is the
// one inverse iteration finds from its start vectors.  Eigenvalues agree with LAPACK to fp32 round-off of the matrix norm.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/scgg.h"
#include "kernels.h"
#include "mfma_chain.h"

namespace scgg {

constexpr int TRI_THREADS = 1024;
constexpr int TRI_MAXN = 256;
constexpr int TRI_MAXSEL = 8;                 // 1 + num_eigenvectors
constexpr int TRI_LDS_MAX = 160 * 1024;
constexpr int TRI_ROUNDS = 12;                // 5^12 > 2^27 subdivisions of the Gershgorin interval

struct TriArgs {
  const float* lap;        // [B,n,n]
  const uint8_t* mask;     // [B,n]
  float* eval_feat;        // row stride ld_ev, offset off_ev : [ncc | first num_ev nonzero eigenvalues / n_valid]
  float* evec_feat;        // node stride ld_vec, offset off_vec : [not-LCC indicator | first num_vec eigenvectors]
  float* w_out;            // [B,n] ascending eigenvalues (optional)
  float* vsel_out;         // [B,n,1+num_vec] the selected eigenvectors before the column mask (optional)
  int B, n, num_ev, num_vec, ld_ev, off_ev, ld_vec, off_vec, lu_batch;
};

__host__ __device__ inline int tri(int i) { return (i * (i + 1)) >> 1; }

struct TriLayout { int np, oV, oP, oD, oE, oT, oW, oRed, oAct, oSel, oZ, oLU, lu_words, total; };   // offsets in 4-byte words
__host__ __device__ inline TriLayout tri_layout(int n, int nsel, int batch) {
  TriLayout L;
  L.np = (n + 63) & ~63;
  int o = (tri(n) + 3) & ~3;
  L.oV = o; o += L.np;
  L.oP = o; o += L.np;
  L.oD = o; o += L.np;
  L.oE = o; o += L.np;
  L.oT = o; o += L.np;
  L.oW = o; o += L.np;
  L.oRed = o; o += 64;
  L.oAct = o; o += L.np / 4;                           // node ids as bytes
  L.oSel = o; o += 32;
  L.oZ = o; o += nsel * L.np;
  L.oLU = o;
  L.lu_words = 3 * L.np + 8;
  o += batch * L.lu_words;
  L.total = o;
  return L;
}

__device__ __forceinline__ float wave_sum(float v) { return other_half_sum(half_sum(v)); }

// ---- LU with partial pivoting of (T - lam I), T = tridiag(e, d, e); one lane, branch-free steps.  dl: multipliers, rd:
// reciprocal pivots, du: first super-diagonal of U, pv: bit i set = rows i and i+1 were swapped (sgttrf's scheme); the second
// super-diagonal of U is e[i+1] where rows were swapped and 0 elsewhere, so it is not stored.
__device__ __forceinline__ float tri_rcp(float x) {    // reciprocal to ~1 ulp: hardware estimate + one Newton step
  const float r = __builtin_amdgcn_rcpf(x);
  return r * (2.0f - x * r);
}
__device__ void tri_lu_factor(const float* d, const float* e, int nv, float lam, float eps_t, float* dl, float* rd, float* du,
                              unsigned* pv) {
  float dcur = d[0] - lam, ucur = nv > 1 ? e[0] : 0.f;
  unsigned word = 0;
  for (int i0 = 0; i0 + 1 < nv; i0 += 4) {
    float sb[4], dn[4], un[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                 // the loads of four steps at once: one LDS round trip per chunk
      const int i = min(i0 + u, nv - 2);
      sb[u] = e[i];
      dn[u] = d[i + 1] - lam;
      un[u] = i + 2 < nv ? e[i + 1] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u;
      if (i + 1 < nv) {
        const bool sw = fabsf(dcur) < fabsf(sb[u]);
        float piv = sw ? sb[u] : dcur;
        piv = fabsf(piv) < eps_t ? eps_t : piv;
        const float r = tri_rcp(piv);
        const float f = (sw ? dcur : sb[u]) * r;
        const float keep = sw ? dn[u] : ucur;      // the super-diagonal entry of the pivot row
        dcur = (sw ? ucur : dn[u]) - f * keep;
        ucur = sw ? -f * un[u] : un[u];
        word |= (sw ? 1u : 0u) << (i & 31);
        dl[i] = f; rd[i] = r; du[i] = keep;
        if ((i & 31) == 31) { pv[i >> 5] = word; word = 0; }
      }
    }
  }
  const float piv = fabsf(dcur) < eps_t ? eps_t : dcur;
  dl[nv - 1] = 0.f; rd[nv - 1] = tri_rcp(piv); du[nv - 1] = 0.f;
  pv[(nv - 1) >> 5] = word;
}

// z <- (T - lam I)^-1 z with the factors above; one lane
__device__ void tri_lu_solve(int nv, const float* e, const float* dl, const float* rd, const float* du, const unsigned* pv,
                             float* z) {
  float bc = z[0];
  for (int i0 = 0; i0 + 1 < nv; i0 += 4) {
    float bn[4], f[4];
    const unsigned word = pv[i0 >> 5];            // i0 is a multiple of 4: one word covers the chunk
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u, nv - 2);
      bn[u] = z[i + 1];
      f[u] = dl[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u;
      if (i + 1 < nv) {
        const bool sw = (word >> (i & 31)) & 1u;
        z[i] = sw ? bn[u] : bc;
        bc = sw ? bc - f[u] * bn[u] : bn[u] - f[u] * bc;
      }
    }
  }
  z[nv - 1] = bc;
  float x1 = 0.f, x2 = 0.f;
  for (int i0 = nv - 1; i0 >= 0; i0 -= 4) {
    float zb[4], r[4], u1[4], u2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = max(i0 - u, 0);
      zb[u] = z[i]; r[u] = rd[i]; u1[u] = du[i];
      u2[u] = (i + 2 < nv && ((pv[i >> 5] >> (i & 31)) & 1u)) ? e[i + 1] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 - u;
      if (i >= 0) {
        const float x = (zb[u] - u1[u] * x1 - u2[u] * x2) * r[u];
        z[i] = x;
        x2 = x1; x1 = x;
      }
    }
  }
}

#ifdef SCGG_TRI_STAMP
#define TRI_STAMP(k) if (t == 0) stamp[k] = wall_clock64()
#else
#define TRI_STAMP(k)
#endif

__global__ __launch_bounds__(TRI_THREADS) void eigen_features_tridiag_kernel(TriArgs a) {
  extern __shared__ float smem[];
#ifdef SCGG_TRI_STAMP
  long long stamp[10];
#endif
  const int b = blockIdx.x, n = a.n, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nsel = 1 + a.num_vec;
  const TriLayout lo = tri_layout(n, nsel, a.lu_batch);
  const int np = lo.np;
  float* Ap = smem;
  float* vv = smem + lo.oV;
  float* pp = smem + lo.oP;
  float* dd = smem + lo.oD;
  float* ee = smem + lo.oE;
  float* tt = smem + lo.oT;
  float* ww = smem + lo.oW;
  float* red = smem + lo.oRed;
  int* redi = reinterpret_cast<int*>(red);
  uint8_t* act = reinterpret_cast<uint8_t*>(smem + lo.oAct);
  int* seli = reinterpret_cast<int*>(smem + lo.oSel);        // [0,8): column, [8,16): -1 own vector / -2 zero / p copy of p
  float* self = smem + lo.oSel + 16;                         // [16,24): the shift used
  float* Z = smem + lo.oZ;
  const uint8_t* m = a.mask + (long long)b * n;
  const float* L = a.lap + (long long)b * n * n;

  TRI_STAMP(0);
  // ---- the valid nodes, compacted
  const bool valid = t < n && m[t] != 0;
  {
    const unsigned long long bal = __ballot(valid);
    if (t < 256 && lane == 0) redi[wv] = __popcll(bal);
    __syncthreads();
    if (valid) {
      int base = 0;
      for (int w = 0; w < wv; ++w) base += redi[w];
      act[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)t;
    }
  }
  const int nv = redi[0] + redi[1] + redi[2] + redi[3];
  const float nvf = (float)nv;
  __syncthreads();
  for (int idx = t; idx < nv * nv; idx += TRI_THREADS) {
    const int r = idx / nv, c = idx - r * nv;
    if (c <= r) Ap[tri(r) + c] = L[act[r] * n + act[c]];
  }
  __syncthreads();

  TRI_STAMP(1);
  // ---- 1. Householder tridiagonalisation.  Step k: v from column k, p = tau A v, A -= v w^T + w v^T with w = p - K v.
  // Three barriers per step: the update of step k also takes the norm of column k + 1 (red[32..48]).
  const int r4 = t >> 2, q = t & 3;
  auto column_norm = [&](int kc) {                     // red[32 + wave] = partial sums of x_i^2, i > kc + 1; red[48] = x_{kc+1}
    const int mc = nv - kc - 1;
    float x = 0.f;
    if (t < mc) x = Ap[tri(kc + 1 + t) + kc];
    const float sq = wave_sum(t > 0 ? x * x : 0.f);
    if (lane == 0) red[32 + wv] = sq;
    if (t == 0) red[48] = x;
  };
  if (nv > 2) column_norm(0);
  __syncthreads();
  for (int k = 0; k + 2 < nv; ++k) {
    const int mk = nv - k - 1;                         // order of the trailing block (rows / columns k+1 .. nv-1)
    float sig = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) sig += red[32 + w];
    const float x0 = red[48];
    if (sig == 0.f) {                                  // nothing to eliminate (uniform: sig comes from LDS)
      if (t == 0) { ee[k] = x0; tt[k] = 0.f; dd[k] = Ap[tri(k) + k]; }
      __syncthreads();                                 // everybody has read red[32..48]
      if (k + 3 < nv) column_norm(k + 1);
      __syncthreads();
      continue;
    }
    const float nrm2 = x0 * x0 + sig;
    const float nrm = sqrtf(nrm2);
    const float alpha = x0 >= 0.f ? -nrm : nrm;
    const float tau = 1.0f / (nrm2 - alpha * x0);      // 2 / v.v
    if (t < mk) {
      const float v = t == 0 ? x0 - alpha : Ap[tri(k + 1 + t) + k];
      vv[k + 1 + t] = v;
      if (t == 0) Ap[tri(k + 1) + k] = v;              // the reflector stays in the eliminated column
    }
    if (t == 0) { ee[k] = alpha; tt[k] = tau; dd[k] = Ap[tri(k) + k]; }
    __syncthreads();
    // p_i = tau sum_j A_ij v_j: thread (row r4, residue q) takes the columns j = k+1+jj, jj = q mod 4; row part from the
    // packed row, the rest from the packed column.  Four elements per trip so that eight LDS reads are in flight.
    const bool on = r4 < mk;
    const int i = k + 1 + r4, base = tri(i);
    float acc = 0.f;
    if (on) {
      const float* Ar = Ap + base + k + 1;
      const float* vr = vv + k + 1;
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
      int jj = q;
      for (; jj + 12 <= r4; jj += 16) {
        const float a0 = Ar[jj], a1 = Ar[jj + 4], a2 = Ar[jj + 8], a3 = Ar[jj + 12];
        const float v0 = vr[jj], v1 = vr[jj + 4], v2 = vr[jj + 8], v3 = vr[jj + 12];
        c0 += a0 * v0; c1 += a1 * v1; c2 += a2 * v2; c3 += a3 * v3;
      }
      for (; jj <= r4; jj += 4) c0 += Ar[jj] * vr[jj];
      jj = r4 + 1 + ((q - r4 - 1) & 3);
      int j = k + 1 + jj, ad = tri(j) + i;
      for (; jj + 12 < mk; jj += 16) {                 // tri(j + 4 s) - tri(j) = 4 s j + 2 s (4 s + 1)
        const float a0 = Ap[ad], a1 = Ap[ad + 4 * j + 10], a2 = Ap[ad + 8 * j + 36], a3 = Ap[ad + 12 * j + 78];
        const float v0 = vv[j], v1 = vv[j + 4], v2 = vv[j + 8], v3 = vv[j + 12];
        c0 += a0 * v0; c1 += a1 * v1; c2 += a2 * v2; c3 += a3 * v3;
        ad += 16 * j + 136;
        j += 16;
      }
      for (; jj < mk; jj += 4) {
        c1 += Ap[ad] * vv[j];
        ad += 4 * j + 10;
        j += 4;
      }
      acc = (c0 + c1) + (c2 + c3);
    }
    acc += SCGG_DPP(acc, 0xB1);
    acc += SCGG_DPP(acc, 0x4E);
    const float p = tau * acc;
    float kp = 0.f;
    if (on && q == 0) { pp[i] = p; kp = p * vv[i]; }
    kp = wave_sum(kp);
    if (lane == 0) red[16 + wv] = kp;
    __syncthreads();
    float K2 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) K2 += red[16 + w];
    K2 *= tau;                                         // 2 K = tau p.v
    float nx = 0.f;                                    // this thread's element of column k + 1 after the update
    if (on) {
      const float av = vv[i], c = pp[i] - K2 * av;     // A_ij -= v_i p_j + (p_i - 2 K v_i) v_j
      float* Ar = Ap + base + k + 1;
      const float* pr = pp + k + 1;
      const float* vr = vv + k + 1;
      int jj = q;
      if (q == 0) {                                    // jj = 0 is column k + 1: next step's column
        nx = Ar[0] - (av * pr[0] + c * vr[0]);
        Ar[0] = nx;
        jj = 4;
      }
      for (; jj + 12 <= r4; jj += 16) {
        const float a0 = Ar[jj], a1 = Ar[jj + 4], a2 = Ar[jj + 8], a3 = Ar[jj + 12];
        const float p0 = pr[jj], p1 = pr[jj + 4], p2 = pr[jj + 8], p3 = pr[jj + 12];
        const float v0 = vr[jj], v1 = vr[jj + 4], v2 = vr[jj + 8], v3 = vr[jj + 12];
        Ar[jj] = a0 - (av * p0 + c * v0);
        Ar[jj + 4] = a1 - (av * p1 + c * v1);
        Ar[jj + 8] = a2 - (av * p2 + c * v2);
        Ar[jj + 12] = a3 - (av * p3 + c * v3);
      }
      for (; jj <= r4; jj += 4) Ar[jj] -= av * pr[jj] + c * vr[jj];
    }
    // norm of column k + 1 below its sub-diagonal element (rows r4 >= 2), sub-diagonal element itself (r4 == 1)
    const float sq = wave_sum((on && q == 0 && r4 >= 2) ? nx * nx : 0.f);
    if (lane == 0) red[32 + wv] = sq;
    if (t == 4) red[48] = nx;                          // r4 == 1, q == 0
    __syncthreads();
  }
  if (t == 0 && nv > 0) {
    if (nv >= 2) { dd[nv - 2] = Ap[tri(nv - 2) + nv - 2]; ee[nv - 2] = Ap[tri(nv - 1) + nv - 2]; tt[nv - 2] = 0.f; }
    dd[nv - 1] = Ap[tri(nv - 1) + nv - 1]; ee[nv - 1] = 0.f; tt[nv - 1] = 0.f;
  }
  __syncthreads();

  TRI_STAMP(2);
  // ---- 2. eigenvalues: Gershgorin interval, then 4 Sturm counts per eigenvalue and round
  {
    float glo = INFINITY, ghi = -INFINITY;
    if (t < nv) {
      const float el = t > 0 ? fabsf(ee[t - 1]) : 0.f, er = t + 1 < nv ? fabsf(ee[t]) : 0.f;
      glo = dd[t] - el - er;
      ghi = dd[t] + el + er;
      pp[t] = ee[t] * ee[t];
    }
    if (t < 256) {
#pragma unroll
      for (int mm = 1; mm < 64; mm <<= 1) {
        glo = fminf(glo, __shfl_xor(glo, mm));
        ghi = fmaxf(ghi, __shfl_xor(ghi, mm));
      }
      if (lane == 0) { red[wv] = glo; red[8 + wv] = ghi; }
    }
  }
  __syncthreads();
  float gl = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  float gu = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
  if (nv == 0) { gl = 0.f; gu = 0.f; }
  const float nrmT = fmaxf(fmaxf(fabsf(gl), fabsf(gu)), 1.0f);
  const float eps_t = nrmT * 6e-8f;
  {
    const float pivmin = 1e-30f;
    const float margin = (gu - gl) * 1e-6f + 2.0f * eps_t;
    float blo = gl - margin, bhi = gu + margin;
    const int j = t >> 2, s = t & 3;
    if (j < nv) {
      for (int it = 0; it < TRI_ROUNDS; ++it) {
        const float h = (bhi - blo) * 0.2f;
        const float x1 = blo + h, x2 = blo + 2.f * h, x3 = blo + 3.f * h, x4 = blo + 4.f * h;
        const float x = s == 0 ? x1 : s == 1 ? x2 : s == 2 ? x3 : x4;
        float qv = dd[0] - x;
        int c = qv < 0.f ? 1 : 0;
#pragma unroll 4
        for (int i = 1; i < nv; ++i) {
          if (fabsf(qv) < pivmin) qv = -pivmin;
          qv = (dd[i] - x) - pp[i - 1] * __builtin_amdgcn_rcpf(qv);
          c += qv < 0.f ? 1 : 0;
        }
        const int f = c > j ? 1 : 0;                  // eigenvalue j lies below x
        const int f0 = __builtin_amdgcn_mov_dpp(f, 0x00, 0xF, 0xF, true), f1 = __builtin_amdgcn_mov_dpp(f, 0x55, 0xF, 0xF, true);
        const int f2 = __builtin_amdgcn_mov_dpp(f, 0xAA, 0xF, 0xF, true), f3 = __builtin_amdgcn_mov_dpp(f, 0xFF, 0xF, 0xF, true);
        if (f0) bhi = x1;
        else if (f1) { blo = x1; bhi = x2; }
        else if (f2) { blo = x2; bhi = x3; }
        else if (f3) { blo = x3; bhi = x4; }
        else blo = x4;
      }
      if (s == 0) ww[j] = 0.5f * (blo + bhi);
    }
  }
  __syncthreads();

  TRI_STAMP(3);
  // ---- 3. which columns: 0 and ncc .. ncc + num_vec - 1 (:159-195), their shifts
  {
    const bool isz = t < nv && ww[t] / nvf < 1e-5f;
    const unsigned long long bal = __ballot(isz);
    if (t < 256 && lane == 0) redi[32 + wv] = __popcll(bal);
  }
  __syncthreads();
  const int ncc = redi[32] + redi[33] + redi[34] + redi[35];
  if (t == 0) {
    for (int s = 0; s < nsel; ++s) {
      const int c = s == 0 ? 0 : ncc + s - 1;
      int src = -1;
      float lam = 0.f;
      if (c >= nv) src = -2;                           // a padding column (unit vector on a masked node) or beyond n: zero
      else {
        for (int p = 0; p < s; ++p) if (seli[p] == c && seli[8 + p] == -1) src = p;
        if (src == -1) {
          lam = ww[c];
          for (int p = 0; p < s; ++p)
            if (seli[8 + p] == -1 && fabsf(lam - self[p]) < 10.f * eps_t) lam = self[p] + 10.f * eps_t;
        }
      }
      seli[s] = c; seli[8 + s] = src; self[s] = lam;
    }
  }
  __syncthreads();

  TRI_STAMP(4);
  // ---- inverse iteration, one wave per vector, lu_batch vectors at a time (their LU workspace is what LDS has left)
  for (int s0 = 0; s0 < nsel; s0 += a.lu_batch) {
    const int s = s0 + wv;
    if (wv < a.lu_batch && s < nsel && seli[8 + s] == -1) {
      float* dl = smem + lo.oLU + wv * lo.lu_words;
      float* rd = dl + np;
      float* du = rd + np;
      unsigned* pv = reinterpret_cast<unsigned*>(du + np);
      float* z = Z + s * np;
      for (int i = lane; i < nv; i += 64) {
        unsigned hsh = (unsigned)(i + 1) * 2654435761u + (unsigned)(s + 1) * 40503u;
        hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        z[i] = (float)(hsh >> 8) * (2.0f / 16777216.0f) - 1.0f + 1e-3f;
      }
      if (lane == 0) tri_lu_factor(dd, ee, nv, self[s], eps_t, dl, rd, du, pv);
      __builtin_amdgcn_wave_barrier();
      for (int it = 0; it < 3; ++it) {
        if (lane == 0) tri_lu_solve(nv, ee, dl, rd, du, pv, z);
        __builtin_amdgcn_wave_barrier();
        float ss = 0.f;
        for (int i = lane; i < nv; i += 64) ss += z[i] * z[i];
        ss = wave_sum(ss);
        const float sc = ss > 0.f && ss < INFINITY ? 1.0f / sqrtf(ss) : 1.0f;
        for (int i = lane; i < nv; i += 64) z[i] *= sc;
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();

  TRI_STAMP(5);
  // ---- Gram-Schmidt inside clusters (wave 0, the vectors in registers: element lane + 64 e), then the reflectors
  if (wv == 0) {
    float zr[TRI_MAXSEL][4];
#pragma unroll
    for (int s = 0; s < TRI_MAXSEL; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = lane + 64 * e;
        zr[s][e] = (s < nsel && seli[8 + s] == -1 && i < nv) ? Z[s * np + i] : 0.f;
      }
#pragma unroll
    for (int s = 1; s < TRI_MAXSEL; ++s) {
      if (s < nsel && seli[8 + s] == -1) {
        bool touched = false;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int p = 0; p < s; ++p) {
            if (seli[8 + p] == -1 && fabsf(ww[seli[s]] - ww[seli[p]]) < 1e-3f * nrmT) {
              float dot = 0.f;
#pragma unroll
              for (int e = 0; e < 4; ++e) dot += zr[s][e] * zr[p][e];
              dot = wave_sum(dot);
#pragma unroll
              for (int e = 0; e < 4; ++e) zr[s][e] -= dot * zr[p][e];
              touched = true;
            }
          }
        if (touched) {
          float ss = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) ss += zr[s][e] * zr[s][e];
          ss = wave_sum(ss);
          const float sc = ss > 0.f ? 1.0f / sqrtf(ss) : 1.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = lane + 64 * e;
            if (i < nv) Z[s * np + i] = zr[s][e] * sc;
            zr[s][e] *= sc;
          }
        }
      }
    }
  }
  __syncthreads();
  TRI_STAMP(6);
  if (wv < nsel && seli[8 + wv] == -1) {               // x = H_0 H_1 ... H_{nv-3} z
    float xr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = lane + 64 * e;
      xr[e] = i < nv ? Z[wv * np + i] : 0.f;
    }
    for (int k = nv - 3; k >= 0; --k) {
      const float tau = tt[k];
      if (tau == 0.f) continue;
      float ve[4], dot = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = lane + 64 * e;
        ve[e] = (i > k && i < nv) ? Ap[tri(i) + k] : 0.f;
        dot += ve[e] * xr[e];
      }
      dot = wave_sum(dot) * tau;
#pragma unroll
      for (int e = 0; e < 4; ++e) xr[e] -= dot * ve[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = lane + 64 * e;
      if (i < nv) Z[wv * np + i] = xr[e];
    }
  }
  __syncthreads();

  TRI_STAMP(7);
  // ---- 4. outputs.  inv[node] = compact index (vv is free now)
  int* inv = reinterpret_cast<int*>(vv);
  if (t < nv) inv[act[t]] = t;
  const float padv = 2.0f * (float)n;                  // eigenvalue of the padding block (:88-94)
  if (a.w_out != nullptr && t < n) a.w_out[(long long)b * n + t] = t < nv ? ww[t] : padv;
  if (a.eval_feat != nullptr) {
    float* ef = a.eval_feat + (long long)b * a.ld_ev + a.off_ev;
    if (t == 0) ef[0] = (float)ncc;
    if (t < a.num_ev) {
      const int r = ncc + t;
      ef[1 + t] = r < n ? (r < nv ? ww[r] : padv) / nvf : 2.0f;            // extension with 2's (:149-151)
    }
  }
  // first column, rounded to 3 decimals (:169), mode over the valid nodes (ties -> smallest value)
  const bool col0 = nv > 0 && m[0] != 0 && seli[8] == -1;                  // eigvectors * mask(column index) (:98)
  float mine = 0.f;
  if (t < nv) {
    mine = col0 ? Z[t] : 0.f;
    mine = nearbyintf(mine * 1000.f) / 1000.f;
    pp[t] = mine;
  }
  __syncthreads();
  if (t < 256) {
    int cnt = -1;
    if (t < nv) {
      cnt = 0;
      for (int j = 0; j < nv; ++j) cnt += pp[j] == mine ? 1 : 0;
    }
    int bestc = cnt;
    float best = mine;
#pragma unroll
    for (int mm = 1; mm < 64; mm <<= 1) {
      const int oc = __shfl_xor(bestc, mm);
      const float ov = __shfl_xor(best, mm);
      if (oc > bestc || (oc == bestc && oc >= 0 && ov < best)) { bestc = oc; best = ov; }
    }
    if (lane == 0) { redi[40 + wv] = bestc; red[48 + wv] = best; }
  }
  __syncthreads();
  int bestc = redi[40];
  float best = red[48];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const int oc = redi[40 + w];
    const float ov = red[48 + w];
    if (oc > bestc || (oc == bestc && oc >= 0 && ov < best)) { bestc = oc; best = ov; }
  }
  if (t < n) {
    const int ai = valid ? inv[t] : -1;
    if (a.evec_feat != nullptr) {
      float* o = a.evec_feat + ((long long)b * n + t) * a.ld_vec + a.off_vec;
      o[0] = (valid && pp[ai] != best) ? 1.f : 0.f;
      for (int s = 1; s < nsel; ++s) {
        const int c = seli[s], src = seli[8 + s];
        float v = 0.f;
        if (valid && src != -2 && c < n && m[c] != 0) v = Z[(src >= 0 ? src : s) * np + ai];
        o[s] = v;
      }
    }
    if (a.vsel_out != nullptr) {
      float* o = a.vsel_out + ((long long)b * n + t) * nsel;
      for (int s = 0; s < nsel; ++s) {
        const int src = seli[8 + s];
        o[s] = (valid && src != -2) ? Z[(src >= 0 ? src : s) * np + ai] : 0.f;
      }
    }
  }
#ifdef SCGG_TRI_STAMP
  __syncthreads();
  TRI_STAMP(8);
  if (t == 0 && b == 0 && a.w_out != nullptr)
    for (int k = 0; k < 8; ++k) a.w_out[k] = (float)(stamp[k + 1] - stamp[k]);      // 100 MHz ticks
#endif
}

}  // namespace scgg

using namespace scgg;

extern "C" int scgg_eigen_features_tridiag(const float* laplacian, const uint8_t* node_mask, int32_t B, int32_t n,
                                           int32_t num_eigenvalues, int32_t num_eigenvectors, float* eval_feat, int32_t ld_ev,
                                           int32_t off_ev, float* evec_feat, int32_t ld_vec, int32_t off_vec, float* w_out,
                                           float* vsel_out, void* stream) {
  if (!laplacian || !node_mask) return fail("eigen_features_tridiag: null argument");
  if (B <= 0 || n <= 0) return fail("eigen_features_tridiag: bad shape");
  if (n > TRI_MAXN) return fail("eigen_features_tridiag: n=%d > %d (one workgroup holds the packed matrix in LDS)", n, TRI_MAXN);
  if (num_eigenvalues < 0 || num_eigenvalues > 256 || num_eigenvectors < 0 || num_eigenvectors + 1 > TRI_MAXSEL)
    return fail("eigen_features_tridiag: unsupported feature counts (num_eigenvectors <= %d)", TRI_MAXSEL - 1);
  if ((eval_feat && ld_ev < off_ev + 1 + num_eigenvalues) || (evec_feat && ld_vec < off_vec + 1 + num_eigenvectors))
    return fail("eigen_features_tridiag: bad destination stride");
  const int nsel = 1 + num_eigenvectors;
  const TriLayout l0 = tri_layout(n, nsel, 0);
  int batch = (TRI_LDS_MAX / 4 - l0.total) / l0.lu_words;
  if (batch < 1) return fail("eigen_features_tridiag: no LDS left for the LU workspace at n=%d", n);
  if (batch > nsel) batch = nsel;
  const TriLayout lay = tri_layout(n, nsel, batch);
  const size_t lds = (size_t)lay.total * 4;
  static LdsAttrOnce once;
  if (set_max_dynamic_lds(once, reinterpret_cast<const void*>(eigen_features_tridiag_kernel), TRI_LDS_MAX) != 0)
    return fail("eigen_features_tridiag: launch setup failed");
  TriArgs a{laplacian, node_mask, eval_feat, evec_feat, w_out, vsel_out, B, n, num_eigenvalues, num_eigenvectors,
            ld_ev, off_ev, ld_vec, off_vec, batch};
  hipLaunchKernelGGL(eigen_features_tridiag_kernel, dim3(B), dim3(TRI_THREADS), lds, static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("eigen_features_tridiag: %s", hipGetErrorString(e));
  return 0;
}
