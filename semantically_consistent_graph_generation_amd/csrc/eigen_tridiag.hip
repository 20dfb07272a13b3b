// Spectral features of the padded graph Laplacians for graphs of up to 256 nodes (vessel graphs: BASELINE configs[2]) on
// gfx950: one 1024-thread workgroup per graph, the whole decomposition in LDS.
//
// Reference: EigenFeatures.compute_features  src/ConStruct/diffusion/extra_features.py:87-112
//   (torch.linalg.eigh :96, eigenvalues_features :138-157, eigenvector_features :159-195).
// The reference takes the FULL decomposition from its library (LAPACK / cuSOLVER syevd; rocSOLVER here: 5.3 ms for four
// 256-node graphs) and then uses n eigenvalues and 1 + num_eigenvectors columns of the eigenvector matrix.  This kernel
// computes exactly that much:
//   1. Householder tridiagonalisation of the valid nv x nv block, lower triangle in LDS with every row padded to whole
//      quads (130 KiB at nv = 256; masked and padding nodes are decoupled 1 x 1 blocks with eigenvalue 2n and never enter),
//      reflectors kept in place of the eliminated columns;
//   2. the eigenvalues the features read (the component count is one Sturm count at the reference's threshold, then
//      eigenvalue 0 and the next max(num_eigenvalues, num_eigenvectors) ones, one wave each: 64 Sturm counts per round
//      cut the interval into 65 parts); the whole spectrum, 4 threads per eigenvalue, only when the caller asks for it;
//   3. the selected eigenvectors (column 0 and columns ncc .. ncc + num_eigenvectors - 1) by inverse iteration on the
//      tridiagonal matrix (LU with partial pivoting, one wave per vector, two iterations from a fixed start vector), made
//      orthonormal inside clusters of close eigenvalues, then multiplied by the reflectors;
//   4. the reference's bookkeeping (:138-195) as in eigen_features.hip.
// The basis inside a degenerate eigenspace is solver-specific in the reference too (SURVEY.md "Hard parts"); here it is the
// one inverse iteration finds from its start vectors.  Eigenvalues agree with LAPACK to fp32 round-off of the matrix norm.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/scgg.h"
#include "kernels.h"
#include "c16.h"

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

// Row i of the lower triangle starts at rs(i) and holds 4 (i / 4 + 1) floats: columns 0 .. i, then zeros up to the end of its
// last quad, so that every quad of four columns is one aligned ds_read_b128.
__host__ __device__ inline int rs(int i) { const int a = i >> 2; return 4 * (a + 1) * (2 * a + (i & 3)); }

struct TriLayout { int np, oV, oP, oD, oE, oW, oRed, oAct, oSel, oZ, oLU, lu_words, total; };   // offsets in 4-byte words
__host__ __device__ inline TriLayout tri_layout(int n, int nsel, int batch) {
  TriLayout L;
  L.np = (n + 63) & ~63;
  int o = rs((n + 3) & ~3);                            // whole row blocks of four
  L.oV = o; o += L.np;
  L.oP = o; o += L.np;
  L.oD = o; o += L.np;
  L.oE = o; o += L.np;
  L.oW = o; o += L.np;
  L.oRed = o; o += 64;
  L.oAct = o; o += L.np / 4;                           // node ids as bytes
  L.oSel = o; o += 48;
  L.oZ = o; o += nsel * L.np;
  L.oLU = o;
  L.lu_words = 3 * L.np + 8;
  o += batch * L.lu_words;
  if (o < L.oZ + 16 * L.np) o = L.oZ + 16 * L.np;      // the same region holds 16 partial p vectors during the tridiagonalisation
  L.total = o;
  return L;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float wave_sum(float v) { return other_half_sum(half_sum(v)); }

// ---- LU with partial pivoting of (T - lam I), T = tridiag(e, d, e); one lane, branch-free steps.  dl: multipliers, rd:
// reciprocal pivots, du: first super-diagonal of U, pv: bit i set = rows i and i+1 were swapped (sgttrf's scheme); the second
// super-diagonal of U is e[i+1] where rows were swapped and 0 elsewhere, so it is not stored.
__device__ __forceinline__ float tri_rcp(float x) {    // reciprocal to ~1 ulp: hardware estimate + one Newton step
  const float r = __builtin_amdgcn_rcpf(x);
  return r * (2.0f - x * r);
}
// Both routines work in chunks of TRI_CH steps whose operands were loaded one chunk earlier: the recurrences are serial, an
// LDS round trip per step would dominate them.
constexpr int TRI_CH = 8;
__device__ void tri_lu_factor(const float* d, const float* e, int nv, float lam, float eps_t, float* dl, float* rd, float* du,
                              unsigned* pv) {
  float dcur = d[0] - lam, ucur = nv > 1 ? e[0] : 0.f;
  unsigned word = 0;
  float ce[TRI_CH + 1], cd[TRI_CH], ne[TRI_CH + 1], nd[TRI_CH];
  auto load = [&](int i0, float (&le)[TRI_CH + 1], float (&ld)[TRI_CH]) {
#pragma unroll
    for (int u = 0; u <= TRI_CH; ++u) le[u] = i0 + u < nv - 1 ? e[i0 + u] : 0.f;      // e[nv-1] does not exist: 0
#pragma unroll
    for (int u = 0; u < TRI_CH; ++u) ld[u] = d[min(i0 + u + 1, nv - 1)];
  };
  load(0, ce, cd);
  for (int i0 = 0; i0 + 1 < nv; i0 += TRI_CH) {
    if (i0 + TRI_CH + 1 < nv) load(i0 + TRI_CH, ne, nd);
#pragma unroll
    for (int u = 0; u < TRI_CH; ++u) {
      const int i = i0 + u;
      if (i + 1 < nv) {
        const float sb = ce[u], dn = cd[u] - lam, un = ce[u + 1];
        const bool sw = fabsf(dcur) < fabsf(sb);
        float piv = sw ? sb : dcur;
        piv = fabsf(piv) < eps_t ? eps_t : piv;
        const float r = tri_rcp(piv);
        const float f = (sw ? dcur : sb) * r;
        const float keep = sw ? dn : ucur;         // the super-diagonal entry of the pivot row
        dcur = (sw ? ucur : dn) - f * keep;
        ucur = sw ? -f * un : un;
        word |= (sw ? 1u : 0u) << (i & 31);
        dl[i] = f; rd[i] = r; du[i] = keep;
        if ((i & 31) == 31) { pv[i >> 5] = word; word = 0; }
      }
    }
#pragma unroll
    for (int u = 0; u <= TRI_CH; ++u) ce[u] = ne[u];
#pragma unroll
    for (int u = 0; u < TRI_CH; ++u) cd[u] = nd[u];
  }
  const float piv = fabsf(dcur) < eps_t ? eps_t : dcur;
  dl[nv - 1] = 0.f; rd[nv - 1] = tri_rcp(piv); du[nv - 1] = 0.f;
  pv[(nv - 1) >> 5] = word;
}

// z <- (T - lam I)^-1 z with the factors above; one lane
__device__ void tri_lu_solve(int nv, const float* e, const float* dl, const float* rd, const float* du, const unsigned* pv,
                             float* z) {
  {
    float bc = z[0];
    float cb[TRI_CH], cf[TRI_CH], nb[TRI_CH], nf[TRI_CH];
    auto load = [&](int i0, float (&lb)[TRI_CH], float (&lf)[TRI_CH]) {
#pragma unroll
      for (int u = 0; u < TRI_CH; ++u) {
        const int i = min(i0 + u, nv - 2);
        lb[u] = z[i + 1];                           // not yet overwritten: the sweep writes z[i] behind itself
        lf[u] = dl[i];
      }
    };
    if (nv > 1) load(0, cb, cf);
    for (int i0 = 0; i0 + 1 < nv; i0 += TRI_CH) {
      const unsigned word = pv[i0 >> 5];            // i0 is a multiple of 8: one word covers the chunk
      if (i0 + TRI_CH + 1 < nv) load(i0 + TRI_CH, nb, nf);
#pragma unroll
      for (int u = 0; u < TRI_CH; ++u) {
        const int i = i0 + u;
        if (i + 1 < nv) {
          const bool sw = (word >> (i & 31)) & 1u;
          z[i] = sw ? cb[u] : bc;
          bc = sw ? bc - cf[u] * cb[u] : cb[u] - cf[u] * bc;
        }
      }
#pragma unroll
      for (int u = 0; u < TRI_CH; ++u) { cb[u] = nb[u]; cf[u] = nf[u]; }
    }
    z[nv - 1] = bc;
  }
  float x1 = 0.f, x2 = 0.f;
  float cz[TRI_CH], cr[TRI_CH], c1[TRI_CH], c2[TRI_CH], nz[TRI_CH], nr[TRI_CH], n1[TRI_CH], n2[TRI_CH];
  auto load = [&](int i0, float (&lz)[TRI_CH], float (&lr)[TRI_CH], float (&l1)[TRI_CH], float (&l2)[TRI_CH]) {
#pragma unroll
    for (int u = 0; u < TRI_CH; ++u) {
      const int i = max(i0 - u, 0);
      lz[u] = z[i]; lr[u] = rd[i]; l1[u] = du[i];
      l2[u] = (i + 2 < nv && ((pv[i >> 5] >> (i & 31)) & 1u)) ? e[i + 1] : 0.f;
    }
  };
  load(nv - 1, cz, cr, c1, c2);
  for (int i0 = nv - 1; i0 >= 0; i0 -= TRI_CH) {
    if (i0 - TRI_CH >= 0) load(i0 - TRI_CH, nz, nr, n1, n2);
#pragma unroll
    for (int u = 0; u < TRI_CH; ++u) {
      const int i = i0 - u;
      if (i >= 0) {
        const float x = (cz[u] - c1[u] * x1 - c2[u] * x2) * cr[u];
        z[i] = x;
        x2 = x1; x1 = x;
      }
    }
#pragma unroll
    for (int u = 0; u < TRI_CH; ++u) { cz[u] = nz[u]; cr[u] = nr[u]; c1[u] = n1[u]; c2[u] = n2[u]; }
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
  long long st2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TRI_ST2(j) if (t == 0 && k == 8) st2[j] = wall_clock64()
#else
#define TRI_ST2(j)
#endif
  const int b = blockIdx.x, n = a.n, t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nsel = 1 + a.num_vec;
  const TriLayout lo = tri_layout(n, nsel, a.lu_batch);
  const int np = lo.np;
  float* Ap = smem;
  float* vv = smem + lo.oV;
  float* pp = smem + lo.oP;
  float* dd = smem + lo.oD;
  float* ee = smem + lo.oE;
  float* ww = smem + lo.oW;
  float* red = smem + lo.oRed;
  int* redi = reinterpret_cast<int*>(red);
  uint8_t* act = reinterpret_cast<uint8_t*>(smem + lo.oAct);
  int* seli = reinterpret_cast<int*>(smem + lo.oSel);        // [0,8): column, [8,16): -1 own vector / -2 zero / p copy of p
  float* self = smem + lo.oSel + 16;                         // [16,24): the shift used
  float* Z = smem + lo.oZ;
  float* zq = smem + lo.oSel + 28;                           // four zeros: what a lane beyond the diagonal reads
  if (t < 4) zq[t] = 0.f;
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
  float tiny2;                                         // a column whose squared norm is below this is taken as zero
  {
    const int nq = (nv + 3) & ~3;                      // rows nv .. nq-1 complete the last row block: zeros
    float amax = 0.f;
    for (int idx = t; idx < nq * nq; idx += TRI_THREADS) {
      const int r = idx / nq, c = idx - r * nq;
      if (c <= (r | 3)) {
        const float val = (c <= r && r < nv) ? L[act[r] * n + act[c]] : 0.f;         // a row's last quad is padded with 0
        Ap[rs(r) + c] = val;
        amax = fmaxf(amax, fabsf(val));
      }
    }
#pragma unroll
    for (int mm = 1; mm < 64; mm <<= 1) amax = fmaxf(amax, __shfl_xor(amax, mm));
    if (lane == 0) red[wv] = amax;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) amax = fmaxf(amax, red[w]);
    // Rounding leaves entries of ~1e-8 |A| where exact arithmetic has zeros (isolated nodes, small components), and their
    // products leave ~1e-16: a reflector built on such a column has tau = 2 / v.v beyond the fp32 range.  Below 1e-10 |A| a
    // column is dropped, which perturbs the matrix by less than fp32 round-off.
    tiny2 = 1e-20f * amax * amax;
  }
  __syncthreads();

  TRI_STAMP(1);
  // ---- 1. Householder tridiagonalisation.  Step k: v from column k, p = tau A v, A -= v w^T + w v^T with w = p - K v.
  // A wave takes row blocks of four rows, lane c the quad of columns 4c .. 4c+3: every stored element is read once per pass
  // (one ds_read_b128 per row and lane) and serves its row sum (reduced over the lanes) and its column sum (kept per lane,
  // the 16 waves' partial vectors are added up afterwards).  v and p sit in registers, one quad per lane.
  float* scr = smem + lo.oZ;                           // [16][np] partial column sums; free until the inverse iteration
  const int l4 = lane * 4;
  // vv holds the raw column of the step at its top: vv[i] = A[i][k] for i > k, 0 elsewhere (written by the update of the
  // step before, which has those elements in hand); red[32..47] the partial sums of squares below the sub-diagonal element.
  auto column_norm = [&](int kc) {                     // the same from the stored matrix (first step, and after a dropped column)
    float x = 0.f;
    if (t < np) {
      if (t > kc && t < nv) x = Ap[rs(t) + kc];
      vv[t] = x;
    }
    const float sq = wave_sum(t > kc + 1 ? x * x : 0.f);
    if (lane == 0) red[32 + wv] = sq;
  };
  if (nv > 2) column_norm(0);
  __syncthreads();
  for (int k = 0; k + 2 < nv; ++k) {
    TRI_ST2(0);
    float sig = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) sig += red[32 + w];
    const float x0 = vv[k + 1];
    if (sig <= tiny2) {                                // nothing to eliminate (uniform: sig comes from LDS)
      if (t == 0) { ee[k] = x0; dd[k] = Ap[rs(k) + k]; Ap[rs(k) + k] = 0.f; }
      __syncthreads();                                 // everybody has read red[32..47] and vv
      if (k + 3 < nv) column_norm(k + 1);
      __syncthreads();
      continue;
    }
    const float nrm2 = x0 * x0 + sig;
    const float nrm = sqrtf(nrm2);
    const float alpha = x0 >= 0.f ? -nrm : nrm;
    const float tau = 1.0f / (nrm2 - alpha * x0);      // 2 / v.v
    const float v0 = x0 - alpha;                       // v = column k with its first element replaced
    TRI_ST2(1);
    f32x4 vreg = l4 < np ? *reinterpret_cast<const f32x4*>(vv + l4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (l4 + e == k + 1) vreg[e] = v0;
    const int A0 = (k + 1) >> 2, A1 = (nv - 1) >> 2;
    {
      // this wave's row blocks: A0 + wv + 16 s.  Rows <= k hold reflectors and rows >= nv zeros: v is 0 there, so they add
      // nothing to the column sums, and their row sums are never read.  Lanes beyond the diagonal quad read the zero quad.
      f32x2 cpl = {0.f, 0.f}, cph = {0.f, 0.f};
      float rp[16];
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const int A = A0 + wv + 16 * sb;               // uniform (wv comes from readfirstlane below)
        if (A <= A1) {
          const float* rowp = Ap + 8 * A * (A + 1) + l4;   // rs(4A) = 8 A (A + 1), a row of this block is 4 (A + 1) floats
          f32x4 qd[4];
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) qd[bb] = *reinterpret_cast<const f32x4*>(lane <= A ? rowp + bb * 4 * (A + 1) : zq);
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {
            const float vi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vreg[bb]), A));
            const f32x2 ql = {qd[bb][0], qd[bb][1]}, qh = {qd[bb][2], qd[bb][3]};
            const f32x2 r2 = ql * f32x2{vreg[0], vreg[1]} + qh * f32x2{vreg[2], vreg[3]};
            rp[4 * sb + bb] = r2[0] + r2[1];
            const f32x2 vi2 = {vi, vi};
            cpl += ql * vi2;
            cph += qh * vi2;
          }
        } else {
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) rp[4 * sb + bb] = 0.f;
        }
      }
      // sixteen wave sums: reduce-scatter inside the 16-lane rows (lane p keeps sum p), then across the four rows
      float r = row_reduce_scatter16(rp, lane & 15);
      r = other_half_sum(swap_rows16_sum(r));
      {
        const int A = A0 + wv + 16 * (lane >> 2);
        if (lane < 16 && A <= A1) pp[4 * A + (lane & 3)] = r;
      }
      if (l4 < np) *reinterpret_cast<f32x4*>(scr + wv * np + l4) = f32x4{cpl[0], cpl[1], cph[0], cph[1]};
    }
    TRI_ST2(3);
    __syncthreads();
    TRI_ST2(4);
    if (t < 256) {                                     // p_j = tau (row sum + column sums - the diagonal term counted twice)
      float kp = 0.f;
      if (t < np) {
        float p = 0.f;
        if (t > k && t < nv) {
          float sacc = pp[t];
#pragma unroll
          for (int w = 0; w < 16; ++w) sacc += scr[w * np + t];
          const float vt = t == k + 1 ? v0 : vv[t];
          sacc -= Ap[rs(t) + t] * vt;
          p = tau * sacc;
          kp = p * vt;
        }
        pp[t] = p;
      }
      // the reflector stays in the eliminated column (only its head differs from what is stored), its tau in the diagonal
      // slot of row k.  Nobody else touches these two words in this phase; the passes on either side read the quads around them.
      if (t == 0) { ee[k] = alpha; dd[k] = Ap[rs(k) + k]; Ap[rs(k) + k] = tau; Ap[rs(k + 1) + k] = v0; }
      kp = wave_sum(kp);
      if (lane == 0) red[16 + wv] = kp;
    }
    TRI_ST2(5);
    __syncthreads();
    TRI_ST2(6);
    {
      const float K2 = tau * ((red[16] + red[17]) + (red[18] + red[19]));            // 2 K = tau p.v
      const f32x4 preg = l4 < np ? *reinterpret_cast<const f32x4*>(pp + l4) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 creg;                                      // A_ij -= v_i p_j + (p_i - 2 K v_i) v_j
#pragma unroll
      for (int e = 0; e < 4; ++e) creg[e] = preg[e] - K2 * vreg[e];
      const f32x2 pl = {preg[0], preg[1]}, ph = {preg[2], preg[3]}, vl = {vreg[0], vreg[1]}, vh = {vreg[2], vreg[3]};
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const int A = A0 + wv + 16 * sb;
        if (A <= A1) {
          float* rowp = Ap + 8 * A * (A + 1) + l4;
          f32x4 qd[4];
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) qd[bb] = *reinterpret_cast<const f32x4*>(lane <= A ? rowp + bb * 4 * (A + 1) : zq);
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {             // rows <= k and >= nv: v_i = c_i = 0, they are written back unchanged
            const float vi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vreg[bb]), A));
            const float ci = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(creg[bb]), A));
            const f32x2 vi2 = {vi, vi}, ci2 = {ci, ci};
            f32x2 ql = {qd[bb][0], qd[bb][1]}, qh = {qd[bb][2], qd[bb][3]};
            ql -= vi2 * pl + ci2 * vl;
            qh -= vi2 * ph + ci2 * vh;
            qd[bb] = f32x4{ql[0], ql[1], qh[0], qh[1]};
#pragma unroll
            for (int e = bb + 1; e < 4; ++e)
              if (lane == A) qd[bb][e] = 0.f;          // beyond the diagonal: padding
          }
          if (lane <= A) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) *reinterpret_cast<f32x4*>(rowp + bb * 4 * (A + 1)) = qd[bb];
          }
        }
      }
      // column k + 1 of this wave's rows (read back behind the wave's own writes): its norm below the sub-diagonal element
      float sq = 0.f;
      {
        const int i = 4 * (A0 + wv + 16 * (lane >> 2)) + (lane & 3);
        if (lane < 16 && i > k && i < nv) {
          const float nx = i >= k + 2 ? Ap[rs(i) + k + 1] : 0.f;             // row k + 1 leaves the active block
          vv[i] = nx;
          if (i > k + 2) sq = nx * nx;
        }
      }
      sq = wave_sum(sq);
      if (lane == 0) red[32 + wv] = sq;
    }
    TRI_ST2(7);
    __syncthreads();
  }
  if (t == 0 && nv > 0) {
    if (nv >= 2) { dd[nv - 2] = Ap[rs(nv - 2) + nv - 2]; ee[nv - 2] = Ap[rs(nv - 1) + nv - 2]; }
    dd[nv - 1] = Ap[rs(nv - 1) + nv - 1]; ee[nv - 1] = 0.f;
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
  const float pivmin = 1e-30f;
  const float margin = (gu - gl) * 1e-6f + 2.0f * eps_t;
  auto sturm = [&](float x) {                          // number of eigenvalues of T below x
    float qv = dd[0] - x;
    int c = qv < 0.f ? 1 : 0;
#pragma unroll 4
    for (int i = 1; i < nv; ++i) {
      if (fabsf(qv) < pivmin) qv = -pivmin;
      qv = (dd[i] - x) - pp[i - 1] * __builtin_amdgcn_rcpf(qv);
      c += qv < 0.f ? 1 : 0;
    }
    return c;
  };
  // The features read eigenvalue 0 and eigenvalues ncc .. ncc + max(num_ev, num_vec) - 1 (:138-195): ncc is ONE Sturm count at
  // the reference's threshold (ev = w / n_valid < 1e-5), and each of those eigenvalues is found by its own wave, the 64
  // lanes cutting the interval into 65 parts per round (5 rounds: 65^5 > 2^30 subdivisions of the Gershgorin interval).
  float* wsel = smem + lo.oSel + 32;                   // [16] the eigenvalues the features read
  const int nneed = 1 + max(a.num_ev, a.num_vec);
  if (wv < nneed && nv > 0) {
    const int ncc_w = sturm(1e-5f * nvf);
    const int idx = wv == 0 ? 0 : ncc_w + wv - 1;
    float lam = 0.f;
    if (idx < nv) {
      float blo = gl - margin, bhi = gu + margin;
      for (int it = 0; it < 5; ++it) {
        const float h = (bhi - blo) * (1.0f / 65.0f);
        const int c = sturm(blo + h * (float)(lane + 1));
        const unsigned long long bal = __ballot(c > idx);          // lanes whose point lies above eigenvalue idx
        const int pfirst = bal == 0ull ? 64 : __ffsll((long long)bal) - 1;
        const float nlo = blo + h * (float)pfirst;
        if (pfirst < 64) bhi = blo + h * (float)(pfirst + 1);
        blo = nlo;
      }
      lam = 0.5f * (blo + bhi);
    }
    if (lane == 0) {
      wsel[wv] = lam;
      if (wv == 0) redi[36] = ncc_w;
    }
  }
  if (nv == 0 && t == 0) redi[36] = 0;
  if (a.w_out != nullptr) {                            // the whole spectrum only on request: 4 Sturm counts per eigenvalue and round
    float blo = gl - margin, bhi = gu + margin;
    const int j = t >> 2, s = t & 3;
    if (j < nv) {
      for (int it = 0; it < TRI_ROUNDS; ++it) {
        const float h = (bhi - blo) * 0.2f;
        const float x1 = blo + h, x2 = blo + 2.f * h, x3 = blo + 3.f * h, x4 = blo + 4.f * h;
        const int c = sturm(s == 0 ? x1 : s == 1 ? x2 : s == 2 ? x3 : x4);
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
  const int ncc = redi[36];
  if (t == 0) {
    for (int s = 0; s < nsel; ++s) {
      const int c = s == 0 ? 0 : ncc + s - 1;
      int src = -1;
      float lam = 0.f;
      if (c >= nv) src = -2;                           // a padding column (unit vector on a masked node) or beyond n: zero
      else {
        for (int p = 0; p < s; ++p) if (seli[p] == c && seli[8 + p] == -1) src = p;
        if (src == -1) {
          lam = wsel[s];
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
      for (int it = 0; it < 2; ++it) {
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
            if (seli[8 + p] == -1 && fabsf(wsel[s] - wsel[p]) < 1e-3f * nrmT) {
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
      const float tau = Ap[rs(k) + k];
      if (tau == 0.f) continue;
      float ve[4], dot = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = lane + 64 * e;
        ve[e] = (i > k && i < nv) ? Ap[rs(i) + k] : 0.f;
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
      ef[1 + t] = r < n ? (r < nv ? wsel[1 + t] : padv) / nvf : 2.0f;      // extension with 2's (:149-151)
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
      o[0] = (valid && pp[ai] != best && ncc != 1) ? 1.f : 0.f;   // one component: constant first vector, see eigen_features.hip
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
    for (int k = 0; k < 8; ++k) {
      a.w_out[k] = (float)(stamp[k + 1] - stamp[k]);      // 100 MHz ticks
      a.w_out[8 + k] = (float)(st2[k] - st2[0]);
    }
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
  if (num_eigenvalues < 0 || num_eigenvalues > 15 || num_eigenvectors < 0 || num_eigenvectors + 1 > TRI_MAXSEL)
    return fail("eigen_features_tridiag: unsupported feature counts (num_eigenvalues <= 15, num_eigenvectors <= %d)", TRI_MAXSEL - 1);
  if ((eval_feat && ld_ev < off_ev + 1 + num_eigenvalues) || (evec_feat && ld_vec < off_vec + 1 + num_eigenvectors))
    return fail("eigen_features_tridiag: bad destination stride");
  const int nsel = 1 + num_eigenvectors;
  const TriLayout l0 = tri_layout(n, nsel, 0);
  int batch = (TRI_LDS_MAX / 4 - l0.oLU) / l0.lu_words;
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
