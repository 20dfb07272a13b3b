// Batched symmetric eigendecomposition of the padded graph Laplacians (n <= 32) and the spectral features built
// from it, for gfx950.  One 256-thread workgroup per graph; the matrix and the eigenvector matrix live in LDS
// (2 x 32 x 33 floats); parallel-order cyclic Jacobi over the nv valid nodes only (nv/2 disjoint rotations per round,
// nv-1 rounds per sweep; masked and padding nodes are diagonal from the start).
//
// Reference: EigenFeatures.compute_features  src/ConStruct/diffusion/extra_features.py:87-112
//   (torch.linalg.eigh :96, eigenvalues_features :138-157, eigenvector_features :159-195).
// The reference's library call (LAPACK / cuSOLVER syevd) costs 50 ms per step for 512 32x32 matrices through
// rocSOLVER on this box (profiles/r01_b); this kernel replaces it.  Eigenvalues agree to fp32 round-off; the
// eigenvector basis inside a degenerate eigenspace is solver-specific in the reference as well (SURVEY.md "Hard
// parts"), here it is the Jacobi basis (rotations between disconnected components are never applied, so null-space
// vectors stay supported on single components).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/scgg.h"
#include "kernels.h"

namespace scgg {

constexpr int EN = 32;
constexpr int ELD = EN + 1;
constexpr int MAX_SWEEPS = 14;

struct EigArgs {
  const float* lap;        // [B,n,n]
  const uint8_t* mask;     // [B,n]
  float* eval_feat;        // row stride ld_ev, offset off_ev : [ncc | first num_ev nonzero eigenvalues / n_valid]
  float* evec_feat;        // node stride ld_vec, offset off_vec : [not-LCC indicator | first num_vec eigenvectors]
  float* w_out;            // [B,n] ascending eigenvalues (optional)
  float* v_out;            // [B,n,n] eigenvectors in columns (optional)
  int B, n, num_ev, num_vec, ld_ev, off_ev, ld_vec, off_vec;
};

constexpr int ETHREADS = 256;

// (p, q), p < q: the k-th pair of round r of the round-robin tournament over the nve active nodes (act[] maps to nodes)
__device__ __forceinline__ void jacobi_pair(int r, int k, int nve, const int* act, int& p, int& q) {
  int x, y;
  if (k == 0) { x = r; y = nve - 1; }
  else { x = (r + k) % (nve - 1); y = (r - k + (nve - 1)) % (nve - 1); }
  x = act[x]; y = act[y];
  p = x < y ? x : y;
  q = x < y ? y : x;
}

__global__ __launch_bounds__(ETHREADS) void eigen_features_kernel(EigArgs a) {
  // A and V ping-pong between two buffers: a round reads one and rewrites every entry of the active submatrix into the
  // other, so a round needs ONE barrier (each thread recomputes the two rotation angles it needs instead of waiting for a
  // separate angle phase); the round-robin schedule is tabulated once
  __shared__ float Abuf[2][EN * ELD];
  __shared__ float Vbuf[2][EN * ELD];
  __shared__ unsigned char ptab[(EN - 1) * 16][2];
  __shared__ float w[EN];
  __shared__ int order[EN];      // order[r] = index of the r-th smallest eigenvalue
  __shared__ int act[EN];        // the nodes that take part in rotations: the valid ones (+ one idle node if odd)
  __shared__ float v0[EN];
  __shared__ float red[ETHREADS / 64];
  __shared__ int flag, s_nve;
  const int b = blockIdx.x, n = a.n, lane = threadIdx.x;
  const uint8_t* m = a.mask + b * n;
  const float* L = a.lap + (long long)b * n * n;

  float* A = Abuf[0];
  float* V = Vbuf[0];
  float fro = 0.f;
  for (int idx = lane; idx < EN * EN; idx += ETHREADS) {
    const int i = idx >> 5, j = idx & 31;
    float v = 0.f;
    if (i < n && j < n) v = L[i * n + j];
    else if (i == j) v = 4.0f * n + i;          // beyond n: decoupled, sorts last
    A[i * ELD + j] = v;
    V[i * ELD + j] = i == j ? 1.f : 0.f;
    Abuf[1][i * ELD + j] = v;                   // entries outside the active submatrix never change
    Vbuf[1][i * ELD + j] = i == j ? 1.f : 0.f;
    if (i < n && j < n) fro += v * v;
  }
#pragma unroll
  for (int mm = 1; mm < 64; mm <<= 1) fro += __shfl_xor(fro, mm);
  if ((lane & 63) == 0) red[lane >> 6] = fro;
  if (lane < 64) {
    // masked and padding nodes are already diagonal (their rows and columns are zero off the diagonal): only the
    // valid nodes are rotated, which cuts a sweep from 31 x 16 rotations to (nv-1) x nv/2
    const bool valid = lane < n && m[lane] != 0;
    const unsigned bal = (unsigned)__ballot(valid);                   // lanes 32..63 contribute 0
    const int nv = __popc(bal);
    if (valid) act[__popc(bal & ((1u << lane) - 1u))] = lane;
    if (lane == 0) {
      if (nv & 1) act[nv] = __ffs((int)~bal) - 1;                      // an idle node: its rotations are identities
      s_nve = nv < 2 ? 0 : (nv + 1) & ~1;
    }
  }
  __syncthreads();
  fro = red[0] + red[1] + red[2] + red[3];
  const float thresh = 2e-7f * sqrtf(fro);
  const int nve = s_nve, npairs = nve >> 1;

  for (int idx = lane; idx < (nve - 1) * npairs; idx += ETHREADS) {
    const int r = idx / npairs, k = idx - r * npairs;
    int p, q;
    jacobi_pair(r, k, nve, act, p, q);
    ptab[r * 16 + k][0] = (unsigned char)p;
    ptab[r * 16 + k][1] = (unsigned char)q;
  }
  // thread = (pair k, pair l): the 2x2 block rows {p_k, q_k} x columns {p_l, q_l} of A, and rows act[l], act[l + 16] of V
  // for the columns {p_k, q_k}
  const int tk = lane >> 4, tl = lane & 15;
  const bool blk_on = tk < npairs && tl < npairs;
  int cur = 0;
  __syncthreads();                                   // ptab is complete
  for (int sweep = 0; sweep < MAX_SWEEPS && nve > 0; ++sweep) {
    for (int r = 0; r < nve - 1; ++r) {
      const float* Ac = Abuf[cur];
      const float* Vc = Vbuf[cur];
      float* An = Abuf[cur ^ 1];
      float* Vn = Vbuf[cur ^ 1];
      // the rotation of pair (p, q): angle from the current A (every thread that needs it computes the same numbers)
      auto rot = [&](int p, int q, float& c, float& s) {
        const float apq = Ac[p * ELD + q];
        c = 1.f; s = 0.f;
        if (fabsf(apq) > thresh) {
          // hardware reciprocal / square root (1 ulp) instead of the correctly rounded sequences: the rotation only has
          // to be orthogonal to fp32 round-off (c from t by one rsq), the iteration corrects everything else, and these
          // four operations are the latency chain of a round
          const float tau = (Ac[q * ELD + q] - Ac[p * ELD + p]) * __builtin_amdgcn_rcpf(2.f * apq);
          const float t = (tau >= 0.f ? 1.f : -1.f) * __builtin_amdgcn_rcpf(fabsf(tau) + __builtin_amdgcn_sqrtf(1.f + tau * tau));
          c = __builtin_amdgcn_rsqf(1.f + t * t);
          s = t * c;
        }
      };
      if (tk < npairs) {
        const int pk = ptab[r * 16 + tk][0], qk = ptab[r * 16 + tk][1];
        float ck, sk;
        rot(pk, qk, ck, sk);
        if (blk_on) {
          const int pl = ptab[r * 16 + tl][0], ql = ptab[r * 16 + tl][1];
          float cl = ck, sl = sk;
          if (tl != tk) rot(pl, ql, cl, sl);
          const float a00 = Ac[pk * ELD + pl], a01 = Ac[pk * ELD + ql], a10 = Ac[qk * ELD + pl], a11 = Ac[qk * ELD + ql];
          // columns first (A J), then rows (J^T .): same operation order as two separate passes
          const float b00 = cl * a00 - sl * a01, b01 = sl * a00 + cl * a01;
          const float b10 = cl * a10 - sl * a11, b11 = sl * a10 + cl * a11;
          float r00 = ck * b00 - sk * b10, r01 = ck * b01 - sk * b11;
          float r10 = sk * b00 + ck * b10, r11 = sk * b01 + ck * b11;
          if (tk == tl && sk != 0.f) { r01 = 0.f; r10 = 0.f; }        // the rotated pair is annihilated exactly
          An[pk * ELD + pl] = r00; An[pk * ELD + ql] = r01;
          An[qk * ELD + pl] = r10; An[qk * ELD + ql] = r11;
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int iv = tl + 16 * hh;
          if (iv < nve) {
            const int i = act[iv];
            const float vip = Vc[i * ELD + pk], viq = Vc[i * ELD + qk];
            Vn[i * ELD + pk] = ck * vip - sk * viq;
            Vn[i * ELD + qk] = sk * vip + ck * viq;
          }
        }
      }
      __syncthreads();
      cur ^= 1;
    }
    // Converged when none of the entries the rotations are decided on exceeds the threshold: the next sweep would rotate
    // nothing.  (Until round 2 that was found out by running that sweep: ~22 us of the kernel's 171.)
    if (lane == 0) flag = 0;
    __syncthreads();
    {
      const float* Ac = Abuf[cur];
      bool big = false;
      for (int idx = lane; idx < (nve - 1) * npairs; idx += ETHREADS) {
        const int r = idx / npairs, k = idx - r * npairs;
        big |= fabsf(Ac[ptab[r * 16 + k][0] * ELD + ptab[r * 16 + k][1]]) > thresh;
      }
      if (big) flag = 1;
    }
    __syncthreads();
    if (flag == 0) break;            // uniform: flag is shared
    __syncthreads();
  }
  A = Abuf[cur];
  V = Vbuf[cur];

  // ---- sort ascending (stable rank), scale, features.  Wave 0 only from here (the mask as a bit set, no global reads
  //      inside the loops).
  if (lane >= 64) return;
  const unsigned vbits = (unsigned)__ballot(lane < n && m[lane < n ? lane : 0] != 0);
  const float nvf = (float)__popc(vbits);
  if (lane < EN) w[lane] = A[lane * ELD + lane];
  __builtin_amdgcn_wave_barrier();
  if (lane < EN) {
    const float wi = w[lane];
    int rank = 0;
    for (int j = 0; j < EN; ++j) rank += (w[j] < wi || (w[j] == wi && j < lane)) ? 1 : 0;
    order[rank] = lane;
  }
  __builtin_amdgcn_wave_barrier();
  const int ncc = __popc((unsigned)__ballot(lane < n && w[order[lane < n ? lane : 0]] / nvf < 1e-5f));

  if (a.w_out != nullptr && lane < n) a.w_out[b * n + lane] = w[order[lane]];
  if (a.v_out != nullptr)
    for (int idx = lane; idx < n * n; idx += 64) {
      const int i = idx / n, r = idx - i * n;
      a.v_out[(long long)b * n * n + idx] = V[i * ELD + order[r]];
    }
  if (a.eval_feat != nullptr) {
    float* ef = a.eval_feat + (long long)b * a.ld_ev + a.off_ev;
    if (lane == 0) ef[0] = (float)ncc;
    if (lane < a.num_ev) {
      const int r = ncc + lane;
      ef[1 + lane] = r < n ? w[order[r]] / nvf : 2.0f;      // extension with 2's (:149-151)
    }
  }
  if (a.evec_feat != nullptr) {
    // eigvectors * mask(row) * mask(column index) (:98), first column rounded to 3 decimals (:169)
    const bool mi = lane < n && ((vbits >> lane) & 1u);
    float mine = 0.f;
    if (lane < EN) {
      if (mi && (vbits & 1u)) mine = V[lane * ELD + order[0]];
      mine = nearbyintf(mine * 1000.f) / 1000.f;
      v0[lane] = mine;
    }
    __builtin_amdgcn_wave_barrier();
    // mode over the valid nodes, ties -> smallest value: lane i counts the valid nodes that share its value, then the
    // (count, -value) maximum is taken across the wave
    int cnt = -1;
    if (mi) {
      cnt = 0;
      for (int j = 0; j < n; ++j) cnt += (((vbits >> j) & 1u) && v0[j] == mine) ? 1 : 0;
    }
    int bestc = cnt;
    float best = mi ? mine : 0.f;
#pragma unroll
    for (int mm = 1; mm < 64; mm <<= 1) {
      const int oc = __shfl_xor(bestc, mm);
      const float ov = __shfl_xor(best, mm);
      if (oc > bestc || (oc == bestc && oc >= 0 && ov < best)) { bestc = oc; best = ov; }
    }
    if (lane < n) {
      float* o = a.evec_feat + ((long long)b * n + lane) * a.ld_vec + a.off_vec;
      // One component (ncc == 1): the first eigenvector is constant on the valid nodes in exact arithmetic, every rounded entry
      // equals the mode and the flag is 0.  The computed vector carries (off-diagonal threshold) / (spectral gap) ~ 3e-5 of
      // error, enough to split the 3-decimal rounding of :169 where 1000 / sqrt(nv) sits next to a half (nv = 11, 17, 23, 27).
      o[0] = (mi && mine != best && ncc != 1) ? 1.f : 0.f;
      for (int k = 0; k < a.num_vec; ++k) {
        const int r = ncc + k;
        float v = 0.f;
        if (r < n && mi && ((vbits >> r) & 1u)) v = V[lane * ELD + order[r]];
        o[1 + k] = v;
      }
    }
  }
}

}  // namespace scgg

using namespace scgg;

extern "C" int scgg_eigen_features(const float* laplacian, const uint8_t* node_mask, int32_t B, int32_t n,
                                   int32_t num_eigenvalues, int32_t num_eigenvectors, float* eval_feat, int32_t ld_ev,
                                   int32_t off_ev, float* evec_feat, int32_t ld_vec, int32_t off_vec, float* w_out,
                                   float* v_out, void* stream) {
  if (!laplacian || !node_mask) return fail("eigen_features: null argument");
  if (B <= 0 || n <= 0) return fail("eigen_features: bad shape");
  if (n > EN) return fail("eigen_features: n=%d > 32 is not built yet (LDS Jacobi kernel only)", n);
  if (num_eigenvalues < 0 || num_eigenvalues > 63 || num_eigenvectors < 0 || num_eigenvectors > 32)
    return fail("eigen_features: unsupported feature counts");
  if ((eval_feat && ld_ev < off_ev + 1 + num_eigenvalues) || (evec_feat && ld_vec < off_vec + 1 + num_eigenvectors))
    return fail("eigen_features: bad destination stride");
  EigArgs a{laplacian, node_mask, eval_feat, evec_feat, w_out, v_out, B, n, num_eigenvalues, num_eigenvectors,
            ld_ev, off_ev, ld_vec, off_vec};
  hipLaunchKernelGGL(eigen_features_kernel, dim3(B), dim3(ETHREADS), 0, static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("eigen_features: %s", hipGetErrorString(e));
  return 0;
}
