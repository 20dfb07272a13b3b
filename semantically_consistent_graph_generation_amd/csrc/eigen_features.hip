// Batched symmetric eigendecomposition of the padded graph Laplacians (n <= 32) and the spectral features built
// from it, for gfx950.  One wave (one 64-thread workgroup) per graph; the matrix and the eigenvector matrix live
// in LDS (2 x 32 x 33 floats); parallel-order cyclic Jacobi (16 disjoint rotations per round, 31 rounds per sweep).
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

__global__ __launch_bounds__(64) void eigen_features_kernel(EigArgs a) {
  __shared__ float A[EN * ELD];
  __shared__ float V[EN * ELD];
  __shared__ float cs[16][2];
  __shared__ float w[EN];
  __shared__ int order[EN];      // order[r] = index of the r-th smallest eigenvalue
  __shared__ float v0[EN];
  __shared__ int flag;
  const int b = blockIdx.x, n = a.n, lane = threadIdx.x;
  const uint8_t* m = a.mask + b * n;
  const float* L = a.lap + (long long)b * n * n;

  float fro = 0.f;
  for (int idx = lane; idx < EN * EN; idx += 64) {
    const int i = idx >> 5, j = idx & 31;
    float v = 0.f;
    if (i < n && j < n) v = L[i * n + j];
    else if (i == j) v = 4.0f * n + i;          // beyond n: decoupled, sorts last
    A[i * ELD + j] = v;
    V[i * ELD + j] = i == j ? 1.f : 0.f;
    if (i < n && j < n) fro += v * v;
  }
#pragma unroll
  for (int mm = 1; mm < 64; mm <<= 1) fro += __shfl_xor(fro, mm);
  const float thresh = 2e-7f * sqrtf(fro);
  __syncthreads();

  for (int sweep = 0; sweep < MAX_SWEEPS; ++sweep) {
    if (lane == 0) flag = 0;
    __syncthreads();
    for (int r = 0; r < EN - 1; ++r) {
      // ---- rotation angles of the 16 disjoint pairs of this round
      if (lane < 16) {
        int p, q;
        if (lane == 0) { p = r; q = EN - 1; }
        else { p = (r + lane) % (EN - 1); q = (r - lane + (EN - 1)) % (EN - 1); }
        if (p > q) { const int t = p; p = q; q = t; }
        const float apq = A[p * ELD + q];
        float c = 1.f, s = 0.f;
        if (fabsf(apq) > thresh) {
          const float tau = (A[q * ELD + q] - A[p * ELD + p]) / (2.f * apq);
          const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
          c = 1.0f / sqrtf(1.f + t * t);
          s = t * c;
          flag = 1;
        }
        cs[lane][0] = c;
        cs[lane][1] = s;
      }
      __syncthreads();
      // ---- A <- A J and V <- V J : items (pair k, row i)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int item = it * 64 + lane, k = item >> 5, i = item & 31;
        int p, q;
        if (k == 0) { p = r; q = EN - 1; }
        else { p = (r + k) % (EN - 1); q = (r - k + (EN - 1)) % (EN - 1); }
        if (p > q) { const int t = p; p = q; q = t; }
        const float c = cs[k][0], s = cs[k][1];
        const float aip = A[i * ELD + p], aiq = A[i * ELD + q];
        A[i * ELD + p] = c * aip - s * aiq;
        A[i * ELD + q] = s * aip + c * aiq;
        const float vip = V[i * ELD + p], viq = V[i * ELD + q];
        V[i * ELD + p] = c * vip - s * viq;
        V[i * ELD + q] = s * vip + c * viq;
      }
      __syncthreads();
      // ---- A <- J^T A : items (pair k, column j)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int item = it * 64 + lane, k = item >> 5, j = item & 31;
        int p, q;
        if (k == 0) { p = r; q = EN - 1; }
        else { p = (r + k) % (EN - 1); q = (r - k + (EN - 1)) % (EN - 1); }
        if (p > q) { const int t = p; p = q; q = t; }
        const float c = cs[k][0], s = cs[k][1];
        const float apj = A[p * ELD + j], aqj = A[q * ELD + j];
        float np_ = c * apj - s * aqj, nq_ = s * apj + c * aqj;
        if (s != 0.f) {               // the rotated pair is annihilated exactly
          if (j == q) np_ = 0.f;
          if (j == p) nq_ = 0.f;
        }
        A[p * ELD + j] = np_;
        A[q * ELD + j] = nq_;
      }
      __syncthreads();
    }
    if (flag == 0) break;            // a whole sweep without a rotation: converged (uniform: flag is shared)
    __syncthreads();
  }

  // ---- sort ascending (stable rank), scale, features
  int nv = 0;
  for (int t = 0; t < n; ++t) nv += m[t] != 0;
  const float nvf = (float)nv;
  if (lane < EN) w[lane] = A[lane * ELD + lane];
  __syncthreads();
  if (lane < EN) {
    const float wi = w[lane];
    int rank = 0;
    for (int j = 0; j < EN; ++j) rank += (w[j] < wi || (w[j] == wi && j < lane)) ? 1 : 0;
    order[rank] = lane;
  }
  __syncthreads();
  int ncc = 0;
  for (int r = 0; r < n; ++r) ncc += (w[order[r]] / nvf < 1e-5f) ? 1 : 0;

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
    if (lane < EN) {
      float v = 0.f;
      if (lane < n && m[lane] && m[0]) v = V[lane * ELD + order[0]];
      v0[lane] = nearbyintf(v * 1000.f) / 1000.f;
    }
    __syncthreads();
    if (lane < n) {
      // mode over the valid nodes, ties -> smallest value
      float best = 0.f;
      int bestc = -1;
      for (int i = 0; i < n; ++i) {
        if (!m[i]) continue;
        int cnt = 0;
        for (int j = 0; j < n; ++j) cnt += (m[j] && v0[j] == v0[i]) ? 1 : 0;
        if (cnt > bestc || (cnt == bestc && v0[i] < best)) { bestc = cnt; best = v0[i]; }
      }
      float* o = a.evec_feat + ((long long)b * n + lane) * a.ld_vec + a.off_vec;
      const bool mi = m[lane] != 0;
      o[0] = (mi && v0[lane] != best) ? 1.f : 0.f;
      for (int k = 0; k < a.num_vec; ++k) {
        const int r = ncc + k;
        float v = 0.f;
        if (r < n && mi && m[r]) v = V[lane * ELD + order[r]];
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
  hipLaunchKernelGGL(eigen_features_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("eigen_features: %s", hipGetErrorString(e));
  return 0;
}
