// Structural extra features on the int8 edge state (gfx950).  One workgroup per graph; the adjacency matrix and its
// powers live in LDS (n <= 32: a 32x32 fp32 tile is 4 KiB), so the whole pass reads the labels once and writes
// the feature tensors once.
//
// Reference: AdjacencyFeatures  src/ConStruct/diffusion/extra_features.py:198-385
//   calculate_kpowers :256-268, k3..k6_cycle :270-327, path_features :329-333, local_neighbors :335-356,
//   get_degree_dist :358-371, get_node_dist :373-378, get_edge_dist :380-385, feature assembly :206-254;
//   Laplacian handed to eigh: EigenFeatures.compute_features :88-94, compute_laplacian :121-127.
// Walk counts are exact in fp32 below 2^24, as in the reference (which also multiplies in fp32).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/scgg.h"
#include "kernels.h"

namespace scgg {

constexpr int NT = 32;        // tile edge: this kernel handles n <= 32
constexpr int LDT = NT + 1;   // padded LDS row (column reads are conflict-free)

struct AdjArgs {
  const int8_t* labels;    // [B,n,n]
  const uint8_t* mask;     // [B,n]
  const int8_t* node_labels;   // [B,n] or null (all nodes class 0)
  float* x_out;            // [B,n,3]
  float* y_out;            // [B, 4 + num_degree+2 + Cx + C]
  float* e_out;            // pair stride e_ld, channel offset e_off, max_degree+1 channels
  float* onehot_out;       // optional: one-hot of the labels into the same destination rows, channels [0,C)
  float* lap_out;          // [B,n,n] or null
  int B, n, C, Cx, num_degree, max_degree, cycle_features, e_ld, e_off, x_ld, x_off, y_ld, y_off;
  int packed;              // the kernel owns whole pair rows [one-hot | max_degree+1 features]: rows are assembled in LDS
                           // and written as contiguous runs instead of 25 scalar stores per pair
};

__device__ __forceinline__ float clamp05(float v) { return fminf(fmaxf(v, 0.f), 5.f) * 0.2f; }

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads, result to all
  const int tid = threadIdx.x;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void adjacency_features_kernel(AdjArgs a) {
  __shared__ float A[NT * LDT];
  __shared__ float P[2][NT * LDT];       // ping-pong powers
  __shared__ float dg[6][NT];            // diagonals of A^1..A^6
  __shared__ float deg[NT];
  __shared__ float gsq[NT];              // Adamic-Adar weight of a common neighbour
  __shared__ float red[4];
  __shared__ float hist[64];
  extern __shared__ float stage[];       // packed mode: [256 pairs][e_ld]
  const int b = blockIdx.x, n = a.n, tid = threadIdx.x;
  const int8_t* lab = a.labels + (long long)b * n * n;
  const uint8_t* m = a.mask + b * n;
  const int K = a.max_degree;

  int nv = 0;
  for (int t = 0; t < n; ++t) nv += m[t] != 0;

  // adjacency: label > 0 on live pairs (labels of dead pairs are -1 in the state; mask applied anyway)
  for (int idx = tid; idx < NT * NT; idx += 256) {
    const int i = idx >> 5, j = idx & 31;
    float v = 0.f;
    if (i < n && j < n && i != j && m[i] && m[j]) v = lab[i * n + j] > 0 ? 1.f : 0.f;
    A[i * LDT + j] = v;
    P[0][i * LDT + j] = v;
  }
  if (tid < 64) hist[tid] = 0.f;
  __syncthreads();
  if (tid < NT) {
    float d = 0.f;
    for (int j = 0; j < NT; ++j) d += A[tid * LDT + j];
    deg[tid] = d;
    gsq[tid] = d > 1.f ? 1.0f / sqrtf(logf(d)) : 0.f;
    dg[0][tid] = 0.f;
  }
  __syncthreads();

  // ---- path features (reachability) for k = 1 and per-class edge histogram, one-hot copy, Laplacian
  const float nvf = (float)nv;
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx - i * n;
    float* e = a.e_out + ((long long)b * n * n + idx) * a.e_ld + a.e_off;
    if (!a.packed) e[0] = a.cycle_features ? (A[i * LDT + j] != 0.f ? 0.2f : 0.f) : 0.f;
    const bool live = i != j && m[i] && m[j];
    const int l = lab[idx];
    if (live && l >= 0 && l < 64) atomicAdd(&hist[l], 1.0f);
    if (a.onehot_out != nullptr && !a.packed) {
      float* oh = a.onehot_out + ((long long)b * n * n + idx) * a.e_ld;
      for (int c = 0; c < a.C; ++c) oh[c] = (live && c == l) ? 1.f : 0.f;
    }
    if (a.lap_out != nullptr) {
      float v;
      if (m[i] && m[j]) v = (i == j ? deg[i] : 0.f) - A[i * LDT + j];
      else v = (i == j && !m[i]) ? 2.0f * n : 0.f;
      a.lap_out[(long long)b * n * n + idx] = v;
    }
    // Adamic-Adar-like index: S_ij = sum_k A_ik A_jk g_k^2, zero diagonal, * log(n_b)/n_b   (:335-356)
    if (!a.packed) {
      float s = 0.f;
      if (i != j)
        for (int k = 0; k < NT; ++k) s += (A[i * LDT + k] * gsq[k]) * (A[j * LDT + k] * gsq[k]);
      e[K] = clamp05(s * (logf(nvf) / nvf));
    }
  }

  // ---- powers A^2..A^K; keep diagonals of A^2..A^6, sum(A^3) and sum(A o (A^2)^2) for the cycle formulas
  float sumA3 = 0.f, sumA_A2sq = 0.f;
  unsigned reach[4] = {0u, 0u, 0u, 0u};   // packed mode: bit k-1 = (A^k)_ij != 0 for this thread's four pairs
  int cur = 0;
  for (int k = 2; k <= K; ++k) {
    float acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = tid + 256 * t, i = idx >> 5, j = idx & 31;
      float s = 0.f;
#pragma unroll 8
      for (int q = 0; q < NT; ++q) s += P[cur][i * LDT + q] * A[q * LDT + j];
      acc[t] = s;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = tid + 256 * t, i = idx >> 5, j = idx & 31;
      P[cur ^ 1][i * LDT + j] = acc[t];
      if (i == j && k <= 6) dg[k - 1][i] = acc[t];
      if (k == 2) sumA_A2sq += A[i * LDT + j] * acc[t] * acc[t];
      if (k == 3) sumA3 += acc[t];
      if (a.packed) {
        reach[t] |= acc[t] != 0.f ? (1u << (k - 1)) : 0u;
      } else if (i < n && j < n) {
        float* e = a.e_out + ((long long)b * n * n + i * n + j) * a.e_ld + a.e_off;
        e[k - 1] = a.cycle_features ? (acc[t] != 0.f ? 0.2f : 0.f) : 0.f;
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  sumA3 = block_sum(sumA3, red);
  sumA_A2sq = block_sum(sumA_A2sq, red);

  // ---- packed mode: the pair rows [one-hot | reachability k = 1..K | Adamic-Adar], eight tile rows (<= 256 pairs) at a
  //      time through LDS, written as one contiguous run of 8 n e_ld floats
  if (a.packed) {
    const float aa_scale = logf(nvf) / nvf;
    const float on = a.cycle_features ? 0.2f : 0.f;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      const int idx = tid + 256 * t, i = idx >> 5, j = idx & 31;
      if (i < n && j < n) {
        float* row = stage + ((i & 7) * n + j) * a.e_ld;
        const bool live = i != j && m[i] && m[j];
        const int l = lab[i * n + j];
        for (int c = 0; c < a.C; ++c) row[c] = (live && c == l) ? 1.f : 0.f;
        float* e = row + a.e_off;
        e[0] = A[i * LDT + j] != 0.f ? on : 0.f;
        for (int k = 2; k <= K; ++k) e[k - 1] = ((reach[t] >> (k - 1)) & 1u) ? on : 0.f;
        float sAA = 0.f;
        if (i != j)
          for (int k = 0; k < NT; ++k) sAA += (A[i * LDT + k] * gsq[k]) * (A[j * LDT + k] * gsq[k]);
        e[K] = clamp05(sAA * aa_scale);
      }
      __syncthreads();
      const int i0 = 8 * t, rows = min(8, n - i0);
      if (rows > 0) {
        float* dst = a.e_out + ((long long)b * n * n + (long long)i0 * n) * a.e_ld;
        const int total = rows * n * a.e_ld;
        for (int q = tid; q < total; q += 256) dst[q] = stage[q];
      }
      __syncthreads();
    }
  }

  // ---- per-node cycle counts and the graph-level sums (one wave)
  const int ny = 4 + a.num_degree + 2 + a.Cx + a.C;
  float* y = a.y_out + (long long)b * a.y_ld + a.y_off;
  (void)ny;
  if (tid < 64) {
    const int i = tid;
    float c3 = 0.f, c4 = 0.f, c5 = 0.f, d2 = 0.f, d3 = 0.f, d4 = 0.f, d6 = 0.f;
    if (i < NT) {
      const float d = deg[i];
      d2 = dg[1][i]; d3 = dg[2][i]; d4 = dg[3][i]; d6 = dg[5][i];
      float Ad = 0.f, At = 0.f;
      for (int q = 0; q < NT; ++q) {
        Ad += A[i * LDT + q] * deg[q];
        At += A[i * LDT + q] * dg[2][q];
      }
      c3 = d3;
      c4 = d4 - d * (d - 1.f) - Ad;
      c5 = dg[4][i] - 2.f * d3 * d - At + d3;
      if (i < n) {
        float* x = a.x_out + ((long long)b * n + i) * a.x_ld + a.x_off;
        const float keep = (m[i] && a.cycle_features) ? 1.f : 0.f;
        x[0] = clamp05(c3 * 0.5f) * keep;
        x[1] = clamp05(c4 * 0.5f) * keep;
        x[2] = clamp05(c5 * 0.5f) * keep;
      }
    }
    float s3 = c3, s4 = c4, s5 = c5, t1 = d6, t2 = d3 * d3, t4 = d2 * d4, t5 = d4, t7 = d2 * d2 * d2, t9 = d2 * d2, t10 = d2;
#pragma unroll
    for (int mm = 1; mm < 64; mm <<= 1) {
      s3 += __shfl_xor(s3, mm); s4 += __shfl_xor(s4, mm); s5 += __shfl_xor(s5, mm);
      t1 += __shfl_xor(t1, mm); t2 += __shfl_xor(t2, mm); t4 += __shfl_xor(t4, mm); t5 += __shfl_xor(t5, mm);
      t7 += __shfl_xor(t7, mm); t9 += __shfl_xor(t9, mm); t10 += __shfl_xor(t10, mm);
    }
    if (tid == 0) {
      const float c6 = t1 - 3.f * t2 + 9.f * sumA_A2sq - 6.f * t4 + 6.f * t5 - 4.f * s3 + 4.f * t7 + 3.f * sumA3 -
                       12.f * t9 + 4.f * t10;
      const float on = a.cycle_features ? 1.f : 0.f;
      y[0] = clamp05(s3 / 6.f) * on;
      y[1] = clamp05(s4 / 8.f) * on;
      y[2] = clamp05(s5 / 10.f) * on;
      y[3] = clamp05(c6 / 12.f) * on;
    }
  }
  // ---- histograms: degree (:358-371), node classes (:373-378), edge classes (:380-385)
  if (tid < a.num_degree + 2) {
    float cnt = 0.f;
    for (int i = 0; i < n; ++i) {
      if (!m[i]) continue;
      int d = (int)deg[i];
      if (d > a.num_degree) d = a.num_degree + 1;
      cnt += d == tid ? 1.f : 0.f;
    }
    y[4 + tid] = nv > 0 ? cnt / nvf : 0.f;
  }
  if (tid >= 64 && tid < 64 + a.Cx) {
    const int c = tid - 64;
    float cnt = 0.f;
    for (int i = 0; i < n; ++i)
      if (m[i]) cnt += (a.node_labels ? a.node_labels[b * n + i] : 0) == c ? 1.f : 0.f;
    y[4 + a.num_degree + 2 + c] = nv > 0 ? cnt / nvf : 0.f;
  }
  __syncthreads();
  if (tid >= 128 && tid < 128 + a.C) {
    const int c = tid - 128;
    const float tot = nvf * (nvf - 1.f);
    y[4 + a.num_degree + 2 + a.Cx + c] = tot > 0.f ? hist[c] / tot : 0.f;
  }
}

// The padded combinatorial Laplacian alone (EigenFeatures.compute_features :88-94, compute_laplacian :121-127), the same
// numbers as adjacency_features_kernel's lap_out: handed out separately so that the eigensolver can start while the
// rest of the structural features are still being computed.  One workgroup per graph, n <= 32.
__global__ __launch_bounds__(256) void laplacian_kernel(const int8_t* labels, const uint8_t* mask, int n, float* lap_out) {
  __shared__ float A[NT * LDT];
  __shared__ float deg[NT];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int8_t* lab = labels + (long long)b * n * n;
  const uint8_t* m = mask + b * n;
  for (int idx = tid; idx < NT * NT; idx += 256) {
    const int i = idx >> 5, j = idx & 31;
    float v = 0.f;
    if (i < n && j < n && i != j && m[i] && m[j]) v = lab[i * n + j] > 0 ? 1.f : 0.f;
    A[i * LDT + j] = v;
  }
  __syncthreads();
  if (tid < NT) {
    float d = 0.f;
    for (int j = 0; j < NT; ++j) d += A[tid * LDT + j];
    deg[tid] = d;
  }
  __syncthreads();
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx - i * n;
    float v;
    if (m[i] && m[j]) v = (i == j ? deg[i] : 0.f) - A[i * LDT + j];
    else v = (i == j && !m[i]) ? 2.0f * n : 0.f;
    lap_out[(long long)b * n * n + idx] = v;
  }
}

}  // namespace scgg

using namespace scgg;

extern "C" int scgg_laplacian(const int8_t* labels, const uint8_t* node_mask, int32_t B, int32_t n, float* lap_out,
                              void* stream) {
  if (!labels || !node_mask || !lap_out) return fail("laplacian: null argument");
  if (B <= 0 || n <= 0) return fail("laplacian: bad shape");
  if (n > NT) return launch_laplacian_rows(labels, node_mask, B, n, lap_out, static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(laplacian_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), labels, node_mask, n, lap_out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("laplacian: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int scgg_adjacency_features(const int8_t* labels, const int8_t* node_labels, const uint8_t* node_mask,
                                       int32_t B, int32_t n, int32_t C, int32_t Cx, int32_t num_degree,
                                       int32_t max_degree, int32_t cycle_features, float* x_out, int32_t x_ld,
                                       int32_t x_off, float* y_out, int32_t y_ld, int32_t y_off, float* e_out,
                                       int32_t e_ld, int32_t e_off, float* onehot_out, float* lap_out, void* stream) {
  if (!labels || !node_mask || !x_out || !y_out || !e_out) return fail("adjacency_features: null argument");
  if (B <= 0 || n <= 0) return fail("adjacency_features: bad shape");
  if (n > NT) return fail("adjacency_features: n=%d > 32 needs scgg_adjacency_features_tiled (workspace argument)", n);
  if (max_degree < 6 || max_degree > 32 || num_degree + 2 > 64 || C > 64 || Cx > 64 || C < 1 || Cx < 1)
    return fail("adjacency_features: unsupported max_degree=%d num_degree=%d C=%d Cx=%d", max_degree, num_degree, C, Cx);
  if (e_ld < e_off + max_degree + 1 || (onehot_out && e_off < C) || x_ld < x_off + 3 ||
      y_ld < y_off + 4 + num_degree + 2 + Cx + C)
    return fail("adjacency_features: bad destination stride");
  const int packed = (onehot_out == e_out && e_off == C && e_ld == C + max_degree + 1 && e_ld <= 48) ? 1 : 0;
  AdjArgs a{labels, node_mask, node_labels, x_out, y_out, e_out, onehot_out, lap_out,
            B, n, C, Cx, num_degree, max_degree, cycle_features, e_ld, e_off, x_ld, x_off, y_ld, y_off, packed};
  hipLaunchKernelGGL(adjacency_features_kernel, dim3(B), dim3(256), packed ? (size_t)256 * e_ld * sizeof(float) : 0,
                     static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("adjacency_features: %s", hipGetErrorString(e));
  return 0;
}
