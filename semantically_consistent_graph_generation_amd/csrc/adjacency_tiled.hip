// Structural extra features for graphs of any size (gfx950): the same numbers as extra_features.hip's one-workgroup
// kernel (n <= 32), with the adjacency powers A^2..A^K and the Adamic-Adar product on the matrix pipe
// (v_mfma_f32_32x32x2_f32: the walk counts are integers, exact in fp32 below 2^24 as in the reference's fp32 matmul) and
// everything else as streaming passes over [B, np, np] planes in a caller workspace (np = n rounded up to 64).
//
// Reference: AdjacencyFeatures  src/ConStruct/diffusion/extra_features.py:198-385
//   calculate_kpowers :256-268 (the K-1 batched products), k3..k6_cycle :270-327, path_features :329-333,
//   local_neighbors :335-356 (w w^T is the (K)th product), histograms :358-385, feature assembly :206-254;
//   Laplacian handed to eigh: EigenFeatures.compute_features :88-94, compute_laplacian :121-127.
//
// Passes (all on the caller's stream, no host synchronisation):
//   1 adj_build      labels -> A (fp32 0/1, zero padded), reach bit 0, degrees, edge-class counts, n_valid
//   2 adj_scale      Wc = A diag(c), Wr = diag(c) A^T with c_j = 1/sqrt(log deg_j) (0 when deg_j <= 1); Laplacian
//   3 adj_gemm x K-1 P_k = P_{k-1} A; epilogue: diag(P_k) for k <= 6, reach bit k-1 = (P_k != 0)
//     adj_gemm       S = Wc Wr
//   4 adj_rows       one wave per row: per-node 3/4/5-cycle counts, the row's terms of the graph-level sums (fp64: exact
//                    wherever the reference's fp32 sums are), histogram votes;  adj_graph: one workgroup per graph sums them
//   5 adj_pair       pair rows [one-hot | reach bits 0..K-1 | Adamic-Adar], contiguous stores
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/scgg.h"
#include "kernels.h"

namespace scgg {

constexpr int TL = 64;                 // GEMM tile edge (one workgroup = 4 waves = 2x2 tiles of 32x32)
constexpr int TKS = 32;                // k-slab
typedef float f32x16_t __attribute__((ext_vector_type(16)));

struct TiledPlanes {                   // carved out of the workspace by plan()
  float *A, *A2, *A3, *Pa, *Pb, *Wc, *Wr, *S;
  uint32_t* reach;                     // [B,np,np] bit k-1 = (A^k)_ij != 0
  float* diag;                         // [6][B][np] diagonals of A^1..A^6
  float* deg;                          // [B][np]
  int* hist;                           // [3][B][64] edge-class counts over live ordered pairs | degree histogram | node-class histogram
  int* nvalid;                         // [B]
  double* rowpart;                     // [B][np][12] per-row terms of the graph-level sums
  size_t bytes;
};

static TiledPlanes plan(void* ws, int B, int np) {
  TiledPlanes p;
  char* c = static_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = c ? c + off : nullptr; off += (bytes + 255) & ~(size_t)255; return r; };
  const size_t plane = (size_t)B * np * np * sizeof(float);
  p.A = (float*)take(plane);  p.A2 = (float*)take(plane); p.A3 = (float*)take(plane);
  p.Pa = (float*)take(plane); p.Pb = (float*)take(plane);
  p.Wc = (float*)take(plane); p.Wr = (float*)take(plane); p.S = (float*)take(plane);
  p.reach = (uint32_t*)take(plane);
  p.diag = (float*)take((size_t)6 * B * np * sizeof(float));
  p.deg = (float*)take((size_t)B * np * sizeof(float));
  p.hist = (int*)take((size_t)3 * B * 64 * sizeof(int));
  p.nvalid = (int*)take((size_t)B * sizeof(int));
  p.rowpart = (double*)take((size_t)B * np * 12 * sizeof(double));
  p.bytes = off;
  return p;
}

__device__ __forceinline__ float clamp05t(float v) { return fminf(fmaxf(v, 0.f), 5.f) * 0.2f; }

// ---- 1: one workgroup per (graph, 16 rows); a wave owns 4 rows
__global__ __launch_bounds__(256) void adj_build_kernel(const int8_t* labels, const uint8_t* mask, int n, int np, float* A,
                                                        uint32_t* reach, float* deg, float* diag1, int* hist, int* nvalid) {
  __shared__ int lh[64];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int8_t* lab = labels + (long long)b * n * n;
  const uint8_t* m = mask + (long long)b * n;
  if (tid < 64) lh[tid] = 0;
  __syncthreads();
  for (int r = 0; r < 4; ++r) {
    const int i = blockIdx.x * 16 + wave * 4 + r;
    if (i >= np) break;
    const bool mi = i < n && m[i];
    float d = 0.f;
    for (int j = lane; j < np; j += 64) {
      float v = 0.f;
      if (mi && j < n && j != i && m[j]) {
        const int l = lab[(long long)i * n + j];
        v = l > 0 ? 1.f : 0.f;
        if (l >= 0 && l < 64) atomicAdd(&lh[l], 1);
      }
      const long long o = ((long long)b * np + i) * np + j;
      A[o] = v;
      reach[o] = v != 0.f ? 1u : 0u;
      d += v;
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) d += __shfl_xor(d, s);
    if (lane == 0) {
      deg[(long long)b * np + i] = d;
      diag1[(long long)b * np + i] = 0.f;
    }
  }
  __syncthreads();
  if (tid < 64 && lh[tid] != 0) atomicAdd(&hist[b * 64 + tid], lh[tid]);
  if (blockIdx.x == 0 && wave == 0) {
    int c = 0;
    for (int j = lane; j < n; j += 64) c += m[j] != 0;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) c += __shfl_xor(c, s);
    if (lane == 0) nvalid[b] = c;
  }
}

// ---- 2
__global__ __launch_bounds__(256) void adj_scale_kernel(const float* A, const float* deg, const uint8_t* mask, int n, int np,
                                                        float* Wc, float* Wr, float* lap_out) {
  const int b = blockIdx.y;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)np * np) return;
  const int i = (int)(idx / np), j = (int)(idx - (long long)i * np);
  const float* Ab = A + (long long)b * np * np;
  const float* dg = deg + (long long)b * np;
  const float a = Ab[idx], at = Ab[(long long)j * np + i];
  const float dj = dg[j], di = dg[i];
  const float cj = dj > 1.f ? 1.0f / sqrtf(logf(dj)) : 0.f;
  const float ci = di > 1.f ? 1.0f / sqrtf(logf(di)) : 0.f;
  Wc[(long long)b * np * np + idx] = a * cj;
  Wr[(long long)b * np * np + idx] = at * ci;
  if (lap_out != nullptr && i < n && j < n) {
    const uint8_t* m = mask + (long long)b * n;
    float v;
    if (m[i] && m[j]) v = (i == j ? di : 0.f) - a;
    else v = (i == j && !m[i]) ? 2.0f * n : 0.f;
    lap_out[((long long)b * n + i) * n + j] = v;
  }
}

// ---- 3: C[b] = X[b] Y[b] on [np,np] planes; grid (np/64, np/64, B)
__global__ __launch_bounds__(256) void adj_gemm_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                       float* __restrict__ C, int np, float* diag, uint32_t* reach,
                                                       uint32_t bit) {
  __shared__ float Xs[TL][TKS + 1];
  __shared__ float Ys[TKS][TL];
  const int b = blockIdx.z, ti = blockIdx.y * TL, tj = blockIdx.x * TL, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hi = lane >> 5;
  const float* Xb = X + (long long)b * np * np;
  const float* Yb = Y + (long long)b * np * np;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int xr = tid >> 2, xc = (tid & 3) * 8;     // X slab [64][32]: 8 floats per thread
  const int yr = tid >> 3, yc = (tid & 7) * 8;     // Y slab [32][64]
  for (int k0 = 0; k0 < np; k0 += TKS) {
    const float4 x0 = *reinterpret_cast<const float4*>(Xb + (long long)(ti + xr) * np + k0 + xc);
    const float4 x1 = *reinterpret_cast<const float4*>(Xb + (long long)(ti + xr) * np + k0 + xc + 4);
    const float4 y0 = *reinterpret_cast<const float4*>(Yb + (long long)(k0 + yr) * np + tj + yc);
    const float4 y1 = *reinterpret_cast<const float4*>(Yb + (long long)(k0 + yr) * np + tj + yc + 4);
    __syncthreads();
    float* xs = &Xs[xr][xc];
    xs[0] = x0.x; xs[1] = x0.y; xs[2] = x0.z; xs[3] = x0.w; xs[4] = x1.x; xs[5] = x1.y; xs[6] = x1.z; xs[7] = x1.w;
    *reinterpret_cast<float4*>(&Ys[yr][yc]) = y0;
    *reinterpret_cast<float4*>(&Ys[yr][yc + 4]) = y1;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TKS; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[wr * 32 + l31][kk + hi], Ys[kk + hi][wc * 32 + l31], acc, 0, 0, 0);
  }
  float* Cb = C + (long long)b * np * np;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ti + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, col = tj + wc * 32 + l31;
    const float v = acc[r];
    Cb[(long long)row * np + col] = v;
    if (reach != nullptr && v != 0.f) reach[((long long)b * np + row) * np + col] |= bit;
    if (diag != nullptr && row == col) diag[(long long)b * np + row] = v;
  }
}

// ---- 4a: one wave per (row, graph): per-node cycle counts, the row's terms of the graph-level sums, histogram votes
struct TiledNodeArgs {
  const float *A, *A2, *A3, *diag, *deg;
  int* hist;                           // [3][B][64]
  const int* nvalid;
  const uint8_t* mask;
  const int8_t* node_labels;
  double* rowpart;
  float *x_out, *y_out;
  int B, n, np, C, Cx, num_degree, cycle_features, x_ld, x_off, y_ld, y_off;
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s);
  return v;
}

__global__ __launch_bounds__(256) void adj_rows_kernel(TiledNodeArgs a) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6), n = a.n, np = a.np;
  if (i >= n) return;
  const long long pl = (long long)b * np * np;
  const float* A = a.A + pl + (long long)i * np;
  const float* A2 = a.A2 + pl + (long long)i * np;
  const float* A3 = a.A3 + pl + (long long)i * np;
  const float* deg = a.deg + (long long)b * np;
  const long long dstride = (long long)a.B * np;
  const float* dg = a.diag + (long long)b * np;
  float Ad = 0.f, At = 0.f;
  double sA3 = 0, sAA = 0;
  for (int j = lane; j < n; j += 64) {
    const float v = A[j], a2 = A2[j];
    Ad += v * deg[j];
    At += v * dg[2 * dstride + j];
    sA3 += A3[j];
    sAA += (double)v * a2 * a2;
  }
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) { Ad += __shfl_xor(Ad, s); At += __shfl_xor(At, s); }
  sA3 = wave_sum_d(sA3);
  sAA = wave_sum_d(sAA);
  if (lane != 0) return;
  const uint8_t* m = a.mask + (long long)b * n;
  const float d = deg[i], d2 = dg[dstride + i], d3 = dg[2 * dstride + i], d4 = dg[3 * dstride + i], d5 = dg[4 * dstride + i],
              d6 = dg[5 * dstride + i];
  const float c3 = d3, c4 = d4 - d * (d - 1.f) - Ad, c5 = d5 - 2.f * d3 * d - At + d3;
  float* x = a.x_out + ((long long)b * n + i) * a.x_ld + a.x_off;
  const float keep = (m[i] && a.cycle_features) ? 1.f : 0.f;
  x[0] = clamp05t(c3 * 0.5f) * keep;
  x[1] = clamp05t(c4 * 0.5f) * keep;
  x[2] = clamp05t(c5 * 0.5f) * keep;
  double* r = a.rowpart + ((long long)b * np + i) * 12;
  r[0] = c3; r[1] = c4; r[2] = c5; r[3] = d6; r[4] = (double)d3 * d3; r[5] = (double)d2 * d4; r[6] = d4;
  r[7] = (double)d2 * d2 * d2; r[8] = (double)d2 * d2; r[9] = d2; r[10] = sA3; r[11] = sAA;
  if (m[i]) {
    int di = (int)d;
    if (di > a.num_degree) di = a.num_degree + 1;
    atomicAdd(&a.hist[((long long)a.B + b) * 64 + di], 1);
    const int xc = a.node_labels ? a.node_labels[(long long)b * n + i] : 0;
    if (xc >= 0 && xc < 64) atomicAdd(&a.hist[((long long)2 * a.B + b) * 64 + xc], 1);
  }
}

// ---- 4b: one workgroup per graph: the row terms summed in a fixed order, graph-level features and histograms
__global__ __launch_bounds__(256) void adj_graph_kernel(TiledNodeArgs a) {
  __shared__ double red[4][12];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = a.n, np = a.np;
  double q[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) q[k] = 0;
  for (int i = tid; i < n; i += 256) {
    const double* r = a.rowpart + ((long long)b * np + i) * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) q[k] += r[k];
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) q[k] = wave_sum_d(q[k]);
  if (lane == 0)
    for (int k = 0; k < 12; ++k) red[wave][k] = q[k];
  __syncthreads();
  float* y = a.y_out + (long long)b * a.y_ld + a.y_off;
  const int nv = a.nvalid[b];
  const float nvf = (float)nv;
  if (tid == 0) {
    for (int k = 0; k < 12; ++k) q[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    const double c6 = q[3] - 3.0 * q[4] + 9.0 * q[11] - 6.0 * q[5] + 6.0 * q[6] - 4.0 * q[0] + 4.0 * q[7] + 3.0 * q[10] -
                      12.0 * q[8] + 4.0 * q[9];
    const float on = a.cycle_features ? 1.f : 0.f;
    y[0] = clamp05t((float)q[0] / 6.f) * on;
    y[1] = clamp05t((float)q[1] / 8.f) * on;
    y[2] = clamp05t((float)q[2] / 10.f) * on;
    y[3] = clamp05t((float)c6 / 12.f) * on;
  }
  if (tid < a.num_degree + 2) y[4 + tid] = nv > 0 ? (float)a.hist[((long long)a.B + b) * 64 + tid] / nvf : 0.f;
  if (tid >= 64 && tid < 64 + a.Cx)
    y[4 + a.num_degree + 2 + (tid - 64)] = nv > 0 ? (float)a.hist[((long long)2 * a.B + b) * 64 + (tid - 64)] / nvf : 0.f;
  if (tid >= 128 && tid < 128 + a.C) {
    const float tot = nvf * (nvf - 1.f);
    y[4 + a.num_degree + 2 + a.Cx + (tid - 128)] = tot > 0.f ? (float)a.hist[b * 64 + (tid - 128)] / tot : 0.f;
  }
}

// ---- 5: one thread per written float; channel q < n_oh: one-hot, then K reach channels, then the Adamic-Adar index
struct TiledPairArgs {
  const int8_t* labels;
  const uint8_t* mask;
  const uint32_t* reach;
  const float* S;
  const int* nvalid;
  float* e_out;
  int n, np, C, K, n_oh, e_ld, e_off, cycle_features;
  long long total;                      // B * n * n * (n_oh + K + 1)
};

__global__ __launch_bounds__(256) void adj_pair_kernel(TiledPairArgs a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.total) return;
  const int nq = a.n_oh + a.K + 1;
  const long long pair = t / nq;
  const int q = (int)(t - pair * nq);
  const long long nn = (long long)a.n * a.n;
  const int b = (int)(pair / nn);
  const int ij = (int)(pair - (long long)b * nn);
  const int i = ij / a.n, j = ij - i * a.n;
  float* row = a.e_out + pair * a.e_ld;
  const long long o = ((long long)b * a.np + i) * a.np + j;
  if (q < a.n_oh) {
    const uint8_t* m = a.mask + (long long)b * a.n;
    const bool live = i != j && m[i] && m[j];
    row[q] = (live && a.labels[pair] == q) ? 1.f : 0.f;
    return;
  }
  const int k = q - a.n_oh;
  if (k < a.K) {
    row[a.e_off + k] = (a.cycle_features && ((a.reach[o] >> k) & 1u)) ? 0.2f : 0.f;
    return;
  }
  const float nvf = (float)a.nvalid[b];
  const float s = i != j ? a.S[o] : 0.f;
  row[a.e_off + a.K] = clamp05t(s * (logf(nvf) / nvf));
}

// The padded Laplacian alone for any n (same numbers as adj_scale_kernel's lap_out): one wave per (row, graph).
__global__ __launch_bounds__(64) void laplacian_rows_kernel(const int8_t* labels, const uint8_t* mask, int n, float* lap_out) {
  const int i = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int8_t* lab = labels + ((long long)b * n + i) * n;
  const uint8_t* m = mask + (long long)b * n;
  const bool mi = m[i] != 0;
  float d = 0.f;
  for (int j = lane; j < n; j += 64) d += (mi && j != i && m[j] && lab[j] > 0) ? 1.f : 0.f;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) d += __shfl_xor(d, s);
  float* out = lap_out + ((long long)b * n + i) * n;
  for (int j = lane; j < n; j += 64) {
    float v;
    if (mi && m[j]) v = (i == j ? d : 0.f) - ((j != i && lab[j] > 0) ? 1.f : 0.f);
    else v = (i == j && !mi) ? 2.0f * n : 0.f;
    out[j] = v;
  }
}

int launch_laplacian_rows(const int8_t* labels, const uint8_t* mask, int B, int n, float* lap_out, hipStream_t st) {
  if (B > 65535) return fail("laplacian: B=%d > 65535", B);
  hipLaunchKernelGGL(laplacian_rows_kernel, dim3(n, B), dim3(64), 0, st, labels, mask, n, lap_out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("laplacian: %s", hipGetErrorString(e));
}

}  // namespace scgg

using namespace scgg;

extern "C" size_t scgg_adjacency_features_workspace_bytes(int32_t B, int32_t n) {
  if (B <= 0 || n <= 0) return 0;
  const int np = (n + TL - 1) / TL * TL;
  return plan(nullptr, B, np).bytes;
}

extern "C" int scgg_adjacency_features_tiled(const int8_t* labels, const int8_t* node_labels, const uint8_t* node_mask,
                                             int32_t B, int32_t n, int32_t C, int32_t Cx, int32_t num_degree,
                                             int32_t max_degree, int32_t cycle_features, float* x_out, int32_t x_ld,
                                             int32_t x_off, float* y_out, int32_t y_ld, int32_t y_off, float* e_out,
                                             int32_t e_ld, int32_t e_off, float* onehot_out, float* lap_out,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  if (!labels || !node_mask || !x_out || !y_out || !e_out || !workspace) return fail("adjacency_features_tiled: null argument");
  if (B <= 0 || n <= 0 || n > 4096 || B > 65535) return fail("adjacency_features_tiled: bad shape B=%d n=%d", B, n);
  if (max_degree < 6 || max_degree > 32 || num_degree + 2 > 64 || C > 64 || Cx > 64 || C < 1 || Cx < 1)
    return fail("adjacency_features_tiled: unsupported max_degree=%d num_degree=%d C=%d Cx=%d", max_degree, num_degree, C, Cx);
  if (e_ld < e_off + max_degree + 1 || (onehot_out && e_off < C) || x_ld < x_off + 3 ||
      y_ld < y_off + 4 + num_degree + 2 + Cx + C)
    return fail("adjacency_features_tiled: bad destination stride");
  if (onehot_out != nullptr && onehot_out != e_out)
    return fail("adjacency_features_tiled: onehot_out must be NULL or e_out (the one-hot goes to channels [0,C) of the same rows)");
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail("adjacency_features_tiled: workspace not 256-byte aligned");
  const int np = (n + TL - 1) / TL * TL;
  const TiledPlanes p = plan(workspace, B, np);
  if (workspace_bytes < p.bytes)
    return fail("adjacency_features_tiled: workspace %zu < %zu bytes", workspace_bytes, p.bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long bnp = (long long)B * np;
  if (hipMemsetAsync(p.hist, 0, (size_t)3 * B * 64 * sizeof(int), st) != hipSuccess) return fail("adjacency_features_tiled: memset");
  hipLaunchKernelGGL(adj_build_kernel, dim3((np + 15) / 16, B), dim3(256), 0, st, labels, node_mask, n, np, p.A, p.reach,
                     p.deg, p.diag, p.hist, p.nvalid);
  const unsigned eb = (unsigned)(((long long)np * np + 255) / 256);
  hipLaunchKernelGGL(adj_scale_kernel, dim3(eb, B), dim3(256), 0, st, p.A, p.deg, node_mask, n, np, p.Wc, p.Wr, lap_out);
  const dim3 gg(np / TL, np / TL, B);
  // P_k = P_{k-1} A: A2 and A3 are kept (cycle formulas), the rest ping-pong
  const float* prev = p.A;
  for (int k = 2; k <= max_degree; ++k) {
    float* out = k == 2 ? p.A2 : k == 3 ? p.A3 : ((k & 1) ? p.Pb : p.Pa);
    hipLaunchKernelGGL(adj_gemm_kernel, gg, dim3(256), 0, st, prev, p.A, out, np, k <= 6 ? p.diag + (k - 1) * bnp : nullptr,
                       p.reach, 1u << (k - 1));
    prev = out;
  }
  hipLaunchKernelGGL(adj_gemm_kernel, gg, dim3(256), 0, st, p.Wc, p.Wr, p.S, np, (float*)nullptr, (uint32_t*)nullptr, 0u);
  TiledNodeArgs na{p.A, p.A2, p.A3, p.diag, p.deg, p.hist, p.nvalid, node_mask, node_labels, p.rowpart, x_out, y_out,
                   B, n, np, C, Cx, num_degree, cycle_features, x_ld, x_off, y_ld, y_off};
  hipLaunchKernelGGL(adj_rows_kernel, dim3((n + 3) / 4, B), dim3(256), 0, st, na);
  hipLaunchKernelGGL(adj_graph_kernel, dim3(B), dim3(256), 0, st, na);
  TiledPairArgs pa{labels, node_mask, p.reach, p.S, p.nvalid, e_out, n, np, C, max_degree, onehot_out ? C : 0, e_ld, e_off,
                   cycle_features, (long long)B * n * n * ((onehot_out ? C : 0) + max_degree + 1)};
  const long long pb = (pa.total + 255) / 256;
  if (pb > 0x7fffffffLL) return fail("adjacency_features_tiled: too many pair channels for one launch");
  hipLaunchKernelGGL(adj_pair_kernel, dim3((unsigned)pb), dim3(256), 0, st, pa);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("adjacency_features_tiled: %s", hipGetErrorString(e));
  return 0;
}
