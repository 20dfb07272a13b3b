// Training-side kernels (gfx950): counter-based dropout and the masked cross-entropy of TrainLoss with its gradient.
//
// Reference:
//   dropout sites  src/ConStruct/models/transformer_model.py:88-105 (layers), :284-340 (in / out MLPs)
//   TrainLoss      src/ConStruct/metrics/train_metrics.py:25-85, CrossEntropyMetric src/ConStruct/metrics/abstract_metrics.py:77-93
//                  (sum of cross entropies over the rows that take part / number of those rows)
// The reference draws dropout masks from torch's global generator; here the mask of site s is a pure function of
// (seed, s, flat element index): Philox4x32-10, counter ((index >> 2) low, (index >> 2) high, s, 0xD0), key = seed, word
// (index & 3) -> u in [0,1) from the top 24 bits, kept iff u >= p, scaled by 1 / (1 - p).  The same function is oracle/training.py:dropout_keep, so
// a train-mode forward can be compared bit for bit, and the backward pass re-derives the mask instead of storing it.
// Both kernels are HBM-bound streaming passes: dropout 8 B per element, the loss (4 C + 1) B per row read and 4 C written.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/scgg.h"
#include "kernels.h"
#include "rng.h"

namespace scgg {

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, long long numel,
                                                      float p, float denom, uint32_t k0, uint32_t k1, uint32_t site,
                                                      const uint32_t* __restrict__ key_dev) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= numel) return;
  if (key_dev != nullptr) { k0 = key_dev[0]; k1 = key_dev[1]; }   // the key read at RUN time: a captured graph replays with new masks
  uint32_t r[4];
  const long long c = idx >> 2;                        // four consecutive elements share one Philox call (train_edge.hip: a lane's f32x4)
  philox4x32_10((uint32_t)(c & 0xffffffffll), (uint32_t)(c >> 32), site, 0xD0u, k0, k1, r);
  out[idx] = u01(r[idx & 3]) >= p ? x[idx] / denom : 0.f;   // a division, as the oracle's x * keep / (1 - p)
}

// rows of C logits, int8 target per row (< 0: the row does not take part).  partial[blockIdx.x] = sum of the cross
// entropies of this block's rows (fixed order inside the block: reproducible); grad (optional) = (softmax - onehot) * g[0]
// on the rows that take part, 0 elsewhere.
__global__ __launch_bounds__(256) void masked_ce_kernel(const float* __restrict__ logits, const int8_t* __restrict__ target,
                                                        long long rows, int C, float* __restrict__ partial,
                                                        float* __restrict__ grad, const float* __restrict__ g) {
  __shared__ float red[4];
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f;
  if (r < rows) {
    const int t = target[r];
    const float* z = logits + r * C;
    if (t >= 0 && t < C) {
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += expf(z[c] - mx);
      const float lse = mx + logf(s);
      loss = lse - z[t];
      if (grad != nullptr) {
        const float gg = g[0];
        for (int c = 0; c < C; ++c) grad[r * C + c] = (expf(z[c] - lse) - (c == t ? 1.f : 0.f)) * gg;
      }
    } else if (grad != nullptr) {
      for (int c = 0; c < C; ++c) grad[r * C + c] = 0.f;
    }
  }
  if (partial != nullptr) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) loss += __shfl_xor(loss, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// out[b,i,j,:] = out[b,j,i,:] = 0.5 (R[b,i,j,:] + R[b,j,i,:]) for valid i != j, 0 elsewhere: the symmetrisation and mask between
// two layers (transformer_model.py:108-112 + PlaceHolder.mask) in ONE pass -- as tensor ops it is a strided add, a scale and a
// masked multiply, three round trips of the [B,n,n,C] tensor, and as many again in the backward.  The map is linear and
// self-adjoint, so the backward is the same kernel on the gradient.  One thread per unordered pair and V channels.
template <int V>
__global__ __launch_bounds__(256) void sym_mask_kernel(const float* __restrict__ R, const uint8_t* __restrict__ mask, int n, int C,
                                                       long long total, float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int CV = C / V;
  const int c = (int)(idx % CV) * V;
  long long r = idx / CV;
  const int j = (int)(r % n); r /= n;
  const int i = (int)(r % n);
  const long long b = r / n;
  if (i > j) return;
  const long long pij = ((b * n + i) * n + j) * C + c, pji = ((b * n + j) * n + i) * C + c;
  const bool live = i != j && mask[b * n + i] != 0 && mask[b * n + j] != 0;
  if (V == 4) {
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (live) v = 0.5f * (*reinterpret_cast<const f32x4_t*>(R + pij) + *reinterpret_cast<const f32x4_t*>(R + pji));
    *reinterpret_cast<f32x4_t*>(out + pij) = v;
    if (i != j) *reinterpret_cast<f32x4_t*>(out + pji) = v;
  } else {
    const float v = live ? 0.5f * (R[pij] + R[pji]) : 0.f;
    out[pij] = v;
    if (i != j) out[pji] = v;
  }
}

// Xtoy's pooling in training mode (layers.py:12-21 as the reference computes it: sums over ALL n rows -- masked rows are zero --
// divided by the number of valid nodes; min / max over x + 1e5 (1 - w) and x - 1e5 (1 - w); variance = sum w (x - mean)^2 / cnt):
// out [B, 4 dx] = mean | min | max | var and, for the backward, aux [B, 3 dx] = mean | sum w (x - mean) | cnt and the rows of
// the two extrema (ties: the smaller row).  As tensor ops this was ~12 launches per layer, two of them 48-us reductions with 512 outputs.
__global__ __launch_bounds__(256) void train_node_stats_kernel(const float* __restrict__ X, const uint8_t* __restrict__ mask, int n, int DX,
                                                               float* __restrict__ out, float* __restrict__ aux, int* __restrict__ arg) {
  // workgroup = (graph, 32 channels); thread = (channel, one of 8 row lanes): a lane takes the rows rl, rl + 8, ..., the lanes
  // are merged through LDS in lane order (one thread per channel walking all rows twice was an 80-us chain of round trips)
  __shared__ float sh[3][8][32];
  __shared__ int shi[2][8][32];
  __shared__ float smean[32];
  const int b = blockIdx.y, cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const bool okc = c < DX;
  const uint8_t* mb = mask + (size_t)b * n;
  int cnt = 0;
  for (int i = 0; i < n; ++i) cnt += mb[i] != 0;
  const float fc = (float)cnt;
  const float* xp = X + (size_t)b * n * DX + (okc ? c : 0);
  float s = 0.f, mn = INFINITY, mx = -INFINITY;
  int imn = 0, imx = 0;
  for (int i = rl; i < n; i += 8) {
    const float v = xp[(size_t)i * DX];
    s += v;
    const float big = mb[i] != 0 ? 0.f : 1e5f;
    if (v + big < mn) { mn = v + big; imn = i; }
    if (v - big > mx) { mx = v - big; imx = i; }
  }
  sh[0][rl][cl] = s; sh[1][rl][cl] = mn; sh[2][rl][cl] = mx; shi[0][rl][cl] = imn; shi[1][rl][cl] = imx;
  __syncthreads();
  if (rl == 0) {
    for (int q = 1; q < 8; ++q) {
      s += sh[0][q][cl];
      if (sh[1][q][cl] < mn || (sh[1][q][cl] == mn && shi[0][q][cl] < imn)) { mn = sh[1][q][cl]; imn = shi[0][q][cl]; }
      if (sh[2][q][cl] > mx || (sh[2][q][cl] == mx && shi[1][q][cl] < imx)) { mx = sh[2][q][cl]; imx = shi[1][q][cl]; }
    }
    smean[cl] = s / fc;
  }
  __syncthreads();
  const float mean = smean[cl];
  float q2 = 0.f, s1 = 0.f;
  for (int k = rl; k < n; k += 8) {
    if (mb[k] == 0) continue;
    const float d = xp[(size_t)k * DX] - mean;
    q2 += d * d; s1 += d;
  }
  sh[0][rl][cl] = q2; sh[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && okc) {
    for (int q = 1; q < 8; ++q) { q2 += sh[0][q][cl]; s1 += sh[1][q][cl]; }
    float* o = out + (size_t)b * 4 * DX + c;
    o[0] = mean; o[DX] = mn; o[2 * DX] = mx; o[3 * DX] = q2 / fc;
    float* ax = aux + (size_t)b * 3 * DX + c;
    ax[0] = mean; ax[DX] = s1; ax[2 * DX] = fc;
    arg[(size_t)b * 2 * DX + c] = imn; arg[(size_t)b * 2 * DX + DX + c] = imx;
  }
}

// dX of the above from g [B, 4 dx] = d mean | d min | d max | d var
__global__ __launch_bounds__(256) void train_node_stats_bwd_kernel(const float* __restrict__ X, const uint8_t* __restrict__ mask, int n, int DX,
                                                                   const float* __restrict__ g, const float* __restrict__ aux,
                                                                   const int* __restrict__ arg, long long total, float* __restrict__ dX) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % DX);
  const long long r = idx / DX;
  const int i = (int)(r % n);
  const long long b = r / n;
  const float* gb = g + b * 4 * DX + c;
  const float* ax = aux + b * 3 * DX + c;
  const float mean = ax[0], s1 = ax[DX], fc = ax[2 * DX];
  const float w = mask[b * n + i] != 0 ? 1.f : 0.f;
  float d = gb[0] / fc + gb[3 * DX] * (2.f * w * (X[idx] - mean) / fc - 2.f * s1 / (fc * fc));
  if (arg[b * 2 * DX + c] == i) d += gb[DX];
  if (arg[b * 2 * DX + DX + c] == i) d += gb[2 * DX];
  dX[idx] = d;
}

}  // namespace scgg

using namespace scgg;

extern "C" int scgg_train_node_stats(const float* X, const uint8_t* mask, int32_t B, int32_t n, int32_t dx, float* out, float* aux,
                                     int32_t* arg, void* stream) {
  if (!X || !mask || !out || !aux || !arg || B <= 0 || n <= 0 || dx <= 0) return fail("train_node_stats: bad arguments");
  if (B > 65535) return fail("train_node_stats: batch too large for one launch");
  hipLaunchKernelGGL(train_node_stats_kernel, dim3((unsigned)((dx + 31) / 32), (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), X,
                     mask, n, dx, out, aux, arg);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_node_stats: %s", hipGetErrorString(e));
}

extern "C" int scgg_train_node_stats_backward(const float* X, const uint8_t* mask, int32_t B, int32_t n, int32_t dx, const float* g,
                                              const float* aux, const int32_t* arg, float* dX, void* stream) {
  if (!X || !mask || !g || !aux || !arg || !dX || B <= 0 || n <= 0 || dx <= 0) return fail("train_node_stats_backward: bad arguments");
  const long long total = (long long)B * n * dx;
  hipLaunchKernelGGL(train_node_stats_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), X, mask,
                     n, dx, g, aux, arg, total, dX);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("train_node_stats_backward: %s", hipGetErrorString(e));
}

extern "C" int scgg_sym_mask(const float* R, const uint8_t* mask, int32_t B, int32_t n, int32_t C, float* out, void* stream) {
  if (!R || !mask || !out || B <= 0 || n <= 0 || C <= 0) return fail("sym_mask: bad arguments");
  if (R == out) return fail("sym_mask: in place is not supported (a pair reads what its mirror writes)");
  const bool v4 = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(R) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const long long total = (long long)B * n * n * (v4 ? C / 4 : C);
  const long long nb = (total + 255) / 256;
  if (nb > 0x7fffffffLL) return fail("sym_mask: tensor too large for one launch");
  if (v4) hipLaunchKernelGGL(sym_mask_kernel<4>, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), R, mask, n, C, total, out);
  else hipLaunchKernelGGL(sym_mask_kernel<1>, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), R, mask, n, C, total, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("sym_mask: %s", hipGetErrorString(e));
}

extern "C" int scgg_dropout(const float* x, float* out, int64_t numel, double p, uint64_t seed, uint32_t site, void* stream) {
  if (!x || !out) return fail("dropout: null argument");
  if (numel < 0 || !(p >= 0.0) || !(p < 1.0)) return fail("dropout: bad numel or p");
  if (numel == 0) return 0;
  const long long nb = (numel + 255) / 256;
  if (nb > 0x7fffffffLL) return fail("dropout: tensor too large for one launch");
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, (long long)numel,
                     (float)p, (float)(1.0 - p), (uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32), site,
                     static_cast<const uint32_t*>(nullptr));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("dropout: %s", hipGetErrorString(e));
}

extern "C" int scgg_dropout_devkey(const float* x, float* out, int64_t numel, double p, const uint32_t* key_dev, uint32_t site,
                                   void* stream) {
  if (!x || !out || !key_dev) return fail("dropout_devkey: null argument");
  if (numel < 0 || !(p >= 0.0) || !(p < 1.0)) return fail("dropout_devkey: bad numel or p");
  if (numel == 0) return 0;
  const long long nb = (numel + 255) / 256;
  if (nb > 0x7fffffffLL) return fail("dropout_devkey: tensor too large for one launch");
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, (long long)numel,
                     (float)p, (float)(1.0 - p), 0u, 0u, site, key_dev);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("dropout_devkey: %s", hipGetErrorString(e));
}

extern "C" int64_t scgg_masked_cross_entropy_blocks(int64_t rows) { return rows <= 0 ? 0 : (rows + 255) / 256; }

extern "C" int scgg_masked_cross_entropy(const float* logits, const int8_t* target, int64_t rows, int32_t C, float* partial,
                                         float* grad, const float* grad_scale, void* stream) {
  if (!logits || !target) return fail("masked_cross_entropy: null argument");
  if (rows < 0 || C < 1 || C > 127) return fail("masked_cross_entropy: bad shape rows=%lld C=%d", (long long)rows, C);
  if (grad != nullptr && grad_scale == nullptr) return fail("masked_cross_entropy: grad needs grad_scale");
  if (rows == 0) return 0;
  const long long nb = (rows + 255) / 256;
  if (nb > 0x7fffffffLL) return fail("masked_cross_entropy: too many rows for one launch");
  hipLaunchKernelGGL(masked_ce_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), logits, target,
                     (long long)rows, C, partial, grad, grad_scale);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("masked_cross_entropy: %s", hipGetErrorString(e));
}
