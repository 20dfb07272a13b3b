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

}  // namespace scgg

using namespace scgg;

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
