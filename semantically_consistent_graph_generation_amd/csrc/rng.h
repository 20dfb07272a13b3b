// Counter-based Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) for the sampling kernels.
// counter = ((i<<16)|j, graph id, step, stream*16 + block), key = (seed lo, seed hi); uniforms from the top 24 bits.
// The numpy twin used by the tests is oracle/noise.py:philox4x32_10 (checked there against the Random123 vectors).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scgg {

constexpr int STREAM_E = 0;
constexpr int STREAM_X = 1;

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                        uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)c0 * 0xD2511F53ull;
    const uint64_t p1 = (uint64_t)c2 * 0xCD9E8D57ull;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

}  // namespace scgg
