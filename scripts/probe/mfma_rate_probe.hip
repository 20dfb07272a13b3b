// Development probe: issue interval of v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x4_f32, dependent (one accumulator)
// and independent (four accumulators), one and two waves per SIMD, and beside a VALU stream of the partner wave.
//   hipcc -O3 --offload-arch=gfx950 mfma_rate_probe.hip -o mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
  float fa = threadIdx.x, fb = 1.f;
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {          // bf16, dependent
#pragma unroll
      for (int r = 0; r < 16; ++r) c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    } else if (MODE == 1) {   // bf16, 4 independent
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      }
    } else if (MODE == 2) {   // f32, dependent
#pragma unroll
      for (int r = 0; r < 16; ++r) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
    } else if (MODE == 3) {   // f32, 4 independent
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c3, 0, 0, 0);
      }
    } else if (MODE == 4 || MODE == 5) {   // waves 0..3: MFMA (4: bf16, 5: f32) independent; waves 4..7: VALU stream
      if (wave < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (MODE == 4) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
          } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c3, 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1"
                       : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      }
    } else if (MODE == 6) {   // one wave: bf16 MFMA interleaved 1:2 with independent VALU
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        asm volatile("v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        asm volatile("v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8 * 16);
  const int iters = 500;
  k<MODE><<<1, threads>>>(out, cyc, iters);
  k<MODE><<<1, threads>>>(out, cyc, iters);
  unsigned long long c[16]; hipMemcpy(c, cyc, 8 * 16, hipMemcpyDeviceToHost);
  printf("%-44s waves/SIMD=%d  wave0: %.1f ticks per MFMA", name, threads / 256, (double)c[0] / (iters * 16.0));
  if (MODE == 4 || MODE == 5) printf("   wave4: %.2f ticks per VALU", (double)c[4] / (iters * 64.0));
  printf("\n");
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int t : {256, 512}) {
    run<0>("bf16 16x16x32 dependent", t);
    run<1>("bf16 16x16x32 4 accumulators", t);
    run<2>("f32 16x16x4 dependent", t);
    run<3>("f32 16x16x4 4 accumulators", t);
  }
  run<4>("bf16 MFMA waves 0-3 | VALU waves 4-7", 512);
  run<5>("f32 MFMA waves 0-3 | VALU waves 4-7", 512);
  run<6>("bf16 MFMA + 2 VALU each, same wave", 256);
  run<6>("bf16 MFMA + 2 VALU each, same wave", 512);
  return 0;
}
