// Development probe: LDS read throughput of one CU for the fragment-read pattern of the edge kernel
// (every lane of a wave reads 16 consecutive bytes, lane-consecutive; all waves of the workgroup stream).
//   hipcc -O3 --offload-arch=gfx950 lds_rate_probe.hip -o lds_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void lds_kernel(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = i;
  __syncthreads();
  f4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (MODE == 0) {
        const f4 v = *reinterpret_cast<const f4*>(smem + ((i * 16 + k) & 63) * 1024 + lane * 16);
        acc += v;
      } else if (MODE == 1) {
        const f2 v = *reinterpret_cast<const f2*>(smem + ((i * 16 + k) & 127) * 512 + lane * 8);
        acc[0] += v[0]; acc[1] += v[1];
      } else {
        const float v = *reinterpret_cast<const float*>(smem + ((i * 16 + k) & 255) * 256 + lane * 4);
        acc[0] += v;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads, int bytes) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
  const int iters = 400;
  lds_kernel<MODE><<<1, threads, 65536>>>(out, cyc, iters);
  lds_kernel<MODE><<<1, threads, 65536>>>(out, cyc, iters);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-14s waves=%2d  %.1f B/tick/CU  (%.1f ticks per wave-instruction)\n", name, threads / 64,
         (double)iters * 16 * threads * bytes / c, (double)c / (iters * 16.0));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int t : {64, 256, 512, 1024}) {
    run<0>("ds_read_b128", t, 16);
    run<1>("ds_read_b64", t, 8);
    run<2>("ds_read_b32", t, 4);
  }
  return 0;
}
