// Probe: do two workgroups of different kernels / streams that share a CU keep their LDS contents when the CU's LDS is
// packed almost full?  Kernel = fill own dynamic LDS with a tag pattern, spin, verify, repeat.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void lds_hold(int words, unsigned tag, int rounds, int spin, int* bad) {
  extern __shared__ unsigned lds[];
  for (int r = 0; r < rounds; ++r) {
    const unsigned t = tag + 977u * r + 131u * blockIdx.x;
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = t ^ (unsigned)i;
    __syncthreads();
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    int b = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) b += lds[i] != (t ^ (unsigned)i);
    if (b) atomicAdd(bad, b);
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  int* bad; CK(hipMalloc(&bad, 8));
  hipStream_t A, B;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi));
  CK(hipFuncSetAttribute((const void*)lds_hold, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int cases[][3] = {   // bytes of kernel on A (x wgs per CU it can fit), bytes of kernel on B, threads of A
    {74496, 12800, 512}, {74496, 36864, 512}, {124928, 36864, 768}, {124928, 30000, 768}, {124928, 20000, 768},
    {65536, 65536, 512}, {81920, 81920, 512}, {100 * 1024, 12800, 512}, {40960, 40960, 256}, {76032, 74496, 512}};
  for (auto& c : cases) {
    CK(hipMemset(bad, 0, 8));
    CK(hipDeviceSynchronize());
    for (int it = 0; it < iters; ++it) {
      hipLaunchKernelGGL(lds_hold, dim3(512), dim3(c[2]), c[0], A, c[0] / 4, 0x1000u * it, 4, 3000, bad);
      hipLaunchKernelGGL(lds_hold, dim3(512), dim3(256), c[1], B, c[1] / 4, 0x77000000u + it, 6, 1500, bad + 1);
    }
    CK(hipDeviceSynchronize());
    int h[2]; CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
    printf("A %6d B (%d thr) + B %6d B: corrupted words A %d, B %d  (%d launches each)\n", c[0], c[2], c[1], h[0], h[1], iters);
  }
  return 0;
}
