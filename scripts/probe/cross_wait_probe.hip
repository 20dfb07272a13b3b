// Probe: the engine's per-layer event pattern (two streams, crossing waits) on synthetic kernels.
//   S: edge(l) -> rec EDGE(l); wait FILM(l+1); node(l) -> rec X(l)          [edge writes E(l+1); node reads F(l), E... writes X(l+1)]
//   A: wait X(l-1); stats(l) ; y(l) -> rec FILM(l+1); wait EDGE(l); pool(l)  [y writes F(l+1); pool checks E(l+1); stats checks X(l)]
// Every buffer element carries the (forward, layer) tag of its writer; readers count elements with an unexpected tag.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void work(int* dst, int n, int tag, const int* src1, int tag1, const int* src2, int tag2, int spin, int* bad) {
  long long t0 = clock64();
  const int s = spin * (1 + (int)(blockIdx.x % 3));
  while (clock64() - t0 < s) {}
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (src1 && src1[(i * 97 + 13) % n] != tag1) atomicAdd(bad, 1);
    if (src2 && src2[(i * 61 + 7) % n] != tag2) atomicAdd(bad + 1, 1);
    if (dst) dst[i] = tag;
  }
}

int main(int argc, char** argv) {
  const int forwards = argc > 1 ? atoi(argv[1]) : 300, L = 5, n = 1 << 16;
  const int edge_spin = argc > 2 ? atoi(argv[2]) : 6000, y_spin = argc > 3 ? atoi(argv[3]) : 2500;
  int *E[2], *X[2], *F[2], *bad;
  for (int k = 0; k < 2; ++k) { CK(hipMalloc(&E[k], n * 4)); CK(hipMalloc(&X[k], n * 4)); CK(hipMalloc(&F[k], n * 4)); }
  CK(hipMalloc(&bad, 64)); CK(hipMemset(bad, 0, 64));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t S = nullptr, A;
  CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, hi));
  std::vector<hipEvent_t> ev(4 * (L + 1) + 4);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto EV = [&](int l, int k) { return ev[4 * (l + 1) + k]; };   // 0 FILM, 1 EDGE, 2 NS, 3 X
  for (int f = 1; f <= forwards; ++f) {
    const int base = f * 100;
    // prologue: S writes E(0), records fork; A writes X(0), F(0)
    CK(hipEventRecord(ev[1], S)); CK(hipStreamWaitEvent(A, ev[1], 0));
    hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, S, E[0], n, base, nullptr, 0, nullptr, 0, 2000, bad);
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, A, X[0], n, base, nullptr, 0, nullptr, 0, 1000, bad);
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, A, F[0], n, base, nullptr, 0, nullptr, 0, 500, bad);
    CK(hipEventRecord(EV(0, 0), A));
    CK(hipStreamWaitEvent(S, EV(0, 0), 0));
    for (int l = 0; l < L; ++l) {
      const int c = l & 1, nx = c ^ 1;
      // A: stats of X(l)
      if (l > 0) CK(hipStreamWaitEvent(A, EV(l - 1, 3), 0));
      hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, A, (int*)nullptr, n, 0, X[c], base + l, nullptr, 0, 300, bad + 2);
      // S: edge(l): reads E(l), F(l); writes E(l+1)
      hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, S, E[nx], n, base + l + 1, E[c], base + l, F[c], base + l, edge_spin, bad + 4);
      CK(hipEventRecord(EV(l, 1), S));
      // A: y(l): writes F(l+1)
      hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, A, F[nx], n, base + l + 1, nullptr, 0, nullptr, 0, y_spin, bad);
      CK(hipEventRecord(EV(l + 1, 0), A));
      // A: pool of E(l+1)
      CK(hipStreamWaitEvent(A, EV(l, 1), 0));
      hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, A, (int*)nullptr, n, 0, E[nx], base + l + 1, nullptr, 0, 300, bad + 6);
      // S: node(l): waits for F(l+1); reads X(l), F(l); writes X(l+1)
      CK(hipStreamWaitEvent(S, EV(l + 1, 0), 0));
      hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, S, X[nx], n, base + l + 1, X[c], base + l, F[c], base + l, 2500, bad + 8);
      CK(hipEventRecord(EV(l, 3), S));
    }
    CK(hipStreamWaitEvent(A, EV(L - 1, 3), 0));
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, A, (int*)nullptr, n, 0, X[L & 1], base + L, nullptr, 0, 300, bad + 10);
    CK(hipEventRecord(ev[0], A)); CK(hipStreamWaitEvent(S, ev[0], 0));
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, S, (int*)nullptr, n, 0, E[L & 1], base + L, F[L & 1], base + L, 300, bad + 12);
    if (f % 7 == 0) CK(hipDeviceSynchronize());
  }
  CK(hipDeviceSynchronize());
  int h[16]; CK(hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost));
  printf("stale reads: stats(X) %d | edge(E,F) %d %d | pool(E) %d | node(X,F) %d %d | tail stats %d | tail check %d %d   over %d forwards\n",
         h[2], h[4], h[5], h[6], h[8], h[9], h[10], h[12], h[13], forwards);
  return 0;
}
