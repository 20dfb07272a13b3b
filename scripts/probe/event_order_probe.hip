// Probe: which cross-stream event patterns order kernels correctly on this runtime.
//   hipcc -O2 --offload-arch=gfx950 event_order_probe.hip -o event_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void slow_write(int* buf, int n, int val, int spin) {   // spins, then writes val to every element
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = val;
}
__global__ void check(const int* buf, int n, int val, int* bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (buf[i] != val) atomicAdd(bad, 1);
}
__global__ void tiny(int* p) { if (threadIdx.x == 9999) p[0] = 1; }

// producer / consumer on ONE stream with a skewed producer (workgroup w spins w-dependent time, then writes slice w); the
// consumer's workgroup w checks slice (w * 37 + 11) % nwg, i.e. data written by another workgroup / XCD
__global__ void skew_write(int* buf, int per, int val, int spin) {
  long long t0 = clock64();
  const int s = spin * (1 + (int)(blockIdx.x % 5));
  while (clock64() - t0 < s) {}
  for (int i = threadIdx.x; i < per; i += blockDim.x) buf[(size_t)blockIdx.x * per + i] = val;
}
__global__ void skew_check(const int* buf, int per, int val, int* bad) {
  const int w = (blockIdx.x * 37 + 11) % gridDim.x;
  for (int i = threadIdx.x; i < per; i += blockDim.x)
    if (buf[(size_t)w * per + i] != val) atomicAdd(bad, 1);
}
__global__ void busy(float* p, int spin) {   // background load on the other stream
  long long t0 = clock64();
  float v = p[threadIdx.x];
  while (clock64() - t0 < spin) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main(int argc, char** argv) {
  if (argc > 2) {   // same-stream producer -> consumer with / without a busy second stream
    const int iters = atoi(argv[1]), nwg = 512, per = 256;
    int *buf, *bad; float* bg;
    CK(hipMalloc(&buf, (size_t)nwg * per * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&bg, 1024 * 256 * 4));
    CK(hipMemset(bg, 0, 1024 * 256 * 4));
    hipStream_t A, B;
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    for (int load = 0; load < 2; ++load) {
      CK(hipMemset(bad, 0, 4));
      CK(hipDeviceSynchronize());
      for (int it = 1; it <= iters; ++it) {
        if (load) hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, B, bg, 30000);
        hipLaunchKernelGGL(skew_write, dim3(nwg), dim3(256), 0, A, buf, per, it, 1500);
        hipLaunchKernelGGL(skew_check, dim3(nwg), dim3(256), 0, A, buf, per, it, bad);
      }
      CK(hipDeviceSynchronize());
      int h = 0; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
      printf("same-stream producer -> consumer, second stream %s: %d stale elements over %d rounds\n", load ? "busy" : "idle", h, iters);
    }
    return 0;
  }

  const int iters = argc > 1 ? atoi(argv[1]) : 500;
  const int n = 1 << 16;
  int *buf, *bad, *junk, *junk2;
  CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&junk, 4)); CK(hipMalloc(&junk2, 4096));
  hipStream_t A, B;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  for (int prio = 0; prio < 2; ++prio)
    for (int bnull = 0; bnull < 2; ++bnull) {
      CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, prio ? hi : lo));
      if (bnull) B = nullptr; else CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
      hipEvent_t E, F, G;
      CK(hipEventCreateWithFlags(&E, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&F, hipEventDisableTiming));
      CK(hipEventCreateWithFlags(&G, hipEventDisableTiming));
      // pattern 4: does a kernel that follows a cross-stream wait still wait for the kernel BEFORE the wait on its own stream?
      //   B: slow_write(it); [wait for an event of A that is already (or soon) complete]; check(it)
      for (int sub = 0; sub < 4; ++sub) {
        CK(hipMemset(bad, 0, 4));
        CK(hipDeviceSynchronize());
        for (int it = 1; it <= iters; ++it) {
          if (sub < 2) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, A, junk);
          else hipLaunchKernelGGL(slow_write, dim3(4), dim3(64), 0, A, junk2, 64, it, sub == 2 ? 8000 : 60000);   // event not complete at enqueue time
          CK(hipEventRecord(G, A));
          hipLaunchKernelGGL(slow_write, dim3(64), dim3(256), 0, B, buf, n, it, 20000 + (it % 7) * 3000);
          if (sub >= 1) CK(hipEventRecord(F, B));          // as the engine does: record, then wait
          CK(hipStreamWaitEvent(B, G, 0));
          hipLaunchKernelGGL(check, dim3(64), dim3(256), 0, B, buf, n, it, bad);
        }
        CK(hipDeviceSynchronize());
        int h = 0; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("A prio %s, B %s, pattern 4.%d (kernel; wait; kernel on one stream): %d stale elements over %d rounds\n", prio ? "high" : "low",
               bnull ? "null stream" : "created", sub, h, iters);
      }
      for (int pat = 0; pat < 4; ++pat) {
        CK(hipMemset(bad, 0, 4));
        CK(hipDeviceSynchronize());
        for (int it = 1; it <= iters; ++it) {
          // fork: A waits for B (like the engine's ev_fork)
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, junk);
          CK(hipEventRecord(F, B)); CK(hipStreamWaitEvent(A, F, 0));
          hipLaunchKernelGGL(slow_write, dim3(64), dim3(256), 0, A, buf, n, it, 20000 + (it % 7) * 3000);
          if (pat == 0) {                       // record directly behind the kernel
            CK(hipEventRecord(E, A));
          } else if (pat == 1) {                // a wait for an (early) event of B sits between the kernel and the record
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, junk);
            CK(hipEventRecord(G, B)); CK(hipStreamWaitEvent(A, G, 0));
            CK(hipEventRecord(E, A));
          } else if (pat == 2) {                // record first, then more work on A
            CK(hipEventRecord(E, A));
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, A, junk);
          } else {                              // wait + tiny kernel + record
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, junk);
            CK(hipEventRecord(G, B)); CK(hipStreamWaitEvent(A, G, 0));
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, A, junk);
            CK(hipEventRecord(E, A));
          }
          CK(hipStreamWaitEvent(B, E, 0));
          hipLaunchKernelGGL(check, dim3(64), dim3(256), 0, B, buf, n, it, bad);
          // join A behind B's check before the next round overwrites buf
          CK(hipEventRecord(F, B)); CK(hipStreamWaitEvent(A, F, 0));
        }
        CK(hipDeviceSynchronize());
        int h = 0; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("A prio %s, B %s, pattern %d: %d stale elements over %d rounds\n", prio ? "high" : "low", bnull ? "null stream" : "created", pat, h, iters);
      }
      CK(hipStreamDestroy(A)); if (!bnull) CK(hipStreamDestroy(B));
    }
  return 0;
}
