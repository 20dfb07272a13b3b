// Probe: does node_linear_x3 (victim, stream A) give the same bits when edge_mlp_in (aggressor, stream B) runs beside it?
// Links against libscgg.so's internal launchers (kernels.h).
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I../../include -I../../semantically_consistent_graph_generation_amd/csrc \
//         pair_probe.hip -L../../semantically_consistent_graph_generation_amd -lscgg -o pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <class T> T* dev(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
static float* dev_rand(size_t n, float scale, std::mt19937& g) {
  std::vector<float> h(n); std::uniform_real_distribution<float> u(-1.f, 1.f);
  for (auto& v : h) v = scale * u(g);
  float* p = dev<float>(n); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p;
}
__global__ void busy(float* p, int spin) {
  long long t0 = clock64(); float v = p[threadIdx.x];
  while (clock64() - t0 < spin) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 300;
  const int mode = argc > 2 ? atoi(argv[2]) : 1;     // 0: victim alone, 1: beside edge_mlp_in, 2: beside a busy kernel
  const int B = 64, n = 32, dx = 256, hid = 256, inE = 25, hidE = 128, de = 64, C = 14;
  std::mt19937 g(7);
  const int M = B * n;
  std::vector<uint8_t> mask(M, 0);
  const int nn[8] = {32, 17, 25, 12, 30, 31, 8, 1};
  for (int b = 0; b < B; ++b) for (int i = 0; i < nn[b % 8]; ++i) mask[b * n + i] = 1;
  uint8_t* d_mask = dev<uint8_t>(M); CK(hipMemcpy(d_mask, mask.data(), M, hipMemcpyHostToDevice));
  const int nch = 2;
  int *rows = dev<int>(M), *rcnt = dev<int>(64), *units = dev<int>(M * nch), *ucnt = dev<int>(64), *goff = dev<int>(2 * B);
  scgg::launch_build_lists(d_mask, B, n, goff, rows, rcnt, units, ucnt, 0);
  // victim: C = relu(A W^T + b), K = N = 256, x3 images
  float* W = dev_rand((size_t)dx * hid, 0.1f, g); float* bias = dev_rand(dx, 0.2f, g);
  const int NB = dx / 32, KB = hid / 32;
  scgg::f32x4_t* Wp = dev<scgg::f32x4_t>((size_t)NB * KB * 4 * 64);
  scgg::f32x4_t* W3 = dev<scgg::f32x4_t>((size_t)NB * (hid / 16) * 192);
  scgg::PackArgs pa{W, Wp, dx, hid, NB, KB}; scgg::launch_pack_linear(pa, 0);
  scgg::PackArgs p3{W, W3, dx, hid, NB, KB}; scgg::launch_pack_linear_x3(p3, 0);
  float* A = dev_rand((size_t)M * hid, 1.f, g);
  float* Cout = dev<float>((size_t)M * dx);
  scgg::NodeLinearArgs la{A, Wp, bias, Cout, rows, rcnt, dx, hid, NB, hid, dx, 2, 1};
  // aggressor: edge_mlp_in
  float* E = dev_rand((size_t)M * n * inE, 1.f, g);
  float *w0 = dev_rand((size_t)hidE * inE, 0.2f, g), *b0 = dev_rand(hidE, 0.2f, g), *w3 = dev_rand((size_t)de * hidE, 0.1f, g), *b3 = dev_rand(de, 0.2f, g);
  const long long img_f4 = scgg::edge_mlp_img_f4(inE, hidE, de);
  scgg::f32x4_t* img = dev<scgg::f32x4_t>((size_t)img_f4);
  scgg::EdgeMlpPackArgs ep{w0, b0, w3, b3, img, inE, hidE, de}; scgg::launch_edge_mlp_pack(ep, 0);
  float* E0 = dev<float>((size_t)M * n * de);
  scgg::EdgeMlpUnitArgs ea{E, nullptr, d_mask, img, E0, units, ucnt, B, n, inE, C};
  float* bg = dev<float>(1024 * 256);
  // the engine's X chain: narrow_linear (in_X = 10 -> 256, relu) -> node_linear_x3 (256 -> 256, relu) -> node_linear_x3 (256 -> 768)
  const int inX = 10;
  float* Xin = dev_rand((size_t)M * inX, 1.f, g);
  float *nw = dev_rand((size_t)hid * inX, 0.3f, g), *nb = dev_rand(hid, 0.2f, g);
  float* hidX = dev<float>((size_t)M * hid);
  scgg::NarrowLinearArgs na{Xin, nw, nb, hidX, rows, rcnt, hid, inX, inX, hid, 1};
  scgg::NodeLinearArgs la2{hidX, Wp, bias, Cout, rows, rcnt, dx, hid, NB, hid, dx, 2, 1};
  float* Wq = dev_rand((size_t)3 * dx * dx, 0.1f, g); float* bq = dev_rand(3 * dx, 0.2f, g);
  scgg::f32x4_t* Wqp = dev<scgg::f32x4_t>((size_t)(3 * dx / 32) * (dx / 32) * 4 * 64);
  scgg::f32x4_t* Wq3 = dev<scgg::f32x4_t>((size_t)(3 * dx / 32) * (dx / 16) * 192);
  scgg::PackArgs pq{Wq, Wqp, 3 * dx, dx, 3 * dx / 32, dx / 32}; scgg::launch_pack_linear(pq, 0);
  scgg::PackArgs pq3{Wq, Wq3, 3 * dx, dx, 3 * dx / 32, dx / 32}; scgg::launch_pack_linear_x3(pq3, 0);
  float* qkv = dev<float>((size_t)M * 3 * dx);
  scgg::NodeLinearArgs lq{Cout, Wqp, bq, qkv, rows, rcnt, 3 * dx, dx, 3 * dx / 32, dx, 3 * dx, 8, 0};
  float* epool = dev<float>((size_t)B * 4 * de);
  scgg::EdgePoolArgs pool{E0, d_mask, epool, B, n, de, 4 * de, 0};

  CK(hipDeviceSynchronize());
  hipStream_t SA, SB; CK(hipStreamCreateWithFlags(&SA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&SB, hipStreamNonBlocking));
  if (scgg::launch_node_linear_x3(la, W3, M, SA) != 0) { fprintf(stderr, "victim launch failed\n"); return 1; }
  if (scgg::launch_edge_mlp_in(ea, inE, hidE, de, SB) != 0) { fprintf(stderr, "aggressor launch failed\n"); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> ref((size_t)M * dx), cur((size_t)M * dx), refE((size_t)M * n * de), curE((size_t)M * n * de);
  CK(hipMemcpy(ref.data(), Cout, ref.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(refE.data(), E0, refE.size() * 4, hipMemcpyDeviceToHost));
  int badV = 0, badA = 0;
  for (int it = 0; it < (mode >= 3 ? 0 : iters); ++it) {
    CK(hipMemsetAsync(Cout, 0, (size_t)M * dx * 4, SA));
    if (mode == 1) scgg::launch_edge_mlp_in(ea, inE, hidE, de, SB);
    if (mode == 2) hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, SB, bg, 40000);
    scgg::launch_node_linear_x3(la, W3, M, SA);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(cur.data(), Cout, cur.size() * 4, hipMemcpyDeviceToHost));
    size_t d = 0; long first = -1;
    for (size_t i = 0; i < cur.size(); ++i) if (memcmp(&cur[i], &ref[i], 4)) { if (first < 0) first = (long)i; ++d; }
    if (d) { if (++badV <= 5) printf("iter %d: victim differs in %zu words, first row %ld col %ld (%g vs %g)\n", it, d, first / dx, first % dx, cur[first], ref[first]); }
    if (mode == 1) {
      CK(hipMemcpy(curE.data(), E0, curE.size() * 4, hipMemcpyDeviceToHost));
      size_t de_ = 0; for (size_t i = 0; i < curE.size(); ++i) de_ += memcmp(&curE[i], &refE[i], 4) != 0;
      if (de_) { if (++badA <= 5) printf("iter %d: aggressor output differs in %zu words\n", it, de_); }
    }
  }
  if (mode >= 3) {   // 3: the chain beside edge_mlp_in + pool; 4: the chain alone
    auto chain = [&]() {
      scgg::launch_narrow_linear(na, M, SA);
      scgg::launch_node_linear_x3(la2, W3, M, SA);
      scgg::launch_node_linear_x3(lq, Wq3, M, SA);
    };
    chain(); CK(hipDeviceSynchronize());
    std::vector<float> rq((size_t)M * 3 * dx), cq((size_t)M * 3 * dx);
    CK(hipMemcpy(ref.data(), Cout, ref.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rq.data(), qkv, rq.size() * 4, hipMemcpyDeviceToHost));
    int badX = 0, badQ = 0;
    for (int it = 0; it < iters; ++it) {
      // leave other values behind, as the engine's later layers do
      CK(hipMemsetAsync(Cout, 0x3f, (size_t)M * dx * 4, SA)); CK(hipMemsetAsync(qkv, 0x3f, (size_t)M * 3 * dx * 4, SA));
      CK(hipMemsetAsync(hidX, 0x3f, (size_t)M * hid * 4, SA));
      CK(hipDeviceSynchronize());
      if (mode == 3) { scgg::launch_edge_mlp_in(ea, inE, hidE, de, SB); scgg::launch_edge_pool_stats(pool, SB); }
      chain();
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(cur.data(), Cout, cur.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(cq.data(), qkv, cq.size() * 4, hipMemcpyDeviceToHost));
      size_t dX = 0, dQ = 0;
      for (size_t i = 0; i < cur.size(); ++i) { const int r = (int)(i / dx); if (mask[r] && memcmp(&cur[i], &ref[i], 4)) ++dX; }
      for (size_t i = 0; i < cq.size(); ++i) { const int r = (int)(i / (3 * dx)); if (mask[r] && memcmp(&cq[i], &rq[i], 4)) ++dQ; }
      if (dX || dQ) { ++badX; if (badX <= 5) printf("iter %d: X(0) words differ %zu, qkv words differ %zu\n", it, dX, dQ); }
      (void)badQ;
    }
    printf("mode %d: chain narrow -> in_X.3 -> qkv wrong in %d of %d runs\n", mode, badX, iters);
    return 0;
  }
  printf("mode %d: victim (node_linear_x3) wrong in %d of %d runs; aggressor (edge_mlp_in) wrong in %d\n", mode, badV, iters, badA);
  return 0;
}
