// Development probe: issue cost (cycles per instruction, one wave on a SIMD) of the VALU forms the softmax part of the edge kernel uses.
//   hipcc -O3 --offload-arch=gfx950 valu_rate_probe.hip -o valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void rate_kernel(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      REP16(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                         "v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 1) {
#define D(op) "v_add_f32_dpp %0, %0, %0 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
              "v_add_f32_dpp %2, %2, %2 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
              "v_add_f32_dpp %4, %4, %4 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
              "v_add_f32_dpp %6, %6, %6 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1"
      REP16(asm volatile(D("quad_perm:[1,0,3,2]") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 2) {
      REP16(asm volatile(D("row_mirror") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 3) {
      REP16(asm volatile(D("row_half_mirror") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 4) {
      REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 5) {
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n"
                         "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 6) {
      REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
    } else if (MODE == 7) {   // dependent chain of plain adds
      REP16(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n"
                         "v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0"
                         : "+v"(a0));)
    } else if (MODE == 8) {   // v_mov_b32_dpp
#undef D
#define D(op) "v_mov_b32_dpp %0, %1 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
              "v_mov_b32_dpp %2, %3 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
              "v_mov_b32_dpp %4, %5 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %6 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
              "v_mov_b32_dpp %6, %7 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %0 " op " row_mask:0xf bank_mask:0xf bound_ctrl:1"
      REP16(asm volatile(D("row_mirror") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
  const int iters = 200;
  rate_kernel<MODE><<<1, threads>>>(out, cyc, iters);
  rate_kernel<MODE><<<1, threads>>>(out, cyc, iters);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s waves/SIMD=%d  %.2f ticks/instr/wave\n", name, threads / 256, (double)c / (iters * 128.0));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int t : {256, 512}) {
    run<0>("v_add_f32 (indep x8)", t);
    run<7>("v_add_f32 (dependent)", t);
    run<5>("v_fma_f32 (chained ring)", t);
    run<1>("v_add_f32_dpp quad_perm", t);
    run<2>("v_add_f32_dpp row_mirror", t);
    run<3>("v_add_f32_dpp row_half_mirror", t);
    run<8>("v_mov_b32_dpp row_mirror", t);
    run<4>("v_exp_f32", t);
    run<6>("cndmask/max3", t);
  }
  return 0;
}
