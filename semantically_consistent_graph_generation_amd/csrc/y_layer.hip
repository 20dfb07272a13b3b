// Fused XEy layer, global-feature (y) part (gfx950): one launch instead of nine small GEMM / LayerNorm launches.
//
//   new_y = y_y(y) + x_y(Xtoy(X)) + e_y(Etoy(E)); y_out MLP          transformer_model.py:239-246
//   y = norm_y1(y + new_y); y = norm_y2(y + lin_y2(relu(lin_y1(y))))  transformer_model.py:101-106
//   FiLM vectors of the NEXT layer: [y_e_add | y_e_mul+1 | y_x_add | y_x_mul+1](y)   transformer_model.py:186-190, :230
//
// The work is tiny (B rows of width <= 2048) and sits on the path between two edge-layer launches (the next edge
// kernel needs the FiLM vectors), so the goal is latency: one workgroup owns G = 2 graphs (256 workgroups at B = 512),
// keeps their vectors in LDS and walks the chain of dense layers with plain fp32 FMAs.  Weights are TRANSPOSED copies
// ([K][N], made once per weight refresh): a thread owns 4 adjacent outputs, so every weight read is a coalesced
// float4, there is no cross-lane reduction, and the input element is an LDS broadcast.  Narrow layers (N/4 < 256
// threads) split K over the spare threads and combine through LDS.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"
#include "pool_stats.h"
#include "x3_tile.h"

namespace scgg {

#ifdef SCGG_STAMP_Y   // development aid: s_memtime at the phase boundaries of y_layer_kernel (scripts/probe/y_stamps.py)
__device__ unsigned long long scgg_y_stamps[4][16];
#define SCGG_YSTAMP(k) do { if (threadIdx.x == 0 && a.update && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 300 || blockIdx.x == 500)) { \
  unsigned long long* q_ = &scgg_y_stamps[blockIdx.x == 0 ? 0 : blockIdx.x == 100 ? 1 : blockIdx.x == 300 ? 2 : 3][k]; if (*q_ == 0) *q_ = __builtin_readcyclecounter(); } } while (0)
extern "C" int scgg_debug_y_stamps(unsigned long long* out, int reset) {
  static unsigned long long zero[4 * 16] = {};
  if (reset) return hipMemcpyToSymbol(HIP_SYMBOL(scgg_y_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(scgg_y_stamps), sizeof(zero)) == hipSuccess ? 0 : -1;
}
#else
#define SCGG_YSTAMP(k) do {} while (0)
#endif


// YG graphs per workgroup, YT threads per workgroup: template parameters of everything below.  Two instances are built:
// <1, 256> (the position-aware block, whose graph merge is written for one graph per workgroup, and the A/B default of
// rounds 1-2) and <2, 512> (the 2D denoiser: the two graphs are pooled side by side by 256 threads each and share every
// weight float4 of the dense part, which is bound by L2 bandwidth: 1.6 MB of weights per graph; +1.5 % on the chain).
constexpr int YW = 256;        // max width of the y vectors (dy); the FFN hidden width dim_ffy has its own buffer

typedef float yf4 __attribute__((ext_vector_type(4)));


// acc[g] += sum_{k0 <= k < k1} Wt[k][o4] * in[g][k], the weight float4s requested YD at a time (the loop is a chain of L2
// round trips otherwise: ~130 ns per iteration measured, i.e. no overlap at all with the compiler's own unrolling)
constexpr int YD = 16;
template <int YG>
__device__ __forceinline__ void y_dot(yf4 (&acc)[YG], const yf4* __restrict__ w4, int N4, int o4, const float* in, int ldin,
                                      int k0, int k1) {
  int k = k0;
  for (; k + YD <= k1; k += YD) {
    yf4 w[YD];
#pragma unroll
    for (int u = 0; u < YD; ++u) w[u] = w4[(size_t)(k + u) * N4 + o4];
#pragma unroll
    for (int u = 0; u < YD; ++u)
#pragma unroll
      for (int g = 0; g < YG; ++g) acc[g] += w[u] * in[g * ldin + k + u];
  }
  for (; k < k1; ++k) {
    const yf4 w = w4[(size_t)k * N4 + o4];
#pragma unroll
    for (int g = 0; g < YG; ++g) acc[g] += w * in[g * ldin + k];
  }
}

// out[g][o] = act(bias[o] + sum_k Wt[k][o] in[g][k]) for o < N, g < YG.  in / out: LDS [YG][ld]; red: LDS scratch of
// YT * YG float4.  Contains barriers: call from uniform control flow.  On return `out` is complete and synchronised.
template <int YG, int YT>
__device__ __forceinline__ void y_dense(const float* __restrict__ Wt, const float* __restrict__ bias, int N, int K,
                                        const float* in, int ldin, float* out, int ldout, bool relu, yf4* red, int tid) {
  const int N4 = N >> 2;
  const yf4* w4 = reinterpret_cast<const yf4*>(Wt);
  if (N4 >= YT || YT % N4 != 0) {
    for (int o4 = tid; o4 < N4; o4 += YT) {
      yf4 acc[YG];
      const yf4 bv = *reinterpret_cast<const yf4*>(bias + 4 * o4);
#pragma unroll
      for (int g = 0; g < YG; ++g) acc[g] = bv;
      y_dot<YG>(acc, w4, N4, o4, in, ldin, 0, K);
#pragma unroll
      for (int g = 0; g < YG; ++g) {
        yf4 v = acc[g];
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        *reinterpret_cast<yf4*>(out + g * ldout + 4 * o4) = v;
      }
    }
    __syncthreads();
  } else {
    const int S = YT / N4, o4 = tid % N4, sl = tid / N4;
    const int k0 = (int)((long long)K * sl / S), k1 = (int)((long long)K * (sl + 1) / S);
    yf4 acc[YG];
#pragma unroll
    for (int g = 0; g < YG; ++g) acc[g] = yf4{0.f, 0.f, 0.f, 0.f};
    y_dot<YG>(acc, w4, N4, o4, in, ldin, k0, k1);
#pragma unroll
    for (int g = 0; g < YG; ++g) red[g * YT + tid] = acc[g];
    __syncthreads();
    if (sl == 0) {
      const yf4 bv = *reinterpret_cast<const yf4*>(bias + 4 * o4);
#pragma unroll
      for (int g = 0; g < YG; ++g) {
        yf4 v = bv;
        for (int t = 0; t < S; ++t) v += red[g * YT + t * N4 + o4];
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        *reinterpret_cast<yf4*>(out + g * ldout + 4 * o4) = v;
      }
    }
    __syncthreads();
  }
}

// y[g][:] <- LN(y[g][:] + t[g][:]) * gamma + beta, D <= 256; wave w handles graphs w, w+4
template <int YG, int YT>
__device__ __forceinline__ void y_add_layernorm(float* y, const float* t, int ld, int D, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  for (int g = wave; g < YG; g += YT / 64) {
    float v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < D ? y[g * ld + c] + t[g * ld + c] : 0.f;
      s += v[i];
    }
    s = wave_sum64(s);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      const float d = c < D ? v[i] - mean : 0.f;
      q += d * d;
    }
    q = wave_sum64(q);
    const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < D) y[g * ld + c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
  }
}

template <int YG, int YT>
__global__ __launch_bounds__(YT) void y_layer_kernel(YLayerArgs a) {
  // Default wave priority: the y round has ~130 us of slack beside the edge kernel (it must be done before node kernel l
  // waits for the FiLM vectors); at s_setprio(3) it finished no sooner that mattered and cost the edge kernel 10 us per
  // launch (0.280 -> 0.270 ms, 452 -> 454.6 graphs/s)
  __builtin_amdgcn_s_setprio(0);
  extern __shared__ __attribute__((aligned(16))) float ysm[];
  const int tid = threadIdx.x;
  const int dy = a.dy, kcat = a.kcat;
  const int ldy = a.dy;                                          // stride of the y-wide vectors
  const int ldf = a.ffy > ldy ? a.ffy : ldy;
  yf4* red = reinterpret_cast<yf4*>(ysm);                       // [YG][YT] float4
  float* cat = ysm + YG * YT * 4;                                // [YG][kcat]  (y | Xtoy | Etoy)
  float* yv = cat + YG * kcat;                                   // [YG][dy]    the y stream
  float* t1 = yv + YG * ldy;                                      // [YG][dy]
  float* t2 = t1 + YG * ldy;                                      // [YG][ldf]   (y_out hidden, then the lin_y1 hidden)
  const int b0 = blockIdx.x * YG;
  SCGG_YSTAMP(0);

  if (a.update) {
    const int e0 = dy + 4 * a.dx;
    const bool own3 = YG == 1 && a.gm_rowagg != nullptr;
    constexpr bool HALVES = YG > 1 && YT == YG * PS_T;          // every graph of the workgroup is pooled by its own PS_T threads
    const bool own_x = own3 || ((YT == PS_T || HALVES) && a.pool_X != nullptr), own_e = own3 || ((YT == PS_T || HALVES) && a.pool_E != nullptr);
    for (int idx = tid; idx < YG * kcat; idx += YT) {
      const int g = idx / kcat, c = idx - g * kcat;
      const int b = min(b0 + g, a.B - 1);
      if (c < dy) cat[idx] = a.y[(size_t)b * dy + c];
      else if (c < e0) { if (!own_x) cat[idx] = a.ycat[(size_t)b * kcat + c]; }
      else if (!own_e) cat[idx] = a.etoy[(size_t)b * (kcat - e0) + (c - e0)];
    }
    if (own3) {                                                    // position-aware block: the graph merge of this graph
      __shared__ int s_nv3;
      const uint8_t* m = a.mask + (size_t)b0 * a.n;
      if (tid == 0) {
        int c = 0;
        for (int i = 0; i < a.n; ++i) c += m[i] != 0;
        s_nv3 = c;
      }
      __syncthreads();
      pool_graph3d(a.gm_rowagg + (size_t)b0 * a.n * 8 * a.de, a.gm_rowaggD + (size_t)b0 * a.n * 4 * a.de,
                   a.gm_newX + (size_t)b0 * a.n * a.dx, m, a.n, s_nv3, a.dx, a.de, cat + dy, cat + e0, tid, YT);
    } else if (own_x || own_e) {                                   // wave-uniform; the statistics of this graph, straight into cat
      SCGG_YSTAMP(9);
      if (HALVES) {                                                  // the graphs side by side, PS_T threads each (the barriers inside
        __shared__ PoolScratch psh[YG];                              // the pooling functions sit at the same places for every graph)
        const int g = tid / PS_T, lt = tid - g * PS_T;
        const size_t bg = (size_t)min(b0 + g, a.B - 1);
        const int nv = pool_compact_mask(psh[g], a.mask + bg * a.n, a.n, lt);
        if (own_x) pool_xtoy_graph(psh[g], a.pool_X + bg * a.n * a.dx, nv, a.dx, cat + g * kcat + dy, lt);
        if (own_e) pool_etoy_graph(psh[g], a.pool_E + bg * a.n * a.n * a.de, a.n, nv, a.de, cat + g * kcat + e0, lt);
      } else {
        __shared__ PoolScratch ps;
        for (int g = 0; g < YG; ++g) {                               // the workgroup's graphs in turn
          const size_t bg = (size_t)min(b0 + g, a.B - 1);
          const int nv = pool_compact_mask(ps, a.mask + bg * a.n, a.n, tid);
          if (own_x) pool_xtoy_graph(ps, a.pool_X + bg * a.n * a.dx, nv, a.dx, cat + g * kcat + dy, tid);
          if (own_e) pool_etoy_graph(ps, a.pool_E + bg * a.n * a.n * a.de, a.n, nv, a.de, cat + g * kcat + e0, tid);
        }
      }
      SCGG_YSTAMP(12);
    }
  }
  for (int idx = tid; idx < YG * dy; idx += YT) {
    const int g = idx / dy, c = idx - g * dy;
    yv[g * ldy + c] = a.y[(size_t)min(b0 + g, a.B - 1) * dy + c];
  }
  __syncthreads();
  SCGG_YSTAMP(1);
  if (a.update) {
    y_dense<YG, YT>(a.Wt_cat, a.b_cat, dy, kcat, cat, kcat, t1, ldy, false, red, tid);        // y_y + x_y + e_y
    SCGG_YSTAMP(2);
    y_dense<YG, YT>(a.Wt_o0, a.b_o0, dy, dy, t1, ldy, t2, ldf, true, red, tid);               // y_out.0 + relu
    y_dense<YG, YT>(a.Wt_o2, a.b_o2, dy, dy, t2, ldf, t1, ldy, false, red, tid);              // y_out.2
    SCGG_YSTAMP(3);
    y_add_layernorm<YG, YT>(yv, t1, ldy, dy, a.ln1_g, a.ln1_b, tid);
    __syncthreads();
    SCGG_YSTAMP(4);
    y_dense<YG, YT>(a.Wt_l1, a.b_l1, a.ffy, dy, yv, ldy, t2, ldf, true, red, tid);            // lin_y1 + relu
    SCGG_YSTAMP(5);
    y_dense<YG, YT>(a.Wt_l2, a.b_l2, dy, a.ffy, t2, ldf, t1, ldy, false, red, tid);           // lin_y2
    SCGG_YSTAMP(6);
    y_add_layernorm<YG, YT>(yv, t1, ldy, dy, a.ln2_g, a.ln2_b, tid);
    __syncthreads();
    for (int idx = tid; idx < YG * dy; idx += YT) {
      const int g = idx / dy, c = idx - g * dy;
      if (b0 + g < a.B) a.y[(size_t)(b0 + g) * dy + c] = yv[g * ldy + c];
    }
  }
  SCGG_YSTAMP(7);
  if (a.Wt_film != nullptr) {
    // FiLM vectors for the next edge / node kernels.  film_n == 0: the 2D block's four dx-wide vectors with +1 folded into
    // the two "mul" quarters; film_n > 0: film_n raw outputs (the position-aware block: y_e_add | y_e_mul | y_x_add | y_x_mul
    // of widths de, de, dx, dx; its kernels add the 1 themselves)
    const int FN = a.film_n > 0 ? a.film_n : 4 * a.dx;
    const int N4 = FN >> 2;
    const yf4* w4 = reinterpret_cast<const yf4*>(a.Wt_film);
    for (int o4 = tid; o4 < N4; o4 += YT) {
      yf4 acc[YG];
      yf4 bv = *reinterpret_cast<const yf4*>(a.b_film + 4 * o4);
      const int quarter = (4 * o4) / a.dx;
      if (a.film_n == 0 && (quarter == 1 || quarter == 3)) bv += 1.0f;
#pragma unroll
      for (int g = 0; g < YG; ++g) acc[g] = bv;
      y_dot<YG>(acc, w4, N4, o4, yv, ldy, 0, dy);
#pragma unroll
      for (int g = 0; g < YG; ++g)
        if (b0 + g < a.B) *reinterpret_cast<yf4*>(a.film + (size_t)(b0 + g) * FN + 4 * o4) = acc[g];
    }
  }
  SCGG_YSTAMP(8);
}

static size_t y_layer_lds(int YG, int YT, int dy, int ffy, int kcat) {
  return (size_t)(YG * YT * 4 + YG * kcat + 2 * YG * dy + YG * (ffy > dy ? ffy : dy)) * 4;
}

bool y_layer_supported(int dy, int ffy, int kcat) {
  return dy % 4 == 0 && ffy % 4 == 0 && kcat % 4 == 0 && dy <= YW && y_layer_lds(1, 256, dy, ffy, kcat) <= 96 * 1024;
}

template <int YG, int YT>
static int launch_y_instance(const YLayerArgs& a, hipStream_t st) {
  const size_t lds = y_layer_lds(YG, YT, a.dy, a.ffy, a.kcat);
  static size_t attr_lds[MAX_DEVICES] = {};   // per device: the largest size asked for so far (engines of different widths share the kernel)
  const int dev = current_device();
  if (dev < 0) return -2;
  if (lds > attr_lds[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(y_layer_kernel<YG, YT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return -2;
    attr_lds[dev] = lds;
  }
  hipLaunchKernelGGL((y_layer_kernel<YG, YT>), dim3((a.B + YG - 1) / YG), dim3(YT), lds, st, a);
  return 0;
}

int launch_y_layer(const YLayerArgs& a, hipStream_t st) {
  if (!y_layer_supported(a.dy, a.ffy, a.kcat)) return -1;
  if (a.gm_rowagg != nullptr && (a.gm_rowaggD == nullptr || a.gm_newX == nullptr || a.mask == nullptr ||
                                 a.kcat != a.dy + 4 * a.dx + 8 * a.de)) return -1;
  if (a.pool_X != nullptr && (a.dx % 4 != 0 || a.dx > 512 || PS_T % (a.dx / 4) != 0 || a.n > 512 || a.mask == nullptr)) return -1;
  if (a.pool_E != nullptr && (a.de % 4 != 0 || a.de > 256 || PS_T % (a.de / 4) != 0 || a.n > 512 || a.mask == nullptr ||
                              a.kcat != a.dy + 4 * a.dx + 4 * a.de)) return -1;
  // two graphs per workgroup when the engine asks for it (the 2D denoiser: every y launch, whatever the schedule switch);
  // one per workgroup for the position-aware block's graph merge.  SCGG_Y_PAIR=0: always one (A/B).
  static const bool pair_on = [] { const char* e = getenv("SCGG_Y_PAIR"); return !(e != nullptr && e[0] == '0'); }();
  if (pair_on && a.pair != 0 && a.gm_rowagg == nullptr && a.B > 1 && y_layer_lds(2, 512, a.dy, a.ffy, a.kcat) <= 64 * 1024)
    return launch_y_instance<2, 512>(a, st);
  return launch_y_instance<1, 256>(a, st);
}

}  // namespace scgg
