// Edge input / output MLPs of the GraphTransformer denoiser on 16-key units (gfx950, fp32 v_mfma_f32_16x16x4_f32).
//
//   mlp_in_E   relu(W2 relu(W1 e + b1) + b2)                                   transformer_model.py:290-297, :358
//              written UNsymmetrised for the pairs of every unit (consumers symmetrise on load, mfma_chain.h)
//   mlp_out_E  W2 relu(W1 sym(E) + b1) + b2 + residual, diagonal mask, symmetrise, mask -> logits
//                                                                              transformer_model.py:372-376, :387-389
//
// Same work decomposition and register layout as edge_unit_layer.hip (unit = valid row x chunk of 16 keys with a valid
// key; key on lane & 15, channel group on lane >> 4, 16 channels per f32x4), but the two small weight matrices of an
// MLP fit LDS whole (<= 48 KB), so a persistent workgroup stages them once and its 8 waves then stream units with no
// barrier at all.  Pairs outside the units are never read downstream (hidden E) or are zero-filled by the caller of
// the launcher (logits).
#include "mfma_chain.h"
#include "kernels.h"
#include "x3_mfma.h"
#include <stdlib.h>

namespace scgg {

namespace {

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// acc += F . x  (16 out x 16 in), dependent chain of four K-steps
__device__ __forceinline__ void mfma4(f32x4& acc, const f32x4& f, const f32x4& x) {
#pragma unroll
  for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f[r], x[r], acc, 0, 0, 0);
}
// two independent chains interleaved (dependent issue is 40 cycles, independent 32)
__device__ __forceinline__ void mfma4x2(f32x4& acc0, f32x4& acc1, const f32x4& f0, const f32x4& f1, const f32x4& x0,
                                        const f32x4& x1) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0[r], x0[r], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1[r], x1[r], acc1, 0, 0, 0);
  }
}

template <int IB, int HB, int OB>
struct MlpGeom {
  static constexpr int F1 = HB * IB * 64, F2 = OB * HB * 64;           // float4s of the two fragment sets
  static constexpr int IMG_F4 = F1 + F2 + HB * 4 + OB * 4;             // + biases (16 floats per block)
};

}  // namespace

// img = [ layer-1 fragments (ob, ib) | layer-2 fragments (ob, ib) | b1 padded | b2 padded ]; fragment (ob, ib), lane l:
// W[16 ob + (l & 15)][16 ib + 4 (l >> 4) + 0..3], zero outside the matrix
template <int IB, int HB, int OB>
__global__ __launch_bounds__(256) void edge_mlp_pack_kernel(EdgeMlpPackArgs a) {
  using G = MlpGeom<IB, HB, OB>;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < G::IMG_F4; idx += gridDim.x * 256) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (idx < G::F1 + G::F2) {
      const bool second = idx >= G::F1;
      const int t = (second ? idx - G::F1 : idx) >> 6, l = idx & 63;
      const int nib = second ? HB : IB;
      const int ob = t / nib, ib = t - ob * nib;
      const float* W = second ? a.W2 : a.W1;
      const int N = second ? a.N2 : a.N1, K = second ? a.N1 : a.K1;
      const int row = 16 * ob + (l & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 16 * ib + 4 * (l >> 4) + e;
        if (row < N && col < K) v[e] = W[(size_t)row * K + col];
      }
    } else {
      const int o = (idx - G::F1 - G::F2) * 4;
      const bool second = o >= HB * 16;
      const float* bsrc = second ? a.b2 : a.b1;
      const int N = second ? a.N2 : a.N1, base = second ? o - HB * 16 : o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (base + e < N) v[e] = bsrc[base + e];
    }
    a.img[idx] = v;
  }
}

template <int IB, int HB, int OB, bool OUT>
__global__ __launch_bounds__(512) void edge_mlp_unit_kernel(EdgeMlpUnitArgs a) {
  using G = MlpGeom<IB, HB, OB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* img = reinterpret_cast<f32x4*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  for (int idx = tid; idx < G::IMG_F4; idx += 512) img[idx] = a.img[idx];
  __syncthreads();
  const f32x4* W1 = img;
  const f32x4* W2 = img + G::F1;
  const float* b1 = reinterpret_cast<const float*>(img + G::F1 + G::F2);
  const float* b2 = b1 + HB * 16;
  const int n = a.n, nch = (n + 15) >> 4;
  const int count = *a.unit_count;
  constexpr int DO = 16 * OB;

  for (int uidx = blockIdx.x * 8 + wave; uidx < count; uidx += gridDim.x * 8) {
    const int u = a.units[uidx];
    const int row = u / nch, jc = u - row * nch;
    const int b = row / n, i = row - b * n;
    const int j = jc * 16 + p;
    const bool inb = j < n;
    const int jj = inb ? j : 0;
    const bool live = inb && a.mask[b * n + jj] != 0 && j != i;
    const long long pij = (long long)row * n + jj;
    const long long pji = ((long long)b * n + jj) * n + i;

    f32x4 x[IB];
    if (OUT) {   // symmetrised hidden E (16 IB channels)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        x[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live) x[ib] = 0.5f * (ldg4(a.src + pij * (16 * IB) + 16 * ib + 4 * g) + ldg4(a.src + pji * (16 * IB) + 16 * ib + 4 * g));
      }
    } else {     // raw input features, row width dE_in (not 16-byte aligned)
      const float* src = a.src + pij * a.dE_in;
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * ib + 4 * g + r;
          x[ib][r] = (live && c < a.dE_in) ? src[c] : 0.f;
        }
    }
    // ---- layer 1 + relu: hidden blocks in pairs (two accumulators, same input block)
    f32x4 hid[HB];
#pragma unroll
    for (int ob = 0; ob < HB; ob += 2) {
      hid[ob] = *reinterpret_cast<const f32x4*>(b1 + 16 * ob + 4 * g);
      hid[ob + 1] = *reinterpret_cast<const f32x4*>(b1 + 16 * (ob + 1) + 4 * g);
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
        mfma4x2(hid[ob], hid[ob + 1], W1[(ob * IB + ib) * 64 + lane], W1[((ob + 1) * IB + ib) * 64 + lane], x[ib], x[ib]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hid[ob][r] = fmaxf(hid[ob][r], 0.f);
        hid[ob + 1][r] = fmaxf(hid[ob + 1][r], 0.f);
      }
    }
    // ---- layer 2
    f32x4 o[OB];
    if (OB >= 2) {
#pragma unroll
      for (int ob = 0; ob + 1 < OB; ob += 2) {
        o[ob] = *reinterpret_cast<const f32x4*>(b2 + 16 * ob + 4 * g);
        o[ob + 1] = *reinterpret_cast<const f32x4*>(b2 + 16 * (ob + 1) + 4 * g);
#pragma unroll
        for (int ib = 0; ib < HB; ++ib)
          mfma4x2(o[ob], o[ob + 1], W2[(ob * HB + ib) * 64 + lane], W2[((ob + 1) * HB + ib) * 64 + lane], hid[ib], hid[ib]);
      }
    } else {       // one output block: two partial sums over the even / odd input blocks
      f32x4 even = *reinterpret_cast<const f32x4*>(b2 + 4 * g), odd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ib = 0; ib < HB; ib += 2)
        mfma4x2(even, odd, W2[ib * 64 + lane], W2[(ib + 1) * 64 + lane], hid[ib], hid[ib + 1]);
      o[0] = even + odd;
    }
    if (OUT) {
      // (mlp + E_in[ij]) and (mlp + E_in[ji]) averaged: the hidden E is symmetric after the sym-load, so the MLP
      // term is bit-identical for (i,j) and (j,i); only the residual needs both orientations.
      if (inb) {
        float* dst = a.dst + pij * a.C;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * ob + 4 * g + r;
            if (c < a.C) {
              float val = 0.f;
              if (live) val = a.E_in == nullptr ? o[ob][r]     // no residual (radius head, pos_transformer_model.py:470-487)
                                                : 0.5f * ((o[ob][r] + a.E_in[pij * a.dE_in + c]) + (o[ob][r] + a.E_in[pji * a.dE_in + c]));
              dst[c] = val;
            }
          }
      }
    } else if (inb) {
      float* dst = a.dst + pij * DO + 4 * g;
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (live) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(o[ob][r], 0.f);
        }
        *reinterpret_cast<f32x4*>(dst + 16 * ob) = v;
      }
    }
  }
}

// =========================================================================================================
// The same MLPs on the bf16 matrix pipe with fp32 accuracy (x3_mfma.h), for even IB (input width a multiple of 32).
// img = [ layer-1 blocks (ob, q) | layer-2 blocks (q, ob) | b1 padded | b2 padded ]; a block = three bf16 terms x 64 lanes
// x 16 bytes of W[16 ob + (l & 15)][32 q + {4g..4g+3, 16+4g..16+4g+3}] (g = l >> 4), zero outside the matrix.  Layer-2
// blocks are chunk-major so that a hidden chunk is split once and used for every output block.
// =========================================================================================================
template <int IB, int HB, int OB>
struct MlpGeomX3 {
  static constexpr int Q1 = IB / 2, Q2 = HB / 2;
  static constexpr int B1 = HB * Q1, B2 = OB * Q2;                      // blocks of the two layers
  static constexpr int IMG_F4 = (B1 + B2) * 192 + HB * 4 + OB * 4;
};

template <int IB, int HB, int OB>
__global__ __launch_bounds__(256) void edge_mlp_pack_x3_kernel(EdgeMlpPackArgs a) {
  using G = MlpGeomX3<IB, HB, OB>;
  u32x4* img = reinterpret_cast<u32x4*>(a.img);
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < G::IMG_F4; idx += gridDim.x * 256) {
    u32x4 out = {0u, 0u, 0u, 0u};
    if (idx < (G::B1 + G::B2) * 192) {
      const int t = idx / 192, term = (idx % 192) >> 6, l = idx & 63, g = l >> 4;
      const bool second = t >= G::B1;
      const int tt = second ? t - G::B1 : t;
      const int ob = second ? tt % OB : tt / G::Q1, q = second ? tt / OB : tt % G::Q1;
      const float* W = second ? a.W2 : a.W1;
      const int N = second ? a.N2 : a.N1, K = second ? a.N1 : a.K1;
      const int row = 16 * ob + (l & 15);
      unsigned hw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int col = 32 * q + (i < 4 ? 4 * g + i : 16 + 4 * g + i - 4);
        float x = (row < N && col < K) ? W[(size_t)row * K + col] : 0.f;
        for (int k = 0; k < term; ++k) x = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
        hw[i] = __float_as_uint(x) >> 16;
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) out[d] = hw[2 * d] | (hw[2 * d + 1] << 16);
    } else {
      const int fo = (idx - (G::B1 + G::B2) * 192) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = fo + e;
        float v = 0.f;
        if (c < HB * 16) v = c < a.N1 ? a.b1[c] : 0.f;
        else if (c - HB * 16 < a.N2) v = a.b2[c - HB * 16];
        out[e] = __float_as_uint(v);
      }
    }
    img[idx] = out;
  }
}

template <int IB, int HB, int OB, bool OUT>
__global__ __launch_bounds__(512) void edge_mlp_unit_x3_kernel(EdgeMlpUnitArgs a) {
  using G = MlpGeomX3<IB, HB, OB>;
  constexpr int Q1 = G::Q1, Q2 = G::Q2;
  static_assert(IB % 2 == 0 && (HB * Q1) % 2 == 0, "x3 MLP: even block counts");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* img = reinterpret_cast<f32x4*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  for (int idx = tid; idx < G::IMG_F4; idx += 512) img[idx] = a.img[idx];
  __syncthreads();
  const f32x4* W1 = img;
  const f32x4* W2 = img + G::B1 * 192;
  const float* b1 = reinterpret_cast<const float*>(img + (G::B1 + G::B2) * 192);
  const float* b2 = b1 + HB * 16;
  const int n = a.n, nch = (n + 15) >> 4;
  const int count = *a.unit_count;
  constexpr int DO = 16 * OB;

  for (int uidx = blockIdx.x * 8 + wave; uidx < count; uidx += gridDim.x * 8) {
    const int u = a.units[uidx];
    const int row = u / nch, jc = u - row * nch;
    const int b = row / n, i = row - b * n;
    const int j = jc * 16 + p;
    const bool inb = j < n;
    const int jj = inb ? j : 0;
    const bool live = inb && a.mask[b * n + jj] != 0 && j != i;
    const long long pij = (long long)row * n + jj;
    const long long pji = ((long long)b * n + jj) * n + i;

    Frag3 fa, fb;
    load_frag3(fa, W1, lane);
    f32x4 x[IB];
    if (OUT) {   // symmetrised hidden E (16 IB channels)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        x[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live) x[ib] = 0.5f * (ldg4(a.src + pij * (16 * IB) + 16 * ib + 4 * g) + ldg4(a.src + pji * (16 * IB) + 16 * ib + 4 * g));
      }
    } else {     // raw input features, row width dE_in (not 16-byte aligned)
      const float* src = a.src + pij * a.dE_in;
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * ib + 4 * g + r;
          x[ib][r] = (live && c < a.dE_in) ? src[c] : 0.f;
        }
    }
    u32x4 xh[Q1], xm[Q1], xl[Q1];
#pragma unroll
    for (int q = 0; q < Q1; ++q) split3(x[2 * q], x[2 * q + 1], xh[q], xm[q], xl[q]);
    // ---- layer 1 + relu
    f32x4 hid[HB];
#pragma unroll
    for (int ob = 0; ob < HB; ++ob) hid[ob] = *reinterpret_cast<const f32x4*>(b1 + 16 * ob + 4 * g);
    block_pipeline<HB * Q1, 0>(W1, W2, fa, fb, lane, [&](int t, const Frag3& w) {
      mfma_x3(hid[t / Q1], w, xh[t % Q1], xm[t % Q1], xl[t % Q1]);
    });
#pragma unroll
    for (int ob = 0; ob < HB; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) hid[ob][r] = fmaxf(hid[ob][r], 0.f);
    // ---- layer 2 (chunk-major blocks: a hidden chunk is split once)
    f32x4 o[OB];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) o[ob] = *reinterpret_cast<const f32x4*>(b2 + 16 * ob + 4 * g);
    {
      u32x4 hh, hm, hl;
      block_pipeline<OB * Q2, 0>(W2, nullptr, fa, fb, lane, [&](int t, const Frag3& w) {
        if (t % OB == 0) split3(hid[2 * (t / OB)], hid[2 * (t / OB) + 1], hh, hm, hl);
        mfma_x3(o[t % OB], w, hh, hm, hl);
      });
    }
    if (OUT) {
      // (mlp + E_in[ij]) and (mlp + E_in[ji]) averaged: the hidden E is symmetric after the sym-load, so the MLP
      // term is bit-identical for (i,j) and (j,i); only the residual needs both orientations.
      if (inb) {
        float* dst = a.dst + pij * a.C;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * ob + 4 * g + r;
            if (c < a.C) {
              float val = 0.f;
              if (live) val = a.E_in == nullptr ? o[ob][r]     // no residual (radius head, pos_transformer_model.py:470-487)
                                                : 0.5f * ((o[ob][r] + a.E_in[pij * a.dE_in + c]) + (o[ob][r] + a.E_in[pji * a.dE_in + c]));
              dst[c] = val;
            }
          }
      }
    } else if (inb) {
      float* dst = a.dst + pij * DO + 4 * g;
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (live) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(o[ob][r], 0.f);
        }
        *reinterpret_cast<f32x4*>(dst + 16 * ob) = v;
      }
    }
  }
}

// SCGG_EDGE_MODE=f32 keeps the fp32-MFMA kernels (development A/B switch, as in edge_unit_layer.hip); odd IB always does
static bool mlp_x3(int ib) {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("SCGG_EDGE_MODE");
    mode = (e != nullptr && e[0] == 'f') ? 0 : 1;
  }
  return mode == 1 && ib % 2 == 0;
}
// The x3 kernels take K in chunks of 32 channels (two 16-channel blocks): an odd block count of the INPUT MLP is rounded
// up, the extra block is zero in the packed weights and in the loaded rows (both are guarded by the true width).
static int in_blocks(int dE_in) {
  const int ib = (dE_in + 15) / 16;
  return (ib % 2 != 0 && mlp_x3(ib + 1)) ? ib + 1 : ib;
}

// ---------------------------------------------------------------------------------------------------------
#define SCGG_MLP_IN_CASES CASE(1, 8, 4) CASE(2, 8, 4) CASE(3, 8, 4) CASE(4, 8, 4) CASE(1, 2, 2) CASE(2, 2, 2)   /* (ceil(in_E/16), hid_E/16, de/16) */
#define SCGG_MLP_OUT_CASES CASE(4, 8, 1) CASE(4, 8, 2) CASE(2, 2, 1)   /* (de/16, hid_E/16, ceil(C/16)) */

static int n_cu_cached() { return device_cu_count(); }

long long edge_mlp_img_f4(int K1, int N1, int N2) {
  const int ib = in_blocks(K1), hb = N1 / 16, ob = (N2 + 15) / 16;
  if (N1 % 32 != 0) return -1;
  if (mlp_x3(ib)) return (long long)(hb * (ib / 2) + ob * (hb / 2)) * 192 + hb * 4 + ob * 4;
  return (long long)hb * ib * 64 + (long long)ob * hb * 64 + hb * 4 + ob * 4;
}

int launch_edge_mlp_pack(const EdgeMlpPackArgs& a, hipStream_t st) {
  const int ib = in_blocks(a.K1), hb = a.N1 / 16, ob = (a.N2 + 15) / 16;
#define CASE(IB, HB, OB)                                                                             \
  if (ib == IB && hb == HB && ob == OB) {                                                            \
    if constexpr (IB % 2 == 0) {                                                                     \
      if (mlp_x3(IB)) {                                                                              \
        hipLaunchKernelGGL((edge_mlp_pack_x3_kernel<IB, HB, OB>), dim3(64), dim3(256), 0, st, a);    \
        return 0;                                                                                    \
      }                                                                                              \
    }                                                                                                \
    hipLaunchKernelGGL((edge_mlp_pack_kernel<IB, HB, OB>), dim3(64), dim3(256), 0, st, a);           \
    return 0;                                                                                        \
  }
  SCGG_MLP_IN_CASES SCGG_MLP_OUT_CASES
#undef CASE
  return -1;
}

template <int IB, int HB, int OB, bool OUT>
static int launch_mlp(const EdgeMlpUnitArgs& a, hipStream_t st) {
  const int n_cu = n_cu_cached();
  if (n_cu <= 0) return -2;
  static LdsAttrOnce attr_f32, attr_x3;
  LdsAttrOnce* attr = &attr_f32;
  void (*kern)(EdgeMlpUnitArgs) = edge_mlp_unit_kernel<IB, HB, OB, OUT>;
  size_t lds = (size_t)MlpGeom<IB, HB, OB>::IMG_F4 * 16;
  if constexpr (IB % 2 == 0) {
    if (mlp_x3(IB)) {
      kern = edge_mlp_unit_x3_kernel<IB, HB, OB, OUT>;
      lds = (size_t)MlpGeomX3<IB, HB, OB>::IMG_F4 * 16;
      attr = &attr_x3;
    }
  }
  if (set_max_dynamic_lds(*attr, reinterpret_cast<const void*>(kern), lds)) return -2;
  hipLaunchKernelGGL(kern, dim3(2 * n_cu), dim3(512), lds, st, a);
  return 0;
}

int launch_edge_mlp_in(const EdgeMlpUnitArgs& a, int dE_in, int hid, int de, hipStream_t st) {
  const int ib = in_blocks(dE_in), hb = hid / 16, ob = de / 16;
  if (hid % 32 != 0 || de % 16 != 0) return -1;
#define CASE(IB, HB, OB) \
  if (ib == IB && hb == HB && ob == OB) return launch_mlp<IB, HB, OB, false>(a, st);
  SCGG_MLP_IN_CASES
#undef CASE
  return -1;
}

int launch_edge_mlp_out(const EdgeMlpUnitArgs& a, int de, int hid, int C, hipStream_t st) {
  const int ib = de / 16, hb = hid / 16, ob = (C + 15) / 16;
  if (hid % 32 != 0 || de % 16 != 0) return -1;
#define CASE(IB, HB, OB) \
  if (ib == IB && hb == HB && ob == OB) return launch_mlp<IB, HB, OB, true>(a, st);
  SCGG_MLP_OUT_CASES
#undef CASE
  return -1;
}

}  // namespace scgg
