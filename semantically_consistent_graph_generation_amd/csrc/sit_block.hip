// One SiT block between two attention launches, fused (gfx950, bf16x3 tiles of x3_tile.h):
//
//   h  += gate_msa (.) proj(attention)                                      sit_model.py:129-130
//   u   = LN(h) (1 + scale_mlp) + shift_mlp ; f = GELU_tanh(fc1(u))          sit_model.py:131, timm Mlp
//   h  += gate_mlp (.) fc2(f)                                                sit_model.py:131-132
//   u'  = LN(h) (1 + scale_msa') + shift_msa' ; q|k|v' = qkv'(u')            the NEXT block's entry, sit_model.py:129, :87
//
// One 8-wave workgroup per tile of 48 valid rows.  The residual stream of a row stays in the registers of the wave that
// owns the row (lane = 4 channels); Linear operands are bf16 term planes in LDS.  The hidden layer (4 H wide) never exists
// whole: fc1 is taken in slices of 256 output channels (8 block pairs = one per wave), each slice goes through GELU into
// the planes of region 2 and at once through its K slice of fc2 onto accumulators that stay in registers.
// Regions: R1 = planes [48][H + 8] (attention rows, then u) and finally the fp32 fc2 result; R2 = the fp32 proj result,
// then planes [48][256 + 8] (hidden slice, then u').  The last block has no next q|k|v: it leaves h (after the attention
// residual) and the MLP output for sit_update_kernel, which applies gate_mlp itself.  During the MLP the residual rows
// wait in global memory (written and read back by the same lanes).
#include "kernels.h"
#include "x3_tile.h"
#include <stdlib.h>

namespace scgg {

template <int HB>
struct SitBlockGeom {
  static constexpr int H = 32 * HB, TR = 48, NRG = 3, RPW = TR / 8;
  static constexpr int KS1 = H + 8, PL1 = TR * KS1;             // planes of an H-wide operand
  static constexpr int KS2 = 256 + 8, PL2 = TR * KS2;           // planes of a 256-wide operand (H <= 256 fits too)
  static constexpr int LD = H + 4;                              // fp32 result rows
  static constexpr size_t F_BYTES = (size_t)TR * LD * 4;
  static constexpr size_t R1 = (size_t)3 * PL1 * 2 > F_BYTES ? (size_t)3 * PL1 * 2 : F_BYTES;
  static constexpr size_t R2 = (size_t)3 * PL2 * 2 > F_BYTES ? (size_t)3 * PL2 * 2 : F_BYTES;
  static constexpr size_t LDS_BYTES = R1 + R2;
};

// GELU, tanh form (timm Mlp's act_layer in sit_model.py:110): 0.5 x (1 + tanh(z)) = x / (1 + exp(-2 z))
__device__ __forceinline__ f32x4 gelu_tanh4(const f32x4& x) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float z = 0.7978845608028654f * (x[e] + 0.044715f * x[e] * x[e] * x[e]);
    o[e] = x[e] / (1.0f + __expf(-2.0f * z));
  }
  return o;
}

template <int HB>
__global__ __launch_bounds__(512) void sit_block_x3_kernel(SitBlockArgs a) {
  using G = SitBlockGeom<HB>;
  constexpr int H = G::H, TR = G::TR, NRG = G::NRG, RPW = G::RPW, KS1 = G::KS1, PL1 = G::PL1, KS2 = G::KS2, PL2 = G::PL2, LD = G::LD;
  constexpr int KCF = 8;                                        // K chunks of an fc2 slice
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* S1 = reinterpret_cast<__bf16*>(smem);
  __bf16* S2 = reinterpret_cast<__bf16*>(smem + G::R1);
  float* F1 = reinterpret_cast<float*>(smem);
  float* F2 = reinterpret_cast<float*>(smem + G::R1);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int count = *a.row_count;
  const int tile = blockIdx.x;
  if (tile * TR >= count) return;
  const bool has_next = a.W3_qkv != nullptr;
  const bool mm = wave < HB;                                    // this wave owns a block pair of the H-wide outputs
  auto fc1_pair = [&](int s) { return w3_block(a.W3_fc1, 2 * (8 * s + wave), HB, lane); };
  auto fc2_pair = [&](int s) { return w3_block(a.W3_fc2 + (size_t)s * a.fc2_stride, 2 * wave, KCF, lane); };
  NWRing ring;
  ring.next = nullptr;
  if (mm) w3_prime<HB>(ring, w3_block(a.W3_proj, 2 * wave, HB, lane));
  else w3_prime<HB>(ring, fc1_pair(0));

  // ---- this wave's rows: attention output -> planes(R1), residual stream -> registers
  const int c = lane * 4;
  const bool in = c < H;
  f32x4 res[RPW];
  // row index / table offset of this wave's row t: recomputed where needed (two cached loads) rather than held in 18 registers
  auto row_ok = [&](int t) { return tile * TR + wave * RPW + t < count; };
  auto row_of = [&](int t) { const int ridx = tile * TR + wave * RPW + t; return a.rows[ridx < count ? ridx : count - 1]; };
  auto tab_of = [&](int row) { return (a.t ? a.t[row / a.n] : a.t_const) * (6 * H); };
#pragma unroll
  for (int t = 0; t < RPW; ++t) {
    res[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (in) {
      const size_t row = row_of(t);
      put_split4<KS1, PL1>(S1, wave * RPW + t, c, *reinterpret_cast<const f32x4*>(a.att + row * H + c));
      res[t] = *reinterpret_cast<const f32x4*>(a.h + row * H + c);
    }
  }
  __syncthreads();

  auto bias_acc = [&](f32x4 (&acc)[2][NRG], const float* bias, int ob0) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * (ob0 + blk) + 4 * g);
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) acc[blk][rg] = bv;
    }
  };
  auto store_f = [&](float* F, const f32x4 (&acc)[2][NRG]) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg)
        *reinterpret_cast<f32x4*>(F + (16 * rg + p) * LD + 16 * (2 * wave + blk) + 4 * g) = acc[blk][rg];
  };
  // rows of this wave: res += gate (.) F ; LayerNorm (eps 1e-6, no affine) ; (1 + scale), + shift -> planes
  auto row_ln = [&](const float* F, const float* tab_gate, int off_gate, const float* tab, int off_shift, int off_scale,
                    auto put) {
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int r = wave * RPW + t, toff = tab_of(row_of(t));
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (in) v = res[t] + *reinterpret_cast<const f32x4*>(tab_gate + toff + off_gate + c) * *reinterpret_cast<const f32x4*>(F + r * LD + c);
      res[t] = v;
      const float mean = wave_sum64((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / H);
      f32x4 d = v - mean;
      if (!in) d = f32x4{0.f, 0.f, 0.f, 0.f};
      const float rstd = 1.0f / sqrtf(wave_sum64((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / H) + 1e-6f);
      if (in) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + toff + off_scale + c),
                    sh = *reinterpret_cast<const f32x4*>(tab + toff + off_shift + c);
        put(r, d * rstd * (sc + 1.0f) + sh);
      }
    }
  };

  // ---- proj: planes(R1) -> fp32(R2)
  if (mm) {
    f32x4 acc[2][NRG];
    bias_acc(acc, a.b_proj, 2 * wave);
    tile_linear_x3<HB, HB, KS1, PL1, NRG>(acc, ring, w3_block(a.W3_proj, 2 * wave, HB, lane), fc1_pair(0), S1, p, g);
    store_f(F2, acc);
  }
  __syncthreads();
  // ---- h += gate_msa (.) proj ; u -> planes(R1)
  row_ln(F2, a.tab, 2 * H, a.tab, 3 * H, 4 * H, [&](int r, const f32x4& u) { put_split4<KS1, PL1>(S1, r, c, u); });
  // the residual stream waits in global memory during the MLP (its 24 registers are what the MLP phase is short of)
#pragma unroll
  for (int t = 0; t < RPW; ++t)
    if (in && row_ok(t)) *reinterpret_cast<f32x4*>(a.h + (size_t)row_of(t) * H + c) = res[t];
  __syncthreads();

  // ---- MLP in slices of 256 hidden channels
  f32x4 acc2[2][NRG];
  if (mm) bias_acc(acc2, a.b_fc2, 2 * wave);
  const nu32x4* w_tail0 = has_next ? w3_block(a.W3_qkv, 2 * wave, HB, lane) : nullptr;
  for (int s = 0; s < a.nslices; ++s) {
    {
      f32x4 acc[2][NRG];
      bias_acc(acc, a.b_fc1, 2 * (8 * s + wave));
      if (mm) tile_linear_x3<HB, KCF, KS1, PL1, NRG>(acc, ring, fc1_pair(s), fc2_pair(s), S1, p, g);
      else tile_linear_x3<HB, HB, KS1, PL1, NRG>(acc, ring, fc1_pair(s), s + 1 < a.nslices ? fc1_pair(s + 1) : w_tail0, S1, p, g);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg)
          put_split4<KS2, PL2>(S2, 16 * rg + p, 16 * (2 * wave + blk) + 4 * g, gelu_tanh4(acc[blk][rg]));
    }
    __syncthreads();
    if (mm) tile_linear_x3<KCF, HB, KS2, PL2, NRG>(acc2, ring, fc2_pair(s), s + 1 < a.nslices ? fc1_pair(s + 1) : w_tail0, S2, p, g);
    __syncthreads();
  }
  // ---- fc2 result -> fp32(R1): the planes of R1 were last read by the final fc1 slice, two barriers ago
  if (mm) store_f(F1, acc2);
  __syncthreads();
  if (!has_next) {
#pragma unroll
    for (int t = 0; t < RPW; ++t)
      if (in && row_ok(t))
        *reinterpret_cast<f32x4*>(a.o + (size_t)row_of(t) * H + c) = *reinterpret_cast<const f32x4*>(F1 + (wave * RPW + t) * LD + c);
    return;
  }
#pragma unroll
  for (int t = 0; t < RPW; ++t)
    if (in) res[t] = *reinterpret_cast<const f32x4*>(a.h + (size_t)row_of(t) * H + c);   // this wave's own store, same lanes
  // ---- h += gate_mlp (.) fc2 -> global ; the next block's u' -> planes(R2)
  row_ln(F1, a.tab, 5 * H, a.tab_next, 0, H, [&](int r, const f32x4& u) { put_split4<KS2, PL2>(S2, r, c, u); });
#pragma unroll
  for (int t = 0; t < RPW; ++t)
    if (in && row_ok(t)) *reinterpret_cast<f32x4*>(a.h + (size_t)row_of(t) * H + c) = res[t];
  __syncthreads();
  // ---- the next block's q | k | v: planes(R2) -> global
  {
    long long qrow[NRG];
    bool qok[NRG];
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg) {
      const int ridx = tile * TR + 16 * rg + p;
      qok[rg] = ridx < count;
      qrow[rg] = a.rows[qok[rg] ? ridx : count - 1];
    }
    for (int pr = wave; pr < 3 * HB; pr += 8) {
      f32x4 acc[2][NRG];
      bias_acc(acc, a.b_qkv, 2 * pr);
      const nu32x4* wn = pr + 8 < 3 * HB ? w3_block(a.W3_qkv, 2 * (pr + 8), HB, lane) : nullptr;
      tile_linear_x3<HB, HB, KS2, PL2, NRG>(acc, ring, w3_block(a.W3_qkv, 2 * pr, HB, lane), wn, S2, p, g);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg)
          if (qok[rg]) *reinterpret_cast<f32x4*>(a.qkv + qrow[rg] * (3 * H) + 16 * (2 * pr + blk) + 4 * g) = acc[blk][rg];
    }
  }
}

// -1: shape not built (then the caller runs the separate kernels)
int launch_sit_block(const SitBlockArgs& a, int H, int M, hipStream_t st) {
  if (H != 192 || a.nslices < 1) return -1;
  using G = SitBlockGeom<6>;
  static LdsAttrOnce attr;
  auto kern = sit_block_x3_kernel<6>;
  if (set_max_dynamic_lds(attr, reinterpret_cast<const void*>(kern), G::LDS_BYTES)) return -2;
  hipLaunchKernelGGL(kern, dim3((M + G::TR - 1) / G::TR), dim3(512), G::LDS_BYTES, st, a);
  return 0;
}

}  // namespace scgg
