// Fused XEy layer, edge part (gfx950, fp32 on v_mfma_f32_16x16x4_f32) -- persistent workgroups over 16-key units.
//
//   NodeEdgeBlock (E part + attention)  transformer_model.py:165-227
//   XEyTransformerLayer E path          transformer_model.py:94-99, masks :110-112 (utils.py:154)
//
// Work unit = (valid query row (b,i), chunk of 16 keys holding at least one valid key).  One wave owns one unit at a
// time: key j on lane & 15, channel group g = lane >> 4, a 16-channel block of activations is one f32x4 whose register
// r holds channel 4g + r ("C16 layout": exactly how v_mfma_f32_16x16x4_f32 leaves its 16x16 result).  A Linear is
// evaluated transposed, out^T = W . in^T, with W as the A operand, so register r of an input block is the B operand of
// K-step r (channels {r, 4+r, 8+r, 12+r}) and a whole Linear -> FiLM -> Linear -> LN -> FFN chain stays in registers.
//
// Why 16-key units and not 32-key rows (the previous edge_layer_kernel): per-wave state halves (about 110 VGPRs), so
// 4 waves fit a SIMD instead of 2 and the scheduler overlaps one wave's VALU / LDS / memory latency with another's
// MFMAs; the key reductions of the attention softmax and of the Etoy pooling are 4 DPP steps inside a 16-lane row (no
// LDS transpose, no cross-row shuffles); graphs are padded to multiples of 16 keys instead of 32.
//
// The layer's 256 KB of weights are cut into NPH phase images (edge_unit_pack_kernel, once per weight refresh):
//   cb = 0..dx/32-1 : e_mul / e_add rows of channel block cb, the matching K-slice of e_out   (96 MFMAs per wave)
//   FFN1 x ffE/64   : four row blocks of linE1                                                (64)
//   FFN2 x de/32    : two row blocks of linE2                                                 (64)
// copied global -> LDS by LDS-DMA into a double buffer while the previous phase computes; one __syncthreads per phase.
//
// Attention: per (row, channel) softmax over the keys with one shift M per (unit, 8 channels of a lane row); a unit
// writes per channel [max | sum p | sum p V] and attn_finish_kernel merges the units of a row (exact rescaling) into
// newX_pre.  If a channel underflows against M the unit falls back to exact per-channel maxima.
#include <stdlib.h>

#include "pool_stats.h"
#include "mfma_chain.h"
#include "c16.h"
#include "kernels.h"
#include "x3_mfma.h"

namespace scgg {

template <int NXB, int NEB, int NFB>
struct UnitGeom {
  static constexpr int DX = 32 * NXB, DE = 32 * NEB, FFE = 32 * NFB;
  static constexpr int IBE = DE / 16, IBF = FFE / 16;      // 16-channel blocks of the E width / FFN width
  static constexpr int W_CB = 6 * IBE * 64;                // float4s of a channel-block phase (e_mul, e_add, e_out slices)
  static constexpr int W_F1 = 4 * IBE * 64;
  static constexpr int W_F2 = 2 * IBF * 64;
  static constexpr int WMAX = W_CB > W_F1 ? (W_CB > W_F2 ? W_CB : W_F2) : (W_F1 > W_F2 ? W_F1 : W_F2);
  static constexpr int PH_F4 = WMAX + 64;                  // + bias area (16 float4 used), whole wave instructions
  static constexpr int NPH = NXB + NFB / 2 + NEB;
  static constexpr size_t lds_bytes(int nw) { return (size_t)2 * PH_F4 * 16 + (size_t)nw * 3 * DX * 4; }
};

// ---------------------------------------------------------------------------------------------------------
// Work lists, rebuilt per forward from the node mask (two short launches):
//   rows[k]  = flat (b*n+i) of the k-th valid node                      (*row_count of them)
//   units[k] = row * nch + jc for every valid row and every 16-key chunk jc of its graph that holds a valid key
//              (*unit_count of them; rows in order, so the units of a row and of a graph are contiguous)
// graph_scan_kernel (one block) counts per graph and scans; graph_fill_kernel (one wave per graph) writes the slices.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void graph_scan_kernel(const uint8_t* mask, int B, int n, int nch, int* goff,
                                                          int* row_count, int* unit_count) {
  __shared__ int wr[16], wu[16];
  __shared__ int base_r, base_u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { base_r = 0; base_u = 0; }
  __syncthreads();
  for (int start = 0; start < B; start += 1024) {
    const int b = start + tid;
    int nv = 0, nu = 0;
    if (b < B) {
      const uint8_t* mb = mask + (size_t)b * n;
      int chunks = 0;
      for (int jc = 0; jc < nch; ++jc) {
        int cnt = 0;
        for (int j = jc * 16; j < min(n, jc * 16 + 16); ++j) cnt += mb[j] != 0;
        nv += cnt;
        chunks += cnt > 0;
      }
      nu = nv * chunks;
    }
    int ir = nv, iu = nu;   // inclusive scans inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int tr = __shfl_up(ir, d), tu = __shfl_up(iu, d);
      if (lane >= d) { ir += tr; iu += tu; }
    }
    if (lane == 63) { wr[wave] = ir; wu[wave] = iu; }
    __syncthreads();
    int orow = base_r + ir - nv, ounit = base_u + iu - nu;
    for (int w = 0; w < wave; ++w) { orow += wr[w]; ounit += wu[w]; }
    if (b < B) { goff[2 * b] = orow; goff[2 * b + 1] = ounit; }
    __syncthreads();
    if (tid == 0) {
      int tr = 0, tu = 0;
      for (int w = 0; w < 16; ++w) { tr += wr[w]; tu += wu[w]; }
      base_r += tr; base_u += tu;
    }
    __syncthreads();
  }
  if (tid == 0) { *row_count = base_r; *unit_count = base_u; }
}

__global__ __launch_bounds__(64) void graph_fill_kernel(const uint8_t* mask, int n, int nch, const int* goff, int* rows,
                                                        int* units, unsigned* chunk_bits) {
  __shared__ int chunk_rank[32];
  __shared__ int s_nvc;
  const int b = blockIdx.x, lane = threadIdx.x;
  const uint8_t* mb = mask + (size_t)b * n;
  if (lane == 0) {
    int k = 0;
    unsigned bits = 0;
    for (int jc = 0; jc < nch; ++jc) {
      bool any = false;
      for (int j = jc * 16; j < min(n, jc * 16 + 16); ++j) any |= mb[j] != 0;
      chunk_rank[jc] = any ? k++ : -1;
      bits |= any ? (1u << jc) : 0u;
    }
    s_nvc = k;
    if (chunk_bits != nullptr) chunk_bits[b] = bits;          // which 16-key chunks of the graph hold a valid key (nch <= 32)
  }
  __syncthreads();
  const int nvc = s_nvc;
  int done = 0;   // valid nodes before this 64-node slab
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    const bool v = i < n && mb[i] != 0;
    const unsigned long long bal = __ballot(v);
    if (v) {
      const int rank = done + __popcll(bal & ((1ull << lane) - 1ull));
      const int row = b * n + i;
      rows[goff[2 * b] + rank] = row;
      int* u = units + goff[2 * b + 1] + rank * nvc;
      for (int jc = 0; jc < nch; ++jc)
        if (chunk_rank[jc] >= 0) u[chunk_rank[jc]] = row * nch + jc;
    }
    done += __popcll(bal);
  }
}

int launch_build_lists(const uint8_t* mask, int B, int n, int* goff, int* rows, int* row_count, int* units,
                       int* unit_count, hipStream_t st, unsigned* chunk_bits) {
  const int nch = (n + 15) / 16;
  if (nch > 32) return -1;
  hipLaunchKernelGGL(graph_scan_kernel, dim3(1), dim3(1024), 0, st, mask, B, n, nch, goff, row_count, unit_count);
  hipLaunchKernelGGL(graph_fill_kernel, dim3(B), dim3(64), 0, st, mask, n, nch, goff, rows, units, chunk_bits);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Phase images from the raw (torch layout) weights.  Fragment t of a phase is 64 float4: lane l holds
// W[row0 + (l & 15)][col0 + 4 (l >> 4) + 0..3]; fragments are stored in the order the kernel consumes them:
// (fragments come in PAIRS that share the input block and feed two different output blocks, see mfma4x2)
//   cb phase : e_mul: for ib: rows 32cb+16ob, ob = 0,1 | same for e_add | e_out: for ob pair, for ib = 0,1 (cols 32cb+16ib):
//              rows 16ob of the pair
//   FFN1 f   : linE1: for ob pair (of 0..3), for ib: rows 64f+16ob of the pair
//   FFN2 rb  : linE2: for ib: rows 32rb+16ob, ob = 0,1
// bias area (floats, at float4 slot WMAX): cb: (e_mul.bias+1)[32cb..] | e_add.bias[32cb..]; FFN1: linE1.bias[64f..];
// FFN2: linE2.bias[32rb..].
// ---------------------------------------------------------------------------------------------------------
template <int NXB, int NEB, int NFB>
__global__ __launch_bounds__(512) void edge_unit_pack_kernel(EdgeUnitPackArgs a) {
  using G = UnitGeom<NXB, NEB, NFB>;
  constexpr int DX = G::DX, DE = G::DE, FFE = G::FFE, IBE = G::IBE, IBF = G::IBF;
  const int ph = blockIdx.x;
  for (int idx = threadIdx.x; idx < G::PH_F4; idx += blockDim.x) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int t = idx >> 6, l = idx & 63;
    const float* W = nullptr;
    int row0 = 0, col0 = 0, K = 0;
    if (ph < NXB) {
      const int cb = ph;
      if (t < 4 * IBE) {
        W = t < 2 * IBE ? a.W_mul : a.W_add;
        const int rem = t % (2 * IBE);
        row0 = 32 * cb + 16 * (rem & 1); col0 = 16 * (rem >> 1); K = DE;
      } else if (t < 6 * IBE) {
        const int u = t - 4 * IBE;
        W = a.W_out; row0 = 16 * (2 * (u >> 2) + (u & 1)); col0 = 32 * cb + 16 * ((u >> 1) & 1); K = DX;
      } else if (idx >= G::WMAX && idx < G::WMAX + 16) {
        const int f = (idx - G::WMAX) * 4;   // float offset in the bias area
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f < 32 ? a.b_mul[32 * cb + f + e] + 1.0f : a.b_add[32 * cb + f - 32 + e];
      }
    } else if (ph < NXB + NFB / 2) {
      const int f = ph - NXB;
      if (t < 4 * IBE) {
        W = a.W_ff1; row0 = 64 * f + 16 * (2 * (t / (2 * IBE)) + (t & 1)); col0 = 16 * ((t >> 1) % IBE); K = DE;
      } else if (idx >= G::WMAX && idx < G::WMAX + 16) {
        const int o = (idx - G::WMAX) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a.b_ff1[64 * f + o + e];
      }
    } else {
      const int rb = ph - NXB - NFB / 2;
      if (t < 2 * IBF) {
        W = a.W_ff2; row0 = 32 * rb + 16 * (t & 1); col0 = 16 * (t >> 1); K = FFE;
      } else if (idx >= G::WMAX && idx < G::WMAX + 8) {
        const int o = (idx - G::WMAX) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a.b_ff2[32 * rb + o + e];
      }
    }
    if (W != nullptr) v = ld4(W + (size_t)(row0 + (l & 15)) * K + col0 + 4 * (l >> 4));
    a.img[(size_t)ph * G::PH_F4 + idx] = v;
  }
}

// LDS-DMA copy of one phase image into buf: wave w issues the 64-float4 pieces w, w+NW, ... (each 1 KiB, lane-linear).
// The global address is (uniform base of the piece row) + (one 32-bit lane offset): the saddr form of the instruction,
// so that no per-phase 64-bit address vectors are kept (hoisted out of the unit loop they cost ~30 VGPRs and spilled, and a
// scratch reload in front of a DMA instruction waits for vmcnt(0), i.e. for every DMA and prefetch in flight).
template <int PH_F4, int NW>
__device__ __forceinline__ void stage_phase_unit(const f32x4* __restrict__ src, f32x4* buf, int wave, int lane) {
  static_assert(PH_F4 % 64 == 0, "phase image must be whole wave instructions");
#ifdef SCGG_ABL_NO_DMA
  return;
#endif
  unsigned voff = (unsigned)(wave * 64 + lane) * 16u;
  asm volatile("" : "+v"(voff));   // opaque per call: keeps (uniform base) + zext(lane offset) from being re-associated
#pragma unroll
  for (int s = 0; s < (PH_F4 / 64 + NW - 1) / NW; ++s) {
    const int piece = wave + NW * s;                                  // wave-uniform
    typedef const __attribute__((address_space(1))) char* gptr;
    const gptr base = (gptr)(reinterpret_cast<const char*>(src)) + (size_t)s * NW * 1024;   // uniform
    if (piece < PH_F4 / 64)
      __builtin_amdgcn_global_load_lds(base + voff, (__attribute__((address_space(3))) void*)(buf + piece * 64), 16, 0, 0);
  }
}

__device__ __forceinline__ void load_frags2(f32x4 (&f)[2], const f32x4* w, int lane) {
  f[0] = w[lane];
  f[1] = w[64 + lane];
}

// NG (even) groups of two fragments starting at W; on entry fa holds group 0, on exit fa holds the group at `after`
// (when not null).  body(g, f0, f1) issues the eight MFMAs of fragment pair g.  The next group's two ds_read_b128 are
// issued before the current group's 8 MFMAs; sched_barrier pins that order (see edge_layer history: left alone hipcc either
// serialises read -> wait -> MFMA or hoists a phase worth of fragments and spills).
template <int NG, class Body>
__device__ __forceinline__ void frag_pipeline(const f32x4* W, const f32x4* after, f32x4 (&fa)[2], f32x4 (&fb)[2],
                                              int lane, Body body) {
  static_assert(NG % 2 == 0, "even number of fragment groups");
#pragma unroll
  for (int gq = 0; gq < NG; gq += 2) {
    load_frags2(fb, W + (gq + 1) * 128, lane);
    __builtin_amdgcn_sched_barrier(0);
    body(gq, fa[0], fa[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (gq + 2 < NG) load_frags2(fa, W + (gq + 2) * 128, lane);
    else if (after != nullptr) load_frags2(fa, after, lane);
    __builtin_amdgcn_sched_barrier(0);
    body(gq + 1, fb[0], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NXB, int NEB, int NFB, int NW>
__global__ __launch_bounds__(NW * 64) void edge_unit_layer_kernel(EdgeLayerArgs a) {
  using G = UnitGeom<NXB, NEB, NFB>;
  constexpr int DX = G::DX, DE = G::DE, IBE = G::IBE, IBF = G::IBF, NPH = G::NPH;
  static_assert(NFB % 2 == 0, "dim_ffE must be a multiple of 64");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* wbuf = reinterpret_cast<f32x4*>(smem);                        // [2][PH_F4]
  float* fl = reinterpret_cast<float*>(wbuf + 2 * G::PH_F4);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  float* vecw = fl + wave * (3 * DX);                                  // Q row / sqrt(df) | y_e_add | y_e_mul+1
  const int n = a.n;
  const int nch = (n + 15) >> 4;
  const int count = *a.unit_count;
  const int per_iter = NW * gridDim.x;
  const int iters = (count + per_iter - 1) / per_iter;
  constexpr float LOG2E = 1.4426950408889634f;

  stage_phase_unit<G::PH_F4, NW>(a.phase_img, wbuf, wave, lane);
  int cur = 0;   // buffer holding the phase about to run

  // everything a wave needs to know about a unit
  struct Unit {
    bool active, inb, vj, live;
    int u, row, b, i;
    long long off_ij, off_ji;
    const float *Kj, *Vj, *film, *qrow;
  };
  auto unit_of = [&](int uidx) {
    Unit t;
    t.active = uidx < count;                                           // wave-uniform
    t.u = t.active ? a.units[uidx] : 0;
    t.row = t.u / nch;
    const int jc = t.u - t.row * nch;
    t.b = t.row / n;
    t.i = t.row - t.b * n;
    const int j = jc * 16 + p;
    t.inb = j < n;
    const int jj = t.inb ? j : 0;
    t.vj = t.active && t.inb && a.mask[t.b * n + jj] != 0;
    t.live = t.vj && j != t.i;
    t.off_ij = ((long long)t.row * n + jj) * DE;
    t.off_ji = (((long long)t.b * n + jj) * n + t.i) * DE;
    t.Kj = a.K + ((long long)t.b * n + jj) * a.ldqkv + 4 * g;
    t.Vj = a.V + ((long long)t.b * n + jj) * a.ldqkv + 4 * g;
    t.film = a.film + (long long)t.b * (4 * DX);
    t.qrow = a.Q + (long long)t.row * a.ldqkv;
    return t;
  };
  // per-row vectors (Q row / sqrt(df) | y_e_add | y_e_mul+1): global -> registers -> this wave's LDS slice
  constexpr int NV = (3 * DX + 255) / 256;
  auto vec_load = [&](const Unit& t, f32x4 (&vr)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = lane * 4 + 256 * k;
      if (t.active && c < 3 * DX) {
        vr[k] = ld4(c < DX ? t.qrow + c : t.film + (c - DX));
        if (c < DX) vr[k] = vr[k] * a.inv_sqrt_df;                     // fold 1/sqrt(df) into the query row
      }
    }
  };
  auto vec_store = [&](const Unit& t, const f32x4 (&vr)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = lane * 4 + 256 * k;
      if (t.active && c < 3 * DX) *reinterpret_cast<f32x4*>(vecw + c) = vr[k];
    }
  };
  // symmetrised input tile (transformer_model.py:99 of the previous layer, applied on load): raw loads / combination
  auto tile_load = [&](const Unit& t, f32x4 (&x0)[IBE], f32x4 (&x1)[IBE]) {
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb)
      if (t.live) {
        x0[eb] = ld4(a.E_in + t.off_ij + 16 * eb + 4 * g);
        x1[eb] = ld4(a.E_in + t.off_ji + 16 * eb + 4 * g);
      }
  };
  auto tile_make = [&](const Unit& t, const f32x4 (&x0)[IBE], const f32x4 (&x1)[IBE], f32x4 (&e)[IBE]) {
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) e[eb] = t.live ? 0.5f * (x0[eb] + x1[eb]) : splat4(0.f);
  };

  // first unit: loaded synchronously; every later unit is fetched during the FFN phases of its predecessor (its loads
  // would otherwise stall all the waves of the workgroup at once, right after the unit boundary)
  Unit un = unit_of(blockIdx.x * NW + wave);
  f32x4 e[IBE];
  {
    f32x4 vr[NV], x0[IBE], x1[IBE];
    vec_load(un, vr);
    tile_load(un, x0, x1);
    vec_store(un, vr);
    tile_make(un, x0, x1, e);
  }

  for (int it = 0; it < iters; ++it) {
    const bool active = un.active, inb = un.inb, vj = un.vj, live = un.live;
    const int u = un.u;
    const long long off_ij = un.off_ij;
    const float *Kj = un.Kj, *Vj = un.Vj;
    const Unit nx = unit_of(((it + 1) * gridDim.x + blockIdx.x) * NW + wave);
    f32x4 accO[IBE];
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) accO[eb] = ld4(a.b_out + 16 * eb + 4 * g);

    f32x4 fa[2], fb[2];                                                 // A fragments: group in use / group in flight
    // ================================================================== channel-block phases
#pragma unroll 1
    for (int cb = 0; cb < NXB; ++cb) {
      __syncthreads();
      stage_phase_unit<G::PH_F4, NW>(a.phase_img + (size_t)(cb + 1) * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
      const f32x4* W = wbuf + cur * G::PH_F4;
      if (active) {
        f32x4 k[2], vn[2], a1[2], a2[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          k[blk] = ld4(Kj + 32 * cb + 16 * blk);
          vn[blk] = ld4(Vj + 32 * cb + 16 * blk);
        }
        load_frags2(fa, W, lane);
        const float* bias = reinterpret_cast<const float*>(W + G::WMAX);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
          a1[ob] = ld4(bias + 16 * ob + 4 * g);                         // e_mul bias + 1
          a2[ob] = ld4(bias + 32 + 16 * ob + 4 * g);                    // e_add bias
        }
        frag_pipeline<2 * IBE>(W, W + 4 * IBE * 64, fa, fb, lane, [&](int gp, const f32x4& f0, const f32x4& f1) {
          if (gp < IBE) mfma4x2(a1[0], a1[1], f0, f1, e[gp]);           // e_mul, input block gp
          else mfma4x2(a2[0], a2[1], f0, f1, e[gp - IBE]);              // e_add
        });
        f32x4 y[2], v[2], z[2];
        float mloc = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const int co = 32 * cb + 16 * blk + 4 * g;
          const f32x4 q = ld4(vecw + co);                               // Q_i / sqrt(df)
          const f32x4 yea = ld4(vecw + DX + co);
          const f32x4 yem = ld4(vecw + 2 * DX + co);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float kk = vj ? k[blk][r] : 0.f;
            const float yy = (q[r] * kk) * a1[blk][r] + a2[blk][r];     // Y = QK/sqrt(df) * (E1+1) + E2
            z[blk][r] = yea[r] + yem[r] * yy;                           // FiLM by y before e_out
            y[blk][r] = vj ? yy : -INFINITY;
            v[blk][r] = vj ? vn[blk][r] : 0.f;
            mloc = fmaxf(mloc, y[blk][r]);
          }
        }
        // newE += W_out[:, block cb] . z
        frag_pipeline<IBE>(W + 4 * IBE * 64, nullptr, fa, fb, lane, [&](int gp, const f32x4& f0, const f32x4& f1) {
          mfma4x2(accO[2 * (gp >> 1)], accO[2 * (gp >> 1) + 1], f0, f1, z[gp & 1]);
        });

        // ---- attention statistics of this (unit, channel block)
        // softmax shift: the maximum over this lane row's 16 keys x 8 channels.  A channel lives in ONE 16-lane row,
        // so the shift only has to be uniform inside the row, and the row maximum needs no cross-row exchange.
        const float M = row_max16_fused(mloc);
        const float nM = M == -INFINITY ? 0.f : -M * LOG2E;
        f32x4 cm[2], cs[2], ca[2];
        bool under = false;
        {
          float red[16];                                                // 8 x exp(y - M) (0 on invalid keys), 8 x p V
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pe = __builtin_amdgcn_exp2f(fmaf(y[blk][r], LOG2E, nM));
              red[4 * blk + r] = pe;
              red[8 + 4 * blk + r] = pe * v[blk][r];
            }
          row_sum16_many<16>(red);
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              cs[blk][r] = red[4 * blk + r];
              ca[blk][r] = red[8 + 4 * blk + r];
              cm[blk][r] = M;
              under |= cs[blk][r] == 0.f;
            }
        }
        if (__any(under && M > -INFINITY)) {
          // a channel underflowed against the row-wide shift: exact per-channel maxima for the whole unit
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float yy = y[blk][r];
              const float mx = row_max16(yy);
              const float ex = yy == -INFINITY ? 0.f : __expf(yy - mx);
              cm[blk][r] = mx;
              cs[blk][r] = row_sum16(ex);
              ca[blk][r] = row_sum16(ex * v[blk][r]);
            }
        }
        if (p == 0) {
          float* dst = a.attn_part + (long long)u * (3 * DX) + 32 * cb + 4 * g;
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
            *reinterpret_cast<f32x4*>(dst + 16 * blk) = cm[blk];
            *reinterpret_cast<f32x4*>(dst + DX + 16 * blk) = cs[blk];
            *reinterpret_cast<f32x4*>(dst + 2 * DX + 16 * blk) = ca[blk];
          }
        }
      }
      cur ^= 1;
    }

    // ================================================================== E <- LN1(E + newE), FFN phases
    f32x4 hid[IBF];
    f32x4 nvr[NV], nx0[IBE], nx1[IBE];                                  // next unit, in flight
#pragma unroll
    for (int f = 0; f < NFB / 2; ++f) {
      __syncthreads();
      stage_phase_unit<G::PH_F4, NW>(a.phase_img + (size_t)(NXB + f + 1) * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
      const f32x4* W = wbuf + cur * G::PH_F4;
      if (f == 0) vec_load(nx, nvr);                                    // FiLM reads of this unit are over
      if (active) {
        load_frags2(fa, W, lane);
        if (f == 0) {
#pragma unroll
          for (int eb = 0; eb < IBE; ++eb) e[eb] += accO[eb];
          layernorm_c16<IBE>(e, a.ln1_g, a.ln1_b, g);
        }
        const float* bias = reinterpret_cast<const float*>(W + G::WMAX);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) hid[4 * f + ob] = ld4(bias + 16 * ob + 4 * g);
        frag_pipeline<2 * IBE>(W, nullptr, fa, fb, lane, [&](int gp, const f32x4& f0, const f32x4& f1) {
          mfma4x2(hid[4 * f + 2 * (gp / IBE)], hid[4 * f + 2 * (gp / IBE) + 1], f0, f1, e[gp % IBE]);
        });
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) hid[4 * f + ob][r] = fmaxf(hid[4 * f + ob][r], 0.f);
      }
      cur ^= 1;
    }
#pragma unroll
    for (int rb = 0; rb < NEB; ++rb) {
      __syncthreads();
      const int nxt = (NXB + NFB / 2 + rb + 1) % NPH;                   // wraps to the first phase of the next unit
      stage_phase_unit<G::PH_F4, NW>(a.phase_img + (size_t)nxt * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
      if (rb == 0) {
        vec_store(nx, nvr);
        tile_load(nx, nx0, nx1);
      }
      const f32x4* W = wbuf + cur * G::PH_F4;
      if (active) {
        load_frags2(fa, W, lane);
        const float* bias = reinterpret_cast<const float*>(W + G::WMAX);
        f32x4 f2[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) f2[ob] = ld4(bias + 16 * ob + 4 * g);
        frag_pipeline<IBF>(W, nullptr, fa, fb, lane, [&](int gp, const f32x4& f0, const f32x4& f1) {
          mfma4x2(f2[0], f2[1], f0, f1, hid[gp]);
        });
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) e[2 * rb + ob] += f2[ob];
      }
      cur ^= 1;
    }
    if (active) {
      layernorm_c16<IBE>(e, a.ln2_g, a.ln2_b, g);
      if (inb) {
        float* dst = a.E_out + off_ij + 4 * g;
#pragma unroll
        for (int eb = 0; eb < IBE; ++eb) *reinterpret_cast<f32x4*>(dst + 16 * eb) = live ? e[eb] : splat4(0.f);
      }
    }
    tile_make(nx, nx0, nx1, e);
    un = nx;
  }
  // the wrap-around weight DMA of the last round (and whatever an idle wave issued) must land before the wave ends: the
  // workgroup's LDS is handed to the next workgroup on the CU as soon as its waves are gone
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// =========================================================================================================
// The same layer on the bf16 matrix pipe with fp32 accuracy ("x3" path).
//
// On gfx950 an fp32 MFMA runs at the vector rate and its time adds to the VALU time of the SIMD; a bf16 MFMA is 16x
// faster and executes beside the VALU.  Every fp32 operand is therefore split EXACTLY into three bf16 terms by
// truncation (x = h + m + l, 8 mantissa bits each, |x - h - m - l| <= 2^-24 |x|) and a product is evaluated as the six
// leading partial products  hl' + lh' + mm' + mh' + hm' + hh'  on v_mfma_f32_16x16x32_bf16 (bf16 x bf16 products are
// exact in the fp32 accumulator): the dropped terms are below 2^-24 relative, i.e. the result has fp32 accuracy, at
// 6/16 of the fp32 MFMA time.  Weights are split once at pack time, activations per 32-channel chunk in registers
// (5.5 VALU per value: and / sub / and / sub + three byte-permute packs per pair).
//
// Layouts.  C16 as above for everything that is not an MFMA operand.  K runs in chunks of 32 channels = two C16 blocks:
// lane (key p, group g) supplies the 8 k-slots  [blk0 r=0..3 | blk1 r=0..3]  = channels 32q + {4g..4g+3, 16+4g..16+4g+3};
// the weight fragment of (16-row output block ob, chunk q) holds, for lane (o = l & 15, g), W[16ob+o][the same 8
// channels] -- one 16-byte LDS read per term.  A phase image is a list of such blocks (3 terms x 1 KiB each).
// =========================================================================================================

template <int NXB, int NEB, int NFB>
struct UnitGeomX3 {
  static constexpr int DX = 32 * NXB, DE = 32 * NEB, FFE = 32 * NFB;
  static constexpr int IBE = DE / 16, IBF = FFE / 16;
  static constexpr int QE = NEB, QF = NFB;                   // 32-channel K chunks of the E width / FFN width
  static constexpr int B_CB = 4 * QE + IBE;                  // blocks of a channel-block phase: e_mul, e_add, e_out slice
  static constexpr int B_FF = 4 * QE + 2 * IBE;              // blocks of an FFN phase: 64 hidden rows of linE1, their columns of linE2
  static constexpr int BMAX = B_CB > B_FF ? B_CB : B_FF;
  static constexpr int BIAS_F4 = 64;                         // fp32 bias area at the head of every phase image
  static constexpr int CB_F4 = BIAS_F4 + B_CB * 192, FF_F4 = BIAS_F4 + B_FF * 192;   // float4 slots actually used
  static constexpr int PH_F4 = BIAS_F4 + BMAX * 192;         // stride of the phase images
  static constexpr int NFF = NFB / 2;                        // FFN phases (64 hidden channels each)
  static constexpr int NPH = NXB + NFF;
  static constexpr size_t lds_bytes(int nw) { return (size_t)2 * PH_F4 * 16 + (size_t)nw * 3 * DX * 4; }
};

// Phase images of the x3 path from the raw (torch layout) weights: 64 float4 of fp32 biases, then the blocks in order of use:
//   cb phase : e_mul (q-major: for q, for ob = 0,1: rows 32cb+16ob) | e_add likewise | e_out rows 16ob (ob = 0..IBE-1), cols
//              32cb..32cb+31;   biases: [0,32) e_mul bias + 1, [32,64) e_add bias
//   FFN f    : linE1: for q, for ob = 0..3: rows 64f+16ob | linE2: for hq = 0,1, for ob = 0..IBE-1: rows 16ob, cols 64f+32hq;
//              biases: [0,64) linE1 bias of the 64 hidden channels, [64,64+DE) linE2 bias
template <int NXB, int NEB, int NFB>
__global__ __launch_bounds__(512) void edge_unit_pack_x3_kernel(EdgeUnitPackArgs a) {
  using G = UnitGeomX3<NXB, NEB, NFB>;
  constexpr int DX = G::DX, DE = G::DE, FFE = G::FFE, IBE = G::IBE, QE = G::QE;
  const int ph = blockIdx.x;
  u32x4* img = reinterpret_cast<u32x4*>(a.img) + (size_t)ph * G::PH_F4;
  for (int idx = threadIdx.x; idx < G::PH_F4; idx += blockDim.x) {
    u32x4 out = {0u, 0u, 0u, 0u};
    if (idx >= G::BIAS_F4) {
      const int bi = idx - G::BIAS_F4;
      const int t = bi / 192, term = (bi % 192) >> 6, l = bi & 63;
      const float* W = nullptr;
      int row0 = 0, col0 = 0, K = 0;
      if (ph < NXB) {
        const int cb = ph;
        if (t < 4 * QE) {
          W = t < 2 * QE ? a.W_mul : a.W_add;
          const int rem = t % (2 * QE);
          row0 = 32 * cb + 16 * (rem & 1); col0 = 32 * (rem >> 1); K = DE;
        } else if (t < 4 * QE + IBE) {
          W = a.W_out; row0 = 16 * (t - 4 * QE); col0 = 32 * cb; K = DX;
        }
      } else {
        const int f = ph - NXB;
        if (t < 4 * QE) { W = a.W_ff1; row0 = 64 * f + 16 * (t & 3); col0 = 32 * (t >> 2); K = DE; }
        else if (t < 4 * QE + 2 * IBE) {
          const int tt = t - 4 * QE;
          W = a.W_ff2; row0 = 16 * (tt % IBE); col0 = 64 * f + 32 * (tt / IBE); K = FFE;
        }
      }
      if (W != nullptr) {
        const int o = row0 + (l & 15), g = l >> 4;
        unsigned hw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int col = col0 + (i < 4 ? 4 * g + i : 16 + 4 * g + i - 4);
          float x = W[(size_t)o * K + col];
          for (int tt = 0; tt < term; ++tt) x = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
          hw[i] = __float_as_uint(x) >> 16;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) out[d] = hw[2 * d] | (hw[2 * d + 1] << 16);
      }
    } else {
      const int fo = idx * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (ph < NXB) {
        if (fo < 64) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fo < 32 ? a.b_mul[32 * ph + fo + e] + 1.0f : a.b_add[32 * ph + fo - 32 + e];
        }
      } else if (fo < 64) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a.b_ff1[64 * (ph - NXB) + fo + e];
      } else if (fo < 64 + DE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a.b_ff2[fo - 64 + e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = __float_as_uint(v[e]);
    }
    img[idx] = out;
  }
}

template <int NXB, int NEB, int NFB, int NW>
__global__ __launch_bounds__(NW * 64) void edge_unit_layer_x3_kernel(EdgeLayerArgs a) {
  using G = UnitGeomX3<NXB, NEB, NFB>;
  constexpr int DX = G::DX, DE = G::DE, IBE = G::IBE, IBF = G::IBF, NPH = G::NPH, QE = G::QE, QF = G::QF;
  static_assert(NFB % 2 == 0, "dim_ffE must be a multiple of 64");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* wbuf = reinterpret_cast<f32x4*>(smem);                        // [2][PH_F4]
  float* fl = reinterpret_cast<float*>(wbuf + 2 * G::PH_F4);
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: everything per unit below stays in SGPRs
  float* vecw = fl + wave * (3 * DX);                                  // Q row / sqrt(df) | y_e_add | y_e_mul+1
  const int n = a.n;
  const int nch = (n + 15) >> 4;
  const int count = *a.unit_count;
  // Rounds: in a full round workgroup b takes the NW consecutive units of its slot.  The units left over for the last,
  // partial round are spread over ALL workgroups (q consecutive units each) instead of filling the first ones: a round's
  // duration grows with the number of active waves on the CU (it is latency-bound), so half-filled workgroups everywhere
  // finish sooner than full ones on half of the CUs.
  const int per_iter = NW * gridDim.x;
  const int full = count / per_iter, rest = count - full * per_iter;
  const int iters = full + (rest > 0 ? 1 : 0);
  const int q_last = (rest + (int)gridDim.x - 1) / (int)gridDim.x;
  // Workgroups go to the 8 XCDs round-robin (workgroup w on XCD w % 8), and every XCD has its own L2.  The unit list is in
  // graph order, so the workgroups of ONE XCD take consecutive stretches of it: a graph's K / V rows and both orientations
  // of its E tiles (E_ij of one unit is E_ji of another unit of the same graph) are then fetched into one L2 instead of four
  // or more.  SCGG_EDGE_XCD=0 (a.xcd_map == 0) keeps the plain order for A/B runs.
  const int nwg = (int)gridDim.x;
  const int wg = (a.xcd_map && nwg % 8 == 0) ? ((int)blockIdx.x % 8) * (nwg / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  auto slot = [&](int it) -> int {                                     // unit-list index of this wave in round `it`
    if (it < full) return (it * nwg + wg) * NW + wave;
    const int r = wg * q_last + wave;
    return (it == full && wave < q_last && r < rest) ? full * per_iter + r : count;
  };
  constexpr float LOG2E = 1.4426950408889634f;
  typedef const char* cptr;
  auto ldo = [](cptr base, unsigned off) { return *reinterpret_cast<const f32x4*>(base + off); };   // scalar base + lane offset

  stage_phase_unit<G::CB_F4, NW>(a.phase_img, wbuf, wave, lane);
  int cur = 0;   // buffer holding the phase about to run

  // A unit = (row (b, i), 16-key chunk jc): all scalar.  The lane's key is j = 16 jc + p.
  struct Unit {
    bool active;
    int u, row, b, i, j0;
  };
  auto unit_of = [&](int uidx) {
    Unit t;
    t.active = uidx < count;
    t.u = t.active ? __builtin_amdgcn_readfirstlane(a.units[uidx]) : 0;
    t.row = t.u / nch;
    t.j0 = (t.u - t.row * nch) * 16;
    t.b = t.row / n;
    t.i = t.row - t.b * n;
    return t;
  };
  auto mask_byte = [&](const Unit& t) -> int {                         // of the lane's key (0 beyond n)
    const int j = t.j0 + p;
    return (t.active && j < n) ? (int)a.mask[t.b * n + j] : 0;
  };
  constexpr int NV = (3 * DX + 255) / 256;
  auto vec_load = [&](const Unit& t, f32x4 (&vr)[NV]) {
    const cptr qrow = reinterpret_cast<cptr>(a.Q + (long long)t.row * a.ldqkv);
    const cptr film = reinterpret_cast<cptr>(a.film + (long long)t.b * (4 * DX));
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = lane * 4 + 256 * k;
      if (c < 3 * DX) {
        vr[k] = c < DX ? ldo(qrow, 4u * c) : ldo(film, 4u * (c - DX));
        if (c < DX) vr[k] = vr[k] * a.inv_sqrt_df;                     // fold 1/sqrt(df) into the query row
      }
    }
  };
  auto vec_store = [&](const f32x4 (&vr)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = lane * 4 + 256 * k;
      if (c < 3 * DX) *reinterpret_cast<f32x4*>(vecw + c) = vr[k];
    }
  };
  // the symmetrised tile row: 0.5 (E[b,i,j,:] + E[b,j,i,:]) on live pairs, 0 elsewhere
  auto tile_load = [&](const Unit& t, int jj, f32x4 (&x0)[IBE], f32x4 (&x1)[IBE]) {
    const cptr bij = reinterpret_cast<cptr>(a.E_in + (long long)t.row * n * DE);
    const cptr bji = reinterpret_cast<cptr>(a.E_in + ((long long)t.b * n * n + t.i) * DE);
    const unsigned oij = 4u * (jj * DE + 4 * g), oji = 4u * (jj * n * DE + 4 * g);
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) {
      x0[eb] = ldo(bij, oij + 64u * eb);
      x1[eb] = ldo(bji, oji + 64u * eb);
    }
  };

  Unit un = unit_of(slot(0));
  int mk = mask_byte(un);
  f32x4 e[IBE];
  if (un.active) {
    f32x4 vr[NV], x0[IBE], x1[IBE];
    const int j = un.j0 + p, jj = j < n ? j : 0;
    const bool live = mk != 0 && j != un.i;
    vec_load(un, vr);
    tile_load(un, jj, x0, x1);
    vec_store(vr);
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) e[eb] = live ? 0.5f * (x0[eb] + x1[eb]) : splat4(0.f);
  }

  for (int it = 0; it < iters; ++it) {
    const bool active = un.active;
    const int j = un.j0 + p;
    const bool inb = j < n, vj = mk != 0, live = vj && j != un.i;
    const int jj = inb ? j : 0;
    // K / V are read without a select per element (a lane without a key has logit -inf, weight 0): its rows must hold
    // finite numbers.  The rows of masked / padding nodes are never written by the row-list Linears, so such a lane reads
    // the query's own row instead (valid whenever the unit is).
    const int jkv = vj ? j : (active ? un.i : 0);
    // the next unit: scalars now, the mask byte of the lane's key in flight until the FFN phases
    const Unit nx = unit_of(slot(it + 1));
    const int nmk = mask_byte(nx);

    const cptr kbase = reinterpret_cast<cptr>(a.K + (long long)un.b * n * a.ldqkv);
    const cptr vbase = reinterpret_cast<cptr>(a.V + (long long)un.b * n * a.ldqkv);
    const unsigned kvoff = 4u * (jkv * a.ldqkv + 4 * g);

    // per-layer vectors are re-read per unit (L1/L2 hits): hoisted out of the unit loop they would hold ~80 VGPRs
    const float *b_out = a.b_out, *ln1_g = a.ln1_g, *ln1_b = a.ln1_b, *ln2_g = a.ln2_g, *ln2_b = a.ln2_b;
    asm volatile("" : "+s"(b_out), "+s"(ln1_g), "+s"(ln1_b), "+s"(ln2_g), "+s"(ln2_b));
    f32x4 accO[IBE];
    u32x4 eh[QE], em[QE], el[QE];                                       // E tile as MFMA operand, kept through the cb phases
    Frag3 fa, fb;
    if (active) {
#pragma unroll
      for (int eb = 0; eb < IBE; ++eb) accO[eb] = ld4(b_out + 16 * eb + 4 * g);
#pragma unroll
      for (int q = 0; q < QE; ++q) split3(e[2 * q], e[2 * q + 1], eh[q], em[q], el[q]);
    }

    // ================================================================== channel-block phases
#pragma unroll 1
    for (int cb = 0; cb < NXB; ++cb) {
      // A wave with a unit has consumed its K / V loads, which are younger than its share of this phase's weight DMA, so
      // that share has landed; a wave WITHOUT a unit waits for nothing on its own and must retire its share here.
      if (!active) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (cb + 1 < NXB)
        stage_phase_unit<G::CB_F4, NW>(a.phase_img + (size_t)(cb + 1) * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
      else
        stage_phase_unit<G::FF_F4, NW>(a.phase_img + (size_t)(cb + 1) * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
      const float* bias = reinterpret_cast<const float*>(wbuf + cur * G::PH_F4);
      const f32x4* W = wbuf + cur * G::PH_F4 + G::BIAS_F4;
      if (active) {
        f32x4 k[2], vn[2], a1[2], a2[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#ifdef SCGG_ABL_NO_KV
          k[blk] = splat4(0.5f); vn[blk] = splat4(0.25f);
#else
          k[blk] = ldo(kbase, kvoff + 4u * (32 * cb + 16 * blk));
          vn[blk] = ldo(vbase, kvoff + 4u * (32 * cb + 16 * blk));
#endif
        }
        load_frag3(fa, W, lane);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
          a1[ob] = ld4(bias + 16 * ob + 4 * g);                         // e_mul bias + 1
          a2[ob] = ld4(bias + 32 + 16 * ob + 4 * g);                    // e_add bias
        }
        block_pipeline<4 * QE, 0>(W, W + 4 * QE * 192, fa, fb, lane, [&](int t, const Frag3& w) {
          const int rem = t % (2 * QE), q = rem >> 1, ob = rem & 1;
          if (t < 2 * QE) mfma_x3(a1[ob], w, eh[q], em[q], el[q]);      // e_mul
          else mfma_x3(a2[ob], w, eh[q], em[q], el[q]);                 // e_add
        });
        f32x4 y[2], v[2], z[2];
        float mloc = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const int co = 32 * cb + 16 * blk + 4 * g;
          const f32x4 q = ld4(vecw + co);                               // Q_i / sqrt(df)
          const f32x4 yea = ld4(vecw + DX + co);
          const f32x4 yem = ld4(vecw + 2 * DX + co);
          // No select on K / V: a lane without a valid key holds the finite data of some other row (q|k|v of masked rows are
          // zero, lanes beyond n read key 0); its logit is masked below (once per lane), so its weight is exactly 0, and
          // an MFMA column depends on its own lane only -- nothing of such a lane is ever stored.
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float yy = (q[r] * k[blk][r]) * a1[blk][r] + a2[blk][r];   // Y = QK/sqrt(df) * (E1+1) + E2
            z[blk][r] = yea[r] + yem[r] * yy;                           // FiLM by y before e_out
            y[blk][r] = yy;
            v[blk][r] = vn[blk][r];
            mloc = fmaxf(mloc, yy);
          }
        }
        // ---- attention statistics of this (unit, channel block), as in the fp32 kernel.  They do not depend on the e_out
        // MFMAs and sit inside that pipeline's scheduling regions: the exponentials behind its first block, the
        // reduce-scatter behind its second (a small, repeatable gain: 0.2457 -> 0.2432 ms per launch).
#ifdef SCGG_ABL_NO_SM
        if (mloc == 12345.f)
#endif
        {
        const float M = row_max16_fused(vj ? mloc : -INFINITY);
        const float nM = !vj ? -INFINITY : (M == -INFINITY ? 0.f : -M * LOG2E);   // a lane without a key: exp2(-inf) = 0
        float mine = 0.f;
        float red[16];                                                  // 8 x exp(y - M) (0 on invalid keys), 8 x p V
        // newE += W_out[:, block cb] . z   (the (4 QE)-block pipeline above left the first e_out block in fa)
        {
          u32x4 zh, zm, zl;
          split3(z[0], z[1], zh, zm, zl);
          block_pipeline<IBE, 0>(W + 4 * QE * 192, nullptr, fa, fb, lane, [&](int t, const Frag3& w) {
            mfma_x3(accO[t], w, zh, zm, zl);
            if (t == 0) {
#pragma unroll
              for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float pe = __builtin_amdgcn_exp2f(fmaf(y[blk][r], LOG2E, nM));
                  red[4 * blk + r] = pe;
                  red[8 + 4 * blk + r] = pe * v[blk][r];
                }
            }
            // reduce-scatter over the 16 keys: lane p gets the row's sum of red[p] (p < 8: sum p of channel (blk = p >> 2,
            // r = p & 3); p >= 8: sum p V of channel p - 8): 33 VALU instead of the 64-80 of sixteen butterflies
            if (t == (IBE > 1 ? 1 : 0)) mine = row_reduce_scatter16(red, p);
          });
        }
        float* part = a.attn_part + (long long)un.u * (3 * DX);
        if (!__any(p < 8 && mine == 0.f && M > -INFINITY)) {
          const int ch = 32 * cb + 16 * ((p >> 2) & 1) + 4 * g + (p & 3);
          part[(p < 8 ? DX : 2 * DX) + ch] = mine;
          if (p < 8) part[ch] = M;
        } else {                                                        // a channel underflowed against the shared shift: exact maxima
          f32x4 cm[2], cs[2], ca[2];
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float yy = vj ? y[blk][r] : -INFINITY;
              const float mx = row_max16(yy);
              const float ex = yy == -INFINITY ? 0.f : __expf(yy - mx);
              cm[blk][r] = mx;
              cs[blk][r] = row_sum16(ex);
              ca[blk][r] = row_sum16(yy == -INFINITY ? 0.f : ex * v[blk][r]);
            }
          if (p == 0) {
            float* dst = part + 32 * cb + 4 * g;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
              *reinterpret_cast<f32x4*>(dst + 16 * blk) = cm[blk];
              *reinterpret_cast<f32x4*>(dst + DX + 16 * blk) = cs[blk];
              *reinterpret_cast<f32x4*>(dst + 2 * DX + 16 * blk) = ca[blk];
            }
          }
        }
        }
      }
      cur ^= 1;
    }

    // ================================================================== E <- LN1(E + newE), FFN phases
    // One FFN phase = 64 hidden channels: linE1 + ReLU into 4 register blocks, then their contribution through linE2
    // accumulated directly onto E (which holds the LN1 output, i.e. the residual).
#pragma unroll
    for (int f = 0; f < G::NFF; ++f) {
      if (!active || f > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // f > 0: no younger load was consumed since
      __syncthreads();
      if (f + 1 < G::NFF)
        stage_phase_unit<G::FF_F4, NW>(a.phase_img + (size_t)(NXB + f + 1) * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
      else                                                              // wraps to the first phase of the next unit
        stage_phase_unit<G::CB_F4, NW>(a.phase_img, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
      const float* bias = reinterpret_cast<const float*>(wbuf + cur * G::PH_F4);
      const f32x4* W = wbuf + cur * G::PH_F4 + G::BIAS_F4;
      if (active) {
        load_frag3(fa, W, lane);
        if (f == 0) {
#pragma unroll
          for (int eb = 0; eb < IBE; ++eb) e[eb] += accO[eb];
          layernorm_c16<IBE>(e, ln1_g, ln1_b, g);
#pragma unroll
          for (int q = 0; q < QE; ++q) split3(e[2 * q], e[2 * q + 1], eh[q], em[q], el[q]);
#pragma unroll
          for (int eb = 0; eb < IBE; ++eb) e[eb] += ld4(bias + 64 + 16 * eb + 4 * g);   // linE2 bias
        }
        f32x4 hid[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) hid[ob] = ld4(bias + 16 * ob + 4 * g);
        block_pipeline<4 * QE, 0>(W, W + 4 * QE * 192, fa, fb, lane, [&](int t, const Frag3& w) {
          mfma_x3(hid[t & 3], w, eh[t >> 2], em[t >> 2], el[t >> 2]);
        });
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) hid[ob][r] = fmaxf(hid[ob][r], 0.f);
        u32x4 hh, hm, hl;
        block_pipeline<2 * IBE, 0>(W + 4 * QE * 192, nullptr, fa, fb, lane, [&](int t, const Frag3& w) {
          if (t % IBE == 0) split3(hid[2 * (t / IBE)], hid[2 * (t / IBE) + 1], hh, hm, hl);   // hidden chunk hq = t / IBE, once
          mfma_x3(e[t % IBE], w, hh, hm, hl);
        });
      }
      cur ^= 1;
    }
    if (active) {
      layernorm_c16<IBE>(e, ln2_g, ln2_b, g);
      if (inb) {
        float* dst = a.E_out + ((long long)un.row * n + jj) * DE + 4 * g;
#pragma unroll
        for (int eb = 0; eb < IBE; ++eb) *reinterpret_cast<f32x4*>(dst + 16 * eb) = live ? e[eb] : splat4(0.f);
      }
    }
    // the next unit's row vectors and E tile.  Not prefetched under the FFN phases: the 44 registers that took are what
    // kept a third wave per SIMD from fitting (149 VGPRs now), and three waves hide this latency better than the prefetch
    // did with two (probe: 331 us against 354)
    if (nx.active) {
      f32x4 nvr[NV], nx0[IBE], nx1[IBE];
      const int nj = nx.j0 + p, njj = nj < n ? nj : 0;
      vec_load(nx, nvr);
      tile_load(nx, njj, nx0, nx1);
      vec_store(nvr);
      const bool nlive = nmk != 0 && nj != nx.i;
#pragma unroll
      for (int eb = 0; eb < IBE; ++eb) e[eb] = nlive ? 0.5f * (nx0[eb] + nx1[eb]) : splat4(0.f);
    }
    un = nx;
    mk = nmk;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the wave ends (see the fp32 kernel's end)
}

// =========================================================================================================
// The x3 layer with the three waves of every SIMD one SUB-STEP apart ("x3s": staggered).
//
// Why.  Cycle stamps of the x3 kernel above (profiles/r02_edge_stamps_before.txt): its three waves per SIMD walk each
// phase in lockstep -- all in the e_mul / e_add MFMAs, then all in the FiLM / softmax VALU -- so the matrix pipe is busy
// ~40 % of a phase and the VALU tail of the slowest wave is exposed at the phase barrier (20 % of a round is barrier
// wait).  Fewer DMA issuers or a register prefetch of the next unit did not help (the first moves the imbalance, the
// second spills).
//
// How.  A phase is cut into three sub-steps with a barrier after each:
//   channel block cb : S0 = K/V requests + e_mul / e_add (48 MFMAs)   S1 = FiLM, split, e_out slice (24 MFMAs)
//                      S2 = attention statistics + partial stores + this wave's share of the weight DMA
//   FFN half phase h : F0 = (h = 0: residual, LayerNorm 1, split) + linE1 for 32 hidden channels (24 MFMAs)
//                      F1 = ReLU, split, their columns of linE2 (24 MFMAs)
//                      F2 = DMA share (+ after the last half phase: LayerNorm 2, E store, next unit's requests)
// Wave group g = wave / 4 (one wave of every SIMD) starts g barriers late, so that at any time a SIMD holds one wave in
// S0, one in S1 and one in S2: matrix work of one wave beside the VALU work of the others, in every slot.
//
// Weight images: three LDS buffers (image q lives in buffer q mod 3).  While group 0 is in phase q the buffers hold
// q - 1 (still read by the groups behind), q and q + 1.  Image q + 2 goes into the buffer of q - 1, whose last reader is
// group 2 in S1 of phase q - 1, i.e. the slot BEFORE group 0's S1 of phase q: groups 0 and 1 issue their share of image
// q + 2 in their S2 of phase q, group 2 in its S2 of phase q - 1 (as "its phase + 3"); every wave retires its DMA with an
// explicit vmcnt(0) at the end of its next S1 / F1, at least one barrier before group 0 reads the image in S0 of phase
// q + 2.  All FFN phases are half phases (<= the size of a channel-block image), so three buffers + the row vectors fit:
// 3 x 37.9 KB + 36 KB = 147 KB at ConStruct dims (13 KB stay free for the y kernel that runs beside this one).
// =========================================================================================================
template <int NXB, int NEB, int NFB>
struct UnitGeomX3S {
  static constexpr int DX = 32 * NXB, DE = 32 * NEB, FFE = 32 * NFB;
  static constexpr int IBE = DE / 16, QE = NEB;
  static constexpr int B_CB = 4 * QE + IBE;                  // e_mul, e_add (2 row blocks x QE chunks each), e_out slice
  static constexpr int B_FH = 2 * QE + IBE;                  // linE1 rows of 32 hidden channels, their columns of linE2
  static constexpr int BMAX = B_CB > B_FH ? B_CB : B_FH;
  static constexpr int BIAS_F4 = 64;
  static constexpr int CB_F4 = BIAS_F4 + B_CB * 192, FH_F4 = BIAS_F4 + B_FH * 192;
  static constexpr int PH_F4 = BIAS_F4 + BMAX * 192;
  static constexpr int NPH = NXB + NFB;                      // phases per unit
  static constexpr int NW = 12;                              // three groups of four waves
  static constexpr size_t lds_bytes() { return (size_t)3 * PH_F4 * 16 + (size_t)NW * 3 * DX * 4; }
};

// Phase images of the staggered kernel: channel-block phases as in edge_unit_pack_x3_kernel; FFN half phase h:
//   linE1: for q, for ob = 0,1: rows 32h+16ob, cols 32q | linE2: for ob = 0..IBE-1: rows 16ob, cols 32h..32h+31;
//   biases: [0,32) linE1 bias of the 32 hidden channels, [64,64+DE) linE2 bias
template <int NXB, int NEB, int NFB>
__global__ __launch_bounds__(512) void edge_unit_pack_x3s_kernel(EdgeUnitPackArgs a) {
  using G = UnitGeomX3S<NXB, NEB, NFB>;
  constexpr int DX = G::DX, DE = G::DE, FFE = G::FFE, IBE = G::IBE, QE = G::QE;
  const int ph = blockIdx.x;
  u32x4* img = reinterpret_cast<u32x4*>(a.img) + (size_t)ph * G::PH_F4;
  for (int idx = threadIdx.x; idx < G::PH_F4; idx += blockDim.x) {
    u32x4 out = {0u, 0u, 0u, 0u};
    if (idx >= G::BIAS_F4) {
      const int bi = idx - G::BIAS_F4;
      const int t = bi / 192, term = (bi % 192) >> 6, l = bi & 63;
      const float* W = nullptr;
      int row0 = 0, col0 = 0, K = 0;
      if (ph < NXB) {
        const int cb = ph;
        if (t < 4 * QE) {
          W = t < 2 * QE ? a.W_mul : a.W_add;
          const int rem = t % (2 * QE);
          row0 = 32 * cb + 16 * (rem & 1); col0 = 32 * (rem >> 1); K = DE;
        } else if (t < 4 * QE + IBE) {
          W = a.W_out; row0 = 16 * (t - 4 * QE); col0 = 32 * cb; K = DX;
        }
      } else {
        const int h = ph - NXB;
        if (t < 2 * QE) { W = a.W_ff1; row0 = 32 * h + 16 * (t & 1); col0 = 32 * (t >> 1); K = DE; }
        else if (t < 2 * QE + IBE) { W = a.W_ff2; row0 = 16 * (t - 2 * QE); col0 = 32 * h; K = FFE; }
      }
      if (W != nullptr) {
        const int o = row0 + (l & 15), g = l >> 4;
        unsigned hw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int col = col0 + (i < 4 ? 4 * g + i : 16 + 4 * g + i - 4);
          float x = W[(size_t)o * K + col];
          for (int tt = 0; tt < term; ++tt) x = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
          hw[i] = __float_as_uint(x) >> 16;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) out[d] = hw[2 * d] | (hw[2 * d + 1] << 16);
      }
    } else {
      const int fo = idx * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (ph < NXB) {
        if (fo < 64) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fo < 32 ? a.b_mul[32 * ph + fo + e] + 1.0f : a.b_add[32 * ph + fo - 32 + e];
        }
      } else if (fo < 32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a.b_ff1[32 * (ph - NXB) + fo + e];
      } else if (fo >= 64 && fo < 64 + DE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a.b_ff2[fo - 64 + e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = __float_as_uint(v[e]);
    }
    img[idx] = out;
  }
}

// ---- diagnostic build only (-DSCGG_STAMP, scripts/probe/edge_layer_probe.hip): per-wave cycle sums of the sub-steps of
//      a round, read with s_memtime; they leave the kernel through g_stamps, which nothing else reads.
#ifdef SCGG_STAMP
constexpr int STAMP_SEGS = 16;
__device__ unsigned long long g_stamps[4][16][STAMP_SEGS];             // [probed workgroup][wave][segment]
#define SCGG_STAMP_DECL unsigned long long st_acc[STAMP_SEGS] = {}; unsigned long long st_last = 0; \
  { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define SCGG_STAMP_AT(k) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
  st_acc[k] += t_ - st_last; st_last = t_; }
#else
#define SCGG_STAMP_DECL
#define SCGG_STAMP_AT(k)
#endif

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB of LDS at the wave-uniform byte address `lds_addr`) issued from inline asm,
// so that hipcc does not see it: with the builtin, every later ds_read of the wave (of ANY LDS address) is preceded by a
// compiler-inserted vmcnt(0), which drains a weight image requested two phases ahead at the very next sub-step.  The
// callers retire these pieces themselves (explicit vmcnt waits).  M0 is saved / restored inside the statement
// (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void lds_dma16(const void* gsrc_lane, unsigned lds_addr_uniform) {
  unsigned keep;
  const unsigned dst = lds_addr_uniform;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc_lane), "s"(dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(unsigned long long)((__attribute__((address_space(3))) const char*)p);
}

template <int NXB, int NEB, int NFB>
__global__ __launch_bounds__(768) void edge_unit_layer_x3s_kernel(EdgeLayerArgs a) {
  using G = UnitGeomX3S<NXB, NEB, NFB>;
  constexpr int DX = G::DX, DE = G::DE, IBE = G::IBE, NPH = G::NPH, QE = G::QE, NW = G::NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* wbuf = reinterpret_cast<f32x4*>(smem);                        // [3][PH_F4]
  float* fl = reinterpret_cast<float*>(wbuf + 3 * G::PH_F4);
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  // Lane-derived LDS offsets are recomputed per sub-step from an opaque copy of the lane id (SCGG_LANE: 3 VALU) instead
  // of living in registers across the unit loop, where they were spilled and their reload (scratch_load + vmcnt(0))
  // stalled every S0 behind its own K / V requests.
  // Sub-step barrier: the LDS reads of this wave are back (lgkmcnt) and nothing else is waited for.  __syncthreads() would
  // also drain vmcnt -- the weight DMA just issued (an LDS-DMA counts as a pending LDS write) -- at every S2 / F2.
#define SCGG_SUBSTEP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SCGG_LANE int lane_ = lane; asm volatile("" : "+v"(lane_)); const int g_ = lane_ >> 4; (void)g_;                                           // 0, 1, 2: runs grp sub-steps behind group 0
  float* vecw = fl + wave * (3 * DX);                                  // Q row / sqrt(df) | y_e_add | y_e_mul+1
  const int n = a.n;
  const int nch = (n + 15) >> 4;
  const int count = *a.unit_count;
  const int per_iter = NW * gridDim.x;
  const int full = count / per_iter, rest = count - full * per_iter;
  const int iters = full + (rest > 0 ? 1 : 0);
  const int q_last = (rest + (int)gridDim.x - 1) / (int)gridDim.x;
  auto slot = [&](int it) -> int {                                     // as in the x3 kernel: last partial round spread out
    if (it < full) return (it * (int)gridDim.x + (int)blockIdx.x) * NW + wave;
    const int r = (int)blockIdx.x * q_last + wave;
    return (it == full && wave < q_last && r < rest) ? full * per_iter + r : count;
  };
  constexpr float LOG2E = 1.4426950408889634f;
  typedef const char* cptr;
  auto ldo = [](cptr base, unsigned off) { return *reinterpret_cast<const f32x4*>(base + off); };

  // this wave's share (pieces wave, wave + 12, ...) of weight image `img` (index in the endless phase sequence) into
  // buffer img mod 3; the image's size follows from its phase
  auto stage_share = [&](int img) {
#ifdef SCGG_ABL_NO_DMA
    return;
#endif
    const int ph = img % NPH;
    const int pieces = (ph < NXB ? G::CB_F4 : G::FH_F4) / 64;
    f32x4* buf = wbuf + (img % 3) * G::PH_F4;
    unsigned voff = (unsigned)(wave * 64 + lane) * 16u;
    asm volatile("" : "+v"(voff));
    typedef const __attribute__((address_space(1))) char* gptr;
    const gptr src = (gptr)(reinterpret_cast<const char*>(a.phase_img + (size_t)ph * G::PH_F4));
#pragma unroll
    for (int s = 0; s < (G::PH_F4 / 64 + NW - 1) / NW; ++s) {
      const int piece = wave + NW * s;                                 // wave-uniform
      if (piece < pieces)
        lds_dma16((const char*)src + (size_t)s * NW * 1024 + voff, lds_addr_of(buf + piece * 64));
    }
  };

  struct Unit {
    bool active;
    int u, row, b, i, j0;
  };
  auto unit_of = [&](int uidx) {
    Unit t;
    t.active = uidx < count;
    t.u = t.active ? __builtin_amdgcn_readfirstlane(a.units[uidx]) : 0;
    t.row = t.u / nch;
    t.j0 = (t.u - t.row * nch) * 16;
    t.b = t.row / n;
    t.i = t.row - t.b * n;
    return t;
  };
  auto mask_byte = [&](const Unit& t) -> int {
    const int j = t.j0 + p;
    return (t.active && j < n) ? (int)a.mask[t.b * n + j] : 0;
  };
  // The unit's row vectors (Q row | y_e_add | y_e_mul + 1: 3 DX floats) go global -> LDS by LDS-DMA, 1 KiB per
  // instruction, no staging registers; only the issuing wave reads them, so its own vmcnt wait orders the reads.
  // 1/sqrt(df) is applied where the query is used.
  auto vec_dma = [&](const Unit& t) {
    typedef const __attribute__((address_space(1))) char* gptr;
    const gptr qrow = (gptr)reinterpret_cast<cptr>(a.Q + (long long)t.row * a.ldqkv);
    const gptr film = (gptr)reinterpret_cast<cptr>(a.film + (long long)t.b * (4 * DX));
    unsigned voff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(voff));
#pragma unroll
    for (int k = 0; k < (3 * DX + 255) / 256; ++k) {
      const int c0 = 256 * k;                                           // first float of this 1 KiB piece
      if (c0 + 256 <= DX)
        lds_dma16((const char*)qrow + 4 * c0 + voff, lds_addr_of(vecw + c0));
      else if (c0 >= DX && c0 + 256 <= 3 * DX)
        lds_dma16((const char*)film + 4 * (c0 - DX) + voff, lds_addr_of(vecw + c0));
      else {                                                            // piece straddles a boundary (DX not a multiple of 256)
        const int c = c0 + lane * 4;
        if (c < 3 * DX) {
          const f32x4 t4 = c < DX ? *reinterpret_cast<const f32x4*>(reinterpret_cast<cptr>(a.Q + (long long)t.row * a.ldqkv) + 4 * c)
                                  : *reinterpret_cast<const f32x4*>(reinterpret_cast<cptr>(a.film + (long long)t.b * (4 * DX)) + 4 * (c - DX));
          *reinterpret_cast<f32x4*>(vecw + c) = t4;
        }
      }
    }
  };
  auto tile_load = [&](const Unit& t, int jj, f32x4 (&x0)[IBE], f32x4 (&x1)[IBE]) {
    const cptr bij = reinterpret_cast<cptr>(a.E_in + (long long)t.row * n * DE);
    const cptr bji = reinterpret_cast<cptr>(a.E_in + ((long long)t.b * n * n + t.i) * DE);
    const unsigned oij = 4u * (jj * DE + 4 * g), oji = 4u * (jj * n * DE + 4 * g);
#pragma unroll
    for (int eb = 0; eb < IBE; ++eb) {
      x0[eb] = ldo(bij, oij + 64u * eb);
      x1[eb] = ldo(bji, oji + 64u * eb);
    }
  };

  // ---- prologue: images 0 and 1 (all waves), image 2 (group 2's share: the others issue theirs in S2 of phase 0)
  stage_share(0);
  stage_share(1);
  if (grp == 2) stage_share(2);

  Unit un = unit_of(slot(0));
  int mk = mask_byte(un);
  f32x4 e[IBE];
  f32x4 nx0[IBE], nx1[IBE];                                            // the next unit's E tile, in flight
  {
    const int j = un.j0 + p, jj = j < n ? j : 0;
    vec_dma(un);
    tile_load(un, jj, nx0, nx1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SCGG_SUBSTEP_BARRIER();
  for (int k = 0; k < grp; ++k) SCGG_SUBSTEP_BARRIER();                       // the stagger: group g starts g sub-steps late

  int ph_glob = 0;                                                     // this wave's phase in the endless image sequence
  const int dma_ahead = grp == 2 ? 3 : 2;
  SCGG_STAMP_DECL
  for (int it = 0; it < iters; ++it) {
    const bool active = un.active;
    const int j = un.j0 + p;
    const bool inb = j < n, vj = mk != 0, live = vj && j != un.i;
    const int jj = inb ? j : 0;
    const Unit nx = unit_of(slot(it + 1));
    const int nmk = mask_byte(nx);

    const cptr kbase = reinterpret_cast<cptr>(a.K + (long long)un.b * n * a.ldqkv);
    const cptr vbase = reinterpret_cast<cptr>(a.V + (long long)un.b * n * a.ldqkv);
    const unsigned kvoff = 4u * (jj * a.ldqkv + 4 * g);
    const float *b_out = a.b_out, *ln1_g = a.ln1_g, *ln1_b = a.ln1_b, *ln2_g = a.ln2_g, *ln2_b = a.ln2_b;
    asm volatile("" : "+s"(b_out), "+s"(ln1_g), "+s"(ln1_b), "+s"(ln2_g), "+s"(ln2_b));
    f32x4 accO[IBE];
    u32x4 eh[QE], em[QE], el[QE];
    if (active) {
      // the unit's symmetrised E tile (requested one sub-step ago, or in the prologue).  Its row vectors were requested
      // BEFORE the tile (vec_dma, then tile_load): vmcnt retires in issue order, so the wait the compiler places in front
      // of the first use of the tile also covers the LDS-DMA of the vectors, which are first read in S1.
#pragma unroll
      for (int eb = 0; eb < IBE; ++eb) e[eb] = live ? 0.5f * (nx0[eb] + nx1[eb]) : splat4(0.f);
#pragma unroll
      for (int eb = 0; eb < IBE; ++eb) accO[eb] = ld4(b_out + 16 * eb + 4 * g);
#pragma unroll
      for (int q = 0; q < QE; ++q) split3(e[2 * q], e[2 * q + 1], eh[q], em[q], el[q]);
    }
    // e is dead from here to LayerNorm 1: the exact split is summed back there (16 registers less in the S0 - S2 sub-steps)
    SCGG_STAMP_AT(0)                                                     // 0: unit set-up (loads consumed, split)

    // ================================================================== channel-block phases
#pragma unroll 1
    for (int cb = 0; cb < NXB; ++cb, ++ph_glob) {
      const f32x4* img = wbuf + (ph_glob % 3) * G::PH_F4;
      const float* bias = reinterpret_cast<const float*>(img);
      const f32x4* W = img + G::BIAS_F4;
      f32x4 k[2], vn[2], a1[2], a2[2];
      // ---------------------------------------------------------------- S0
      if (active) {
        SCGG_LANE
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          k[blk] = ldo(kbase, kvoff + 4u * (32 * cb + 16 * blk));
          vn[blk] = ldo(vbase, kvoff + 4u * (32 * cb + 16 * blk));
        }
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
          a1[ob] = ld4(bias + 16 * ob + 4 * g_);                        // e_mul bias + 1
          a2[ob] = ld4(bias + 32 + 16 * ob + 4 * g_);                   // e_add bias
        }
        block_pipeline_deep<4 * QE>(W, lane_, [&](int t, const Frag3& w) {
          const int rem = t % (2 * QE), q = rem >> 1, ob = rem & 1;
          if (t < 2 * QE) mfma_x3(a1[ob], w, eh[q], em[q], el[q]);      // e_mul
          else mfma_x3(a2[ob], w, eh[q], em[q], el[q]);                 // e_add
        });
      }
      SCGG_STAMP_AT(1)                                                   // 1: S0 work
      SCGG_SUBSTEP_BARRIER();
      SCGG_STAMP_AT(2)                                                   // 2: barrier after S0
      // ---------------------------------------------------------------- S1
      f32x4 y[2], v[2];
      if (active) {
        f32x4 z[2];
        Frag3 fa, fb;
        SCGG_LANE
        load_frag3(fa, W + 4 * QE * 192, lane_);                         // first e_out block: in flight under the FiLM
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const int co = 32 * cb + 16 * blk + 4 * g_;
          const f32x4 q = ld4(vecw + co);                               // Q_i
          const f32x4 yea = ld4(vecw + DX + co);
          const f32x4 yem = ld4(vecw + 2 * DX + co);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float kk = vj ? k[blk][r] * a.inv_sqrt_df : 0.f;
            const float yy = (q[r] * kk) * a1[blk][r] + a2[blk][r];     // Y = QK/sqrt(df) * (E1+1) + E2
            z[blk][r] = yea[r] + yem[r] * yy;                           // FiLM by y before e_out
            y[blk][r] = vj ? yy : -INFINITY;
            v[blk][r] = vj ? vn[blk][r] : 0.f;
          }
        }
        u32x4 zh, zm, zl;
        split3(z[0], z[1], zh, zm, zl);
        block_pipeline<IBE, 0>(W + 4 * QE * 192, nullptr, fa, fb, lane_,
                               [&](int t, const Frag3& w) { mfma_x3(accO[t], w, zh, zm, zl); });
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // retires this wave's weight DMA (see the header)
      SCGG_STAMP_AT(3)                                                   // 3: S1 work
      SCGG_SUBSTEP_BARRIER();
      SCGG_STAMP_AT(4)                                                   // 4: barrier after S1
      // ---------------------------------------------------------------- S2
      stage_share(ph_glob + dma_ahead);
      if (active) {
        float mloc = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, y[blk][r]);
        const float M = row_max16_fused(mloc);
        const float nM = M == -INFINITY ? 0.f : -M * LOG2E;
        f32x4 cm[2], cs[2], ca[2];
        bool under = false;
        {
          float red[16];
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pe = __builtin_amdgcn_exp2f(fmaf(y[blk][r], LOG2E, nM));
              red[4 * blk + r] = pe;
              red[8 + 4 * blk + r] = pe * v[blk][r];
            }
          row_sum16_many<16>(red);
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              cs[blk][r] = red[4 * blk + r];
              ca[blk][r] = red[8 + 4 * blk + r];
              cm[blk][r] = M;
              under |= cs[blk][r] == 0.f;
            }
        }
        if (__any(under && M > -INFINITY)) {
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float yy = y[blk][r];
              const float mx = row_max16(yy);
              const float ex = yy == -INFINITY ? 0.f : __expf(yy - mx);
              cm[blk][r] = mx;
              cs[blk][r] = row_sum16(ex);
              ca[blk][r] = row_sum16(ex * v[blk][r]);
            }
        }
        if (p == 0) {
          float* dst = a.attn_part + (long long)un.u * (3 * DX) + 32 * cb + 4 * g;
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
            *reinterpret_cast<f32x4*>(dst + 16 * blk) = cm[blk];
            *reinterpret_cast<f32x4*>(dst + DX + 16 * blk) = cs[blk];
            *reinterpret_cast<f32x4*>(dst + 2 * DX + 16 * blk) = ca[blk];
          }
        }
      }
      SCGG_STAMP_AT(5)                                                   // 5: S2 work
      SCGG_SUBSTEP_BARRIER();
      SCGG_STAMP_AT(6)                                                   // 6: barrier after S2
    }

    // ================================================================== E <- LN1(E + newE), FFN half phases
#pragma unroll
    for (int h = 0; h < NFB; ++h, ++ph_glob) {
      const f32x4* img = wbuf + (ph_glob % 3) * G::PH_F4;
      const float* bias = reinterpret_cast<const float*>(img);
      const f32x4* W = img + G::BIAS_F4;
      f32x4 hid[2];
      // ---------------------------------------------------------------- F0
      if (active) {
        SCGG_LANE
        Frag3 fa, fb;
        load_frag3(fa, W, lane_);
        if (h == 0) {
#pragma unroll
          for (int q = 0; q < QE; ++q) unsplit3(eh[q], em[q], el[q], e[2 * q], e[2 * q + 1]);
#pragma unroll
          for (int eb = 0; eb < IBE; ++eb) e[eb] += accO[eb];
          layernorm_c16<IBE>(e, ln1_g, ln1_b, g_);
#pragma unroll
          for (int q = 0; q < QE; ++q) split3(e[2 * q], e[2 * q + 1], eh[q], em[q], el[q]);
#pragma unroll
          for (int eb = 0; eb < IBE; ++eb) e[eb] += ld4(bias + 64 + 16 * eb + 4 * g_);  // linE2 bias
        }
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) hid[ob] = ld4(bias + 16 * ob + 4 * g_);
        block_pipeline<2 * QE, 0>(W, nullptr, fa, fb, lane_, [&](int t, const Frag3& w) {
          mfma_x3(hid[t & 1], w, eh[t >> 1], em[t >> 1], el[t >> 1]);
        });
      }
      SCGG_STAMP_AT(7)                                                   // 7: F0 work
      SCGG_SUBSTEP_BARRIER();
      SCGG_STAMP_AT(8)                                                   // 8: barrier after F0
      // ---------------------------------------------------------------- F1
      if (active) {
        SCGG_LANE
        Frag3 fa, fb;
        load_frag3(fa, W + 2 * QE * 192, lane_);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
          for (int r = 0; r < 4; ++r) hid[ob][r] = fmaxf(hid[ob][r], 0.f);
        u32x4 hh, hm, hl;
        split3(hid[0], hid[1], hh, hm, hl);
        block_pipeline<IBE, 0>(W + 2 * QE * 192, nullptr, fa, fb, lane_,
                               [&](int t, const Frag3& w) { mfma_x3(e[t], w, hh, hm, hl); });
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // retires this wave's weight DMA
      SCGG_STAMP_AT(9)                                                   // 9: F1 work
      SCGG_SUBSTEP_BARRIER();
      SCGG_STAMP_AT(10)                                                  // 10: barrier after F1
      // ---------------------------------------------------------------- F2
      stage_share(ph_glob + dma_ahead);
      if (h == NFB - 1) {
        {   // the next unit's requests, consumed after the barrier.  Unconditional (a wave without a next unit reads unit
            // 0's rows, which exist): a conditional definition would keep the 44 registers live through the whole round.
          const int nj = nx.j0 + p, njj = nj < n ? nj : 0;
          vec_dma(nx);
          tile_load(nx, njj, nx0, nx1);
        }
        if (active) {
          layernorm_c16<IBE>(e, ln2_g, ln2_b, g);
          if (inb) {
            float* dst = a.E_out + ((long long)un.row * n + jj) * DE + 4 * g;
#pragma unroll
            for (int eb = 0; eb < IBE; ++eb) *reinterpret_cast<f32x4*>(dst + 16 * eb) = live ? e[eb] : splat4(0.f);
          }
        }
      }
      SCGG_STAMP_AT(11)                                                  // 11: F2 work
      SCGG_SUBSTEP_BARRIER();
      SCGG_STAMP_AT(12)                                                  // 12: barrier after F2
    }
    un = nx;
    mk = nmk;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // DMA issued for phases nobody runs: let it land before exit
  for (int k = grp; k < 2; ++k) SCGG_SUBSTEP_BARRIER();                         // every wave has passed the same number of barriers
#ifdef SCGG_STAMP
  {
    const int slot_ = blockIdx.x == 0 ? 0 : blockIdx.x == 1 ? 1 : blockIdx.x == gridDim.x / 2 ? 2 : blockIdx.x == gridDim.x - 1 ? 3 : -1;
    if (slot_ >= 0 && lane == 0) {
#pragma unroll
      for (int k = 0; k < STAMP_SEGS; ++k) g_stamps[slot_][wave][k] = st_acc[k];
      g_stamps[slot_][wave][15] = (unsigned long long)iters;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// newX_pre[row] = y_x_add + (y_x_mul + 1) * softmax-weighted V, merged over the row's units
// (transformer_model.py:207-214, :230-232).  One block per valid row, one thread per channel.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_finish_kernel(AttnFinishArgs a) {
  __shared__ unsigned cv;
  const int ridx = blockIdx.x;
  if (ridx >= *a.row_count) return;
  const int row = a.rows[ridx];
  const int n = a.n, DX = a.dx, nch = (n + 15) >> 4;
  const int b = row / n;
  if (threadIdx.x == 0) {
    unsigned bits = 0;
    for (int jc = 0; jc < nch; ++jc) {
      bool any = false;
      for (int j = jc * 16; j < min(n, jc * 16 + 16); ++j) any |= a.mask[b * n + j] != 0;
      bits |= any ? (1u << jc) : 0u;
    }
    cv = bits;
  }
  __syncthreads();
  const unsigned bits = cv;
  const float* film = a.film + (long long)b * (4 * DX);
  for (int c = threadIdx.x; c < DX; c += blockDim.x) {
    float M = -INFINITY, S = 0.f, A = 0.f;
    for (int jc = 0; jc < nch; ++jc) {
      if (!((bits >> jc) & 1u)) continue;
      const float* pt = a.attn_part + ((long long)row * nch + jc) * (3 * DX) + c;
      const float m = pt[0], s = pt[DX], ac = pt[2 * DX];
      const float mn = fmaxf(M, m);
      const float fo = M == -INFINITY ? 0.f : __expf(M - mn);
      const float fn = m == -INFINITY ? 0.f : __expf(m - mn);
      S = S * fo + s * fn;
      A = A * fo + ac * fn;
      M = mn;
    }
    a.newX_pre[(long long)row * DX + c] = film[2 * DX + c] + film[3 * DX + c] * (A / S);
  }
}

int launch_attn_finish(const AttnFinishArgs& a, int M, hipStream_t st) {
  hipLaunchKernelGGL(attn_finish_kernel, dim3(M), dim3(256), 0, st, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Etoy pooling of a hidden E (layers.py:34-41): z[b] = [mean | min | max | biased variance] per channel over the n_b x n_b
// pairs of the valid nodes of the SYMMETRISED masked tensor (the stored one is not: mfma_chain.h load_sym_tile), zero
// diagonal included.  One workgroup per graph; thread = (channel, slice of the unordered pairs); each stored element
// is read once (pair {i,j} = 0.5 (E_ij + E_ji), weight 2); one pass, shifted sums merged pairwise (see below).
// ---------------------------------------------------------------------------------------------------------
constexpr int POOL_T = PS_T;  // threads per graph (pool_stats.h, where the arithmetic lives: y_layer_kernel shares it)
__global__ __launch_bounds__(POOL_T) void edge_pool_stats_kernel(EdgePoolArgs a) {
  __shared__ PoolScratch ps;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nv = pool_compact_mask(ps, a.mask + (long long)b * a.n, a.n, tid);
  pool_etoy_graph(ps, a.E + (long long)b * a.n * a.n * a.de, a.n, nv, a.de, a.z + (long long)b * a.z_stride + a.z_offset, tid);
}

// several workgroups per graph (pool_stats.h): grid (W, B) parts, then one workgroup per graph merges them
__global__ __launch_bounds__(POOL_T) void edge_pool_part_kernel(EdgePoolArgs a, float* __restrict__ parts, int W) {
  __shared__ PoolScratch ps;
  const int w = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int nv = pool_compact_mask(ps, a.mask + (long long)b * a.n, a.n, tid);
  pool_etoy_part(ps, a.E + (long long)b * a.n * a.n * a.de, a.n, nv, a.de, w, W, parts + ((size_t)b * W + w) * 5 * a.de, tid);
}
__global__ __launch_bounds__(POOL_T) void edge_pool_finish_kernel(EdgePoolArgs a, const float* __restrict__ parts, int W) {
  __shared__ int s_nv;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int c = 0;
    for (int i = 0; i < a.n; ++i) c += a.mask[(long long)b * a.n + i] != 0;
    s_nv = c;
  }
  __syncthreads();
  pool_etoy_finish(parts + (size_t)b * W * 5 * a.de, W, s_nv, a.de, a.z + (long long)b * a.z_stride + a.z_offset, tid, POOL_T);
}
int launch_edge_pool_stats_sliced(const EdgePoolArgs& a, float* parts, int W, hipStream_t st) {
  if (a.de > 256 || a.de % 4 != 0 || POOL_T % (a.de / 4) != 0 || a.n > 512 || W < 1 || parts == nullptr) return -1;
  hipLaunchKernelGGL(edge_pool_part_kernel, dim3(W, a.B), dim3(POOL_T), 0, st, a, parts, W);
  hipLaunchKernelGGL(edge_pool_finish_kernel, dim3(a.B), dim3(POOL_T), 0, st, a, parts, W);
  return 0;
}

int launch_edge_pool_stats(const EdgePoolArgs& a, hipStream_t st) {
  if (a.de > 256 || a.de % 4 != 0 || POOL_T % (a.de / 4) != 0 || a.n > 512) return -1;
  hipLaunchKernelGGL(edge_pool_stats_kernel, dim3(a.B), dim3(POOL_T), 0, st, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
#define SCGG_UNIT_CASES CASE(8, 2, 4) CASE(4, 2, 4) CASE(1, 1, 2)

// SCGG_EDGE_MODE (development A/B switch, read once): "f32" = the plain fp32-MFMA kernel, "x3s" = the staggered bf16x3
// kernel (round 2 experiment: correct, 13-15 % slower), default = "x3", the bf16x3 kernel of round 1
enum EdgeMode { EDGE_F32 = 0, EDGE_X3 = 1, EDGE_X3S = 2 };
static int edge_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("SCGG_EDGE_MODE");
    mode = (e == nullptr || e[0] == 0) ? EDGE_X3 : (e[0] == 'f' ? EDGE_F32 : (e[0] == 'x' && e[1] == '3' && e[2] == 's' ? EDGE_X3S : EDGE_X3));
  }
  return mode;
}
static bool edge_x3() { return edge_mode() == EDGE_X3; }

long long edge_phase_img_f4(int dx, int de, int ffe) {
#define CASE(NXB, NEB, NFB)                                                                                   \
  if (dx == 32 * NXB && de == 32 * NEB && ffe == 32 * NFB)                                                    \
    return edge_mode() == EDGE_X3S ? (long long)UnitGeomX3S<NXB, NEB, NFB>::NPH * UnitGeomX3S<NXB, NEB, NFB>::PH_F4 \
           : edge_x3() ? (long long)UnitGeomX3<NXB, NEB, NFB>::NPH * UnitGeomX3<NXB, NEB, NFB>::PH_F4           \
                       : (long long)UnitGeom<NXB, NEB, NFB>::NPH * UnitGeom<NXB, NEB, NFB>::PH_F4;
  SCGG_UNIT_CASES
#undef CASE
  return -1;
}

int launch_edge_unit_pack(const EdgeUnitPackArgs& a, int dx, int de, int ffe, hipStream_t st) {
#define CASE(NXB, NEB, NFB)                                                                                        \
  if (dx == 32 * NXB && de == 32 * NEB && ffe == 32 * NFB) {                                                       \
    if (edge_mode() == EDGE_X3S)                                                                                   \
      hipLaunchKernelGGL((edge_unit_pack_x3s_kernel<NXB, NEB, NFB>), dim3(UnitGeomX3S<NXB, NEB, NFB>::NPH), dim3(512), 0, st, a); \
    else if (edge_x3())                                                                                            \
      hipLaunchKernelGGL((edge_unit_pack_x3_kernel<NXB, NEB, NFB>), dim3(UnitGeomX3<NXB, NEB, NFB>::NPH), dim3(512), 0, st, a); \
    else                                                                                                           \
      hipLaunchKernelGGL((edge_unit_pack_kernel<NXB, NEB, NFB>), dim3(UnitGeom<NXB, NEB, NFB>::NPH), dim3(512), 0, st, a); \
    return 0;                                                                                                      \
  }
  SCGG_UNIT_CASES
#undef CASE
  return -1;
}

#ifndef SCGG_UNIT_WAVES
#define SCGG_UNIT_WAVES 8      // fp32 kernel: 2 waves per SIMD (up to 256 VGPRs)
#endif
#ifndef SCGG_X3_WAVES
#define SCGG_X3_WAVES 12       // x3 kernel: 3 waves per SIMD (168 VGPRs; 149 used, no scratch)
#endif

template <int NXB, int NEB, int NFB>
static int launch_one(const EdgeLayerArgs& a, int n_cu, hipStream_t st) {
  static LdsAttrOnce attr_x3, attr_f32, attr_x3s;
  if (edge_mode() == EDGE_X3S) {
    auto kern = edge_unit_layer_x3s_kernel<NXB, NEB, NFB>;
    const size_t lds = UnitGeomX3S<NXB, NEB, NFB>::lds_bytes();
    if (set_max_dynamic_lds(attr_x3s, reinterpret_cast<const void*>(kern), lds)) return -2;
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(UnitGeomX3S<NXB, NEB, NFB>::NW * 64), lds, st, a);   // one persistent workgroup per CU
  } else if (edge_x3()) {
    constexpr int NW = SCGG_X3_WAVES;
    auto kern = edge_unit_layer_x3_kernel<NXB, NEB, NFB, NW>;
    const size_t lds = UnitGeomX3<NXB, NEB, NFB>::lds_bytes(NW);
    if (set_max_dynamic_lds(attr_x3, reinterpret_cast<const void*>(kern), lds)) return -2;
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(NW * 64), lds, st, a);        // one persistent workgroup per CU
  } else {
    constexpr int NW = SCGG_UNIT_WAVES;
    auto kern = edge_unit_layer_kernel<NXB, NEB, NFB, NW>;
    const size_t lds = UnitGeom<NXB, NEB, NFB>::lds_bytes(NW);
    if (set_max_dynamic_lds(attr_f32, reinterpret_cast<const void*>(kern), lds)) return -2;
    hipLaunchKernelGGL(kern, dim3(n_cu * (16 / NW)), dim3(NW * 64), lds, st, a);
  }
  return 0;
}

int launch_edge_layer(const EdgeLayerArgs& a_in, int dx, int de, int ffe, hipStream_t st) {
  const int n_cu = device_cu_count();
  if (n_cu <= 0) return -2;
  static const bool xcd_on = [] { const char* e = getenv("SCGG_EDGE_XCD"); return !(e != nullptr && e[0] == '0'); }();
  EdgeLayerArgs a = a_in;
  a.xcd_map = xcd_on ? 1 : 0;
#define CASE(NXB, NEB, NFB) \
  if (dx == 32 * NXB && de == 32 * NEB && ffe == 32 * NFB) return launch_one<NXB, NEB, NFB>(a, n_cu, st);
  SCGG_UNIT_CASES
#undef CASE
  return -1;
}

// =========================================================================================================
// Stress configuration (BASELINE.json configs[4]): the NodeEdgeBlock alone on a bf16 edge tensor.
//
//   E_in / newE are stored as bf16 [B,n,n,de]; Q, K, V, the FiLM vectors and every accumulator are fp32; weights are
//   rounded to bf16 once at pack time; each product is ONE v_mfma_f32_16x16x32_bf16 (no three-term split).  The tile of a
//   lane is read from HBM already in MFMA operand form (two 8-byte loads per 32-channel chunk) and newE leaves as 8-byte
//   stores: 2 de bytes in and 2 de bytes out per pair -- the "E-tensor HBM stream" of the north star: 4 de B per pair
//   against 6 de dx flop per pair (196 608 flop / 512 B at de = 128, dx = 256: 384 flop/B, above the bf16 ridge of ~310).
//   Semantics = NodeEdgeBlock.forward (transformer_model.py:165-248) without the y update: newE = e_out(...) * m_i m_j
//   (the diagonal is NOT removed there), attention partials per (unit, channel) as in the layer kernels.  The input is
//   used as given (the block of the reference does not symmetrise).  Same units, lanes, phases and LDS-DMA double buffer as
//   edge_unit_layer_x3_kernel, channel-block phases only.
// =========================================================================================================
template <int NXB, int NEB>
struct BlockGeomBF {
  static constexpr int DX = 32 * NXB, DE = 32 * NEB, IBE = DE / 16, QE = NEB;
  static constexpr int B_CB = 4 * QE + IBE;                  // e_mul, e_add (q-major, two row blocks each), e_out slice
  static constexpr int BIAS_F4 = 64;
  static constexpr int PH_F4 = BIAS_F4 + B_CB * 64;          // one bf16 block = 64 lanes x 16 B
  // When all NXB phase images fit LDS beside the row vectors and partials (vessel dims: 53 + 37 KB) they are loaded once per
  // workgroup and the unit loop has NO barrier: the waves of a SIMD drift apart, one's vector work meets another's MFMAs.
  static constexpr bool resident(int nw) { return (size_t)NXB * PH_F4 * 16 + (size_t)nw * 6 * DX * 4 <= 144 * 1024; }
  static constexpr size_t lds_bytes(int nw) {                // images, row vectors, partials
    return (size_t)(resident(nw) ? NXB : 2) * PH_F4 * 16 + (size_t)nw * 6 * DX * 4;
  }
};

__device__ __forceinline__ unsigned bf16_rn(float x) {       // round to nearest even, as torch's .to(bfloat16)
  const unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// two fp32 -> packed bf16, round to nearest even (gfx950 has the instruction: v_cvt_pk_bf16_f32)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_rn(float x0, float x1) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
}

struct EdgeBlockPackArgs { const float *W_mul, *b_mul, *W_add, *b_add, *W_out; u32x4* img; };

template <int NXB, int NEB>
__global__ __launch_bounds__(512) void edge_block_pack_bf16_kernel(EdgeBlockPackArgs a) {
  using G = BlockGeomBF<NXB, NEB>;
  constexpr int DX = G::DX, DE = G::DE, IBE = G::IBE, QE = G::QE;
  const int cb = blockIdx.x;
  u32x4* img = a.img + (size_t)cb * G::PH_F4;
  for (int idx = threadIdx.x; idx < G::PH_F4; idx += blockDim.x) {
    u32x4 out = {0u, 0u, 0u, 0u};
    if (idx >= G::BIAS_F4) {
      const int bi = idx - G::BIAS_F4, t = bi >> 6, l = bi & 63;
      const float* W; int row0, col0, K;
      if (t < 4 * QE) {
        W = t < 2 * QE ? a.W_mul : a.W_add;
        const int rem = t % (2 * QE);
        row0 = 32 * cb + 16 * (rem & 1); col0 = 32 * (rem >> 1); K = DE;
      } else { W = a.W_out; row0 = 16 * (t - 4 * QE); col0 = 32 * cb; K = DX; }
      const int o = row0 + (l & 15), g = l >> 4;
      unsigned hw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) hw[i] = bf16_rn(W[(size_t)o * K + col0 + (i < 4 ? 4 * g + i : 16 + 4 * g + i - 4)]);
#pragma unroll
      for (int d = 0; d < 4; ++d) out[d] = hw[2 * d] | (hw[2 * d + 1] << 16);
    } else if (idx < 16) {
      const int fo = idx * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        out[e] = __float_as_uint(fo < 32 ? a.b_mul[32 * cb + fo + e] + 1.0f : a.b_add[32 * cb + fo - 32 + e]);
    }
    img[idx] = out;
  }
}

struct EdgeBlockArgs {
  const unsigned short* E_in;   // bf16 [B,n,n,de]
  unsigned short* E_out;        // bf16 [B,n,n,de]: newE of the valid rows' pairs (0 beyond the valid keys); other rows untouched
  const uint8_t* mask;
  const float *Q, *K, *V;       // [B,n,dx] with row stride ldqkv, masked by the caller as the reference does (:181-183)
  int ldqkv;
  const float* film;            // [B, 4 dx] = [y_e_add | y_e_mul+1 | y_x_add | y_x_mul+1]
  const f32x4* phase_img;
  const float* b_out;           // e_out.bias
  float* attn_part;             // [B*n*nch, 3, dx]
  const int* units;
  const int* unit_count;
  float inv_sqrt_df;
  int B, n;
};

template <int NXB, int NEB, int NW>
__global__ __launch_bounds__(NW * 64) void edge_block_bf16_kernel(EdgeBlockArgs a) {
  using G = BlockGeomBF<NXB, NEB>;
  constexpr int DX = G::DX, DE = G::DE, IBE = G::IBE, QE = G::QE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool RES = G::resident(NW);
  f32x4* wbuf = reinterpret_cast<f32x4*>(smem);
  float* fl = reinterpret_cast<float*>(wbuf + (RES ? NXB : 2) * G::PH_F4);
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* vecw = fl + wave * (3 * DX);
  // The unit's attention partials are collected in LDS and leave as three 1-KiB stores at the END of the unit.  Stored from
  // inside a phase (16 bytes from 4 lanes) they were still in flight at the next phase barrier, whose vmcnt(0) -- stores
  // count in vmcnt on gfx9 -- then waited out a full store latency in every phase.
  float* statw = fl + NW * (3 * DX) + wave * (3 * DX);
  const int n = a.n, nch = (n + 15) >> 4;
  const int count = *a.unit_count;
  const int per_iter = NW * gridDim.x;
  const int iters = (count + per_iter - 1) / per_iter;
  constexpr float LOG2E = 1.4426950408889634f;
  typedef const char* cptr;
  auto ldo = [](cptr base, unsigned off) { return *reinterpret_cast<const f32x4*>(base + off); };
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define SCGG_BF(v) __builtin_bit_cast(bf16x8, v)

  if (RES) {
#pragma unroll 1
    for (int cb = 0; cb < NXB; ++cb)
      stage_phase_unit<G::PH_F4, NW>(a.phase_img + (size_t)cb * G::PH_F4, wbuf + cb * G::PH_F4, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // the only barrier of the kernel in this mode
  } else {
    stage_phase_unit<G::PH_F4, NW>(a.phase_img, wbuf, wave, lane);
  }
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int uidx = (it * (int)gridDim.x + (int)blockIdx.x) * NW + wave;
    const bool active = uidx < count;
    if (RES && !active) break;                             // no barrier ahead: a wave without a unit is done
    const int u = active ? __builtin_amdgcn_readfirstlane(a.units[uidx]) : 0;
    const int row = u / nch, j0 = (u - row * nch) * 16, b = row / n;
    const int j = j0 + p;
    const bool inb = j < n;
    const int jj = inb ? j : 0;
    const bool vj = active && inb && a.mask[b * n + jj] != 0;
    u32x4 eq[QE];
    f32x4 accO[IBE];
    if (active) {
      // row vectors: Q_i / sqrt(df) | y_e_add | y_e_mul + 1 (this wave's LDS), then the E tile in operand form
      const cptr qrow = reinterpret_cast<cptr>(a.Q + (long long)row * a.ldqkv);
      const cptr film = reinterpret_cast<cptr>(a.film + (long long)b * (4 * DX));
      for (int c = lane * 4; c < 3 * DX; c += 256) {
        f32x4 v = c < DX ? ldo(qrow, 4u * c) * a.inv_sqrt_df : ldo(film, 4u * (c - DX));
        *reinterpret_cast<f32x4*>(vecw + c) = v;
      }
      const unsigned short* e = a.E_in + ((long long)row * n + jj) * DE + 4 * g;
#pragma unroll
      for (int q = 0; q < QE; ++q) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(e + 32 * q), hi = *reinterpret_cast<const u32x2*>(e + 32 * q + 16);
        eq[q] = vj ? u32x4{lo[0], lo[1], hi[0], hi[1]} : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int eb = 0; eb < IBE; ++eb) accO[eb] = ld4(a.b_out + 16 * eb + 4 * g);
    }
    const cptr kbase = reinterpret_cast<cptr>(a.K + (long long)b * n * a.ldqkv);
    const cptr vbase = reinterpret_cast<cptr>(a.V + (long long)b * n * a.ldqkv);
    // no select per K / V element: a lane without a key (logit -inf, weight 0) reads the query's own row, which is written
    const unsigned kvoff = 4u * ((vj ? j : (active ? row - b * n : 0)) * a.ldqkv + 4 * g);
    // K / V rows of the lane's key, one phase AHEAD: a load that is consumed in the phase that issues it is younger than the
    // phase's weight DMA, so waiting for it (vmcnt counts in order) waited for the DMA to land as well -- a full memory
    // latency in the middle of every phase.  Requested a phase early they are covered by the vmcnt(0) in front of the
    // barrier, which the DMA needs anyway.
    f32x4 kc[2], vc[2];
    if (active) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        kc[blk] = ldo(kbase, kvoff + 4u * (16 * blk));
        vc[blk] = ldo(vbase, kvoff + 4u * (16 * blk));
      }
    }
#pragma unroll 1
    for (int cb = 0; cb < NXB; ++cb) {
      if (!RES) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the phase image (and the unit's loads) has landed
        __syncthreads();
        stage_phase_unit<G::PH_F4, NW>(a.phase_img + (size_t)((cb + 1) % NXB) * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
      }
      const int slot = RES ? cb : cur;
      const float* bias = reinterpret_cast<const float*>(wbuf + slot * G::PH_F4);
      const u32x4* W = reinterpret_cast<const u32x4*>(wbuf + slot * G::PH_F4 + G::BIAS_F4);
      if (active) {
        f32x4 k[2], vn[2], a1[2], a2[2], kn[2], vnn[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          k[blk] = kc[blk];
          vn[blk] = vc[blk];
          if (cb + 1 < NXB) {
            kn[blk] = ldo(kbase, kvoff + 4u * (32 * (cb + 1) + 16 * blk));
            vnn[blk] = ldo(vbase, kvoff + 4u * (32 * (cb + 1) + 16 * blk));
          }
          a1[blk] = ld4(bias + 16 * blk + 4 * g);
          a2[blk] = ld4(bias + 32 + 16 * blk + 4 * g);
        }
#pragma unroll
        for (int t = 0; t < 4 * QE; ++t) {                       // e_mul then e_add: blocks (q, ob), q-major
          const int rem = t % (2 * QE), q = rem >> 1, ob = rem & 1;
          const u32x4 w = W[t * 64 + lane];
#ifdef SCGG_ABL_BF_NO_MFMA      // timing build: operands stay live, the matrix pipe idles
          if (t < 2 * QE) a1[ob][0] += __uint_as_float((w[0] ^ eq[q][0]) & 0x3f800000u);
          else a2[ob][0] += __uint_as_float((w[1] ^ eq[q][1]) & 0x3f800000u);
          continue;
#endif
          if (t < 2 * QE) a1[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(w), SCGG_BF(eq[q]), a1[ob], 0, 0, 0);
          else a2[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(w), SCGG_BF(eq[q]), a2[ob], 0, 0, 0);
        }
        f32x4 y[2], v[2], z[2];
        float mloc = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const int co = 32 * cb + 16 * blk + 4 * g;
          const f32x4 q = ld4(vecw + co), yea = ld4(vecw + DX + co), yem = ld4(vecw + 2 * DX + co);
          // No select on K / V: a lane without a valid key holds finite data of some other row (the caller zeroes masked
          // rows; lanes beyond n read key 0), its logit is forced to -inf, so its weight is exactly 0, and an MFMA column
          // depends on its own lane only -- nothing of such a lane is ever stored.
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float yy = (q[r] * k[blk][r]) * a1[blk][r] + a2[blk][r];
            z[blk][r] = yea[r] + yem[r] * yy;
            y[blk][r] = yy;                                        // masked below: once per lane, not once per value
            v[blk][r] = vn[blk][r];
            mloc = fmaxf(mloc, yy);
          }
        }
        {
          u32x4 zb;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const float x0 = d < 2 ? z[0][2 * d] : z[1][2 * d - 4], x1 = d < 2 ? z[0][2 * d + 1] : z[1][2 * d - 3];
            zb[d] = pack_bf16_rn(x0, x1);
          }
#pragma unroll
          for (int eb = 0; eb < IBE; ++eb) {
            const u32x4 w = W[(4 * QE + eb) * 64 + lane];
#ifdef SCGG_ABL_BF_NO_MFMA
            accO[eb][0] += __uint_as_float((w[0] ^ zb[0]) & 0x3f800000u);
            continue;
#endif
            accO[eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SCGG_BF(w), SCGG_BF(zb), accO[eb], 0, 0, 0);
          }
        }
#ifdef SCGG_ABL_BF_NO_SM        // timing build: no attention statistics
        if (mloc == 12345.f)
#endif
        {
        const float M = row_max16_fused(vj ? mloc : -INFINITY);
        const float nM = !vj ? -INFINITY : (M == -INFINITY ? 0.f : -M * LOG2E);   // a lane without a key: exp2(-inf) = 0
        f32x4 cm[2], cs[2], ca[2];
        bool under = false;
        {
          float red[16];                                                // 8 x exp(y - M) (0 on invalid keys), 8 x p V
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pe = __builtin_amdgcn_exp2f(fmaf(y[blk][r], LOG2E, nM));
              red[4 * blk + r] = pe;
              red[8 + 4 * blk + r] = pe * v[blk][r];
            }
          // lane p of a row now gets the row's sum of red[p]: p < 8: sum p of channel (blk = p >> 2, r = p & 3), p >= 8: sum p V
          const float mine = row_reduce_scatter16(red, p);
          const int ch = 32 * cb + 16 * ((p >> 2) & 1) + 4 * g + (p & 3);
          statw[(p < 8 ? DX : 2 * DX) + ch] = mine;
          if (p < 8) statw[ch] = M;
          under = p < 8 && mine == 0.f;
        }
        (void)cm; (void)cs; (void)ca;
        if (__any(under && M > -INFINITY)) {                         // a channel underflowed against the shared shift: exact maxima
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float yy = vj ? y[blk][r] : -INFINITY;
              const float mx = row_max16(yy);
              const float ex = yy == -INFINITY ? 0.f : __expf(yy - mx);
              cm[blk][r] = mx;
              cs[blk][r] = row_sum16(ex);
              ca[blk][r] = row_sum16(yy == -INFINITY ? 0.f : ex * v[blk][r]);
            }
          if (p == 0) {
            float* dst = statw + 32 * cb + 4 * g;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
              *reinterpret_cast<f32x4*>(dst + 16 * blk) = cm[blk];
              *reinterpret_cast<f32x4*>(dst + DX + 16 * blk) = cs[blk];
              *reinterpret_cast<f32x4*>(dst + 2 * DX + 16 * blk) = ca[blk];
            }
          }
        }
        }
        if (cb + 1 < NXB) {
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) { kc[blk] = kn[blk]; vc[blk] = vnn[blk]; }
        }
      }
      cur ^= 1;
    }
    if (active) {
      float* dst = a.attn_part + (long long)u * (3 * DX);
      for (int c = lane * 4; c < 3 * DX; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(statw + c);
    }
    if (active && inb) {
      unsigned short* dst = a.E_out + ((long long)row * n + jj) * DE + 4 * g;
#pragma unroll
      for (int eb = 0; eb < IBE; ++eb) {
        u32x2 o = {0u, 0u};
        if (vj) o = u32x2{pack_bf16_rn(accO[eb][0], accO[eb][1]), pack_bf16_rn(accO[eb][2], accO[eb][3])};
        *reinterpret_cast<u32x2*>(dst + 16 * eb) = o;
      }
    }
  }
#undef SCGG_BF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA in flight when the wave ends
}

// newX_pre[row] = y_x_add + (y_x_mul + 1) * softmax-weighted V merged over the row's units, up to 64 key chunks (n <= 1024)
__global__ __launch_bounds__(256) void attn_merge64_kernel(const float* attn_part, const uint8_t* mask, const float* film_all,
                                                           float* newX_pre, const int* rows, const int* row_count, int n, int DX) {
  __shared__ unsigned long long cv;
  const int ridx = blockIdx.x;
  if (ridx >= *row_count) return;
  const int row = rows[ridx], nch = (n + 15) >> 4, b = row / n;
  if (threadIdx.x == 0) {
    unsigned long long bits = 0;
    for (int jc = 0; jc < nch; ++jc) {
      bool any = false;
      for (int j = jc * 16; j < min(n, jc * 16 + 16); ++j) any |= mask[b * n + j] != 0;
      bits |= any ? (1ull << jc) : 0ull;
    }
    cv = bits;
  }
  __syncthreads();
  const unsigned long long bits = cv;
  const float* film = film_all + (long long)b * (4 * DX);
  for (int c = threadIdx.x; c < DX; c += blockDim.x) {
    float M = -INFINITY, S = 0.f, A = 0.f;
    for (int jc = 0; jc < nch; ++jc) {
      if (!((bits >> jc) & 1ull)) continue;
      const float* pt = attn_part + ((long long)row * nch + jc) * (3 * DX) + c;
      const float m = pt[0], s = pt[DX], ac = pt[2 * DX];
      const float mn = fmaxf(M, m);
      const float fo = M == -INFINITY ? 0.f : __expf(M - mn), fn = m == -INFINITY ? 0.f : __expf(m - mn);
      S = S * fo + s * fn; A = A * fo + ac * fn; M = mn;
    }
    newX_pre[(long long)row * DX + c] = film[2 * DX + c] + film[3 * DX + c] * (A / S);
  }
}

}  // namespace scgg

using namespace scgg;

extern "C" size_t scgg_edge_block_bf16_image_bytes(int32_t dx, int32_t de) {
  if (dx == 256 && de == 128) return (size_t)8 * BlockGeomBF<8, 4>::PH_F4 * 16;
  if (dx == 256 && de == 64) return (size_t)8 * BlockGeomBF<8, 2>::PH_F4 * 16;
  if (dx == 128 && de == 64) return (size_t)4 * BlockGeomBF<4, 2>::PH_F4 * 16;      // vessel dims (discrete_vessels.yaml)
  return 0;
}

extern "C" int scgg_edge_block_bf16_pack(const float* W_mul, const float* b_mul, const float* W_add, const float* b_add,
                                         const float* W_out, int32_t dx, int32_t de, void* image, void* stream) {
  if (!W_mul || !b_mul || !W_add || !b_add || !W_out || !image) return fail("edge_block_bf16_pack: null argument");
  EdgeBlockPackArgs a{W_mul, b_mul, W_add, b_add, W_out, static_cast<u32x4*>(image)};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dx == 256 && de == 128) hipLaunchKernelGGL((edge_block_pack_bf16_kernel<8, 4>), dim3(8), dim3(512), 0, st, a);
  else if (dx == 256 && de == 64) hipLaunchKernelGGL((edge_block_pack_bf16_kernel<8, 2>), dim3(8), dim3(512), 0, st, a);
  else if (dx == 128 && de == 64) hipLaunchKernelGGL((edge_block_pack_bf16_kernel<4, 2>), dim3(4), dim3(512), 0, st, a);
  else return fail("edge_block_bf16_pack: unsupported dims dx=%d de=%d (built: 256/128, 256/64, 128/64)", dx, de);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("edge_block_bf16_pack: %s", hipGetErrorString(e));
}

template <int NXB, int NEB, int NW = 12>
static int launch_edge_block_bf16(const EdgeBlockArgs& a, hipStream_t st) {
  const int n_cu = device_cu_count();
  if (n_cu <= 0) return fail("edge_block_bf16: device query failed");
  static LdsAttrOnce attr;
  auto kern = edge_block_bf16_kernel<NXB, NEB, NW>;
  const size_t lds = BlockGeomBF<NXB, NEB>::lds_bytes(NW);
  if (set_max_dynamic_lds(attr, reinterpret_cast<const void*>(kern), lds)) return fail("edge_block_bf16: LDS attribute");
  hipLaunchKernelGGL(kern, dim3(n_cu), dim3(NW * 64), lds, st, a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("edge_block_bf16: %s", hipGetErrorString(e));
}

extern "C" int scgg_edge_block_bf16(const uint16_t* E_in, uint16_t* newE, const uint8_t* node_mask, const float* Q,
                                    const float* K, const float* V, int32_t ldqkv, const float* film, const void* image,
                                    const float* b_out, float* attn_part, const int32_t* units, const int32_t* unit_count,
                                    int32_t B, int32_t n, int32_t dx, int32_t de, int32_t n_head, void* stream) {
  if (!E_in || !newE || !node_mask || !Q || !K || !V || !film || !image || !b_out || !attn_part || !units || !unit_count)
    return fail("edge_block_bf16: null argument");
  if (B <= 0 || n <= 0 || n > 1024 || n_head <= 0 || dx % n_head) return fail("edge_block_bf16: bad shape");
  if (ldqkv % 4 != 0) return fail("edge_block_bf16: ldqkv must be a multiple of 4");
  EdgeBlockArgs a{E_in, newE, node_mask, Q, K, V, ldqkv, film, static_cast<const f32x4*>(image), b_out, attn_part, units,
                  unit_count, 1.0f / sqrtf((float)(dx / n_head)), B, n};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dx == 256 && de == 128) return launch_edge_block_bf16<8, 4>(a, st);
  if (dx == 256 && de == 64) return launch_edge_block_bf16<8, 2>(a, st);
  if (dx == 128 && de == 64) return launch_edge_block_bf16<4, 2>(a, st);
  return fail("edge_block_bf16: unsupported dims dx=%d de=%d (built: 256/128, 256/64, 128/64)", dx, de);
}

extern "C" int scgg_attn_merge(const float* attn_part, const uint8_t* node_mask, const float* film, float* newX_pre,
                               const int32_t* rows, const int32_t* row_count, int32_t max_rows, int32_t n, int32_t dx,
                               void* stream) {
  if (!attn_part || !node_mask || !film || !newX_pre || !rows || !row_count) return fail("attn_merge: null argument");
  if (n <= 0 || n > 1024 || max_rows <= 0) return fail("attn_merge: bad shape");
  hipLaunchKernelGGL(attn_merge64_kernel, dim3(max_rows), dim3(256), 0, static_cast<hipStream_t>(stream), attn_part, node_mask,
                     film, newX_pre, rows, row_count, n, dx);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("attn_merge: %s", hipGetErrorString(e));
}
