// Fused XEy layer, edge part (gfx950, fp32 on v_mfma_f32_32x32x2_f32) -- persistent, LDS-staged weights.
//
//   NodeEdgeBlock (E part + attention)  transformer_model.py:165-227
//   XEyTransformerLayer E path          transformer_model.py:94-99, masks :110-112 (utils.py:154)
//
// Structure: one workgroup of 8 waves per CU, persistent over the list of VALID query rows; every
// wave owns one row (b,i) at a time, keys j on the lanes (chunks of 32), channels in registers ("C layout",
// mfma_chain.h).  The 256 KB of layer weights do not fit LDS, so the row's work is cut into NPH phases
//   cb = 0..dx/32-1 : e_mul/e_add rows of channel block cb, the matching K-slice of e_out   (96 MFMAs per wave)
//   FFN1 x ffE/64   : two row blocks of linE1                                               (64)
//   FFN2 x de/32    : one row block of linE2                                                (64)
// whose weights (<= 25 KB, pre-arranged once per weight refresh into per-phase images by edge_phase_pack_kernel) are
// copied global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers) into a double buffer while the
// previous phase computes; one __syncthreads per phase (it also retires the DMA).  All 8 waves read their MFMA A
// fragments from the same LDS image (ds_read_b128, conflict-free), so L2 sees each weight once per workgroup
// iteration instead of once per wave; the fragments of MFMA block t+1 are read under the 16 MFMAs of block t.
//
// Attention: per (row, channel) softmax over the keys.  Fast path: one wave-uniform shift M = max over the block
// (softmax is shift invariant), p = exp(y-M), and the two key-sums (sum p, sum p*V) are taken by transposing the
// 32x32 block through a per-wave LDS scratch (lane <- channel), which costs ~70 VALU per half block instead of ~400
// for per-register cross-lane reductions.  If a channel underflows completely (its maximum is > ~87 below M) the block
// falls back to exact per-channel maxima.  Rows with more than 32 keys merge chunk partials online in LDS.
#include "mfma_chain.h"
#include "kernels.h"

namespace scgg {

// ---------------------------------------------------------------------------------------------------------
// valid-row list: rows[k] = flat (b*n+i) of the k-th valid node, *count = number of valid rows.  One block.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void build_row_list_kernel(const uint8_t* mask, int M, int* rows, int* count) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int start = 0; start < M; start += 1024) {
    const int idx = start + tid;
    const bool v = idx < M && mask[idx] != 0;
    const unsigned long long bal = __ballot(v);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (v) rows[off + before] = idx;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wsum[w];
      base += t;
    }
    __syncthreads();
  }
  if (tid == 0) *count = base;
}

int launch_build_row_list(const uint8_t* mask, int M, int* rows, int* count, hipStream_t st) {
  hipLaunchKernelGGL(build_row_list_kernel, dim3(1), dim3(1024), 0, st, mask, M, rows, count);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 lds_vec_block(const float* vec, int h) {
  f32x16 x;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(vec + 8 * g + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[4 * g + e] = t[e];
  }
  return x;
}

// acc += F . x for the 32x32 block whose A fragments f are already in registers, while the fragments of the NEXT block
// (at `next` in LDS) are read into nf underneath the 16 MFMAs.  sched_barrier pins that order: left alone, hipcc either
// serialises  ds_read -> lgkmcnt(0) -> 4 MFMAs  or hoists a whole phase of fragments into ~100 VGPRs and spills.
__device__ __forceinline__ void mfma_block_pipe(f32x16& acc, const f32x4 (&f)[4], const f32x16& x, f32x4 (&nf)[4],
                                                const f32x4* next, int lane) {
  load_frag(nf, next, lane);
  __builtin_amdgcn_sched_barrier(0);
  mfma16(acc, f, x);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void mfma_block_last(f32x16& acc, const f32x4 (&f)[4], const f32x16& x) {
  __builtin_amdgcn_sched_barrier(0);
  mfma16(acc, f, x);
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ float other_half_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Development probe (scripts/probe): cycle stamps of the second tile of workgroup 0, one slot per code point.
#ifdef SCGG_STAMP
__device__ unsigned long long g_stamps[8][96];
#define STAMP(k)                                                                                   \
  do {                                                                                             \
    if (blockIdx.x == 0 && it == 1 && jc == 0 && lane == 0) g_stamps[wave][k] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

template <int NXB, int NEB, int NFB>
struct LayerGeom {
  static constexpr int DX = 32 * NXB, DE = 32 * NEB;
  static constexpr int W_CB = 3 * NEB * 256;         // float4s of a channel-block phase
  static constexpr int W_F1 = 2 * NEB * 256;
  static constexpr int W_F2 = NFB * 256;
  static constexpr int WMAX = W_CB > W_F1 ? (W_CB > W_F2 ? W_CB : W_F2) : (W_F1 > W_F2 ? W_F1 : W_F2);
  static constexpr int PH_F4 = WMAX + 64;            // + bias area (16 float4 used, padded to one wave instruction)
  static constexpr int NPH = NXB + NFB / 2 + NEB;
  static constexpr int SCR_KIND = 32 * 20 + 16;      // floats per kind in the softmax transpose scratch
  static constexpr int SCR = 2 * SCR_KIND + 64;      // + 3x16 fallback slots
  static constexpr size_t LDS_BYTES = (size_t)2 * PH_F4 * 16 + (size_t)8 * (3 * DX + SCR + 3 * DX) * 4;
};

// ---------------------------------------------------------------------------------------------------------
// Phase images.  img[ph][0..WMAX) = the phase's MFMA A fragments in the order the kernel walks them,
// img[ph][WMAX..WMAX+16) = its bias vectors (C layout source order), rest zero:
//   cb phase  : e_mul[cb][kb=0..NEB) | e_add[cb][kb] | e_out[rb=0..NEB)[cb] ; bias = (e_mul.bias+1)[cb] | e_add.bias[cb]
//   FFN1 f    : linE1 row blocks 2f, 2f+1 (all kb)                          ; bias = linE1.bias[64f .. 64f+64)
//   FFN2 rb   : linE2 row block rb (all kb)                                 ; bias = linE2.bias[32rb .. 32rb+32)
// ---------------------------------------------------------------------------------------------------------
template <int NXB, int NEB, int NFB>
__global__ __launch_bounds__(512) void edge_phase_pack_kernel(EdgePhasePackArgs a) {
  using G = LayerGeom<NXB, NEB, NFB>;
  const int ph = blockIdx.x;
  for (int idx = threadIdx.x; idx < G::PH_F4; idx += blockDim.x) {
    const f32x4* src = nullptr;
    if (ph < NXB) {
      const int cb = ph;
      if (idx < NEB * 256) src = a.W_mul + cb * NEB * 256 + idx;
      else if (idx < 2 * NEB * 256) src = a.W_add + cb * NEB * 256 + (idx - NEB * 256);
      else if (idx < 3 * NEB * 256) {
        const int j = idx - 2 * NEB * 256;
        src = a.W_out + ((j >> 8) * NXB + cb) * 256 + (j & 255);
      } else if (idx >= G::WMAX && idx < G::WMAX + 8) src = reinterpret_cast<const f32x4*>(a.b_mul1) + cb * 8 + (idx - G::WMAX);
      else if (idx >= G::WMAX + 8 && idx < G::WMAX + 16) src = reinterpret_cast<const f32x4*>(a.b_add) + cb * 8 + (idx - G::WMAX - 8);
    } else if (ph < NXB + NFB / 2) {
      const int f = ph - NXB;
      if (idx < G::W_F1) src = a.W_ff1 + (2 * f * NEB) * 256 + idx;
      else if (idx >= G::WMAX && idx < G::WMAX + 16) src = reinterpret_cast<const f32x4*>(a.b_ff1) + f * 16 + (idx - G::WMAX);
    } else {
      const int rb = ph - NXB - NFB / 2;
      if (idx < G::W_F2) src = a.W_ff2 + (rb * NFB) * 256 + idx;
      else if (idx >= G::WMAX && idx < G::WMAX + 8) src = reinterpret_cast<const f32x4*>(a.b_ff2) + rb * 8 + (idx - G::WMAX);
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src != nullptr) v = *src;
    a.img[(size_t)ph * G::PH_F4 + idx] = v;
  }
}

// LDS-DMA copy of one phase image into buf: wave w issues the 64-float4 pieces w, w+8, ... (each 1 KiB, lane-linear)
template <int PH_F4>
__device__ __forceinline__ void stage_phase(const f32x4* __restrict__ src, f32x4* buf, int wave, int lane) {
  static_assert(PH_F4 % 64 == 0, "phase image must be whole wave instructions");
#pragma unroll
  for (int s = 0; s < (PH_F4 / 64 + 7) / 8; ++s) {
    const int piece = wave + 8 * s;                                   // wave-uniform
    if (piece < PH_F4 / 64)
      __builtin_amdgcn_global_load_lds(
          reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(src + piece * 64 + lane)),
          (__attribute__((address_space(3))) void*)(buf + piece * 64), 16, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Attention statistics of one (row, key chunk, 32-channel block): per channel  sum_j p  and  sum_j p V_j  with
// p = exp(y - shift), merged into this wave's running [max | sum | acc] slots (transformer_model.py:207-214, the
// masked_softmax of utils.py over the key axis).  y is -inf on invalid keys, v is 0 there.
// ---------------------------------------------------------------------------------------------------------
template <int DX, int SCR_KIND>
__device__ __forceinline__ void softmax_block(const f32x16& y, const f32x16& v, int cb, bool first_chunk, bool multi,
                                              float* scr, float* run, int lane) {
  const int p = lane & 31, h = lane >> 5;
  constexpr float LOG2E = 1.4426950408889634f;
  float mloc = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, y[r]);
  const float M = other_half_max(half_max(mloc));                      // wave-uniform shift
  const float nM = M == -INFINITY ? 0.f : -M * LOG2E;
#pragma unroll
  for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      f32x4 w0, w1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = 8 * hp + 4 * gg + t;
        w0[t] = __builtin_amdgcn_exp2f(fmaf(y[r], LOG2E, nM));         // exp(y - M); exactly 0 on invalid keys
        w1[t] = w0[t] * v[r];
      }
      *reinterpret_cast<f32x4*>(scr + p * 20 + 8 * gg + 4 * h) = w0;
      *reinterpret_cast<f32x4*>(scr + SCR_KIND + p * 20 + 8 * gg + 4 * h) = w1;
    }
    __builtin_amdgcn_wave_barrier();
    // lane <- (channel c16, kind, key half): sum 16 keys, then fold the two key halves
    const int c16 = lane & 15, kind = (lane >> 4) & 1, jh = lane >> 5;
    const float* S = scr + kind * SCR_KIND + (jh * 16) * 20 + c16;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc += S[t * 20];
    __builtin_amdgcn_wave_barrier();
    acc = other_half_sum(acc);
    auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
    float csum = __uint_as_float(sw[0]);      // rows 0,2: sum p      (own value)
    float cacc = __uint_as_float(sw[1]);      // rows 0,2: sum p*V    (value of the next 16-lane row)
    float cmax = M;
    if (M > -INFINITY && __any(lane < 16 && csum == 0.f)) {
      // a channel underflowed against the block-wide shift: exact per-channel maxima for this half block
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float yy = y[8 * hp + t];
        const float mx = half_max(yy);
        const float ex = yy == -INFINITY ? 0.f : __expf(yy - mx);
        const float s1 = half_sum(ex);
        const float s2 = half_sum(ex * v[8 * hp + t]);
        if (p == 0) {
          const int cc = (t & 3) + 8 * (t >> 2) + 4 * h;        // channel within the half block
          scr[2 * SCR_KIND + cc] = mx;
          scr[2 * SCR_KIND + 16 + cc] = s1;
          scr[2 * SCR_KIND + 32 + cc] = s2;
        }
      }
      __builtin_amdgcn_wave_barrier();
      cmax = scr[2 * SCR_KIND + c16];
      csum = scr[2 * SCR_KIND + 16 + c16];
      cacc = scr[2 * SCR_KIND + 32 + c16];
    }
    if (lane < 16) {
      const int c = 32 * cb + 16 * hp + c16;
      if (first_chunk) {
        run[DX + c] = csum; run[2 * DX + c] = cacc;
        if (multi) run[c] = cmax;
      } else {
        const float mo = run[c], mn = fmaxf(mo, cmax);
        const float fo = mo == -INFINITY ? 0.f : __expf(mo - mn);
        const float fn = cmax == -INFINITY ? 0.f : __expf(cmax - mn);
        run[c] = mn;
        run[DX + c] = run[DX + c] * fo + csum * fn;
        run[2 * DX + c] = run[2 * DX + c] * fo + cacc * fn;
      }
    }
  }
}

template <int NXB, int NEB, int NFB>
__global__ __launch_bounds__(512, 2) void edge_layer_kernel(EdgeLayerArgs a) {
  using G = LayerGeom<NXB, NEB, NFB>;
  constexpr int DX = G::DX, DE = G::DE, NPH = G::NPH;
  static_assert(NFB % 2 == 0, "dim_ffE must be a multiple of 64");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* wbuf = reinterpret_cast<f32x4*>(smem);                        // [2][PH_F4]
  float* fl = reinterpret_cast<float*>(wbuf + 2 * G::PH_F4);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 31, h = lane >> 5;
  float* vecw = fl + wave * (3 * DX);                                  // Q row | y_e_add | y_e_mul+1
  float* scr = fl + 8 * (3 * DX) + wave * G::SCR;                      // softmax transpose scratch
  float* run = fl + 8 * (3 * DX) + 8 * G::SCR + wave * (3 * DX);       // running max | sum | acc (n > 32)
  const bool late = wave >= 4;                                         // second wave of each SIMD (see the cb loop)
  const int n = a.n;
  const int nchunks = (n + 31) >> 5;
  const int count = *a.row_count;
  const int rows_per_iter = 8 * gridDim.x;
  const int iters = (count + rows_per_iter - 1) / rows_per_iter;

  stage_phase<G::PH_F4>(a.phase_img, wbuf, wave, lane);
  int cur = 0;   // buffer holding the phase about to run

  for (int it = 0; it < iters; ++it) {
    const int ridx = (it * gridDim.x + blockIdx.x) * 8 + wave;
    const bool active = ridx < count;                                  // wave-uniform
    const int row = active ? a.rows[ridx] : 0;
    const int b = row / n, i = row - b * n;
    const float* film = a.film + (long long)b * (4 * DX);
    if (active) {   // per-row vectors into this wave's LDS slice (read after >= 1 barrier)
      for (int c = lane * 4; c < 3 * DX; c += 256) {
        const float* src = c < DX ? a.Q + (long long)row * a.ldqkv + c : film + (c - DX);
        f32x4 val = *reinterpret_cast<const f32x4*>(src);
        if (c < DX) val = val * a.inv_sqrt_df;                           // fold 1/sqrt(df) into the query row
        *reinterpret_cast<f32x4*>(vecw + c) = val;
      }
    }
    for (int jc = 0; jc < nchunks; ++jc) {
      const int j = jc * 32 + p;
      const bool inb = j < n;
      const int jj = inb ? j : 0;
      const bool vj = active && inb && a.mask[b * n + jj] != 0;
      const bool live = vj && j != i;
      const long long off_ij = ((long long)row * n + jj) * DE;
      const long long off_ji = (((long long)b * n + jj) * n + i) * DE;
      const float* Kj = a.K + ((long long)b * n + jj) * a.ldqkv;
      const float* Vj = a.V + ((long long)b * n + jj) * a.ldqkv;

      f32x16 e[NEB], accO[NEB];
      float pf0 = 0.f, pf1 = 0.f;
      STAMP(0);
      load_sym_tile<NEB>(e, a.E_in, off_ij, off_ji, live, h);
#pragma unroll
      for (int rb = 0; rb < NEB; ++rb) accO[rb] = load_vec_block(a.b_out + 32 * rb, h);

      // ---- Etoy pooling partials of the layer INPUT for this (row, chunk): per channel sum | min | max | M2 about
      //      the row mean over the valid keys (the zero diagonal included, layers.py:34-40).  Same LDS transpose as the
      //      attention sums; merged per graph by edge_stats_finish_kernel (Chan's parallel variance).
      if (a.stats_part != nullptr && active) {
        const unsigned vmask = (unsigned)__ballot(vj);                  // keys of this chunk, bit j (both halves agree)
        const float cntf = (float)__popc(vmask);
        const int c16 = lane & 15, slab = (lane >> 4) & 1, jh = lane >> 5;
#pragma unroll
        for (int ps = 0; ps < NEB; ++ps) {
#pragma unroll
          for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
              f32x4 w0;
#pragma unroll
              for (int t = 0; t < 4; ++t) w0[t] = e[ps][8 * hp + 4 * gg + t];
              *reinterpret_cast<f32x4*>(scr + hp * G::SCR_KIND + p * 20 + 8 * gg + 4 * h) = w0;
            }
          __builtin_amdgcn_wave_barrier();
          const float* S = scr + slab * G::SCR_KIND + (jh * 16) * 20 + c16;
          float xs[16];
          float sm = 0.f, mn = INFINITY, mx = -INFINITY;
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            xs[t] = S[t * 20];
            const bool ok = (vmask >> (jh * 16 + t)) & 1u;
            sm += ok ? xs[t] : 0.f;
            mn = fminf(mn, ok ? xs[t] : INFINITY);
            mx = fmaxf(mx, ok ? xs[t] : -INFINITY);
          }
          __builtin_amdgcn_wave_barrier();
          sm = other_half_sum(sm);
          mn = -other_half_max(-mn);
          mx = other_half_max(mx);
          const float mean = cntf > 0.f ? sm / cntf : 0.f;
          float m2 = 0.f;
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const bool ok = (vmask >> (jh * 16 + t)) & 1u;
            const float d = xs[t] - mean;
            m2 += ok ? d * d : 0.f;
          }
          m2 = other_half_sum(m2);
          if (lane < 32) {
            float* dst = a.stats_part + ((long long)row * nchunks + jc) * (4 * DE) + 32 * ps + 16 * slab + c16;
            dst[0] = sm; dst[DE] = mn; dst[2 * DE] = mx; dst[3 * DE] = m2;
          }
        }
      }

      // ================================================================== channel-block phases
      // The two waves of a SIMD (w and w+4) share its matrix pipe and VALU port, and the phase barrier keeps them in
      // step.  Waves 4-7 therefore run the attention statistics of block cb at the START of phase cb+1, while their
      // partner is already issuing that phase's MFMAs; waves 0-3 run theirs at the end of the phase, under the
      // partner's MFMA tail (MI355X_MICROARCH.md "Two waves per SIMD").
      f32x16 y, v;
      f32x4 fa[4], fb[4];                                               // A fragments: block in use / block in flight
#pragma unroll 1
      for (int cb = 0; cb < NXB; ++cb) {
        STAMP(1 + 6 * cb);
        __syncthreads();
        STAMP(2 + 6 * cb);
        stage_phase<G::PH_F4>(a.phase_img + (size_t)(cb + 1) * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
        const f32x4* W = wbuf + cur * G::PH_F4;
        if (active) {
          f32x16 k = load_vec_block(Kj + 32 * cb, h);
          f32x16 vn = load_vec_block(Vj + 32 * cb, h);
          if (late && cb > 0) softmax_block<DX, G::SCR_KIND>(y, v, cb - 1, jc == 0, nchunks > 1, scr, run, lane);
          load_frag(fa, W, lane);
          f32x16 a1 = lds_vec_block(reinterpret_cast<const float*>(W + G::WMAX), h);        // e_mul bias + 1
          f32x16 a2 = lds_vec_block(reinterpret_cast<const float*>(W + G::WMAX + 8), h);    // e_add bias
          // blocks in image order of use: (e_mul,kb0) (e_add,kb0) (e_mul,kb1) ... then e_out rb0..
#pragma unroll
          for (int t = 0; t < 2 * NEB; ++t) {
            const int nb = t + 1 < 2 * NEB ? (((t + 1) & 1) ? NEB : 0) + ((t + 1) >> 1) : 2 * NEB;
            if ((t & 1) == 0) mfma_block_pipe(a1, fa, e[t >> 1], fb, W + nb * 256, lane);
            else mfma_block_pipe(a2, fb, e[t >> 1], fa, W + nb * 256, lane);
          }
          STAMP(3 + 6 * cb);
          f32x16 z;
#pragma unroll
          for (int g = 0; g < 4; ++g) {   // per group of 4 channels: keeps only 3 float4 of row vectors live
            const int co = 32 * cb + 8 * g + 4 * h;
            const f32x4 q = *reinterpret_cast<const f32x4*>(vecw + co);            // Q_i / sqrt(df)
            const f32x4 yea = *reinterpret_cast<const f32x4*>(vecw + DX + co);
            const f32x4 yem = *reinterpret_cast<const f32x4*>(vecw + 2 * DX + co);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int r = 4 * g + t;
              const float kk = vj ? k[r] : 0.f;
              const float yy = (q[t] * kk) * a1[r] + a2[r];            // Y = QK/sqrt(df) * (E1+1) + E2
              z[r] = yea[t] + yem[t] * yy;                             // FiLM by y before e_out
              y[r] = vj ? yy : -INFINITY;
              v[r] = vj ? vn[r] : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          // newE += W_out[:, block cb] . z
          STAMP(4 + 6 * cb);
#pragma unroll
          for (int rb = 0; rb < NEB; ++rb) {
            if (rb + 1 < NEB) {
              if ((rb & 1) == 0) mfma_block_pipe(accO[rb], fa, z, fb, W + (2 * NEB + rb + 1) * 256, lane);
              else mfma_block_pipe(accO[rb], fb, z, fa, W + (2 * NEB + rb + 1) * 256, lane);
            } else {
              mfma_block_last(accO[rb], (rb & 1) == 0 ? fa : fb, z);
            }
          }
          STAMP(5 + 6 * cb);
          if (!late) softmax_block<DX, G::SCR_KIND>(y, v, cb, jc == 0, nchunks > 1, scr, run, lane);
        }
        STAMP(6 + 6 * cb);
        cur ^= 1;
      }

      // ================================================================== E <- LN1(E + newE), FFN phases
      f32x16 hid[NFB];
#pragma unroll
      for (int f = 0; f < NFB / 2; ++f) {
        STAMP(1 + 6 * NXB + 2 * f);
        __syncthreads();
        STAMP(2 + 6 * NXB + 2 * f);
        stage_phase<G::PH_F4>(a.phase_img + (size_t)(NXB + f + 1) * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
        const f32x4* W = wbuf + cur * G::PH_F4;
        if (active) {
          if (f == 0) {
            if (late) softmax_block<DX, G::SCR_KIND>(y, v, NXB - 1, jc == 0, nchunks > 1, scr, run, lane);
#pragma unroll
            for (int rb = 0; rb < NEB; ++rb)
#pragma unroll
              for (int r = 0; r < 16; ++r) e[rb][r] += accO[rb][r];
            layernorm_c<NEB>(e, a.ln1_g, a.ln1_b, h);
          }
          load_frag(fa, W, lane);
#pragma unroll
          for (int t = 0; t < 2; ++t)
            hid[2 * f + t] = lds_vec_block(reinterpret_cast<const float*>(W + G::WMAX) + 32 * t, h);
#pragma unroll
          for (int u = 0; u < 2 * NEB; ++u) {                           // u = t * NEB + kb
            f32x16& acc = hid[2 * f + u / NEB];
            if (u + 1 < 2 * NEB) {
              if ((u & 1) == 0) mfma_block_pipe(acc, fa, e[u % NEB], fb, W + (u + 1) * 256, lane);
              else mfma_block_pipe(acc, fb, e[u % NEB], fa, W + (u + 1) * 256, lane);
            } else {
              mfma_block_last(acc, (u & 1) == 0 ? fa : fb, e[u % NEB]);
            }
          }
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[2 * f + t][r] = fmaxf(hid[2 * f + t][r], 0.f);
        }
        cur ^= 1;
      }
#pragma unroll
      for (int rb = 0; rb < NEB; ++rb) {
        STAMP(1 + 6 * NXB + NFB + 2 * rb);
        __syncthreads();
        STAMP(2 + 6 * NXB + NFB + 2 * rb);
        const int nxt = (NXB + NFB / 2 + rb + 1) % NPH;                 // wraps to the first phase of the next row
        stage_phase<G::PH_F4>(a.phase_img + (size_t)nxt * G::PH_F4, wbuf + (cur ^ 1) * G::PH_F4, wave, lane);
        if (rb == 0 && nchunks == 1) {
          // touch the next row's E lines now (one dword per 128-B line) so that its tile load two phases from here is
          // an L2 hit instead of a chip-wide HBM burst at the top of the next iteration; values are only consumed by
          // the never-true test after the tile's stores
          const int nridx = ((it + 1) * gridDim.x + blockIdx.x) * 8 + wave;
          if (nridx < count) {
            const int nrow = a.rows[nridx];
            const int nb = nrow / n, ni = nrow - nb * n;
            if (lane * 32 < n * DE) pf0 = *reinterpret_cast<const volatile float*>(a.E_in + (long long)nrow * n * DE + lane * 32);
            if (p < n) pf1 = *reinterpret_cast<const volatile float*>(a.E_in + (((long long)nb * n + p) * n + ni) * DE + 32 * h);
          }
        }
        const f32x4* W = wbuf + cur * G::PH_F4;
        if (active) {
          load_frag(fa, W, lane);
          f32x16 f2 = lds_vec_block(reinterpret_cast<const float*>(W + G::WMAX), h);
#pragma unroll
          for (int kb = 0; kb < NFB; ++kb) {
            if (kb + 1 < NFB) {
              if ((kb & 1) == 0) mfma_block_pipe(f2, fa, hid[kb], fb, W + (kb + 1) * 256, lane);
              else mfma_block_pipe(f2, fb, hid[kb], fa, W + (kb + 1) * 256, lane);
            } else {
              mfma_block_last(f2, (kb & 1) == 0 ? fa : fb, hid[kb]);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) e[rb][r] += f2[r];
        }
        cur ^= 1;
      }
      STAMP(1 + 6 * NXB + NFB + 2 * NEB);
      if (active) {
        layernorm_c<NEB>(e, a.ln2_g, a.ln2_b, h);
        if (inb) {
          float* dst = a.E_out + off_ij;
#pragma unroll
          for (int rb = 0; rb < NEB; ++rb) {
            if (!live) e[rb] = splat16(0.f);
            store_vec_block(dst + 32 * rb, h, e[rb]);
          }
        }
      }
      if (__float_as_uint(pf0) + __float_as_uint(pf1) == 0x7fedcba9u) a.E_out[0] = pf0;   // keeps the prefetch loads alive
      STAMP(2 + 6 * NXB + NFB + 2 * NEB);
    }
    if (active) {   // newX_pre = y_x_add + (y_x_mul + 1) * weighted_V   (transformer_model.py:230-232)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int c = lane; c < DX; c += 64)
        a.newX_pre[(long long)row * DX + c] = film[2 * DX + c] + film[3 * DX + c] * (run[2 * DX + c] / run[DX + c]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// merge the per-(row, chunk) partials of one graph into z[b] = [mean | min | max | biased variance] (layers.py:34-41)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_stats_finish_kernel(EdgeStatsArgs a) {
  // thread = (channel c, row slice s); slices merge through LDS in a fixed order (Chan's formula is order dependent
  // only in round-off, the fixed order keeps it reproducible)
  __shared__ float sN[256], sMean[256], sM2[256], sMn[256], sMx[256];
  const int b = blockIdx.x, n = a.n, DE = a.de, tid = threadIdx.x;
  const int S = 256 / DE, c = tid % DE, s = tid / DE;
  const uint8_t* m = a.mask + b * n;
  const int nchunks = (n + 31) >> 5;
  float N = 0.f, mean = 0.f, M2 = 0.f, mn = INFINITY, mx = -INFINITY;
  if (s < S)
    for (int jc = 0; jc < nchunks; ++jc) {
      int cnt = 0;
      for (int j = jc * 32; j < min(n, jc * 32 + 32); ++j) cnt += m[j] != 0;
      if (cnt == 0) continue;
      const float nb = (float)cnt;
      for (int i = s; i < n; i += S) {
        if (!m[i]) continue;
        const float* p = a.part + (((long long)b * n + i) * nchunks + jc) * (4 * DE) + c;
        const float mb = p[0] / nb, tot = N + nb, delta = mb - mean;
        M2 += p[3 * DE] + delta * delta * (N * nb / tot);
        mean += delta * (nb / tot);
        N = tot;
        mn = fminf(mn, p[DE]);
        mx = fmaxf(mx, p[2 * DE]);
      }
    }
  sN[tid] = N; sMean[tid] = mean; sM2[tid] = M2; sMn[tid] = mn; sMx[tid] = mx;
  __syncthreads();
  if (s == 0) {
    for (int t = 1; t < S; ++t) {
      const int o = t * DE + c;
      const float nb = sN[o];
      if (nb == 0.f) continue;
      const float tot = N + nb, delta = sMean[o] - mean;
      M2 += sM2[o] + delta * delta * (N * nb / tot);
      mean += delta * (nb / tot);
      N = tot;
      mn = fminf(mn, sMn[o]);
      mx = fmaxf(mx, sMx[o]);
    }
    float* z = a.z + (long long)b * a.z_stride + a.z_offset;
    z[c] = mean; z[DE + c] = mn; z[2 * DE + c] = mx; z[3 * DE + c] = M2 / N;
  }
}

int launch_edge_stats_finish(const EdgeStatsArgs& a, hipStream_t st) {
  if (a.de > 256 || 256 % a.de != 0) return -1;
  hipLaunchKernelGGL(edge_stats_finish_kernel, dim3(a.B), dim3(256), 0, st, a);
  return 0;
}

template <int NXB, int NEB, int NFB>
static int launch_one(const EdgeLayerArgs& a, int n_cu, hipStream_t st) {
  using G = LayerGeom<NXB, NEB, NFB>;
  static bool attr_set = false;
  auto kern = edge_layer_kernel<NXB, NEB, NFB>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)G::LDS_BYTES) != hipSuccess)
      return -2;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), G::LDS_BYTES, st, a);
  return 0;
}

long long edge_phase_img_f4(int dx, int de, int ffe) {
#define CASE(NXB, NEB, NFB)                                   \
  if (dx == 32 * NXB && de == 32 * NEB && ffe == 32 * NFB)    \
    return (long long)LayerGeom<NXB, NEB, NFB>::NPH * LayerGeom<NXB, NEB, NFB>::PH_F4;
  CASE(8, 2, 4) CASE(4, 2, 4) CASE(1, 1, 2)
#undef CASE
  return -1;
}

int launch_edge_phase_pack(const EdgePhasePackArgs& a, int dx, int de, int ffe, hipStream_t st) {
#define CASE(NXB, NEB, NFB)                                                                                   \
  if (dx == 32 * NXB && de == 32 * NEB && ffe == 32 * NFB) {                                                  \
    hipLaunchKernelGGL((edge_phase_pack_kernel<NXB, NEB, NFB>), dim3(LayerGeom<NXB, NEB, NFB>::NPH), dim3(512), 0, st, a); \
    return 0;                                                                                                 \
  }
  CASE(8, 2, 4) CASE(4, 2, 4) CASE(1, 1, 2)
#undef CASE
  return -1;
}

int launch_edge_layer(const EdgeLayerArgs& a, int dx, int de, int ffe, hipStream_t st) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -2;
    n_cu = prop.multiProcessorCount;
  }
#define CASE(NXB, NEB, NFB) \
  if (dx == 32 * NXB && de == 32 * NEB && ffe == 32 * NFB) return launch_one<NXB, NEB, NFB>(a, n_cu, st);
  CASE(8, 2, 4) CASE(4, 2, 4) CASE(1, 1, 2)
#undef CASE
  return -1;
}

}  // namespace scgg
