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
// whose weights (<= 24 KB) are staged global -> registers -> LDS into a double buffer by all 512 threads while the
// previous phase computes; one __syncthreads per phase.  All 8 waves read their MFMA A fragments from the same LDS
// image (ds_read_b128, conflict-free), so L2 sees each weight once per workgroup-iteration instead of once per wave.
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

// acc += W[kb-th 256-float4 block at w] . x   with the A fragments read from LDS
__device__ __forceinline__ void mfma_block_lds(f32x16& acc, const f32x4* w, const f32x16& x, int lane) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 a = w[g * 64 + lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], x[4 * g + e], acc, 0, 0, 0);
  }
  // keep the scheduler from hoisting the fragment reads of later blocks (it would hold ~100 VGPRs of weights and spill)
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ float other_half_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int NXB, int NEB, int NFB>
struct LayerGeom {
  static constexpr int DX = 32 * NXB, DE = 32 * NEB;
  static constexpr int W_CB = 3 * NEB * 256;         // float4s of a channel-block phase
  static constexpr int W_F1 = 2 * NEB * 256;
  static constexpr int W_F2 = NFB * 256;
  static constexpr int WMAX = W_CB > W_F1 ? (W_CB > W_F2 ? W_CB : W_F2) : (W_F1 > W_F2 ? W_F1 : W_F2);
  static constexpr int PH_F4 = WMAX + 32;            // + bias area (128 floats)
  static constexpr int NPH = NXB + NFB / 2 + NEB;
  static constexpr int SCR_KIND = 32 * 20 + 16;      // floats per kind in the softmax transpose scratch
  static constexpr int SCR = 2 * SCR_KIND + 64;      // + 3x16 fallback slots
  static constexpr size_t LDS_BYTES = (size_t)2 * PH_F4 * 16 + (size_t)8 * (3 * DX + SCR + 3 * DX) * 4;
};

// stage loads of phase ph: thread tid fetches float4 slots tid, tid+512, ... of the phase image
template <int NXB, int NEB, int NFB>
__device__ __forceinline__ void stage_load(const EdgeLayerArgs& a, int ph, int tid, f32x4 (&st)[4]) {
  using G = LayerGeom<NXB, NEB, NFB>;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int idx = tid + 512 * s;
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
    if (src != nullptr) st[s] = *src;
  }
}

template <int NXB, int NEB, int NFB>
__device__ __forceinline__ void stage_store(f32x4* buf, int ph, int tid, const f32x4 (&st)[4]) {
  using G = LayerGeom<NXB, NEB, NFB>;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int idx = tid + 512 * s;
    bool on;
    if (ph < NXB) on = idx < G::W_CB || (idx >= G::WMAX && idx < G::WMAX + 16);
    else if (ph < NXB + NFB / 2) on = idx < G::W_F1 || (idx >= G::WMAX && idx < G::WMAX + 16);
    else on = idx < G::W_F2 || (idx >= G::WMAX && idx < G::WMAX + 8);
    if (on && idx < G::PH_F4) buf[idx] = st[s];
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
  const int n = a.n;
  const int nchunks = (n + 31) >> 5;
  const int count = *a.row_count;
  const int rows_per_iter = 8 * gridDim.x;
  const int iters = (count + rows_per_iter - 1) / rows_per_iter;

  f32x4 st[4];
  stage_load<NXB, NEB, NFB>(a, 0, tid, st);
  stage_store<NXB, NEB, NFB>(wbuf, 0, tid, st);
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
#pragma unroll 1
      for (int cb = 0; cb < NXB; ++cb) {
        __syncthreads();
        const int nxt = cb + 1;                                         // next phase index (< NPH since NPH > NXB)
        stage_load<NXB, NEB, NFB>(a, nxt, tid, st);
        const f32x4* W = wbuf + cur * G::PH_F4;
        if (active) {
          f32x16 k = load_vec_block(Kj + 32 * cb, h);
          f32x16 v = load_vec_block(Vj + 32 * cb, h);
          f32x16 a1 = lds_vec_block(reinterpret_cast<const float*>(W + G::WMAX), h);        // e_mul bias + 1
          f32x16 a2 = lds_vec_block(reinterpret_cast<const float*>(W + G::WMAX + 8), h);    // e_add bias
#pragma unroll
          for (int kb = 0; kb < NEB; ++kb) {
            mfma_block_lds(a1, W + kb * 256, e[kb], lane);
            mfma_block_lds(a2, W + (NEB + kb) * 256, e[kb], lane);
          }
          f32x16 y, z;
          float mloc = -INFINITY;
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
              y[r] = (q[t] * kk) * a1[r] + a2[r];                      // Y = QK/sqrt(df) * (E1+1) + E2
              z[r] = yea[t] + yem[t] * y[r];                           // FiLM by y before e_out
              mloc = fmaxf(mloc, vj ? y[r] : -INFINITY);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          // newE += W_out[:, block cb] . z
#pragma unroll
          for (int rb = 0; rb < NEB; ++rb) mfma_block_lds(accO[rb], W + (2 * NEB + rb) * 256, z, lane);

          // ---- attention statistics of this (row, chunk, channel block)
          const float M = other_half_max(half_max(mloc));              // wave-uniform shift
#pragma unroll
          for (int hp = 0; hp < 2; ++hp) {
            float pe[8], pv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              pe[t] = vj ? __expf(y[8 * hp + t] - M) : 0.f;
              pv[t] = pe[t] * (vj ? v[8 * hp + t] : 0.f);
            }
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
              f32x4 w0, w1;
#pragma unroll
              for (int t = 0; t < 4; ++t) { w0[t] = pe[4 * gg + t]; w1[t] = pv[4 * gg + t]; }
              *reinterpret_cast<f32x4*>(scr + p * 20 + 8 * gg + 4 * h) = w0;
              *reinterpret_cast<f32x4*>(scr + G::SCR_KIND + p * 20 + 8 * gg + 4 * h) = w1;
            }
            __builtin_amdgcn_wave_barrier();
            // lane <- (channel c16, kind, key half): sum 16 keys, then fold the two key halves
            const int c16 = lane & 15, kind = (lane >> 4) & 1, jh = lane >> 5;
            const float* S = scr + kind * G::SCR_KIND + (jh * 16) * 20 + c16;
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
                const float mx = half_max(vj ? yy : -INFINITY);
                const float ex = vj ? __expf(yy - mx) : 0.f;
                const float s1 = half_sum(ex);
                const float s2 = half_sum(ex * (vj ? v[8 * hp + t] : 0.f));
                if (p == 0) {
                  const int cc = (t & 3) + 8 * (t >> 2) + 4 * h;        // channel within the half block
                  scr[2 * G::SCR_KIND + cc] = mx;
                  scr[2 * G::SCR_KIND + 16 + cc] = s1;
                  scr[2 * G::SCR_KIND + 32 + cc] = s2;
                }
              }
              __builtin_amdgcn_wave_barrier();
              cmax = scr[2 * G::SCR_KIND + c16];
              csum = scr[2 * G::SCR_KIND + 16 + c16];
              cacc = scr[2 * G::SCR_KIND + 32 + c16];
            }
            if (lane < 16) {
              const int c = 32 * cb + 16 * hp + c16;
              if (nchunks == 1) {
                // newX_pre = y_x_add + (y_x_mul + 1) * weighted_V   (transformer_model.py:230-232)
                a.newX_pre[(long long)row * DX + c] = film[2 * DX + c] + film[3 * DX + c] * (cacc / csum);
              } else if (jc == 0) {
                run[c] = cmax; run[DX + c] = csum; run[2 * DX + c] = cacc;
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
        stage_store<NXB, NEB, NFB>(wbuf + (cur ^ 1) * G::PH_F4, nxt, tid, st);
        cur ^= 1;
      }

      // ================================================================== E <- LN1(E + newE), FFN phases
      f32x16 hid[NFB];
#pragma unroll
      for (int f = 0; f < NFB / 2; ++f) {
        __syncthreads();
        const int nxt = NXB + f + 1;
        stage_load<NXB, NEB, NFB>(a, nxt, tid, st);
        const f32x4* W = wbuf + cur * G::PH_F4;
        if (active) {
          if (f == 0) {
#pragma unroll
            for (int rb = 0; rb < NEB; ++rb)
#pragma unroll
              for (int r = 0; r < 16; ++r) e[rb][r] += accO[rb][r];
            layernorm_c<NEB>(e, a.ln1_g, a.ln1_b, h);
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            hid[2 * f + t] = lds_vec_block(reinterpret_cast<const float*>(W + G::WMAX) + 32 * t, h);
#pragma unroll
            for (int kb = 0; kb < NEB; ++kb) mfma_block_lds(hid[2 * f + t], W + (t * NEB + kb) * 256, e[kb], lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[2 * f + t][r] = fmaxf(hid[2 * f + t][r], 0.f);
          }
        }
        stage_store<NXB, NEB, NFB>(wbuf + (cur ^ 1) * G::PH_F4, nxt, tid, st);
        cur ^= 1;
      }
#pragma unroll
      for (int rb = 0; rb < NEB; ++rb) {
        __syncthreads();
        const int nxt = (NXB + NFB / 2 + rb + 1) % NPH;                 // wraps to the first phase of the next row
        stage_load<NXB, NEB, NFB>(a, nxt, tid, st);
        const f32x4* W = wbuf + cur * G::PH_F4;
        if (active) {
          f32x16 f2 = lds_vec_block(reinterpret_cast<const float*>(W + G::WMAX), h);
#pragma unroll
          for (int kb = 0; kb < NFB; ++kb) mfma_block_lds(f2, W + kb * 256, hid[kb], lane);
#pragma unroll
          for (int r = 0; r < 16; ++r) e[rb][r] += f2[r];
        }
        stage_store<NXB, NEB, NFB>(wbuf + (cur ^ 1) * G::PH_F4, nxt, tid, st);
        cur ^= 1;
      }
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
    }
    if (active && nchunks > 1) {   // finish the multi-chunk rows
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
