// Node / global-feature path kernels (gfx950): dense fp32 GEMM on v_mfma_f32_32x32x2_f32 with a fused
// bias / ReLU / row-mask epilogue, residual + LayerNorm, Xtoy pooling, and the weight packers.
//
// These serve q/k/v, x_out, linX1/2, the y-FiLM vectors, the whole y path and the X in/out MLPs
// (transformer_model.py:179-180,218,230-246,88-92,101-106,281-305,322-342); they are O(B*n*d^2) and small next
// to the edge path, but still dense contractions, so they run on the matrix cores in exact fp32.
#include "pool_stats.h"
#include "mfma_chain.h"
#include "c16.h"
#include "kernels.h"

namespace scgg {

// ------------------------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] . W[N,K]^T + bias).  Block = 4 waves (2x2), wave tile 64(m) x 64(n), no LDS: operands are
// L2-resident and each float4 load feeds 4 MFMAs (k-steps kc+4h .. kc+4h+3 for lane half h).
// ------------------------------------------------------------------------------------------------
template <bool FAST>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const int m_base = blockIdx.x * 128 + (wave >> 1) * 64;
  const int n_base = blockIdx.y * 128 + (wave & 1) * 64;
  if (m_base >= a.M || n_base >= a.N) return;
  f32x16 acc[2][2];   // [nt][mt]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = splat16(0.f);

  int mrow[2], nrow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    mrow[t] = min(m_base + 32 * t + p, a.M - 1);
    nrow[t] = min(n_base + 32 * t + p, a.N - 1);
  }
  const bool n1 = n_base + 32 < a.N, m1 = m_base + 32 < a.M;   // wave-uniform: second tile present?

  for (int kc = 0; kc < a.K; kc += 8) {
    f32x4 av[2], wv[2];
    const int k0 = kc + 4 * h;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (FAST) {
        av[t] = *reinterpret_cast<const f32x4*>(a.A + (long long)mrow[t] * a.lda + k0);
        wv[t] = *reinterpret_cast<const f32x4*>(a.W + (long long)nrow[t] * a.ldw + k0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = k0 + e < a.K;
          av[t][e] = ok ? a.A[(long long)mrow[t] * a.lda + k0 + e] : 0.f;
          wv[t][e] = ok ? a.W[(long long)nrow[t] * a.ldw + k0 + e] : 0.f;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[0][e], av[0][e], acc[0][0], 0, 0, 0);
      if (m1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[0][e], av[1][e], acc[0][1], 0, 0, 0);
      if (n1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[1][e], av[0][e], acc[1][0], 0, 0, 0);
      if (n1 && m1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[1][e], av[1][e], acc[1][1], 0, 0, 0);
    }
  }

#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m_base + 32 * mt + p;
    if (m >= a.M) continue;
    const bool keep = !(a.flags & GEMM_ROWMASK) || a.rowmask[m] != 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = n_base + 32 * nt + ch_of(r, h);
        if (nn >= a.N) continue;
        float v = acc[nt][mt][r];
        if (a.bias) v += a.bias[nn];
        if (a.flags & GEMM_RELU) v = fmaxf(v, 0.f);
        if (a.flags & GEMM_ADD_ONE_COLS)
          if ((nn >= a.one_lo && nn < a.one_hi) || (nn >= a.one_lo2 && nn < a.one_hi2)) v += 1.0f;
        a.C[(long long)m * a.ldc + nn] = keep ? v : 0.f;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Small-M GEMM for the global-feature (y) path, M = number of graphs: the K loop is the long axis, so it is split over
// the 16 waves of a workgroup that owns one 32(m) x 32(n) output tile; partial tiles meet in LDS and are summed in a
// FIXED wave order (bit-reproducible, no atomics) before the bias / ReLU / +1 epilogue.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gemm_splitk_kernel(GemmArgs a) {
  __shared__ float part[16][32 * 33];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const int m_base = blockIdx.x * 32, n_base = blockIdx.y * 32;
  const int mrow = min(m_base + p, a.M - 1), nrow = min(n_base + p, a.N - 1);
  const int ksteps = (a.K + 7) / 8;                       // steps of 8 k
  const int per = (ksteps + 15) / 16;
  const int s0 = wave * per, s1 = min(ksteps, s0 + per);
  f32x16 acc = splat16(0.f);
  const float* Ar = a.A + (long long)mrow * a.lda;
  const float* Wr = a.W + (long long)nrow * a.ldw;
  for (int sb = s0; sb < s1; sb += 4) {                   // 4 steps (32 k) per trip: 8 independent loads in flight
    f32x4 av[4], wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k0 = (sb + u) * 8 + 4 * h;
      const bool on = sb + u < s1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = on && k0 + e < a.K;
        av[u][e] = ok ? Ar[k0 + e] : 0.f;
        wv[u][e] = ok ? Wr[k0 + e] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][e], av[u][e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][ch_of(r, h) * 33 + p] = acc[r];     // [n][m]
  __syncthreads();
  const int nn_l = threadIdx.x >> 5, m_l = threadIdx.x & 31;                   // 1024 threads = 32 x 32 outputs
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) v += part[w][nn_l * 33 + m_l];
  const int m = m_base + m_l, nn = n_base + nn_l;
  if (m < a.M && nn < a.N) {
    if (a.bias) v += a.bias[nn];
    if (a.flags & GEMM_RELU) v = fmaxf(v, 0.f);
    if (a.flags & GEMM_ADD_ONE_COLS)
      if ((nn >= a.one_lo && nn < a.one_hi) || (nn >= a.one_lo2 && nn < a.one_hi2)) v += 1.0f;
    const bool keep = !(a.flags & GEMM_ROWMASK) || a.rowmask[m] != 0;
    a.C[(long long)m * a.ldc + nn] = keep ? v : 0.f;
  }
}

int launch_gemm(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return 0;
  if (a.M <= 1024) {   // few rows (graphs): split K over the waves of a workgroup
    hipLaunchKernelGGL(gemm_splitk_kernel, dim3((a.M + 31) / 32, (a.N + 31) / 32), dim3(1024), 0, st, a);
    return 0;
  }
  dim3 g((a.M + 127) / 128, (a.N + 127) / 128), b(256);
  const bool fast = (a.K % 8 == 0) && (a.lda % 4 == 0) && (a.ldw % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.W)) % 16 == 0);
  if (fast) hipLaunchKernelGGL(gemm_kernel<true>, g, b, 0, st, a);
  else hipLaunchKernelGGL(gemm_kernel<false>, g, b, 0, st, a);
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Row-list Linear for the node path: C[row, :] = epi(A[row, :] . W^T + b) for the VALID rows only.
// One wave = 32 rows (from the valid-row list) x `rbw` 32-wide output blocks.  The 32 x K activation tile is held in
// registers in C layout (row on the lane), the weights come pre-packed in A-fragment order (mfma_chain.h) as 1 KiB
// coalesced loads from L2.  Rows that are not in the list are not touched (their buffers keep the zeros written by the
// input MLP), which removes the padded third of the node work.
// ------------------------------------------------------------------------------------------------
template <int KB>
__global__ __launch_bounds__(256) void node_linear_kernel(NodeLinearArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const int count = *a.row_count;
  const int tile = blockIdx.x * 4 + wave;
  if (tile * 32 >= count) return;
  const int ridx = tile * 32 + p;
  const bool ok = ridx < count;
  const long long row = a.rows[ok ? ridx : count - 1];
  f32x16 x[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) x[kb] = load_vec_block(a.A + row * a.lda + 32 * kb, h);
  const int rb0 = blockIdx.y * a.rbw;
#pragma unroll 1
  for (int rb = rb0; rb < rb0 + a.rbw && rb < a.NB; ++rb) {
    f32x16 acc = load_vec_block(a.bias + 32 * rb, h);
    linear_rb<KB>(acc, a.Wp, rb, x, lane);
    if (a.relu) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
    }
    if (ok) {
      float* dst = a.C + row * a.ldc + 32 * rb;
      if (32 * rb + 32 <= a.N) {
        store_vec_block(dst, h, acc);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * rb + ch_of(r, h) < a.N) dst[ch_of(r, h)] = acc[r];
      }
    }
  }
}

int launch_node_linear(const NodeLinearArgs& a, int M, hipStream_t st) {
  if (a.K % 32 != 0 || a.lda % 4 != 0 || a.ldc % 4 != 0) return -1;
  dim3 g((M + 127) / 128, (a.NB + a.rbw - 1) / a.rbw), b(256);
  switch (a.K / 32) {
    case 1: hipLaunchKernelGGL(node_linear_kernel<1>, g, b, 0, st, a); return 0;
    case 2: hipLaunchKernelGGL(node_linear_kernel<2>, g, b, 0, st, a); return 0;
    case 4: hipLaunchKernelGGL(node_linear_kernel<4>, g, b, 0, st, a); return 0;
    case 8: hipLaunchKernelGGL(node_linear_kernel<8>, g, b, 0, st, a); return 0;
    default: return -1;
  }
}

// ------------------------------------------------------------------------------------------------
// out = LayerNorm(x + r) (eps 1e-5), optionally zeroing masked rows.  One wave per row, D <= 1024.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_layernorm_kernel(AddLnArgs a) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= a.M) return;
  if (a.rows != nullptr) {            // valid-row list: rows beyond the count do nothing, the others are remapped
    if (row >= *a.row_count) return;
    row = a.rows[row];
  }
  const int D = a.D;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int c = lane + 64 * t;
    v[t] = 0.f;
    if (c < D) {
      v[t] = a.x[(long long)row * D + c] + a.r[(long long)row * D + c];
      s += v[t];
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int c = lane + 64 * t;
    if (c < D) {
      const float d = v[t] - mean;
      q += d * d;
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) q += __shfl_xor(q, m);
  const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
  const bool keep = a.rowmask == nullptr || a.rowmask[row] != 0;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int c = lane + 64 * t;
    if (c < D) a.out[(long long)row * D + c] = keep ? (v[t] - mean) * rstd * a.gamma[c] + a.beta[c] : 0.f;
  }
}

// The same for a valid-row list and D = 64 F4 (128 or 256): 16 lanes per row (F4 float4 each), four rows per wave, the two
// reductions are 4 DPP steps inside the 16-lane row instead of 6 cross-lane shuffles; one wave per row left most lanes idle
// and took 22 us for 11 k rows of 128.
template <int F4>
__global__ __launch_bounds__(256) void add_layernorm_rows_kernel(AddLnArgs a) {
  constexpr int D = 64 * F4;
  const int ridx = blockIdx.x * 16 + (threadIdx.x >> 4), p = threadIdx.x & 15;
  const int count = *a.row_count;
  const bool on = ridx < count;
  const long long row = a.rows[on ? ridx : 0];
  f32x4 v[F4];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < F4; ++t) {
    const long long o = row * D + 64 * t + 4 * p;
    v[t] = *reinterpret_cast<const f32x4*>(a.x + o) + *reinterpret_cast<const f32x4*>(a.r + o);
    s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
  }
  const float mean = row_sum16(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < F4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[t][e] - mean;
      q += d * d;
    }
  const float rstd = 1.0f / sqrtf(row_sum16(q) * (1.0f / D) + 1e-5f);
  if (!on) return;
  const bool keep = a.rowmask == nullptr || a.rowmask[row] != 0;
#pragma unroll
  for (int t = 0; t < F4; ++t) {
    const int c = 64 * t + 4 * p;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + c), bt = *reinterpret_cast<const f32x4*>(a.beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = keep ? (v[t][e] - mean) * rstd * gm[e] + bt[e] : 0.f;
    *reinterpret_cast<f32x4*>(a.out + row * D + c) = o;
  }
}

int launch_add_layernorm(const AddLnArgs& a, hipStream_t st) {
  if (a.D > 1024) return -1;
  if (a.rows != nullptr && (a.D == 128 || a.D == 256) &&
      ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.r) | reinterpret_cast<uintptr_t>(a.out) |
        reinterpret_cast<uintptr_t>(a.gamma) | reinterpret_cast<uintptr_t>(a.beta)) % 16 == 0)) {
    if (a.D == 128) hipLaunchKernelGGL(add_layernorm_rows_kernel<2>, dim3((a.M + 15) / 16), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(add_layernorm_rows_kernel<4>, dim3((a.M + 15) / 16), dim3(256), 0, st, a);
    return 0;
  }
  hipLaunchKernelGGL(add_layernorm_kernel, dim3((a.M + 3) / 4), dim3(256), 0, st, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Xtoy pooling (layers.py:11-19): [mean|min|max|biased variance] over the valid nodes of each graph
// ------------------------------------------------------------------------------------------------
// One workgroup per graph; thread = (4 channels, slice of the rows): float4 loads, the rows of a slice are independent
// loads in flight together (the kernel runs beside the edge kernel's start and sits in front of the y round, so its
// latency chain matters, not its bytes).  dx % 4 == 0, dx / 4 divides NS_T; the valid nodes come first in a list in LDS.
constexpr int NS_T = PS_T;     // pool_stats.h, where the arithmetic lives (y_layer_kernel shares it)
__global__ __launch_bounds__(NS_T) void node_stats_kernel(NodeStatsArgs a) {
  __builtin_amdgcn_s_setprio(3);   // two-stream schedule: runs beside the edge kernel and in front of the y round: issue first
  __shared__ PoolScratch ps;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nv = pool_compact_mask(ps, a.mask + (long long)b * a.n, a.n, tid);
  pool_xtoy_graph(ps, a.X + (long long)b * a.n * a.dx, nv, a.dx, a.z + (long long)b * a.z_stride + a.z_offset, tid);
}

int launch_node_stats(const NodeStatsArgs& a, hipStream_t st) {
  if (a.dx % 4 != 0 || a.dx > 512 || NS_T % (a.dx / 4) != 0 || a.n > 512) return -1;
  hipLaunchKernelGGL(node_stats_kernel, dim3(a.B), dim3(NS_T), 0, st, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// packers
// ------------------------------------------------------------------------------------------------
__global__ void pack_linear_kernel(PackArgs a) {
  const long long total = (long long)a.NB * a.KB * 4 * 64;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int lane = idx & 63;
    const int g = (idx >> 6) & 3;
    const long long rk = idx >> 8;
    const int kb = rk % a.KB, rb = rk / a.KB;
    const int nn = 32 * rb + (lane & 31);
    f32x4_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 32 * kb + 8 * g + 4 * (lane >> 5) + e;
      v[e] = (nn < a.N && k < a.K) ? a.W[(long long)nn * a.K + k] : 0.f;
    }
    a.Wp[idx] = v;
  }
}

// Three-term bf16 image (x = h + m + l by truncation, exact) for v_mfma_f32_16x16x32_bf16 with W as the A operand:
// slot ((ob * KC + kc) * 3 + term) * 64 + lane holds, for output channel 16 ob + (lane & 15), the 8 input channels
// 32 kc + 8 (lane >> 4) + 0..7 as packed bf16 pairs (ob < 2 NB: 16-channel blocks, KC = K / 32).
__global__ void pack_linear_x3_kernel(PackArgs a) {
  const int KC = a.K / 32;
  const long long total = (long long)2 * a.NB * KC * 192;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int lane = idx & 63;
    const long long q = idx >> 6;
    const int term = (int)(q % 3);
    const long long rk = q / 3;
    const int kc = (int)(rk % KC), ob = (int)(rk / KC);
    const int nn = 16 * ob + (lane & 15);
    unsigned hw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 32 * kc + 8 * (lane >> 4) + i;
      float x = (nn < a.N && k < a.K) ? a.W[(long long)nn * a.K + k] : 0.f;
      for (int t = 0; t < term; ++t) x = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
      hw[i] = __float_as_uint(x) >> 16;
    }
    u4 v;
#pragma unroll
    for (int d = 0; d < 4; ++d) v[d] = hw[2 * d] | (hw[2 * d + 1] << 16);
    reinterpret_cast<u4*>(a.Wp)[idx] = v;
  }
}

int launch_pack_linear_x3(const PackArgs& a, hipStream_t st) {
  if (a.K % 32 != 0) return -1;
  const long long total = (long long)2 * a.NB * (a.K / 32) * 192;
  hipLaunchKernelGGL(pack_linear_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
  return 0;
}

int launch_pack_linear(const PackArgs& a, hipStream_t st) {
  const long long total = (long long)a.NB * a.KB * 256;
  hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
  return 0;
}

__global__ void pad_vec_kernel(const float* src, float* dst, int n, int n_pad, float add) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) dst[i] = i < n ? src[i] + add : 0.f;
}

int launch_pad_vec(const float* src, float* dst, int n, int n_pad, float add, hipStream_t st) {
  hipLaunchKernelGGL(pad_vec_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, st, src, dst, n, n_pad, add);
  return 0;
}

__global__ void copy_cols_kernel(const float* src, int lds, float* dst, int ldd, int M, int ncols) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)M * ncols) return;
  const int m = idx / ncols, c = idx - (long long)m * ncols;
  dst[(long long)m * ldd + c] = src[(long long)m * lds + c];
}

// ------------------------------------------------------------------------------------------------
// Row-list Linear for a very narrow input (K <= 64: the raw node features).  The transposed weights ([K][N]) and a
// slab of 64 input rows sit in LDS; thread = (row of the slab, 4 adjacent outputs).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void narrow_linear_kernel(NarrowLinearArgs a) {
  extern __shared__ __attribute__((aligned(16))) float nsm[];
  float* wt = nsm;                       // [K][N]
  float* xs = nsm + a.K * a.N;           // [64][K]
  const int tid = threadIdx.x, N4 = a.N >> 2;
  const int count = *a.row_count;
  const int r0 = blockIdx.x * 64;
  if (r0 >= count) return;
  for (int idx = tid; idx < a.N * a.K; idx += 256) {
    const int o = idx / a.K, k = idx - o * a.K;
    wt[k * a.N + o] = a.W[idx];
  }
  for (int idx = tid; idx < 64 * a.K; idx += 256) {
    const int r = idx / a.K, k = idx - r * a.K;
    const int ridx = r0 + r;
    xs[idx] = ridx < count ? a.A[(long long)a.rows[ridx] * a.lda + k] : 0.f;
  }
  __syncthreads();
  for (int item = tid; item < 64 * N4; item += 256) {
    const int r = item / N4, o4 = item - r * N4;
    if (r0 + r >= count) break;
    f32x4 acc = *reinterpret_cast<const f32x4*>(a.bias + 4 * o4);
    for (int k = 0; k < a.K; ++k) acc += *reinterpret_cast<const f32x4*>(wt + k * a.N + 4 * o4) * xs[r * a.K + k];
    if (a.relu) { acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f); }
    *reinterpret_cast<f32x4*>(a.C + (long long)a.rows[r0 + r] * a.ldc + 4 * o4) = acc;
  }
}

int launch_narrow_linear(const NarrowLinearArgs& a, int M, hipStream_t st) {
  if (a.N % 4 != 0 || a.ldc % 4 != 0 || a.K > 64 || (size_t)(a.K * a.N + 64 * a.K) * 4 > 64 * 1024) return -1;
  hipLaunchKernelGGL(narrow_linear_kernel, dim3((M + 63) / 64), dim3(256), (size_t)(a.K * a.N + 64 * a.K) * 4, st, a);
  return 0;
}

// rows of X[M, D] whose mask byte is 0 <- 0 (one wave per row)
__global__ __launch_bounds__(256) void zero_masked_rows_kernel(float* X, const uint8_t* mask, int M, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M || mask[row] != 0) return;
  for (int c = lane; c < D; c += 64) X[(long long)row * D + c] = 0.f;
}

int launch_zero_masked_rows(float* X, const uint8_t* mask, int M, int D, hipStream_t st) {
  hipLaunchKernelGGL(zero_masked_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, st, X, mask, M, D);
  return 0;
}

__global__ void transpose_kernel(const float* src, float* dst, int rows, int cols) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= (long long)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx - (long long)r * cols);
  dst[(size_t)c * rows + r] = src[idx];
}

int launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)(((long long)rows * cols + 255) / 256)), dim3(256), 0, st, src, dst,
                     rows, cols);
  return 0;
}

int launch_copy_cols(const float* src, int lds, float* dst, int ldd, int M, int ncols, hipStream_t st) {
  const long long total = (long long)M * ncols;
  if (total == 0) return 0;
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, lds, dst, ldd, M,
                     ncols);
  return 0;
}

// X_out = (mlp_out_X(X) + X_in[..., :nout]) * mask     (transformer_model.py:374, :387-389)
__global__ void x_out_residual_kernel(const float* mlp, const float* X_in, int dX_in, const uint8_t* mask,
                                      float* out, int M, int nout) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)M * nout) return;
  const int m = idx / nout, c = idx - (long long)m * nout;
  const bool keep = mask == nullptr || mask[m] != 0;
  out[idx] = keep ? mlp[idx] + X_in[(long long)m * dX_in + c] : 0.f;
}

// Narrow output head on the valid rows (mlp_out_X.3 + residual, transformer_model.py:374, :387): one wave per row,
// out[row, c] = W[c, :] . hid[row, :] + b[c] + X_in[row, c] for c < N (N small); masked rows are zero-filled by the caller.
__global__ __launch_bounds__(256) void row_head_kernel(const float* hid, int ldh, const float* W, const float* b,
                                                       const float* X_in, int dX_in, const int* rows, const int* row_count,
                                                       float* out, int N, int K) {
  const int ridx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ridx >= *row_count) return;
  const long long row = rows[ridx];
  const float* hrow = hid + row * ldh;
  for (int c = 0; c < N; ++c) {
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const f32x4_t hv = *reinterpret_cast<const f32x4_t*>(hrow + k), wv = *reinterpret_cast<const f32x4_t*>(W + (long long)c * K + k);
      acc += hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
    }
#pragma unroll
    for (int mm = 1; mm < 64; mm <<= 1) acc += __shfl_xor(acc, mm);
    if (lane == 0) out[row * N + c] = acc + b[c] + X_in[row * dX_in + c];
  }
}

int launch_row_head(const float* hid, int ldh, const float* W, const float* b, const float* X_in, int dX_in, const int* rows,
                    const int* row_count, float* out, int M, int N, int K, hipStream_t st) {
  if (K % 4 != 0 || ldh % 4 != 0 || N > 16) return -1;
  hipLaunchKernelGGL(row_head_kernel, dim3((M + 3) / 4), dim3(256), 0, st, hid, ldh, W, b, X_in, dX_in, rows, row_count, out,
                     N, K);
  return 0;
}

int launch_x_out_residual(const float* mlp, const float* X_in, int dX_in, const uint8_t* mask, float* out, int M,
                          int nout, hipStream_t st) {
  const long long total = (long long)M * nout;
  if (total == 0) return 0;
  hipLaunchKernelGGL(x_out_residual_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, mlp, X_in, dX_in,
                     mask, out, M, nout);
  return 0;
}

}  // namespace scgg
