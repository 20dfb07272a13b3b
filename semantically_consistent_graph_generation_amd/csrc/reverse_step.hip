// Reverse-diffusion step kernels for gfx950: softmax -> posterior p(z_s | z_t, x0) summed over x0 ->
// symmetric categorical edge sampling with top-k candidates.  HBM-bound, elementwise, one thread per unordered
// pair (i<j) which writes both (i,j) and (j,i).
//
// Reference: NoiseModel.sample_zs_from_zt_and_pred  noise_model.py:342-447,
//            compute_batched_over0_posterior_distribution  diffusion_utils.py:315-343,
//            sample_discrete_features  diffusion_utils.py:200-285,
//            get_Qt / get_Qt_bar  noise_model.py:101-154 and :524-552 (absorbing edges).
//
// All transitions of the reference have the form  Q = a*I + (1-a)*1 m^T  (m = limit marginals), so for a
// one-hot z_t = k the sum over x0 collapses to O(C) per pair (DESIGN.md "Reverse step"):
//   den[a]  = (1-abar_t) m_k + abar_t [a==k]              (0 -> 1e-6, diffusion_utils.py:340)
//   S       = sum_a pred[a] / den[a]
//   un[s]   = (beta_t m_k + (1-beta_t)[s==k]) * ((1-abar_s) m_s S + abar_s pred[s] / den[s])
//   prob    = un / sum(un)                                 (all-zero -> uniform, noise_model.py:395)
// Sampling is the documented deterministic rule (sequential fp32 inverse CDF without replacement) driven either by
// caller-supplied uniforms or by Philox4x32-10 keyed on (seed; i, j, graph id, step).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.h"
#include "sampling.h"

namespace scgg {

constexpr int MAXC = 32;
constexpr int MAXK = 8;

// sequential inverse-CDF draws without replacement; p is consumed.  Must match oracle/noise.py:inverse_cdf_draws.
// MC: compile-time bound of the class loops (>= C): 16 for the usual edge-class counts, MAXC otherwise; the operations on the
// classes c < C and their order do not depend on it
template <int MC>
__device__ __forceinline__ void draw_sequence(float (&p)[MC], int C, const float* u, int k, int8_t* out) {
  uint32_t picked = 0;
  for (int r = 0; r < k; ++r) {
    float total = 0.f;
#pragma unroll
    for (int c = 0; c < MC; ++c)
      if (c < C) total = total + p[c];
    const float thr = u[r] * total;
    float acc = 0.f;
    int sel = -1, last = -1;
#pragma unroll
    for (int c = 0; c < MC; ++c)
      if (c < C) {
        acc = acc + p[c];
        if (sel < 0 && thr < acc) sel = c;
        if (p[c] > 0.f) last = c;
      }
    if (sel < 0) sel = last;
    if (sel < 0) sel = __ffs(~picked) - 1;
    out[r] = (int8_t)sel;
#pragma unroll
    for (int c = 0; c < MC; ++c)
      if (c == sel) p[c] = 0.f;
    picked |= 1u << sel;
  }
}

__device__ __forceinline__ void get_uniforms(const SampleRng& g, int b, int i, int j, long long pair, int top_k,
                                             float* u /*[1+top_k]*/) {
  if (g.u0 != nullptr) {
    u[0] = g.u0[pair];
    for (int r = 0; r < top_k; ++r) u[1 + r] = g.uk[pair * top_k + r];
  } else {
    const uint32_t k0 = (uint32_t)(g.seed & 0xFFFFFFFFull), k1 = (uint32_t)(g.seed >> 32);
    const int nblk = (1 + top_k + 3) >> 2;
    for (int blk = 0; blk < nblk; ++blk) {
      uint32_t o[4];
      philox4x32_10(((uint32_t)i << 16) | (uint32_t)j, g.graph_id_base + (uint32_t)b, g.step,
                    (uint32_t)(STREAM_E * 16 + blk), k0, k1, o);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * blk + e < 1 + top_k) u[4 * blk + e] = u01(o[e]);
    }
  }
}

__device__ __forceinline__ void write_pair(const SampleOut& o, long long pij, long long pji, int top_k, int8_t lab,
                                           const int8_t* cand) {
  o.labels[pij] = lab;
  o.labels[pji] = lab;
  if (o.cands != nullptr) {
    if (top_k == 4 && (reinterpret_cast<uintptr_t>(o.cands) & 3) == 0) {      // the usual case: one 32-bit store per orientation
      const uint32_t v = (uint32_t)(uint8_t)cand[0] | ((uint32_t)(uint8_t)cand[1] << 8) | ((uint32_t)(uint8_t)cand[2] << 16) |
                         ((uint32_t)(uint8_t)cand[3] << 24);
      reinterpret_cast<uint32_t*>(o.cands)[pij] = v;
      reinterpret_cast<uint32_t*>(o.cands)[pji] = v;
    } else {
      for (int r = 0; r < top_k; ++r) {
        o.cands[pij * top_k + r] = cand[r];
        o.cands[pji * top_k + r] = cand[r];
      }
    }
  }
}

// sample one live pair from p (consumed) and write it
template <int MC>
__device__ __forceinline__ void sample_and_write(float (&p)[MC], int C, int top_k, const SampleRng& g, int b, int i,
                                                 int j, long long pij, long long pji, const SampleOut& o) {
  float u[1 + MAXK];
  get_uniforms(g, b, i, j, pij, top_k, u);
  int8_t cand[MAXK];
  int8_t lab;
  if (top_k > 0) {
    float p1[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) p1[c] = p[c];
    draw_sequence(p1, C, u, 1, &lab);              // E_t = multinomial(1)            diffusion_utils.py:239
    draw_sequence(p, C, u + 1, top_k, cand);       // multinomial(top_k)              :246
    cand[0] = lab;                                 // slot 0 overwritten              :249
  } else {
    draw_sequence(p, C, u, 1, &lab);
  }
  write_pair(o, pij, pji, top_k, lab, cand);
}

// --------------------------------------------------------------------------------------------------
// one live pair: softmax of its logit row (lrow, in LDS), posterior, and for i < j the draws
template <int MC>
__device__ __forceinline__ void reverse_step_pair(const ReverseStepArgs& a, const float* lrow, long long pij, int b, int i, int j, int k) {
  const int C = a.C;
  const long long pji = (long long)b * a.n * a.n + (long long)j * a.n + i;
  float pr[MC];
  float mxl = -INFINITY;
#pragma unroll
  for (int c = 0; c < MC; ++c) {
    pr[c] = c < C ? lrow[c] : -INFINITY;
    mxl = fmaxf(mxl, pr[c]);
  }
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < MC; ++c) {
    pr[c] = c < C ? expf(pr[c] - mxl) : 0.f;
    se += pr[c];
  }
  const float inv = 1.0f / se;
  const int t = a.t_int[b], s = a.s_int[b];
  const float beta_t = a.beta_tab[t], abar_t = a.abar_tab[t], abar_s = a.abar_tab[s];
  const float mk = a.marg[k];
  float S = 0.f;
  float q[MC];   // pred[a]/den[a]
#pragma unroll
  for (int c = 0; c < MC; ++c) {
    q[c] = 0.f;
    if (c < C) {
      float den = (1.0f - abar_t) * mk + (c == k ? abar_t : 0.f);
      if (den == 0.f) den = 1e-6f;
      q[c] = pr[c] * inv / den;
      S += q[c];
    }
  }
  float tot = 0.f;
#pragma unroll
  for (int c = 0; c < MC; ++c) {
    float un = 0.f;
    if (c < C) {
      const float qt = beta_t * mk + (c == k ? 1.0f - beta_t : 0.f);
      un = qt * ((1.0f - abar_s) * a.marg[c] * S + abar_s * q[c]);
    }
    pr[c] = un;
    tot += un;
  }
  if (tot == 0.f) {
#pragma unroll
    for (int c = 0; c < MC; ++c) pr[c] = c < C ? 1e-5f : 0.f;
    tot = 1e-5f * C;
  }
#pragma unroll
  for (int c = 0; c < MC; ++c) pr[c] = pr[c] / tot;
  if (a.prob_out != nullptr)
    for (int c = 0; c < C; ++c) a.prob_out[pij * C + c] = pr[c];
  if (i < j) sample_and_write(pr, C, a.top_k, a.rng, b, i, j, pij, pji, a.out);
}

template <int MC>
__global__ __launch_bounds__(256) void reverse_step_kernel(ReverseStepArgs a) {
  // the workgroup's 256 logit rows (consecutive pairs) through LDS: coalesced reads; a thread's own row is C floats at a
  // stride of C between lanes otherwise (28 cache lines per load instruction at C = 14)
  __shared__ float lg[256 * MC];
  __shared__ short queue[256];
  __shared__ int wcount[4];
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  const long long nn = (long long)a.n * a.n;
  const int C = a.C;
  {
    const long long base = blockIdx.x * 256ll * C, total = a.B * nn * C;
    for (int k = threadIdx.x; k < 256 * C; k += 256) lg[k] = base + k < total ? a.logits[base + k] : 0.f;
  }
  // this thread's pair: dead pairs are written at once; a live pair that needs the arithmetic is queued
  bool work = false;
  int b = 0, i = 0, j = 0, k = -1;
  if (idx < a.B * nn) {
    b = (int)(idx / nn);
    const int ij = (int)(idx - b * nn);
    i = ij / a.n; j = ij - i * a.n;
    if (!(i > j && a.prob_out == nullptr)) {
      const bool live = i != j && a.mask[b * a.n + i] != 0 && a.mask[b * a.n + j] != 0;
      k = live ? a.labels_t[idx] : -1;
      if (k < 0) {   // diagonal / padded pair (or a pair the caller marked dead): zero one-hot row -> label -1
        if (i <= j) {
          int8_t zero[MAXK] = {0};
          write_pair(a.out, idx, (long long)b * nn + (long long)j * a.n + i, a.top_k, -1, zero);
        }
        if (a.prob_out != nullptr)
          for (int c = 0; c < C; ++c) a.prob_out[idx * C + c] = 1.0f / C;   // what the reference leaves there
      } else {
        work = true;
      }
    }
  }
  // Live pairs are a quarter of the threads in the sampling loop (upper triangle of the valid block): they are compacted
  // over the workgroup so that the class loops and the Philox draws run in full waves and the other waves leave.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(work);
  if (lane == 0) wcount[wave] = __popcll(bal);
  __syncthreads();                                           // also: lg is complete
  int off = 0, total_work = 0;
  for (int w = 0; w < 4; ++w) {
    if (w < wave) off += wcount[w];
    total_work += wcount[w];
  }
  if (work) queue[off + __popcll(bal & ((1ull << lane) - 1ull))] = (short)threadIdx.x;
  __syncthreads();
  if ((int)threadIdx.x >= total_work) return;
  const int li = queue[threadIdx.x];
  const long long pij = blockIdx.x * 256ll + li;
  const int b2 = (int)(pij / nn);
  const int ij2 = (int)(pij - b2 * nn);
  const int i2 = ij2 / a.n, j2 = ij2 - i2 * a.n;
  reverse_step_pair<MC>(a, lg + li * C, pij, b2, i2, j2, a.labels_t[pij]);
}

// generic form: probabilities supplied (any transition; also the training-time apply_noise draw)
__global__ __launch_bounds__(256) void sample_from_probs_kernel(SampleProbsArgs a) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  const long long nn = (long long)a.n * a.n;
  if (idx >= a.B * nn) return;
  const int b = idx / nn;
  const int ij = idx - b * nn;
  const int i = ij / a.n, j = ij - i * a.n;
  if (i > j) return;
  const long long pij = idx, pji = (long long)b * nn + (long long)j * a.n + i;
  const bool live = i != j && a.mask[b * a.n + i] != 0 && a.mask[b * a.n + j] != 0;
  if (!live) {
    int8_t zero[MAXK] = {0};
    write_pair(a.out, pij, pji, a.top_k, -1, zero);
    return;
  }
  float pr[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) pr[c] = c < a.C ? a.prob[pij * a.C + c] : 0.f;
  sample_and_write(pr, a.C, a.top_k, a.rng, b, i, j, pij, pji, a.out);
}

// labels int8 (-1 = zero row) -> masked one-hot float E[B,n,n,C] (+ optional int64 extra_info from the int8 candidates)
__global__ __launch_bounds__(256) void labels_to_onehot_kernel(const int8_t* labels, const int8_t* cands, long long P,
                                                               int C, int top_k, float* E, long long* extra) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= P) return;
  const int lab = labels[idx];
  for (int c = 0; c < C; ++c) E[idx * C + c] = (c == lab) ? 1.0f : 0.0f;
  if (extra != nullptr)
    for (int r = 0; r < top_k; ++r) extra[idx * top_k + r] = lab < 0 ? 0 : (long long)cands[idx * top_k + r];
}

// one-hot float rows -> int8 labels; an all-zero row (masked) -> -1
__global__ __launch_bounds__(256) void onehot_to_labels_kernel(const float* E, long long P, int C, int8_t* labels) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= P) return;
  int best = -1;
  float bv = 0.f;
  for (int c = 0; c < C; ++c) {
    const float v = E[idx * C + c];
    if (v > bv) { bv = v; best = c; }
  }
  labels[idx] = (int8_t)best;
}

static inline dim3 grid_for(long long total) { return dim3((unsigned)((total + 255) / 256)); }

int launch_reverse_step(const ReverseStepArgs& a, hipStream_t st) {
  if (a.C > MAXC || a.top_k > MAXK) return -1;
  if (a.C <= 16) hipLaunchKernelGGL(reverse_step_kernel<16>, grid_for((long long)a.B * a.n * a.n), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(reverse_step_kernel<MAXC>, grid_for((long long)a.B * a.n * a.n), dim3(256), 0, st, a);
  return 0;
}
int launch_sample_from_probs(const SampleProbsArgs& a, hipStream_t st) {
  if (a.C > MAXC || a.top_k > MAXK) return -1;
  hipLaunchKernelGGL(sample_from_probs_kernel, grid_for((long long)a.B * a.n * a.n), dim3(256), 0, st, a);
  return 0;
}
int launch_labels_to_onehot(const int8_t* labels, const int8_t* cands, long long P, int C, int top_k, float* E,
                            long long* extra, hipStream_t st) {
  hipLaunchKernelGGL(labels_to_onehot_kernel, grid_for(P), dim3(256), 0, st, labels, cands, P, C, top_k, E, extra);
  return 0;
}
int launch_onehot_to_labels(const float* E, long long P, int C, int8_t* labels, hipStream_t st) {
  hipLaunchKernelGGL(onehot_to_labels_kernel, grid_for(P), dim3(256), 0, st, E, P, C, labels);
  return 0;
}

}  // namespace scgg
