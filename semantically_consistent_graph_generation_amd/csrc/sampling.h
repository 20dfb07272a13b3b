// Argument blocks of the reverse-step / sampling kernels (internal; the C ABI is include/scgg.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scgg {

struct SampleRng {
  const float* u0;          // [B,n,n] uniforms of the single draw, or null -> Philox
  const float* uk;          // [B,n,n,top_k] uniforms of the top-k draw
  uint64_t seed;
  uint32_t step;
  uint32_t graph_id_base;   // global index of graph 0 of this batch (keeps results independent of the sharding)
};

struct SampleOut {
  int8_t* labels;           // [B,n,n]  class index, -1 where the one-hot row is the zero vector (diag / padding)
  int8_t* cands;            // [B,n,n,top_k] or null
};

struct ReverseStepArgs {
  const float* logits;      // [B,n,n,C] denoiser output (x0 logits)
  const int8_t* labels_t;   // [B,n,n]   z_t
  const uint8_t* mask;      // [B,n]
  const int32_t* t_int;     // [B]
  const int32_t* s_int;     // [B]
  const float* beta_tab;    // [T+1] one-step beta of the E branch (unclamped beta for absorbing edges)
  const float* abar_tab;    // [T+1] alpha_bar of the E branch
  const float* marg;        // [C]   limit marginals m of the E transition
  float* prob_out;          // [B,n,n,C] or null (debug / tests)
  SampleRng rng;
  SampleOut out;
  int B, n, C, top_k;
};

struct SampleProbsArgs {
  const float* prob;        // [B,n,n,C]
  const uint8_t* mask;
  SampleRng rng;
  SampleOut out;
  int B, n, C, top_k;
};

int launch_reverse_step(const ReverseStepArgs& a, hipStream_t st);
int launch_sample_from_probs(const SampleProbsArgs& a, hipStream_t st);
int launch_labels_to_onehot(const int8_t* labels, const int8_t* cands, long long P, int C, int top_k, float* E,
                            long long* extra, hipStream_t st);
int launch_onehot_to_labels(const float* E, long long P, int C, int8_t* labels, hipStream_t st);

}  // namespace scgg
