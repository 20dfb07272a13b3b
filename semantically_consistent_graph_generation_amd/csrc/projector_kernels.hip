// Device side of a projector step: the reference finds the new edges with
// (argmax(z_s.E) - z_t_adj).nonzero() on the full one-hot tensors and copies whole [B,n,n] / [B,n,n,k] tensors to the
// host per graph (projector_3d_utils.py:259-279); here the int8 state is compared in place, the new edges are
// compacted IN THE ORDER nonzero() gives (graph, i, j; upper triangle) into int32 records, and the host's edits are
// scattered back.  HBM-bound byte work: 2 B read per pair, nothing else moves unless an edge is new.
#include "../../include/scgg.h"
#include "kernels.h"

using scgg::fail;

namespace {

constexpr int T = 256;

__device__ inline bool is_new(int s, int t, int binary) { return binary ? (s > 0 && t <= 0) : (s > t); }
__device__ inline bool is_lost(int s, int t, int binary) { return binary ? (t > 0 && s <= 0) : (s < t); }

// counts[b] = new edges of graph b (upper triangle); status[0] |= 1 when an edge of z_t is missing or lowered in z_s
__global__ __launch_bounds__(T) void new_edges_count_kernel(const int8_t* ls, const int8_t* lt, int n, int binary,
                                                            int32_t* counts, int32_t* status) {
  const int b = blockIdx.x;
  const long long base = (long long)b * n * n;
  int c = 0, lost = 0;
  for (int p = threadIdx.x; p < n * n; p += T) {
    const int i = p / n, j = p - i * n;
    const int s = ls[base + p], t = lt[base + p];
    lost |= is_lost(s, t, binary);
    c += (i < j) && is_new(s, t, binary);
  }
  __shared__ int red[T];
  red[threadIdx.x] = c;
  __syncthreads();
  for (int w = T / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[b] = red[0];
  if (lost) atomicOr(status, 1);
}

// in place: counts[0..B) -> exclusive offsets, counts[B] = total.  One workgroup (B is a batch size).
__global__ __launch_bounds__(T) void new_edges_scan_kernel(int32_t* counts, int B) {
  __shared__ int part[T];
  const int per = (B + T - 1) / T;
  const int lo = threadIdx.x * per, hi = min(lo + per, B);
  int s = 0;
  for (int k = lo; k < hi; ++k) s += counts[k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k < T; ++k) { const int v = part[k]; part[k] = run; run += v; }
    counts[B] = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int k = lo; k < hi; ++k) { const int v = counts[k]; counts[k] = run; run += v; }
}

// rec[offsets[b] + rank] = (b, i, j, label, cands[0..k)) for the new edges of graph b in row-major order
__global__ __launch_bounds__(T) void new_edges_fill_kernel(const int8_t* ls, const int8_t* lt, const int8_t* cands, int n,
                                                           int top_k, int binary, const int32_t* offsets, int32_t* rec,
                                                           long long cap) {
  const int b = blockIdx.x;
  if (offsets[b + 1] == offsets[b]) return;
  const long long base = (long long)b * n * n;
  const int stride = 4 + top_k;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int wsum[T / 64];
  int run = offsets[b];
  for (int p0 = 0; p0 < n * n; p0 += T) {
    const int p = p0 + threadIdx.x;
    bool hit = false;
    int i = 0, j = 0, s = 0;
    if (p < n * n) {
      i = p / n;
      j = p - i * n;
      s = ls[base + p];
      hit = (i < j) && is_new(s, lt[base + p], binary);
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < T / 64; ++w) {
      before += w < wave ? wsum[w] : 0;
      total += wsum[w];
    }
    if (hit) {
      const long long slot = run + before + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < cap) {
        int32_t* r = rec + slot * stride;
        r[0] = b; r[1] = i; r[2] = j; r[3] = s;
        for (int k = 0; k < top_k; ++k) r[4 + k] = cands[(base + p) * top_k + k];
      }
    }
    run += total;
    __syncthreads();
  }
}

__global__ __launch_bounds__(T) void apply_edits_kernel(int8_t* labels, const int32_t* edits, long long ne, int B, int n) {
  const long long k = blockIdx.x * (long long)T + threadIdx.x;
  if (k >= ne) return;
  const int b = edits[k * 4], u = edits[k * 4 + 1], v = edits[k * 4 + 2], c = edits[k * 4 + 3];
  if (b < 0 || b >= B || u < 0 || u >= n || v < 0 || v >= n) return;       // never write outside the state
  labels[((long long)b * n + u) * n + v] = (int8_t)c;
  labels[((long long)b * n + v) * n + u] = (int8_t)c;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
  return 0;
}

}  // namespace

extern "C" {

int scgg_new_edges(const int8_t* labels_s, const int8_t* labels_t, const int8_t* cands, int32_t B, int32_t n, int32_t top_k,
                   int32_t binary, int32_t* offsets, int32_t* rec, int64_t cap, int32_t* status, void* stream) {
  if (!labels_s || !labels_t || !offsets || !rec || !status || B <= 0 || n <= 0 || cap < 0) return fail("new_edges: bad argument");
  if (top_k < 0 || (top_k > 0 && !cands)) return fail("new_edges: top_k=%d needs the candidate tensor", top_k);
  if ((long long)n * n > 0x7fffffffll / 2) return fail("new_edges: n=%d is too large", n);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(status, 0, sizeof(int32_t), st) != hipSuccess) return fail("new_edges: clearing the status word failed");
  hipLaunchKernelGGL(new_edges_count_kernel, dim3(B), dim3(T), 0, st, labels_s, labels_t, n, binary, offsets, status);
  hipLaunchKernelGGL(new_edges_scan_kernel, dim3(1), dim3(T), 0, st, offsets, B);
  hipLaunchKernelGGL(new_edges_fill_kernel, dim3(B), dim3(T), 0, st, labels_s, labels_t, cands, n, top_k, binary, offsets, rec,
                     (long long)cap);
  return check_launch("new_edges");
}

int scgg_apply_edge_edits(int8_t* labels, const int32_t* edits, int64_t n_edits, int32_t B, int32_t n, void* stream) {
  if (!labels || n_edits < 0 || (n_edits && !edits) || B <= 0 || n <= 0) return fail("apply_edge_edits: bad argument");
  if (n_edits == 0) return 0;
  hipLaunchKernelGGL(apply_edits_kernel, dim3((unsigned)((n_edits + T - 1) / T)), dim3(T), 0, static_cast<hipStream_t>(stream),
                     labels, edits, (long long)n_edits, B, n);
  return check_launch("apply_edge_edits");
}

}  // extern "C"
