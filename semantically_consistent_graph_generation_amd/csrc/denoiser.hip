// Host-side engine behind scgg_denoiser_* (include/scgg.h): owns the packed-weight arena, lays out the caller's
// workspace and enqueues the kernels of one GraphTransformer forward on one HIP stream.  No torch types, no
// synchronisation, no allocation after finalize.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <atomic>
#include <stdlib.h>
#include <string>
#include <vector>

#include "../../include/scgg.h"
#include "kernels.h"
#include "sampling.h"

// Scratch of the allocation-free entry point scgg_denoiser_forward_static (include/scgg.h): the forward's workspace as a
// file-scope device array.  Every GPU the library runs on holds its own copy until the process ends; host code never
// uses the array's name as an address, only the per-device pointer hipGetSymbolAddress returned (static_scratch()).
#define SCGG_STATIC_SCRATCH_BYTES (32u << 20)
static __device__ __align__(256) unsigned char scgg_static_scratch[SCGG_STATIC_SCRATCH_BYTES];

namespace scgg {

static thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
const char* last_error() { return g_err; }

static int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
  return 0;
}

// device address of this GPU's copy of scgg_static_scratch: looked up once per device, with that device current
static void* static_scratch() {
  constexpr int MAX_DEV = 64;
  static std::atomic<void*> ptr[MAX_DEV] = {};              // threads may race to the first lookup: both find the same address
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) { fail("static scratch: no current device"); return nullptr; }
  void* p = ptr[dev].load(std::memory_order_acquire);
  if (p == nullptr) {
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scgg_static_scratch));
    if (e != hipSuccess || p == nullptr) { fail("hipGetSymbolAddress(scgg_static_scratch): %s", hipGetErrorString(e)); return nullptr; }
    ptr[dev].store(p, std::memory_order_release);
  }
  return p;
}

static inline int pad32(int v) { return (v + 31) / 32 * 32; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Param {
  const float* p;
  int64_t numel;
};

struct PackedLinear {   // chain-kernel operand
  f32x4_t* W = nullptr;
  f32x4_t* W3 = nullptr;  // three-term bf16 image of the same matrix (node_layer_x3_kernel)
  float* b = nullptr;
  int N = 0, K = 0;
};

struct LayerPack {
  bool predicts_y = true;
  f32x4_t* phase_img = nullptr;   // edge_unit_layer_kernel's per-phase weight images (edge_unit_layer.hip)
  float* b_out = nullptr;         // e_out.bias (padded copy)
  PackedLinear qkv, x_out, linX1, linX2;     // node path, row-list kernel (W == nullptr: shape not supported, generic GEMM)
  float *ln1_g, *ln1_b, *ln2_g, *ln2_b;      // padded copies
  float *Wqkv, *bqkv;                        // [3dx,dx]
  float *Wfilm, *bfilm;                      // [4dx,dy]
  float *Wycat, *bycat;                      // [dy, dy+4dx+4de]
  float *Wt_cat, *Wt_o0, *Wt_o2, *Wt_l1, *Wt_l2, *Wt_film;   // transposed copies for y_layer_kernel ([K][N])
};

}  // namespace scgg

using namespace scgg;

struct scgg_denoiser {
  scgg_denoiser_config c;
  std::map<std::string, Param> params;
  bool finalized = false;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  f32x4_t *in_E_img = nullptr, *out_E_img = nullptr;   // LDS images of the edge input / output MLPs (edge_mlp_unit.hip)
  PackedLinear in_X2, out_X1;
  std::vector<LayerPack> layers;
  // optional live timing of the dominant kernel (edge_layer_kernel): HIP events on the launch stream
  int prof_every = 0;            // 0 = off; k = time the layer launches of every k-th forward
  long long prof_forwards = 0;
  std::vector<hipEvent_t> prof_ev;   // pairs (start, stop), created by scgg_denoiser_profile
  size_t prof_used = 0;
  // second stream for the global-feature (y) path, which only meets the node / edge paths at the FiLM vectors and at
  // the pooled statistics: it runs beside the X path instead of after it (fork / join by events, capture-safe)
  hipStream_t aux = nullptr;
  std::vector<hipEvent_t> sync_ev;
};

namespace {

struct Arena {   // two-pass bump allocator: first pass measures, second pass hands out pointers
  char* base;
  size_t off = 0;
  explicit Arena(char* b) : base(b) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

const Param* find(const scgg_denoiser* h, const std::string& name, int64_t numel) {
  auto it = h->params.find(name);
  if (it == h->params.end()) {
    fail("missing parameter '%s'", name.c_str());
    return nullptr;
  }
  if (it->second.numel != numel) {
    fail("parameter '%s' has %lld elements, expected %lld", name.c_str(), (long long)it->second.numel,
         (long long)numel);
    return nullptr;
  }
  return &it->second;
}

// allocate (+ optionally fill) one packed Linear for the chain kernels
int pack_linear(scgg_denoiser* h, Arena& ar, bool fill, const std::string& name, int N, int K, float bias_add,
                PackedLinear& out, hipStream_t st) {
  const int NB = (N + 31) / 32, KB = (K + 31) / 32;
  out.N = N;
  out.K = K;
  out.W = ar.take<f32x4_t>((size_t)NB * KB * 256);
  out.b = ar.take<float>((size_t)NB * 32);
  if (!fill) return 0;
  const Param* w = find(h, name + ".weight", (int64_t)N * K);
  const Param* b = find(h, name + ".bias", N);
  if (!w || !b) return -1;
  PackArgs pa{w->p, out.W, N, K, NB, KB};
  launch_pack_linear(pa, st);
  launch_pad_vec(b->p, out.b, N, NB * 32, bias_add, st);
  return check_launch("pack_linear");
}

static inline bool node_kernel_ok(int K) { return K == 32 || K == 64 || K == 128 || K == 256; }

// packed copy of a raw (engine-owned or borrowed) weight / bias pair for the row-list node kernel
int pack_raw(Arena& ar, bool fill, const float* W, const float* b, int N, int K, PackedLinear& out, hipStream_t st) {
  out = PackedLinear();
  if (!node_kernel_ok(K)) return 0;
  const int NB = (N + 31) / 32, KB = K / 32;
  out.N = N;
  out.K = K;
  out.W = ar.take<f32x4_t>((size_t)NB * KB * 256);
  out.W3 = ar.take<f32x4_t>((size_t)NB * (K / 16) * 192);
  out.b = ar.take<float>((size_t)NB * 32);
  if (!fill) return 0;
  PackArgs pa{W, out.W, N, K, NB, KB};
  launch_pack_linear(pa, st);
  PackArgs p3{W, out.W3, N, K, NB, KB};
  launch_pack_linear_x3(p3, st);
  launch_pad_vec(b, out.b, N, NB * 32, 0.f, st);
  return check_launch("pack_raw");
}

int pack_named(scgg_denoiser* h, Arena& ar, bool fill, const std::string& name, int N, int K, PackedLinear& out,
               hipStream_t st) {
  const float *W = nullptr, *b = nullptr;
  if (fill && node_kernel_ok(K)) {
    const Param* w = find(h, name + ".weight", (int64_t)N * K);
    const Param* bb = find(h, name + ".bias", N);
    if (!w || !bb) return -1;
    W = w->p;
    b = bb->p;
  }
  return pack_raw(ar, fill, W, b, N, K, out, st);
}

int pad_vec(scgg_denoiser* h, Arena& ar, bool fill, const std::string& name, int n, float*& out, hipStream_t st) {
  out = ar.take<float>((size_t)pad32(n));
  if (!fill) return 0;
  const Param* v = find(h, name, n);
  if (!v) return -1;
  launch_pad_vec(v->p, out, n, pad32(n), 0.f, st);
  return check_launch("pad_vec");
}

// dst[N, ldd] columns [col0, col0+K) <- W[N,K]
int put_cols(scgg_denoiser* h, const std::string& name, int N, int K, float* dst, int ldd, int col0, hipStream_t st) {
  const Param* w = find(h, name, (int64_t)N * K);
  if (!w) return -1;
  launch_copy_cols(w->p, K, dst + col0, ldd, N, K, st);
  return 0;
}

__global__ void vec_sum3_kernel(const float* a, const float* b, const float* c, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i] + c[i];
}

int build(scgg_denoiser* h, bool fill, hipStream_t st) {
  const scgg_denoiser_config& c = h->c;
  Arena ar(fill ? h->arena : nullptr);
  {
    const long long n_in = edge_mlp_img_f4(c.in_E, c.hid_E, c.de), n_out = edge_mlp_img_f4(c.de, c.hid_E, c.out_E);
    if (n_in < 0 || n_out < 0) return fail("edge MLP kernels have no build for hid_E=%d", c.hid_E);
    h->in_E_img = ar.take<f32x4_t>((size_t)n_in);
    h->out_E_img = ar.take<f32x4_t>((size_t)n_out);
    if (fill) {
      const Param* w0 = find(h, "mlp_in_E.0.weight", (int64_t)c.hid_E * c.in_E); const Param* b0 = find(h, "mlp_in_E.0.bias", c.hid_E);
      const Param* w3 = find(h, "mlp_in_E.3.weight", (int64_t)c.de * c.hid_E); const Param* b3 = find(h, "mlp_in_E.3.bias", c.de);
      const Param* v0 = find(h, "mlp_out_E.0.weight", (int64_t)c.hid_E * c.de); const Param* d0 = find(h, "mlp_out_E.0.bias", c.hid_E);
      const Param* v3 = find(h, "mlp_out_E.3.weight", (int64_t)c.out_E * c.hid_E); const Param* d3 = find(h, "mlp_out_E.3.bias", c.out_E);
      if (!w0 || !b0 || !w3 || !b3 || !v0 || !d0 || !v3 || !d3) return -1;
      EdgeMlpPackArgs pi{w0->p, b0->p, w3->p, b3->p, h->in_E_img, c.in_E, c.hid_E, c.de};
      EdgeMlpPackArgs po{v0->p, d0->p, v3->p, d3->p, h->out_E_img, c.de, c.hid_E, c.out_E};
      if (launch_edge_mlp_pack(pi, st) || launch_edge_mlp_pack(po, st))
        return fail("edge MLP pack: unsupported dims in_E=%d hid_E=%d de=%d out_E=%d", c.in_E, c.hid_E, c.de, c.out_E);
    }
  }
  if (pack_named(h, ar, fill, "mlp_in_X.3", c.dx, c.hid_X, h->in_X2, st)) return -1;
  if (pack_named(h, ar, fill, "mlp_out_X.0", c.hid_X, c.dx, h->out_X1, st)) return -1;
  h->layers.resize(c.n_layers);
  const int dx = c.dx, de = c.de, dy = c.dy;
  const int kcat = dy + 4 * dx + 4 * de;
  for (int l = 0; l < c.n_layers; ++l) {
    LayerPack& L = h->layers[l];
    L.predicts_y = (l != c.n_layers - 1) || c.out_y > 0;
    const std::string p = "tf_layers." + std::to_string(l) + ".";
    const std::string sa = p + "self_attn.";
    {
      const long long nimg = edge_phase_img_f4(dx, de, c.dim_ffE);
      if (nimg < 0) return fail("edge layer kernel has no build for dx=%d de=%d dim_ffE=%d", dx, de, c.dim_ffE);
      L.phase_img = ar.take<f32x4_t>((size_t)nimg);
      if (fill) {
        const Param* wm = find(h, sa + "e_mul.weight", (int64_t)dx * de); const Param* bm = find(h, sa + "e_mul.bias", dx);
        const Param* wa = find(h, sa + "e_add.weight", (int64_t)dx * de); const Param* ba = find(h, sa + "e_add.bias", dx);
        const Param* wo = find(h, sa + "e_out.weight", (int64_t)de * dx);
        const Param* w1 = find(h, p + "linE1.weight", (int64_t)c.dim_ffE * de); const Param* b1 = find(h, p + "linE1.bias", c.dim_ffE);
        const Param* w2 = find(h, p + "linE2.weight", (int64_t)de * c.dim_ffE); const Param* b2 = find(h, p + "linE2.bias", de);
        if (!wm || !bm || !wa || !ba || !wo || !w1 || !b1 || !w2 || !b2) return -1;
        EdgeUnitPackArgs pa{wm->p, bm->p, wa->p, ba->p, wo->p, w1->p, b1->p, w2->p, b2->p, L.phase_img};
        if (launch_edge_unit_pack(pa, dx, de, c.dim_ffE, st)) return fail("edge phase pack failed");
      }
    }
    if (pad_vec(h, ar, fill, sa + "e_out.bias", de, L.b_out, st)) return -1;
    if (pad_vec(h, ar, fill, p + "normE1.weight", de, L.ln1_g, st)) return -1;
    if (pad_vec(h, ar, fill, p + "normE1.bias", de, L.ln1_b, st)) return -1;
    if (pad_vec(h, ar, fill, p + "normE2.weight", de, L.ln2_g, st)) return -1;
    if (pad_vec(h, ar, fill, p + "normE2.bias", de, L.ln2_b, st)) return -1;
    L.Wqkv = ar.take<float>((size_t)3 * dx * dx);
    L.bqkv = ar.take<float>((size_t)3 * dx);
    L.Wfilm = ar.take<float>((size_t)4 * dx * dy);
    L.bfilm = ar.take<float>((size_t)4 * dx);
    L.Wycat = ar.take<float>((size_t)dy * kcat);
    L.bycat = ar.take<float>((size_t)dy);
    L.Wt_cat = ar.take<float>((size_t)dy * kcat);
    L.Wt_o0 = ar.take<float>((size_t)dy * dy);
    L.Wt_o2 = ar.take<float>((size_t)dy * dy);
    L.Wt_l1 = ar.take<float>((size_t)dy * c.dim_ffy);
    L.Wt_l2 = ar.take<float>((size_t)dy * c.dim_ffy);
    L.Wt_film = ar.take<float>((size_t)4 * dx * dy);
    if (pack_named(h, ar, fill, sa + "x_out", dx, dx, L.x_out, st)) return -1;
    if (pack_named(h, ar, fill, p + "linX1", c.dim_ffX, dx, L.linX1, st)) return -1;
    if (pack_named(h, ar, fill, p + "linX2", dx, c.dim_ffX, L.linX2, st)) return -1;
    {   // reserve the packed q|k|v slot in both passes; it is filled after the concatenation below
      PackedLinear tmp;
      if (pack_raw(ar, false, nullptr, nullptr, 3 * dx, dx, tmp, st)) return -1;
      L.qkv = tmp;
    }
    if (!fill) continue;
    const char* qkv[3] = {"q", "k", "v"};
    for (int t = 0; t < 3; ++t) {
      if (put_cols(h, sa + qkv[t] + ".weight", dx, dx, L.Wqkv + (size_t)t * dx * dx, dx, 0, st)) return -1;
      if (put_cols(h, sa + qkv[t] + ".bias", 1, dx, L.bqkv + (size_t)t * dx, dx, 0, st)) return -1;
    }
    if (L.qkv.W != nullptr) {
      PackArgs pa{L.Wqkv, L.qkv.W, 3 * dx, dx, (3 * dx + 31) / 32, dx / 32};
      launch_pack_linear(pa, st);
      PackArgs p3{L.Wqkv, L.qkv.W3, 3 * dx, dx, (3 * dx + 31) / 32, dx / 32};
      launch_pack_linear_x3(p3, st);
      launch_pad_vec(L.bqkv, L.qkv.b, 3 * dx, (3 * dx + 31) / 32 * 32, 0.f, st);
    }
    const char* fl[4] = {"y_e_add", "y_e_mul", "y_x_add", "y_x_mul"};
    for (int t = 0; t < 4; ++t) {
      if (put_cols(h, sa + fl[t] + ".weight", dx, dy, L.Wfilm + (size_t)t * dx * dy, dy, 0, st)) return -1;
      if (put_cols(h, sa + fl[t] + ".bias", 1, dx, L.bfilm + (size_t)t * dx, dx, 0, st)) return -1;
    }
    if (L.predicts_y) {
      if (put_cols(h, sa + "y_y.weight", dy, dy, L.Wycat, kcat, 0, st)) return -1;
      if (put_cols(h, sa + "x_y.lin.weight", dy, 4 * dx, L.Wycat, kcat, dy, st)) return -1;
      if (put_cols(h, sa + "e_y.lin.weight", dy, 4 * de, L.Wycat, kcat, dy + 4 * dx, st)) return -1;
      const Param* b0 = find(h, sa + "y_y.bias", dy);
      const Param* b1 = find(h, sa + "x_y.lin.bias", dy);
      const Param* b2 = find(h, sa + "e_y.lin.bias", dy);
      if (!b0 || !b1 || !b2) return -1;
      hipLaunchKernelGGL(vec_sum3_kernel, dim3((dy + 255) / 256), dim3(256), 0, st, b0->p, b1->p, b2->p, L.bycat, dy);
    }
    launch_transpose(L.Wfilm, L.Wt_film, 4 * dx, dy, st);
    if (L.predicts_y) {
      const Param* o0 = find(h, sa + "y_out.0.weight", (int64_t)dy * dy);
      const Param* o2 = find(h, sa + "y_out.2.weight", (int64_t)dy * dy);
      const Param* l1 = find(h, p + "lin_y1.weight", (int64_t)c.dim_ffy * dy);
      const Param* l2 = find(h, p + "lin_y2.weight", (int64_t)dy * c.dim_ffy);
      if (!o0 || !o2 || !l1 || !l2) return -1;
      launch_transpose(L.Wycat, L.Wt_cat, dy, kcat, st);
      launch_transpose(o0->p, L.Wt_o0, dy, dy, st);
      launch_transpose(o2->p, L.Wt_o2, dy, dy, st);
      launch_transpose(l1->p, L.Wt_l1, c.dim_ffy, dy, st);
      launch_transpose(l2->p, L.Wt_l2, dy, c.dim_ffy, st);
    }
    if (check_launch("concat weights")) return -1;
  }
  h->arena_bytes = align_up(ar.off, 256);
  return 0;
}

struct Workspace {
  float *Xh, *qkv, *newX, *t1, *t2, *E0, *E1, *y, *ycat, *yt1, *yt2, *film, *hidX, *hidy, *xo;
  int *rows, *row_count, *units, *unit_count, *goff;
  unsigned* chunk_bits;   // [B] valid-chunk bits per graph (launch_build_lists)
  float *epool, *attn, *epart;
  size_t bytes;
};

Workspace carve(const scgg_denoiser_config& c, int B, int n, char* base) {
  Arena ar(base);
  Workspace w;
  const size_t M = (size_t)B * n, P = M * n;
  const int kcat = c.dy + 4 * c.dx + 4 * c.de;
  w.Xh = ar.take<float>(M * c.dx);
  w.qkv = ar.take<float>(M * 3 * c.dx);
  w.newX = ar.take<float>(M * c.dx);
  w.t1 = ar.take<float>(M * c.dx);
  w.t2 = ar.take<float>(M * (size_t)(c.dim_ffX > c.hid_X ? c.dim_ffX : c.hid_X));
  w.E0 = ar.take<float>(P * c.de);
  w.E1 = ar.take<float>(P * c.de);
  w.y = ar.take<float>((size_t)B * c.dy);
  w.ycat = ar.take<float>((size_t)B * kcat);
  w.yt1 = ar.take<float>((size_t)B * c.dy);
  const int ywide = c.dim_ffy > c.hid_y ? c.dim_ffy : c.hid_y;
  w.yt2 = ar.take<float>((size_t)B * (ywide > c.dy ? ywide : c.dy));
  w.film = ar.take<float>((size_t)2 * B * 4 * c.dx);        // double-buffered: layer l+1's FiLM vectors are produced
                                                            // on the aux stream while layer l's are still read
  w.hidX = ar.take<float>(M * c.hid_X);
  w.hidy = ar.take<float>((size_t)B * c.hid_y);
  w.xo = ar.take<float>(M * (size_t)(c.out_X > 0 ? c.out_X : 1));
  const size_t nch = (size_t)(n + 15) / 16;                 // 16-key chunks per row (edge_unit_layer.hip)
  w.epool = ar.take<float>((size_t)2 * B * 4 * c.de);       // Etoy stats of E(l), E(l+1)
  w.attn = ar.take<float>(M * nch * 3 * c.dx);
  w.epart = ar.take<float>(n > POOL_SLICED_MIN_N ? (size_t)B * POOL_SLICED_W * 5 * c.de : 0);   // Etoy parts of large graphs
  w.rows = ar.take<int>(M);
  w.row_count = ar.take<int>(64);
  w.units = ar.take<int>(M * nch);
  w.unit_count = ar.take<int>(64);
  w.goff = ar.take<int>((size_t)2 * B);
  w.chunk_bits = ar.take<unsigned>((size_t)B);
  w.bytes = align_up(ar.off, 256);
  return w;
}

int gemm(const float* A, int lda, const Param* W, const Param* bias, float* C, int ldc, int M, int N, int K, int flags,
         const uint8_t* rowmask, hipStream_t st) {
  if (!W || !bias) return -1;
  GemmArgs g{A, W->p, bias->p, C, rowmask, M, N, K, lda, K, ldc, flags, 0, 0, 0, 0};
  return launch_gemm(g, st);
}

int gemm_raw(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K,
             int flags, const uint8_t* rowmask, hipStream_t st, int lo = 0, int hi = 0, int lo2 = 0, int hi2 = 0) {
  GemmArgs g{A, W, bias, C, rowmask, M, N, K, lda, K, ldc, flags, lo, hi, lo2, hi2};
  return launch_gemm(g, st);
}

const Param* P_(scgg_denoiser* h, const std::string& name, int64_t numel) { return find(h, name, numel); }

// row-list Linear on the valid rows (falls back to the generic GEMM over all rows when the shape is not packed)
int node_lin(scgg_denoiser* h, const PackedLinear& pk, const std::string& name, const float* A, int lda, float* C, int ldc,
             int M, int N, int K, bool relu, bool rowmask, const uint8_t* mask, const int* rows, const int* row_count,
             hipStream_t st) {
  if (pk.W != nullptr) {
    const int NB = (N + 31) / 32;
    const int rbw = NB >= 24 ? 8 : (NB >= 4 ? 2 : 1);
    NodeLinearArgs a{A, pk.W, pk.b, C, rows, row_count, N, K, NB, lda, ldc, rbw, relu ? 1 : 0};
    if (launch_node_linear(a, M, st)) return fail("node_linear: unsupported K=%d", K);
    return 0;
  }
  return gemm(A, lda, P_(h, name + ".weight", (int64_t)N * K), P_(h, name + ".bias", N), C, ldc, M, N, K,
              (relu ? GEMM_RELU : 0) | (rowmask ? GEMM_ROWMASK : 0), mask, st);
}

// sym + mask copy of an unsymmetrised hidden E into a caller buffer (debug tap only)
__global__ void sym_copy_kernel(const float* E, const uint8_t* mask, float* out, int B, int n, int de) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  const long long total = (long long)B * n * n * de;
  if (idx >= total) return;
  const int c = idx % de;
  const long long pr = idx / de;
  const int j = pr % n;
  const long long bi = pr / n;
  const int i = bi % n, b = bi / n;
  const bool live = i != j && mask[b * n + i] && mask[b * n + j];
  out[idx] = live ? 0.5f * (E[idx] + E[(((long long)b * n + j) * n + i) * de + c]) : 0.f;
}

// the second stream and the fork / join events: created on first use, on the device the caller has made current
int ensure_streams(scgg_denoiser* h) {
  if (h->aux != nullptr) return 0;
  int prio_lo = 0, prio_hi = 0;   // the y kernels are short and feed the next edge launch: let them jump the queue
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, prio_hi) != hipSuccess)
    return fail("aux stream creation failed");
  h->sync_ev.resize((size_t)(h->c.n_layers + 1) * 4 + 1);   // [0..3]: join, fork, E0, second stream caught up; 4 per layer; [last]: y0
  for (hipEvent_t& e : h->sync_ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail("event creation failed");
  return 0;
}

int run_forward(scgg_denoiser* h, const float* X, const float* E, const float* y, const uint8_t* mask, int B, int n,
                int n_run_layers, bool tap, float* o1, float* o2, float* o3, void* workspace, size_t ws_bytes,
                hipStream_t st) {
  if (!h || !h->finalized) return fail("denoiser not finalized");
  if (B <= 0 || n <= 0) return fail("bad batch shape B=%d n=%d", B, n);
  const scgg_denoiser_config& c = h->c;
  if (ensure_streams(h)) return -1;
  if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return fail("workspace must be 256-byte aligned");
  Workspace w = carve(c, B, n, static_cast<char*>(workspace));
  if (ws_bytes < w.bytes) return fail("workspace too small: %zu < %zu", ws_bytes, w.bytes);
  const int M = B * n;
  const int dx = c.dx, de = c.de, dy = c.dy;
  const int kcat = dy + 4 * dx + 4 * de;

  // ---- streams.  The forward runs on ONE stream, the caller's.  The engine's second stream (scgg_denoiser_side_stream) only
  //      carries what the CALLER queues there -- the sampling loop's eigen features, which produce parts of X and y -- and is
  //      joined in front of the first kernel that reads X or y, i.e. BEHIND the work lists and the edge input MLP, which
  //      need E and the mask only (SCGG_LATE_JOIN=0 joins at the start).  SCGG_TWO_STREAMS=y is a development switch: the y
  //      round of layer l on the second stream beside that layer's edge kernel (reproducible, slower).
  //      The round-1 schedule that spread the whole node / global path over the second stream is gone: a few per cent of its
  //      forwards computed single wrong node rows (up to 1e-2 in the logits) and the cause was never found.
  static const int stream_mode = [] { const char* e = getenv("SCGG_TWO_STREAMS"); return (e != nullptr && e[0] == 'y') ? 2 : 0; }();
  static const bool late_join = [] { const char* e = getenv("SCGG_LATE_JOIN"); return !(e != nullptr && e[0] == '0'); }();
  if (hipEventRecord(h->sync_ev[0], h->aux) != hipSuccess) return fail("stream join failed");
  if (!late_join && hipStreamWaitEvent(st, h->sync_ev[0], 0) != hipSuccess) return fail("stream join failed");
  enum { EV_FILM = 0, EV_EDGE = 1, EV_NS = 2, EV_X = 3, EV_PER_LAYER = 4 };
  auto ev = [&](int l, int k) { return h->sync_ev[(size_t)(l + 1) * EV_PER_LAYER + k]; };
  hipEvent_t ev_join = h->sync_ev[1];

  // ---- work lists (valid rows, 16-key units), then fork
  if (launch_build_lists(mask, B, n, w.goff, w.rows, w.row_count, w.units, w.unit_count, st, w.chunk_bits))
    return fail("n=%d too large for the unit list (max 512)", n);
  // The residual stream X ping-pongs between two buffers when every layer runs the fused node kernel (w.t1 is only
  // used by the unfused path): node kernel l then never overwrites the X whose pooled statistics the y path is taking.
  bool dbl_X = node_layer_supported(dx, c.dim_ffX) && h->in_X2.W != nullptr && c.in_X <= 64 && c.hid_X % 4 == 0;
  for (int l = 0; l < n_run_layers; ++l) {
    const LayerPack& L = h->layers[l];
    dbl_X = dbl_X && L.x_out.W && L.linX1.W && L.linX2.W && L.qkv.W;
  }
  float* Xcur = w.Xh;
  float* Xalt = w.t1;
  // valid rows only: the masked rows of the residual stream are zeroed once per forward, nothing writes them afterwards
  if (h->in_X2.W != nullptr && c.in_X <= 64 && c.hid_X % 4 == 0) {
    launch_zero_masked_rows(w.Xh, mask, M, dx, st);
    if (dbl_X) launch_zero_masked_rows(w.t1, mask, M, dx, st);
  }
  // the fused y path (y input MLP, FiLM vectors of layer 0 and the y rounds in y_layer_kernel)
  const bool y_on_main = y_layer_supported(dy, c.dim_ffy, dy + 4 * dx + 4 * de) && n_run_layers > 0;
  // One stream: the y round takes the pooled statistics of X(l) and E(l) itself (pool_stats.h) -- the separate statistics
  // kernels exist to run BESIDE other work, and here they would only add two launches per layer in front of it.
  // SCGG_Y_POOL=0 keeps them (development A/B switch).
  const bool y_beside = stream_mode == 2 && y_layer_supported(dy, c.dim_ffy, kcat);
  auto rec = [&](hipEvent_t e, hipStream_t s2) { return hipEventRecord(e, s2) != hipSuccess; };
  auto hold = [&](hipStream_t s2, hipEvent_t e) { return hipStreamWaitEvent(s2, e, 0) != hipSuccess; };
  static const bool y_pool_on = [] { const char* e = getenv("SCGG_Y_POOL"); return !(e != nullptr && e[0] == '0'); }();
  const bool y_pools = y_pool_on && y_layer_supported(dy, c.dim_ffy, kcat) && dx % 4 == 0 && dx <= 512 &&
                       256 % (dx / 4) == 0 && de % 4 == 0 && de <= 256 && 256 % (de / 4) == 0 && n <= 512;
  bool pooled_e0 = false;
  bool head_fused = false;   // the last node kernel also ran mlp_out_X.0
  {
    EdgeMlpUnitArgs a{E, nullptr, mask, h->in_E_img, w.E0, w.units, w.unit_count, B, n, c.in_E, c.out_E};
    if (launch_edge_mlp_in(a, c.in_E, c.hid_E, de, st)) return fail("edge_mlp_in: unsupported dims");
    // Etoy of E(0) for the fused y path when the y round does not pool for itself (SCGG_Y_POOL=0)
    pooled_e0 = !y_pools && y_layer_supported(dy, c.dim_ffy, kcat) && n_run_layers > 0 && h->layers[0].predicts_y;
    if (pooled_e0) {
      EdgePoolArgs ep{w.E0, mask, w.epool, B, n, de, 4 * de, 0};
      if (launch_edge_pool_stats(ep, st)) return fail("edge_pool_stats: unsupported de=%d or n=%d", de, n);
    }
  }
  // the output tensors are zero outside the valid rows / live pairs (utils.py:154) and only their valid part is written
  // at the end: zero-filled now, while the caller's stream waits for the prologue of the second stream anyway
  bool out_zeroed = false;
  if (!tap && y_on_main) {
    if (c.out_X > 0 && o1 && hipMemsetAsync(o1, 0, (size_t)M * c.out_X * sizeof(float), st) != hipSuccess) return fail("memset failed");
    if (hipMemsetAsync(o2, 0, (size_t)M * n * c.out_E * sizeof(float), st) != hipSuccess) return fail("memset failed");
    out_zeroed = true;
  }
  if (late_join && hipStreamWaitEvent(st, h->sync_ev[0], 0) != hipSuccess) return fail("stream join failed");
  // ---- input MLPs (transformer_model.py:358-366): E on the main stream, X and y beside it on the aux stream
  {
    const Param* w0 = P_(h, "mlp_in_X.0.weight", (int64_t)c.hid_X * c.in_X); const Param* b0 = P_(h, "mlp_in_X.0.bias", c.hid_X);
    const Param* w3 = P_(h, "mlp_in_X.3.weight", (int64_t)dx * c.hid_X); const Param* b3 = P_(h, "mlp_in_X.3.bias", dx);
    if (!w0 || !b0 || !w3 || !b3) return -1;
    if (h->in_X2.W != nullptr && c.in_X <= 64 && c.hid_X % 4 == 0) {
      NarrowLinearArgs na{X, w0->p, b0->p, w.hidX, w.rows, w.row_count, c.hid_X, c.in_X, c.in_X, c.hid_X, 1};
      if (launch_narrow_linear(na, M, st)) return fail("narrow_linear rejected");
      NodeLinearArgs la{w.hidX, h->in_X2.W, h->in_X2.b, w.Xh, w.rows, w.row_count, dx, c.hid_X, (dx + 31) / 32, c.hid_X, dx,
                        (dx + 31) / 32 >= 4 ? 2 : 1, 1};
      if (launch_node_linear_x3(la, h->in_X2.W3, M, st) != 0 && launch_node_linear(la, M, st))
        return fail("node_linear(mlp_in_X.3) rejected");
    } else {
      if (gemm(X, c.in_X, w0, b0, w.hidX, c.hid_X, M, c.hid_X, c.in_X, GEMM_RELU, nullptr, st)) return -1;
      if (gemm(w.hidX, c.hid_X, w3, b3, w.Xh, dx, M, dx, c.hid_X, GEMM_RELU | GEMM_ROWMASK, mask, st)) return -1;
    }
  }
  {
    hipStream_t ys = st;
    if (gemm(y, c.in_y, P_(h, "mlp_in_y.0.weight", (int64_t)c.hid_y * c.in_y), P_(h, "mlp_in_y.0.bias", c.hid_y), w.hidy,
             c.hid_y, B, c.hid_y, c.in_y, GEMM_RELU, nullptr, ys)) return -1;
    if (gemm(w.hidy, c.hid_y, P_(h, "mlp_in_y.3.weight", (int64_t)dy * c.hid_y), P_(h, "mlp_in_y.3.bias", dy), w.y, dy, B,
             dy, c.hid_y, GEMM_RELU, nullptr, ys)) return -1;
  }
  float* Ecur = w.E0;
  float* Enext = w.E1;
  const bool prof_now = !tap && h->prof_every > 0 && (h->prof_forwards++ % h->prof_every) == 0;

  // ---- layers (transformer_model.py:76-112, :165-248)
  for (int l = 0; l < n_run_layers; ++l) {
    LayerPack& L = h->layers[l];
    const std::string p = "tf_layers." + std::to_string(l) + ".";
    const std::string sa = p + "self_attn.";
    float* film = w.film + (size_t)(l & 1) * B * 4 * dx;
    // ---- q|k|v masked: layer 0 on the aux stream (beside the edge input MLP), later layers come out of the previous
    //      layer's fused node kernel; on the unfused path every layer computes them on the main stream
    const bool fused_node = node_layer_supported(dx, c.dim_ffX) && L.x_out.W && L.linX1.W && L.linX2.W && L.qkv.W;
    {
      hipStream_t qs = st;
      if (fused_node && l > 0) {
      } else if (L.qkv.W != nullptr) {
        NodeLinearArgs qa{Xcur, L.qkv.W, L.qkv.b, w.qkv, w.rows, w.row_count, 3 * dx, dx, (3 * dx + 31) / 32, dx, 3 * dx,
                          dx >= 256 ? 8 : (3 * dx + 31) / 32, 0};
        if (launch_node_linear_x3(qa, L.qkv.W3, M, qs) != 0 && launch_node_linear(qa, M, qs))
          return fail("node_linear(qkv) rejected");
      } else if (gemm_raw(Xcur, dx, L.Wqkv, L.bqkv, w.qkv, 3 * dx, M, 3 * dx, dx, GEMM_ROWMASK, mask, qs)) return -1;
    }
    // ---- aux: y-FiLM vectors (with the +1 folded in), Xtoy pooled statistics of the layer INPUT (:239-243)
    const bool fused_y = y_layer_supported(dy, c.dim_ffy, kcat);
    if (!fused_y || l == 0) {   // (fused: layer l's vectors were produced by layer l-1's y kernel)
      if (fused_y) {
        YLayerArgs ya{};
        ya.y = w.y; ya.Wt_film = L.Wt_film; ya.b_film = L.bfilm; ya.film = film;
        ya.B = B; ya.dy = dy; ya.ffy = c.dim_ffy; ya.kcat = kcat; ya.dx = dx; ya.update = 0; ya.pair = 1;
        // y on the caller's stream: what that stream needs from the second one before edge kernel 0 is X(0) and q|k|v(0);
        // the event goes directly behind those kernels, then the second stream waits for y(0)
        if (launch_y_layer(ya, st)) return fail("y_layer launch failed");
      } else if (gemm_raw(w.y, dy, L.Wfilm, L.bfilm, film, 4 * dx, B, 4 * dx, dy, GEMM_ADD_ONE_COLS, nullptr, st, dx,
                          2 * dx, 3 * dx, 4 * dx)) return -1;
    }
    if (L.predicts_y && !y_pools) {
      if (!fused_y) launch_copy_cols(w.y, dy, w.ycat, kcat, B, dy, st);
      NodeStatsArgs ns{Xcur, mask, w.ycat, B, n, dx, kcat, dy};
      if (launch_node_stats(ns, st)) return fail("node_stats: unsupported dx=%d or n=%d", dx, n);
    }
    if (y_beside && rec(ev(l, EV_NS), st)) return fail("event record failed");      // X(l), E(l), y are final here
    // Etoy of the layer INPUT E(l) (:239-243).  Fused y path: E(0) here, E(l+1) at the end of round l (beside the node
    // kernel); unfused: every layer here, straight into the ycat row.
    if (L.predicts_y && !y_pools && (l == 0 || !fused_y) && !(l == 0 && pooled_e0)) {
      EdgePoolArgs ep{Ecur, mask, fused_y ? w.epool : w.ycat, B, n, de, fused_y ? 4 * de : kcat, fused_y ? 0 : dy + 4 * dx};
      if (launch_edge_pool_stats(ep, st)) return fail("edge_pool_stats: unsupported de=%d or n=%d", de, n);
    }

    // ---- main: fused edge kernel.  Its FiLM vectors: layer 0 (and the unfused y path) waits here; on the fused y path
    //      the wait for layer l + 1 sits in front of node kernel l (the y round of layer l runs beside edge kernel l),
    //      so node kernel l and edge kernel l + 1 go back to back (a cross-stream wait costs ~10 us of idle queue)
    {
      EdgeLayerArgs a;
      a.E_in = Ecur; a.E_out = Enext; a.mask = mask;
      a.Q = w.qkv; a.K = w.qkv + dx; a.V = w.qkv + 2 * dx; a.ldqkv = 3 * dx;
      a.film = film;
      a.phase_img = L.phase_img;
      a.b_out = L.b_out;
      a.attn_part = w.attn; a.units = w.units; a.unit_count = w.unit_count;
      a.ln1_g = L.ln1_g; a.ln1_b = L.ln1_b; a.ln2_g = L.ln2_g; a.ln2_b = L.ln2_b;
      a.inv_sqrt_df = 1.0f / sqrtf((float)(dx / c.n_head));
      a.B = B; a.n = n;
      const bool timed = prof_now && h->prof_used + 2 <= h->prof_ev.size();
      if (timed) (void)hipEventRecord(h->prof_ev[h->prof_used], st);
      if (launch_edge_layer(a, dx, de, c.dim_ffE, st)) return fail("edge_layer: unsupported dims dx=%d de=%d ffE=%d (or launch setup failed)", dx, de, c.dim_ffE);
      if (timed) { (void)hipEventRecord(h->prof_ev[h->prof_used + 1], st); h->prof_used += 2; }
      if (!fused_node) {
        AttnFinishArgs af{w.attn, mask, film, w.newX, w.rows, w.row_count, n, dx};
        launch_attn_finish(af, M, st);
      }
    }
    // ---- aux: y update (transformer_model.py:239-246, :101-106) and the next layer's FiLM vectors
    const bool next_layer = l + 1 < n_run_layers;
    float* film_next = w.film + (size_t)((l + 1) & 1) * B * 4 * dx;
    if (fused_y) {
      if (L.predicts_y || next_layer) {
        YLayerArgs ya{};
        ya.y = w.y; ya.ycat = w.ycat; ya.etoy = w.epool + (size_t)(l & 1) * B * 4 * de;
        ya.B = B; ya.dy = dy; ya.ffy = c.dim_ffy; ya.kcat = kcat; ya.dx = dx;
        ya.update = L.predicts_y ? 1 : 0;
        ya.pair = 1;
        if (y_pools && L.predicts_y) {
          ya.pool_X = Xcur; ya.pool_E = Ecur; ya.mask = mask; ya.n = n; ya.de = de;
          if (n > POOL_SLICED_MIN_N) {   // few large graphs: the pooled E statistics by W workgroups per graph, the y round reads them
            EdgePoolArgs ep{Ecur, mask, w.epool + (size_t)(l & 1) * B * 4 * de, B, n, de, 4 * de, 0};
            if (launch_edge_pool_stats_sliced(ep, w.epart, POOL_SLICED_W, st)) return fail("edge_pool_stats: unsupported de=%d or n=%d", de, n);
            ya.pool_E = nullptr;
          }
        }
        // The y round needs nothing from edge kernel l and is not held back for it: one y workgroup (50 VGPRs, 36 KB of
        // LDS at ConStruct dims) fits on a CU beside one edge workgroup (2 x 200 VGPRs per SIMD lane, 122 KB).
        if (L.predicts_y) {
          const Param* o0b = P_(h, sa + "y_out.0.bias", dy); const Param* o2b = P_(h, sa + "y_out.2.bias", dy);
          const Param* g1 = P_(h, p + "norm_y1.weight", dy); const Param* b1 = P_(h, p + "norm_y1.bias", dy);
          const Param* g2 = P_(h, p + "norm_y2.weight", dy); const Param* b2 = P_(h, p + "norm_y2.bias", dy);
          const Param* l1b = P_(h, p + "lin_y1.bias", c.dim_ffy); const Param* l2b = P_(h, p + "lin_y2.bias", dy);
          if (!o0b || !o2b || !g1 || !b1 || !g2 || !b2 || !l1b || !l2b) return -1;
          ya.Wt_cat = L.Wt_cat; ya.b_cat = L.bycat;
          ya.Wt_o0 = L.Wt_o0; ya.b_o0 = o0b->p; ya.Wt_o2 = L.Wt_o2; ya.b_o2 = o2b->p;
          ya.ln1_g = g1->p; ya.ln1_b = b1->p; ya.ln2_g = g2->p; ya.ln2_b = b2->p;
          ya.Wt_l1 = L.Wt_l1; ya.b_l1 = l1b->p; ya.Wt_l2 = L.Wt_l2; ya.b_l2 = l2b->p;
        }
        if (next_layer) { ya.Wt_film = h->layers[l + 1].Wt_film; ya.b_film = h->layers[l + 1].bfilm; ya.film = film_next; }
        if (y_beside && hold(h->aux, ev(l, EV_NS))) return fail("stream wait failed");
        if (launch_y_layer(ya, y_beside ? h->aux : st)) return fail("y_layer launch failed");
        if (y_beside && next_layer && rec(ev(l + 1, EV_FILM), h->aux)) return fail("event record failed");
      }
      if (next_layer && h->layers[l + 1].predicts_y && !y_pools) {   // Etoy of E(l+1), the tensor edge kernel l has just written
        EdgePoolArgs ep{Enext, mask, w.epool + (size_t)((l + 1) & 1) * B * 4 * de, B, n, de, 4 * de, 0};
        if (launch_edge_pool_stats(ep, st)) return fail("edge_pool_stats: unsupported de=%d or n=%d", de, n);
      }
    } else if (L.predicts_y) {
      if (gemm_raw(w.ycat, kcat, L.Wycat, L.bycat, w.yt1, dy, B, dy, kcat, 0, nullptr, st)) return -1;
      if (gemm(w.yt1, dy, P_(h, sa + "y_out.0.weight", (int64_t)dy * dy), P_(h, sa + "y_out.0.bias", dy), w.yt2, dy, B, dy,
               dy, GEMM_RELU, nullptr, st)) return -1;
      if (gemm(w.yt2, dy, P_(h, sa + "y_out.2.weight", (int64_t)dy * dy), P_(h, sa + "y_out.2.bias", dy), w.yt1, dy, B, dy,
               dy, 0, nullptr, st)) return -1;
      const Param* g1 = P_(h, p + "norm_y1.weight", dy); const Param* b1 = P_(h, p + "norm_y1.bias", dy);
      const Param* g2 = P_(h, p + "norm_y2.weight", dy); const Param* b2 = P_(h, p + "norm_y2.bias", dy);
      if (!g1 || !b1 || !g2 || !b2) return -1;
      AddLnArgs ln1{w.y, w.yt1, g1->p, b1->p, nullptr, w.y, B, dy, nullptr, nullptr};
      launch_add_layernorm(ln1, st);
      if (gemm(w.y, dy, P_(h, p + "lin_y1.weight", (int64_t)c.dim_ffy * dy), P_(h, p + "lin_y1.bias", c.dim_ffy), w.yt2,
               c.dim_ffy, B, c.dim_ffy, dy, GEMM_RELU, nullptr, st)) return -1;
      if (gemm(w.yt2, c.dim_ffy, P_(h, p + "lin_y2.weight", (int64_t)dy * c.dim_ffy), P_(h, p + "lin_y2.bias", dy), w.yt1,
               dy, B, dy, c.dim_ffy, 0, nullptr, st)) return -1;
      AddLnArgs ln2{w.y, w.yt1, g2->p, b2->p, nullptr, w.y, B, dy, nullptr, nullptr};
      launch_add_layernorm(ln2, st);
    }
    // ---- main: X path (transformer_model.py:234-235, :88-92)
    {
      const Param* g1 = P_(h, p + "normX1.weight", dx); const Param* b1 = P_(h, p + "normX1.bias", dx);
      const Param* g2 = P_(h, p + "normX2.weight", dx); const Param* b2 = P_(h, p + "normX2.bias", dx);
      if (!g1 || !b1 || !g2 || !b2) return -1;
      if (y_beside && l + 1 < n_run_layers && hold(st, ev(l + 1, EV_FILM))) return fail("stream wait failed");
      // in place: X is overwritten from here on, its pooled statistics must have been taken; double-buffered: nothing
      // to wait for (the buffer written now was last read two layers ago)
      if (fused_node) {
        const bool next = l + 1 < n_run_layers;
        NodeLayerArgs na{w.attn, mask, film, Xcur, w.rows, w.row_count, L.x_out.W, L.x_out.b, g1->p, b1->p,
                         L.linX1.W, L.linX1.b, L.linX2.W, L.linX2.b, g2->p, b2->p,
                         next ? h->layers[l + 1].qkv.W : nullptr, next ? h->layers[l + 1].qkv.b : nullptr, w.qkv, n,
                         L.x_out.W3, L.linX1.W3, L.linX2.W3, next ? h->layers[l + 1].qkv.W3 : nullptr,
                         dbl_X ? Xalt : Xcur, 0, 0, 0};
        na.chunk_bits = w.chunk_bits;
        // after the last layer the trailing Linear is mlp_out_X.0 (+ ReLU) into the hidden buffer of the X output head
        head_fused = false;
        if (!next && !tap && l + 1 == c.n_layers && c.out_X > 0 && c.out_X <= 16 && h->out_X1.W3 != nullptr &&
            c.hid_X % 32 == 0 && node_layer_x3_enabled()) {
          na.W_qkv = h->out_X1.W; na.b_qkv = h->out_X1.b; na.W3_qkv = h->out_X1.W3; na.qkv = w.hidX;
          na.tail_pairs = c.hid_X / 32; na.tail_ld = c.hid_X; na.tail_relu = 1;
          head_fused = true;
        }
        if (launch_node_layer(na, dx, c.dim_ffX, M, st)) return fail("node_layer launch failed");
        if (dbl_X) { float* tx = Xcur; Xcur = Xalt; Xalt = tx; }
      } else {
        if (node_lin(h, L.x_out, sa + "x_out", w.newX, dx, w.t1, dx, M, dx, dx, false, true, mask, w.rows, w.row_count, st))
          return -1;
        AddLnArgs ln1{Xcur, w.t1, g1->p, b1->p, nullptr, Xcur, M, dx, w.rows, w.row_count};
        if (launch_add_layernorm(ln1, st)) return fail("layernorm width %d unsupported", dx);
        if (node_lin(h, L.linX1, p + "linX1", Xcur, dx, w.t2, c.dim_ffX, M, c.dim_ffX, dx, true, false, mask, w.rows,
                     w.row_count, st)) return -1;
        if (node_lin(h, L.linX2, p + "linX2", w.t2, c.dim_ffX, w.t1, dx, M, dx, c.dim_ffX, false, false, mask, w.rows,
                     w.row_count, st)) return -1;
        AddLnArgs ln2{Xcur, w.t1, g2->p, b2->p, mask, Xcur, M, dx, w.rows, w.row_count};
        launch_add_layernorm(ln2, st);
      }
    }
    float* t = Ecur; Ecur = Enext; Enext = t;
    if (check_launch("layer")) return -1;
  }
  if (y_beside && (rec(ev_join, h->aux) || hold(st, ev_join))) return fail("stream join failed");   // the last y round
  // the aux stream needs the final X (the main stream's last node kernel) for the X output head

  if (tap) {
    if (o1) launch_copy_cols(Xcur, dx, o1, dx, M, dx, st);
    if (o2) {
      const long long total = (long long)M * n * de;
      hipLaunchKernelGGL(sym_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Ecur, mask, o2, B, n, de);
    }
    if (o3) launch_copy_cols(w.y, dy, o3, dy, B, dy, st);
    return check_launch("tap");
  }

  // ---- output MLPs (transformer_model.py:371-389): E on the main stream, X and y beside it on the aux stream
  if (c.out_X > 0) {
    // valid rows only when the shapes allow: x3 row-list Linear + a one-wave-per-row head with the residual fused
    // (masked rows of the output are zero: utils.py:154); else the dense GEMMs
    const Param* w3 = P_(h, "mlp_out_X.3.weight", (int64_t)c.out_X * c.hid_X); const Param* b3 = P_(h, "mlp_out_X.3.bias", c.out_X);
    if (!w3 || !b3) return -1;
    NodeLinearArgs la{Xcur, h->out_X1.W, h->out_X1.b, w.hidX, w.rows, w.row_count, c.hid_X, dx, (c.hid_X + 31) / 32, dx, c.hid_X,
                      1, 1};
    bool rows_only = head_fused || (h->out_X1.W3 != nullptr && c.out_X <= 16 && c.hid_X % 4 == 0 &&
                                    launch_node_linear_x3(la, h->out_X1.W3, M, st) == 0);
    if (rows_only) {
      if (!out_zeroed && hipMemsetAsync(o1, 0, (size_t)M * c.out_X * sizeof(float), st) != hipSuccess) return fail("memset failed");
      if (launch_row_head(w.hidX, c.hid_X, w3->p, b3->p, X, c.in_X, w.rows, w.row_count, o1, M, c.out_X, c.hid_X, st))
        return fail("row_head rejected");
    } else {
      if (gemm(Xcur, dx, P_(h, "mlp_out_X.0.weight", (int64_t)c.hid_X * dx), P_(h, "mlp_out_X.0.bias", c.hid_X), w.hidX,
               c.hid_X, M, c.hid_X, dx, GEMM_RELU, nullptr, st)) return -1;
      if (gemm(w.hidX, c.hid_X, w3, b3, w.xo, c.out_X, M, c.out_X, c.hid_X, 0, nullptr, st)) return -1;
      launch_x_out_residual(w.xo, X, c.in_X, mask, o1, M, c.out_X, st);
    }
  }
  if (c.out_y > 0) {
    if (!o3) return fail("out_y requested by the config but out_y pointer is NULL");
    if (gemm(w.y, dy, P_(h, "mlp_out_y.0.weight", (int64_t)c.hid_y * dy), P_(h, "mlp_out_y.0.bias", c.hid_y), w.hidy,
             c.hid_y, B, c.hid_y, dy, GEMM_RELU, nullptr, st)) return -1;
    if (gemm(w.hidy, c.hid_y, P_(h, "mlp_out_y.3.weight", (int64_t)c.out_y * c.hid_y), P_(h, "mlp_out_y.3.bias", c.out_y),
             w.yt1, c.out_y, B, c.out_y, c.hid_y, 0, nullptr, st)) return -1;
    launch_x_out_residual(w.yt1, y, c.in_y, nullptr, o3, B, c.out_y, st);
  }
  {
    // logits are zero outside the live pairs (utils.py:154); the unit kernel writes the chunks that hold valid keys
    if (!out_zeroed && hipMemsetAsync(o2, 0, (size_t)M * n * c.out_E * sizeof(float), st) != hipSuccess) return fail("memset failed");
    EdgeMlpUnitArgs a{Ecur, E, mask, h->out_E_img, o2, w.units, w.unit_count, B, n, c.in_E, c.out_E};
    if (launch_edge_mlp_out(a, de, c.hid_E, c.out_E, st)) return fail("edge_mlp_out: unsupported dims");
  }
  // join: everything the aux stream did is ordered before what follows on the caller's stream
  return check_launch("forward");
}

}  // namespace

// ====================================================================================================== C ABI
extern "C" {

int scgg_version(void) { return 1; }
const char* scgg_last_error(void) { return last_error(); }

scgg_denoiser* scgg_denoiser_create(const scgg_denoiser_config* cfg) {
  if (!cfg) { fail("null config"); return nullptr; }
  const scgg_denoiser_config& c = *cfg;
  if (c.n_layers < 1 || c.dx % 32 || c.de % 32 || c.dim_ffE % 32 || c.hid_E % 32 || c.n_head < 1 || c.dx % c.n_head ||
      c.out_E < 1 || c.out_E > 32 || c.in_E < c.out_E || c.in_E > 64 || c.in_X < c.out_X || c.in_y < c.out_y) {
    fail("unsupported denoiser dims (need dx,de,dim_ffE,hid_E multiples of 32, out_E<=32, in_E<=64, in>=out)");
    return nullptr;
  }
  scgg_denoiser* h = new scgg_denoiser();
  h->c = c;
  return h;
}

void scgg_denoiser_destroy(scgg_denoiser* h) {
  if (!h) return;
  for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->sync_ev) (void)hipEventDestroy(e);
  if (h->aux) (void)hipStreamDestroy(h->aux);
  if (h->arena) (void)hipFree(h->arena);
  delete h;
}

int scgg_denoiser_set_param(scgg_denoiser* h, const char* name, const float* data, int64_t numel) {
  if (!h || !name || !data) return fail("set_param: null argument");
  h->params[name] = Param{data, numel};
  return 0;
}

int scgg_denoiser_finalize(scgg_denoiser* h, void* stream) {
  if (!h) return fail("null handle");
  if (build(h, false, nullptr)) return -1;   // measure
  if (h->arena) (void)hipFree(h->arena);
  h->arena = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&h->arena), h->arena_bytes) != hipSuccess)
    return fail("hipMalloc of %zu bytes for the packed weights failed", h->arena_bytes);
  if (build(h, true, static_cast<hipStream_t>(stream))) return -1;
  // set-up work the compute calls must not do: the second stream and its events, and this device's scratch address
  if (ensure_streams(h)) return -1;
  if (static_scratch() == nullptr) return -1;
  h->finalized = true;
  return 0;
}

int scgg_denoiser_refresh(scgg_denoiser* h, void* stream) {
  if (!h || !h->finalized) return fail("denoiser not finalized");
  return build(h, true, static_cast<hipStream_t>(stream));
}

size_t scgg_denoiser_workspace_bytes(const scgg_denoiser* h, int32_t B, int32_t n) {
  if (!h || B <= 0 || n <= 0) return 0;
  return carve(h->c, B, n, nullptr).bytes;
}

int scgg_denoiser_forward(scgg_denoiser* h, const float* X, const float* E, const float* y, const uint8_t* node_mask,
                          int32_t B, int32_t n, float* out_X, float* out_E, float* out_y, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (!X || !E || !y || !node_mask || !out_E) return fail("forward: null tensor");
  if (h && h->c.out_X > 0 && !out_X) return fail("forward: out_X is NULL");
  return run_forward(h, X, E, y, node_mask, B, n, h ? h->c.n_layers : 0, false, out_X, out_E, out_y, workspace,
                     workspace_bytes, static_cast<hipStream_t>(stream));
}

int scgg_denoiser_forward_static(scgg_denoiser* h, const float* X, const float* E, const float* y,
                                 const uint8_t* node_mask, int32_t B, int32_t n, float* out_X, float* out_E,
                                 float* out_y, void* stream) {
  if (!X || !E || !y || !node_mask || !out_E) return fail("forward_static: null tensor");
  if (!h || !h->finalized) return fail("forward_static: denoiser not finalized");
  if (h->c.out_X > 0 && !out_X) return fail("forward_static: out_X is NULL");
  if (B <= 0 || n <= 0) return fail("forward_static: bad batch shape B=%d n=%d", B, n);
  const size_t need = carve(h->c, B, n, nullptr).bytes;
  if (need > (size_t)SCGG_STATIC_SCRATCH_BYTES)
    return fail("forward_static: B=%d n=%d needs %zu bytes of scratch, the static buffer holds %zu (use scgg_denoiser_forward "
                "with a caller workspace)", B, n, need, (size_t)SCGG_STATIC_SCRATCH_BYTES);
  void* ws = static_scratch();          // per-device pointer from hipGetSymbolAddress, never the array's name
  if (ws == nullptr) return -1;
  return run_forward(h, X, E, y, node_mask, B, n, h->c.n_layers, false, out_X, out_E, out_y, ws,
                     (size_t)SCGG_STATIC_SCRATCH_BYTES, static_cast<hipStream_t>(stream));
}

size_t scgg_denoiser_static_scratch_bytes(void) { return (size_t)SCGG_STATIC_SCRATCH_BYTES; }

int scgg_denoiser_forward_hidden(scgg_denoiser* h, const float* X, const float* E, const float* y,
                                 const uint8_t* node_mask, int32_t B, int32_t n, int32_t n_run_layers, float* hid_X,
                                 float* hid_E, float* hid_y, void* workspace, size_t workspace_bytes, void* stream) {
  if (!X || !E || !y || !node_mask) return fail("forward_hidden: null tensor");
  if (!h || n_run_layers < 0 || n_run_layers > h->c.n_layers) return fail("forward_hidden: bad layer count");
  return run_forward(h, X, E, y, node_mask, B, n, n_run_layers, true, hid_X, hid_E, hid_y, workspace, workspace_bytes,
                     static_cast<hipStream_t>(stream));
}

void* scgg_denoiser_side_stream(scgg_denoiser* h) {
  if (!h || !h->finalized) { fail("side_stream: denoiser not finalized"); return nullptr; }
  if (ensure_streams(h)) return nullptr;
  return h->aux;
}

int scgg_denoiser_profile(scgg_denoiser* h, int32_t every_n_forwards, int32_t max_launches) {
  if (!h || every_n_forwards < 0 || max_launches < 0) return fail("profile: bad argument");
  for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
  h->prof_ev.clear();
  h->prof_used = 0;
  h->prof_forwards = 0;
  h->prof_every = every_n_forwards;
  if (every_n_forwards == 0) return 0;
  h->prof_ev.resize((size_t)2 * max_launches);
  for (hipEvent_t& e : h->prof_ev)
    if (hipEventCreate(&e) != hipSuccess) return fail("profile: hipEventCreate failed");
  return 0;
}

int scgg_denoiser_profile_read(scgg_denoiser* h, int32_t* n_launches, double* total_ms) {
  if (!h || !n_launches || !total_ms) return fail("profile_read: null argument");
  double tot = 0.0;
  for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]) != hipSuccess)
      return fail("profile_read: events not complete (synchronise the stream first)");
    tot += ms;
  }
  *n_launches = (int32_t)(h->prof_used / 2);
  *total_ms = tot;
  h->prof_used = 0;
  return 0;
}

// ---------------------------------------------------------------------------------------------- reverse step
static SampleRng to_rng(const scgg_rng* r) {
  SampleRng g{nullptr, nullptr, 0, 0, 0};
  if (r) { g.u0 = r->u0; g.uk = r->uk; g.seed = r->seed; g.step = r->step; g.graph_id_base = r->graph_id_base; }
  return g;
}

int scgg_reverse_step_edges(const float* logits_E, const int8_t* labels_t, const uint8_t* node_mask,
                            const int32_t* t_int, const int32_t* s_int, const float* beta_tab, const float* abar_tab,
                            const float* marginals, int32_t B, int32_t n, int32_t C, int32_t top_k,
                            const scgg_rng* rng, int8_t* labels_s, int8_t* cands_s, float* prob_out, void* stream) {
  if (!logits_E || !labels_t || !node_mask || !t_int || !s_int || !beta_tab || !abar_tab || !marginals || !labels_s)
    return fail("reverse_step: null argument");
  if (B <= 0 || n <= 0 || n > 65535 || C < 1 || C > 32 || top_k < 0 || top_k > 8 || top_k > C)
    return fail("reverse_step: unsupported shape B=%d n=%d C=%d top_k=%d", B, n, C, top_k);
  if (top_k > 0 && !cands_s) return fail("reverse_step: top_k > 0 needs a candidate buffer");
  if (rng && rng->u0 && top_k > 0 && !rng->uk) return fail("reverse_step: u0 given without uk");
  ReverseStepArgs a{logits_E, labels_t, node_mask, t_int, s_int, beta_tab, abar_tab, marginals, prob_out, to_rng(rng),
                    SampleOut{labels_s, cands_s}, B, n, C, top_k};
  if (launch_reverse_step(a, static_cast<hipStream_t>(stream))) return fail("reverse_step: launch rejected");
  return check_launch("reverse_step");
}

int scgg_sample_edges_from_probs(const float* prob_E, const uint8_t* node_mask, int32_t B, int32_t n, int32_t C,
                                 int32_t top_k, const scgg_rng* rng, int8_t* labels, int8_t* cands, void* stream) {
  if (!prob_E || !node_mask || !labels) return fail("sample_edges: null argument");
  if (B <= 0 || n <= 0 || n > 65535 || C < 1 || C > 32 || top_k < 0 || top_k > 8 || top_k > C)
    return fail("sample_edges: unsupported shape B=%d n=%d C=%d top_k=%d", B, n, C, top_k);
  if (top_k > 0 && !cands) return fail("sample_edges: top_k > 0 needs a candidate buffer");
  if (rng && rng->u0 && top_k > 0 && !rng->uk) return fail("sample_edges: u0 given without uk");
  SampleProbsArgs a{prob_E, node_mask, to_rng(rng), SampleOut{labels, cands}, B, n, C, top_k};
  if (launch_sample_from_probs(a, static_cast<hipStream_t>(stream))) return fail("sample_edges: launch rejected");
  return check_launch("sample_edges");
}

int scgg_labels_to_onehot(const int8_t* labels, const int8_t* cands, int64_t n_pairs, int32_t C, int32_t top_k,
                          float* E_onehot, int64_t* extra_info, void* stream) {
  if (!labels || !E_onehot || n_pairs <= 0) return fail("labels_to_onehot: bad argument");
  if (extra_info && !cands) return fail("labels_to_onehot: extra_info needs cands");
  launch_labels_to_onehot(labels, cands, n_pairs, C, top_k, E_onehot, reinterpret_cast<long long*>(extra_info),
                          static_cast<hipStream_t>(stream));
  return check_launch("labels_to_onehot");
}

int scgg_onehot_to_labels(const float* E_onehot, int64_t n_pairs, int32_t C, int8_t* labels, void* stream) {
  if (!labels || !E_onehot || n_pairs <= 0) return fail("onehot_to_labels: bad argument");
  launch_onehot_to_labels(E_onehot, n_pairs, C, labels, static_cast<hipStream_t>(stream));
  return check_launch("onehot_to_labels");
}

}  // extern "C"
