// Host-side engine behind scgg_denoiser3d_* (include/scgg.h): Graph3DTransformer / EquivariantGraph3DTransformer of the
// reference (src/ConStruct/models/pos_transformer_model.py:338-493, equivariant_pos_transformer_model.py:334-486), eval
// mode, on ONE stream.  The edge path of every layer is one fused kernel (block3d.hip); the node and global paths, which
// are O(B n d^2) and small next to it, run as fp32-MFMA GEMM launches (node_kernels.hip) between small merge kernels.
// No torch types, no synchronisation, no allocation after finalize.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/scgg.h"
#include "kernels.h"

namespace scgg {
namespace {

struct Param3 { const float* p; int64_t numel; };
static inline size_t align_up3(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Bump {   // two-pass bump allocator (base == nullptr: measuring pass)
  char* base;
  size_t off = 0;
  explicit Bump(char* b) : base(b) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up3(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

struct RowLin {                  // row-list Linear on the bf16x3 scheme (node_layer.hip), null W3: generic GEMM instead
  f32x4_t* W3 = nullptr;
  float* b = nullptr;            // padded to a multiple of 32
  int N = 0, K = 0;
};

struct Layer3 {
  bool predicts_y = true;
  f32x4_t* img = nullptr;        // block3d_edge_kernel's LDS image
  float *Wnl, *bnl;              // [3dx+2de, dx]: q | k | v | x_e_mul1 | x_e_mul2
  float *Wfilm, *bfilm;          // [2de+2dx, dy]: y_e_add | y_e_mul | y_x_add | y_x_mul
  float *Wycat, *bycat;          // [dy, dy+4dx+8de]: y_y | x_y.lin | e_y.lin | dist_y.lin, summed biases
  float *Wt_cat, *Wt_o0, *Wt_o2, *Wt_l1, *Wt_l2, *Wt_film;   // transposed copies for y_layer_kernel ([K][N])
  RowLin nl, out, ex, px, x_out, linX1, linX2;
};

// [w_dist | w_cos | b_dist + b_n1 + b_n2 | w_n1 | w_n2] of one layer (the same folding as the edge kernel's LDS image)
__global__ void dvec_kernel(const float* W_dist, const float* b_dist, const float* W_n1, const float* b_n1, const float* W_n2,
                            const float* b_n2, float* out, int de) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= de) return;
  out[c] = W_dist[2 * c];
  out[de + c] = W_dist[2 * c + 1];
  out[2 * de + c] = (b_dist[c] + b_n1[c]) + b_n2[c];
  out[3 * de + c] = W_n1[c];
  out[4 * de + c] = W_n2[c];
}

__global__ void vec_sum4_kernel(const float* a, const float* b, const float* c, const float* d, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ((a[i] + b[i]) + c[i]) + d[i];
}

}  // namespace
}  // namespace scgg

using namespace scgg;

struct scgg_denoiser3d {
  scgg_denoiser3d_config c;
  std::map<std::string, Param3> params;
  bool finalized = false;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  f32x4_t *in_E_img = nullptr, *out_E_img = nullptr, *rad_img = nullptr;
  RowLin in_X2, pos2, out_X1;      // mlp_in_X.3, mlp_in_pos_high_dim.3, mlp_out_X.0
  float* dvec = nullptr;           // [n_layers][5][de]: the distance-feature vectors of every layer (block3d_dist_stats_kernel)
  std::vector<Layer3> layers;
};

namespace {

const Param3* find3(const scgg_denoiser3d* h, const std::string& name, int64_t numel) {
  auto it = h->params.find(name);
  if (it == h->params.end()) { fail("missing parameter '%s'", name.c_str()); return nullptr; }
  if (it->second.numel != numel) {
    fail("parameter '%s' has %lld elements, expected %lld", name.c_str(), (long long)it->second.numel, (long long)numel);
    return nullptr;
  }
  return &it->second;
}

int check3(const char* what) {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("%s: %s", what, hipGetErrorString(e));
}

// dst[N, ldd] columns [col0, col0+K) <- W[N,K]
int put3(scgg_denoiser3d* h, const std::string& name, int N, int K, float* dst, int ldd, int col0, hipStream_t st) {
  const Param3* w = find3(h, name, (int64_t)N * K);
  if (!w) return -1;
  launch_copy_cols(w->p, K, dst + col0, ldd, N, K, st);
  return 0;
}

// three-term bf16 image + padded bias of W[N,K] for the row-list Linear (K = 128 / 256 / 512 only)
int pack_rowlin(Bump& ar, bool fill, const float* W, const float* b, int N, int K, RowLin& out, hipStream_t st) {
  out = RowLin();
  if (K != 128 && K != 256 && K != 512) return 0;
  const int NB = (N + 31) / 32;
  out.N = N;
  out.K = K;
  out.W3 = ar.take<f32x4_t>((size_t)NB * (K / 16) * 192);
  out.b = ar.take<float>((size_t)NB * 32);
  if (!fill) return 0;
  if (!W || !b) return -1;
  PackArgs p3{W, out.W3, N, K, NB, K / 32};
  launch_pack_linear_x3(p3, st);
  launch_pad_vec(b, out.b, N, NB * 32, 0.f, st);
  return 0;
}
int pack_rowlin_named(scgg_denoiser3d* h, Bump& ar, bool fill, const std::string& name, int N, int K, RowLin& out, hipStream_t st) {
  const float *W = nullptr, *b = nullptr;
  if (fill && (K == 128 || K == 256 || K == 512)) {
    const Param3 *w = find3(h, name + ".weight", (int64_t)N * K), *bb = find3(h, name + ".bias", N);
    if (!w || !bb) return -1;
    W = w->p;
    b = bb->p;
  }
  return pack_rowlin(ar, fill, W, b, N, K, out, st);
}

int build3(scgg_denoiser3d* h, bool fill, hipStream_t st) {
  const scgg_denoiser3d_config& c = h->c;
  Bump ar(fill ? h->arena : nullptr);
  const int dx = c.dx, de = c.de, dy = c.dy, H = c.n_head;
  {
    const long long n_in = edge_mlp_img_f4(c.in_E, c.hid_E, de), n_out = edge_mlp_img_f4(de, c.hid_E, c.out_E),
                    n_rad = edge_mlp_img_f4(de, c.hid_E, 1);
    if (n_in < 0 || n_out < 0 || n_rad < 0) return fail("edge MLP kernels have no build for hid_E=%d", c.hid_E);
    h->in_E_img = ar.take<f32x4_t>((size_t)n_in);
    h->out_E_img = ar.take<f32x4_t>((size_t)n_out);
    h->rad_img = ar.take<f32x4_t>((size_t)n_rad);
    if (fill) {
      const Param3 *w0 = find3(h, "mlp_in_E.0.weight", (int64_t)c.hid_E * c.in_E), *b0 = find3(h, "mlp_in_E.0.bias", c.hid_E);
      const Param3 *w3 = find3(h, "mlp_in_E.3.weight", (int64_t)de * c.hid_E), *b3 = find3(h, "mlp_in_E.3.bias", de);
      const Param3 *v0 = find3(h, "mlp_out_E.0.weight", (int64_t)c.hid_E * de), *d0 = find3(h, "mlp_out_E.0.bias", c.hid_E);
      const Param3 *v3 = find3(h, "mlp_out_E.3.weight", (int64_t)c.out_E * c.hid_E), *d3 = find3(h, "mlp_out_E.3.bias", c.out_E);
      const Param3 *r0 = find3(h, "mlp_out_radius.0.weight", (int64_t)c.hid_E * de), *e0 = find3(h, "mlp_out_radius.0.bias", c.hid_E);
      const Param3 *r3 = find3(h, "mlp_out_radius.3.weight", c.hid_E), *e3 = find3(h, "mlp_out_radius.3.bias", 1);
      if (!w0 || !b0 || !w3 || !b3 || !v0 || !d0 || !v3 || !d3 || !r0 || !e0 || !r3 || !e3) return -1;
      EdgeMlpPackArgs pi{w0->p, b0->p, w3->p, b3->p, h->in_E_img, c.in_E, c.hid_E, de};
      EdgeMlpPackArgs po{v0->p, d0->p, v3->p, d3->p, h->out_E_img, de, c.hid_E, c.out_E};
      EdgeMlpPackArgs pr{r0->p, e0->p, r3->p, e3->p, h->rad_img, de, c.hid_E, 1};
      if (launch_edge_mlp_pack(pi, st) || launch_edge_mlp_pack(po, st) || launch_edge_mlp_pack(pr, st))
        return fail("edge MLP pack: unsupported dims in_E=%d hid_E=%d de=%d out_E=%d", c.in_E, c.hid_E, de, c.out_E);
    }
  }
  if (pack_rowlin_named(h, ar, fill, "mlp_in_X.3", dx, c.hid_X, h->in_X2, st)) return -1;
  if (pack_rowlin_named(h, ar, fill, "mlp_in_pos_high_dim.3", dx, c.hid_X, h->pos2, st)) return -1;
  if (pack_rowlin_named(h, ar, fill, "mlp_out_X.0", c.hid_X, dx, h->out_X1, st)) return -1;
  h->layers.resize(c.n_layers);
  h->dvec = ar.take<float>((size_t)c.n_layers * 5 * de);
  const int NL = 3 * dx + 2 * de, FL = 2 * de + 2 * dx, kcat = dy + 4 * dx + 8 * de;
  for (int l = 0; l < c.n_layers; ++l) {
    Layer3& L = h->layers[l];
    L.predicts_y = l != c.n_layers - 1;
    const std::string p = "tf_layers." + std::to_string(l) + ".", sa = p + "self_attn.";
    const long long nimg = block3d_img_f4(dx, de, c.dim_ffE);
    if (nimg < 0) return fail("3D edge kernel has no build for dx=%d de=%d dim_ffE=%d", dx, de, c.dim_ffE);
    L.img = ar.take<f32x4_t>((size_t)nimg);
    L.Wnl = ar.take<float>((size_t)NL * dx);
    L.bnl = ar.take<float>((size_t)NL);
    L.Wfilm = ar.take<float>((size_t)FL * dy);
    L.bfilm = ar.take<float>((size_t)FL);
    L.Wycat = ar.take<float>((size_t)dy * kcat);
    L.bycat = ar.take<float>((size_t)dy);
    L.Wt_cat = ar.take<float>((size_t)dy * kcat);
    L.Wt_o0 = ar.take<float>((size_t)dy * dy);
    L.Wt_o2 = ar.take<float>((size_t)dy * dy);
    L.Wt_l1 = ar.take<float>((size_t)dy * c.dim_ffy);
    L.Wt_l2 = ar.take<float>((size_t)dy * c.dim_ffy);
    L.Wt_film = ar.take<float>((size_t)FL * dy);
    if (pack_rowlin(ar, false, nullptr, nullptr, NL, dx, L.nl, st)) return -1;      // filled after the concatenation below
    if (pack_rowlin_named(h, ar, fill, sa + "out", dx, H * dx, L.out, st)) return -1;
    if (pack_rowlin_named(h, ar, fill, sa + "e_x_mul.lin", dx, 4 * de, L.ex, st)) return -1;
    if (pack_rowlin_named(h, ar, fill, sa + "pos_x_mul.lin", dx, 4 * de, L.px, st)) return -1;
    if (pack_rowlin_named(h, ar, fill, sa + "x_out", dx, dx, L.x_out, st)) return -1;
    if (pack_rowlin_named(h, ar, fill, p + "linX1", c.dim_ffX, dx, L.linX1, st)) return -1;
    if (pack_rowlin_named(h, ar, fill, p + "linX2", dx, c.dim_ffX, L.linX2, st)) return -1;
    if (!fill) continue;
    {
      auto W = [&](const std::string& nm, int64_t numel) { const Param3* q = find3(h, nm, numel); return q ? q->p : nullptr; };
      Block3DPackArgs a;
      a.W_in = W(sa + "in_E.weight", (int64_t)de * de);            a.b_in = W(sa + "in_E.bias", de);
      a.W_dadd = W(sa + "dist_add_e.weight", (int64_t)de * de);    a.b_dadd = W(sa + "dist_add_e.bias", de);
      a.W_dmul = W(sa + "dist_mul_e.weight", (int64_t)de * de);    a.b_dmul = W(sa + "dist_mul_e.bias", de);
      a.W_eout = W(sa + "e_out.weight", (int64_t)de * de);         a.b_eout = W(sa + "e_out.bias", de);
      a.W_ff1 = W(p + "linE1.weight", (int64_t)c.dim_ffE * de);    a.b_ff1 = W(p + "linE1.bias", c.dim_ffE);
      a.W_ff2 = W(p + "linE2.weight", (int64_t)de * c.dim_ffE);    a.b_ff2 = W(p + "linE2.bias", de);
      a.W_eatt = W(sa + "e_att_mul.weight", (int64_t)H * de);      a.b_eatt = W(sa + "e_att_mul.bias", H);
      a.W_patt = W(sa + "pos_att_mul.weight", (int64_t)H * de);    a.b_patt = W(sa + "pos_att_mul.bias", H);
      a.W_a = W(sa + "a.weight", (int64_t)H * dx);
      a.ln1g = W(p + "normE1.weight", de);  a.ln1b = W(p + "normE1.bias", de);
      a.ln2g = W(p + "normE2.weight", de);  a.ln2b = W(p + "normE2.bias", de);
      a.W_dist = W(sa + "lin_dist1.weight", (int64_t)de * 2);      a.b_dist = W(sa + "lin_dist1.bias", de);
      a.W_n1 = W(sa + "lin_norm_pos1.weight", de);                 a.b_n1 = W(sa + "lin_norm_pos1.bias", de);
      a.W_n2 = W(sa + "lin_norm_pos2.weight", de);                 a.b_n2 = W(sa + "lin_norm_pos2.bias", de);
      a.img = L.img;
      a.n_head = H;
      const float* all[] = {a.W_in, a.b_in, a.W_dadd, a.b_dadd, a.W_dmul, a.b_dmul, a.W_eout, a.b_eout, a.W_ff1, a.b_ff1,
                            a.W_ff2, a.b_ff2, a.W_eatt, a.b_eatt, a.W_patt, a.b_patt, a.W_a, a.ln1g, a.ln1b, a.ln2g, a.ln2b,
                            a.W_dist, a.b_dist, a.W_n1, a.b_n1, a.W_n2, a.b_n2};
      for (const float* q : all)
        if (!q) return -1;
      if (launch_block3d_pack(a, dx, de, c.dim_ffE, st)) return fail("3D edge pack failed");
      hipLaunchKernelGGL(dvec_kernel, dim3((de + 255) / 256), dim3(256), 0, st, a.W_dist, a.b_dist, a.W_n1, a.b_n1, a.W_n2, a.b_n2,
                         h->dvec + (size_t)l * 5 * de, de);
    }
    const char* nl[5] = {"q", "k", "v", "x_e_mul1", "x_e_mul2"};
    int r0 = 0;
    for (int t = 0; t < 5; ++t) {
      const int N = t < 3 ? dx : de;
      if (put3(h, sa + nl[t] + ".weight", N, dx, L.Wnl + (size_t)r0 * dx, dx, 0, st)) return -1;
      if (put3(h, sa + nl[t] + ".bias", 1, N, L.bnl + r0, N, 0, st)) return -1;
      r0 += N;
    }
    if (L.nl.W3 != nullptr) {
      PackArgs p3{L.Wnl, L.nl.W3, NL, dx, (NL + 31) / 32, dx / 32};
      launch_pack_linear_x3(p3, st);
      launch_pad_vec(L.bnl, L.nl.b, NL, (NL + 31) / 32 * 32, 0.f, st);
    }
    const char* fl[4] = {"y_e_add", "y_e_mul", "y_x_add", "y_x_mul"};
    r0 = 0;
    for (int t = 0; t < 4; ++t) {
      const int N = t < 2 ? de : dx;
      if (put3(h, sa + fl[t] + ".weight", N, dy, L.Wfilm + (size_t)r0 * dy, dy, 0, st)) return -1;
      if (put3(h, sa + fl[t] + ".bias", 1, N, L.bfilm + r0, N, 0, st)) return -1;
      r0 += N;
    }
    if (L.predicts_y) {
      if (put3(h, sa + "y_y.weight", dy, dy, L.Wycat, kcat, 0, st)) return -1;
      if (put3(h, sa + "x_y.lin.weight", dy, 4 * dx, L.Wycat, kcat, dy, st)) return -1;
      if (put3(h, sa + "e_y.lin.weight", dy, 4 * de, L.Wycat, kcat, dy + 4 * dx, st)) return -1;
      if (put3(h, sa + "dist_y.lin.weight", dy, 4 * de, L.Wycat, kcat, dy + 4 * dx + 4 * de, st)) return -1;
      const Param3 *b0 = find3(h, sa + "y_y.bias", dy), *b1 = find3(h, sa + "x_y.lin.bias", dy),
                   *b2 = find3(h, sa + "e_y.lin.bias", dy), *b3 = find3(h, sa + "dist_y.lin.bias", dy);
      if (!b0 || !b1 || !b2 || !b3) return -1;
      hipLaunchKernelGGL(vec_sum4_kernel, dim3((dy + 255) / 256), dim3(256), 0, st, b0->p, b1->p, b2->p, b3->p, L.bycat, dy);
      const Param3 *o0 = find3(h, sa + "y_out.0.weight", (int64_t)dy * dy), *o2 = find3(h, sa + "y_out.2.weight", (int64_t)dy * dy),
                   *l1 = find3(h, p + "lin_y1.weight", (int64_t)c.dim_ffy * dy), *l2 = find3(h, p + "lin_y2.weight", (int64_t)dy * c.dim_ffy);
      if (!o0 || !o2 || !l1 || !l2) return -1;
      launch_transpose(L.Wycat, L.Wt_cat, dy, kcat, st);
      launch_transpose(o0->p, L.Wt_o0, dy, dy, st);
      launch_transpose(o2->p, L.Wt_o2, dy, dy, st);
      launch_transpose(l1->p, L.Wt_l1, c.dim_ffy, dy, st);
      launch_transpose(l2->p, L.Wt_l2, dy, c.dim_ffy, st);
    }
    launch_transpose(L.Wfilm, L.Wt_film, FL, dy, st);
    if (check3("3D weight concatenation")) return -1;
  }
  h->arena_bytes = align_up3(ar.off, 256);
  return 0;
}

struct Work3 {
  float *X, *X1, *nodelin, *film, *pos4, *E0, *E1, *attn, *stat, *wvcat, *rowstat, *rowagg, *rowstatD, *rowaggD, *t_out, *t_ex, *t_px, *newX, *t_ff,
      *y, *y1, *zcat, *etoy, *yt1, *yt2, *hidX, *hidy, *xo;
  int *rows, *row_count, *units, *unit_count, *goff, *chunk_cnt;
  size_t bytes;
};

Work3 carve3(const scgg_denoiser3d_config& c, int B, int n, char* base) {
  Bump ar(base);
  Work3 w;
  const size_t M = (size_t)B * n, P = M * n, nch = (size_t)(n + 15) / 16;
  const int dx = c.dx, de = c.de, dy = c.dy;
  const int wideX = std::max(std::max(c.dim_ffX, c.hid_X), dx);
  w.X = ar.take<float>(M * dx);
  w.X1 = ar.take<float>(M * dx);
  w.nodelin = ar.take<float>(M * (3 * dx + 2 * de));
  w.film = ar.take<float>((size_t)2 * B * (2 * de + 2 * dx));     // double-buffered: layer l+1's vectors are written while layer l's are read
  w.pos4 = ar.take<float>((size_t)c.n_layers * M * 4);
  w.E0 = ar.take<float>(P * de);
  w.E1 = ar.take<float>(P * de);
  w.attn = ar.take<float>(M * nch * ((size_t)c.n_head * dx + 32));
  w.stat = ar.take<float>(M * nch * 8 * de);
  w.wvcat = ar.take<float>(M * (size_t)c.n_head * dx);
  w.rowstat = ar.take<float>(M * 8 * de);
  w.rowagg = ar.take<float>(M * 8 * de);
  w.rowstatD = ar.take<float>((size_t)c.n_layers * M * 4 * de);
  w.rowaggD = ar.take<float>((size_t)c.n_layers * M * 4 * de);
  w.t_out = ar.take<float>(M * dx);
  w.t_ex = ar.take<float>(M * dx);
  w.t_px = ar.take<float>(M * dx);
  w.newX = ar.take<float>(M * dx);
  w.t_ff = ar.take<float>(M * wideX);
  w.hidX = ar.take<float>(M * wideX);
  w.xo = ar.take<float>(M * (size_t)std::max(c.out_X, 1));
  const int ywide = std::max(std::max(c.dim_ffy, c.hid_y), dy);
  w.y = ar.take<float>((size_t)B * dy);
  w.y1 = ar.take<float>((size_t)B * dy);
  w.zcat = ar.take<float>((size_t)B * (dy + 4 * dx + 8 * de));
  w.etoy = ar.take<float>((size_t)B * 8 * de);
  w.yt1 = ar.take<float>((size_t)B * ywide);
  w.yt2 = ar.take<float>((size_t)B * ywide);
  w.hidy = ar.take<float>((size_t)B * ywide);
  w.rows = ar.take<int>(M);
  w.row_count = ar.take<int>(64);
  w.units = ar.take<int>(M * nch);
  w.unit_count = ar.take<int>(64);
  w.goff = ar.take<int>((size_t)2 * B);
  w.chunk_cnt = ar.take<int>((size_t)B * nch);
  w.bytes = align_up3(ar.off, 256);
  return w;
}

int gemm3(scgg_denoiser3d* h, const float* A, int lda, const std::string& name, float* C, int ldc, int M, int N, int K, int flags,
          const uint8_t* rowmask, hipStream_t st) {
  const Param3 *W = find3(h, name + ".weight", (int64_t)N * K), *b = find3(h, name + ".bias", N);
  if (!W || !b) return -1;
  GemmArgs g{A, W->p, b->p, C, rowmask, M, N, K, lda, K, ldc, flags, 0, 0, 0, 0};
  return launch_gemm(g, st);
}
int gemm3_raw(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K, int flags,
              const uint8_t* rowmask, hipStream_t st) {
  GemmArgs g{A, W, bias, C, rowmask, M, N, K, lda, K, ldc, flags, 0, 0, 0, 0};
  return launch_gemm(g, st);
}
// sym + mask copy of an unsymmetrised hidden E into a caller buffer (debug tap)
__global__ void sym_copy3_kernel(const float* E, const uint8_t* mask, float* out, int B, int n, int de) {
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

#define TRY(x) do { if ((x) != 0) return -1; } while (0)

int run3(scgg_denoiser3d* h, const float* X, const float* E, const float* y, const float* pos, const uint8_t* mask, int B, int n,
         int n_run_layers, bool tap, float* o1, float* o2, float* o3, void* workspace, size_t ws_bytes, hipStream_t st) {
  if (!h || !h->finalized) return fail("denoiser3d not finalized");
  if (B <= 0 || n <= 0) return fail("bad batch shape B=%d n=%d", B, n);
  if (!X || !E || !y || !pos || !mask) return fail("denoiser3d: null input");
  const scgg_denoiser3d_config& c = h->c;
  if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return fail("workspace must be 256-byte aligned");
  Work3 w = carve3(c, B, n, static_cast<char*>(workspace));
  if (ws_bytes < w.bytes) return fail("workspace too small: %zu < %zu", ws_bytes, w.bytes);
  const int M = B * n, dx = c.dx, de = c.de, dy = c.dy, H = c.n_head;
  const int NL = 3 * dx + 2 * de, FL = 2 * de + 2 * dx, kcat = dy + 4 * dx + 8 * de;
  const bool eq = c.equivariant != 0;

  TRY(launch_build_lists(mask, B, n, w.goff, w.rows, w.row_count, w.units, w.unit_count, st));
  TRY(launch_block3d_chunk_count(mask, B, n, w.chunk_cnt, st));
  TRY(launch_block3d_pos(pos, mask, B, n, c.n_layers, w.pos4, st));
  TRY(launch_block3d_dist_stats(w.pos4, mask, h->dvec, w.rows, w.row_count, w.rowstatD, w.rowaggD, B, n, de, c.n_layers, st));
  // ---- input MLPs (:451-461).  Hidden E is stored unsymmetrised; every consumer symmetrises and masks on load.
  {
    EdgeMlpUnitArgs a{E, nullptr, mask, h->in_E_img, w.E0, w.units, w.unit_count, B, n, c.in_E, 0};
    if (launch_edge_mlp_in(a, c.in_E, c.hid_E, de, st)) return fail("edge input MLP: unsupported dims");
  }
  TRY(gemm3(h, y, c.in_y, "mlp_in_y.0", w.hidy, c.hid_y, B, c.hid_y, c.in_y, GEMM_RELU, nullptr, st));
  TRY(gemm3(h, w.hidy, c.hid_y, "mlp_in_y.3", w.y, dy, B, dy, c.hid_y, GEMM_RELU, nullptr, st));
  // row-list Linear on the valid rows (bf16x3 scheme) where the shape is packed, else the generic GEMM over all rows
  auto rowlin = [&](const RowLin& pk, const float* Wraw, const float* braw, const float* A, int lda, float* C, int ldc, int N, int K,
                    bool relu) -> int {
    if (pk.W3 != nullptr) {
      NodeLinearArgs a{A, nullptr, pk.b, C, w.rows, w.row_count, N, K, (N + 31) / 32, lda, ldc, 1, relu ? 1 : 0};
      const int rc = launch_node_linear_x3(a, pk.W3, M, st);
      if (rc == 0) return 0;
      if (rc != -1) return fail("row-list Linear launch failed (N=%d K=%d)", N, K);
    }
    return gemm3_raw(A, lda, Wraw, braw, C, ldc, M, N, K, (relu ? GEMM_RELU : 0) | GEMM_ROWMASK, mask, st);
  };
  auto rowlin_named = [&](const RowLin& pk, const std::string& name, const float* A, int lda, float* C, int ldc, int N, int K,
                          bool relu) -> int {
    const Param3 *W = find3(h, name + ".weight", (int64_t)N * K), *b = find3(h, name + ".bias", N);
    if (!W || !b) return -1;
    return rowlin(pk, W->p, b->p, A, lda, C, ldc, N, K, relu);
  };
  auto ln_rows = [&](const float* x, const float* r, const std::string& name, float* out) -> int {
    const Param3 *g = find3(h, name + ".weight", dx), *b = find3(h, name + ".bias", dx);
    if (!g || !b) return -1;
    AddLnArgs a{x, r, g->p, b->p, mask, out, M, dx, w.rows, w.row_count};
    return launch_add_layernorm(a, st);
  };
  // first Linear of a node MLP: K <= 64 inputs per row (raw node features / coordinates), valid rows only when possible
  auto first_lin = [&](const std::string& name, const float* A, int K, float* C, int N) -> int {
    const Param3 *W = find3(h, name + ".weight", (int64_t)N * K), *b = find3(h, name + ".bias", N);
    if (!W || !b) return -1;
    if (K <= 64 && N % 4 == 0) {
      NarrowLinearArgs na{A, W->p, b->p, C, w.rows, w.row_count, N, K, K, N, 1};
      if (launch_narrow_linear(na, M, st) == 0) return 0;
    }
    return gemm3_raw(A, K, W->p, b->p, C, N, M, N, K, GEMM_RELU, nullptr, st);
  };
  auto x_head = [&](const float* hidden_X) -> int {     // mlp_out_X on `hidden_X` + residual to the input, masked (:467-472, :492)
    TRY(rowlin_named(h->out_X1, "mlp_out_X.0", hidden_X, dx, w.hidX, c.hid_X, c.hid_X, dx, true));
    const Param3 *w3 = find3(h, "mlp_out_X.3.weight", (int64_t)c.out_X * c.hid_X), *b3 = find3(h, "mlp_out_X.3.bias", c.out_X);
    if (!w3 || !b3) return -1;
    if (c.out_X <= 16 && c.hid_X % 4 == 0) {
      if (hipMemsetAsync(o1, 0, (size_t)M * c.out_X * sizeof(float), st) != hipSuccess) return fail("memset failed");
      if (launch_row_head(w.hidX, c.hid_X, w3->p, b3->p, X, c.in_X, w.rows, w.row_count, o1, M, c.out_X, c.hid_X, st) == 0) return 0;
    }
    TRY(gemm3_raw(w.hidX, c.hid_X, w3->p, b3->p, w.xo, c.out_X, M, c.out_X, c.hid_X, 0, nullptr, st));
    return launch_x_out_residual(w.xo, X, c.in_X, mask, o1, M, c.out_X, st);
  };
  if (hipMemsetAsync(w.X, 0, (size_t)M * dx * sizeof(float), st) != hipSuccess) return fail("memset failed");   // padded rows stay zero
  if (eq) {
    TRY(first_lin("mlp_in_X.0", X, c.in_X, w.hidX, c.hid_X));
    TRY(rowlin_named(h->in_X2, "mlp_in_X.3", w.hidX, c.hid_X, w.X, dx, dx, c.hid_X, true));
  } else {
    TRY(first_lin("mlp_in_pos_high_dim.0", pos, 3, w.hidX, c.hid_X));
    TRY(rowlin_named(h->pos2, "mlp_in_pos_high_dim.3", w.hidX, c.hid_X, w.X, dx, dx, c.hid_X, true));
    if (!tap && c.out_X > 0) {   // :467: the node head never sees the layers
      TRY(first_lin("mlp_in_X.0", X, c.in_X, w.hidX, c.hid_X));
      TRY(rowlin_named(h->in_X2, "mlp_in_X.3", w.hidX, c.hid_X, w.t_ff, dx, dx, c.hid_X, true));
      TRY(x_head(w.t_ff));
    }
  }
  float *Ecur = w.E0, *Enext = w.E1;
  const int L_run = tap ? n_run_layers : c.n_layers;
  // SCGG_Y_POOL=0: the graph merge as a kernel of its own in front of the y round (development A/B switch)
  static const bool y_pool_on = [] { const char* e = getenv("SCGG_Y_POOL"); return !(e != nullptr && e[0] == '0'); }();
  auto y_round = [&](int l, bool update, float* film_dst) -> int {   // y(l) -> y(l+1) (update) and the FiLM vectors of `film_of`
    const int film_of = update ? l + 1 : l;
    YLayerArgs ya{};
    if (update && y_pool_on) {        // the y workgroup of a graph merges its graph's row aggregates itself (pool_stats.h)
      ya.gm_rowagg = w.rowagg; ya.gm_rowaggD = w.rowaggD + (size_t)l * M * 4 * de; ya.gm_newX = w.newX; ya.mask = mask; ya.n = n;
      ya.de = de;
    }
    ya.y = w.y; ya.ycat = w.zcat; ya.etoy = w.etoy; ya.B = B; ya.dy = dy; ya.ffy = c.dim_ffy; ya.kcat = kcat; ya.dx = dx;
    ya.update = update ? 1 : 0;
    ya.film_n = FL;
    if (update) {
      const Layer3& L = h->layers[l];
      const std::string p = "tf_layers." + std::to_string(l) + ".", sa = p + "self_attn.";
      const Param3 *bo0 = find3(h, sa + "y_out.0.bias", dy), *bo2 = find3(h, sa + "y_out.2.bias", dy),
                   *g1 = find3(h, p + "norm_y1.weight", dy), *b1 = find3(h, p + "norm_y1.bias", dy),
                   *bl1 = find3(h, p + "lin_y1.bias", c.dim_ffy), *bl2 = find3(h, p + "lin_y2.bias", dy),
                   *g2 = find3(h, p + "norm_y2.weight", dy), *b2 = find3(h, p + "norm_y2.bias", dy);
      if (!bo0 || !bo2 || !g1 || !b1 || !bl1 || !bl2 || !g2 || !b2) return -1;
      ya.Wt_cat = L.Wt_cat; ya.b_cat = L.bycat; ya.Wt_o0 = L.Wt_o0; ya.b_o0 = bo0->p; ya.Wt_o2 = L.Wt_o2; ya.b_o2 = bo2->p;
      ya.ln1_g = g1->p; ya.ln1_b = b1->p; ya.Wt_l1 = L.Wt_l1; ya.b_l1 = bl1->p; ya.Wt_l2 = L.Wt_l2; ya.b_l2 = bl2->p;
      ya.ln2_g = g2->p; ya.ln2_b = b2->p;
    }
    if (film_of < c.n_layers) {
      ya.Wt_film = h->layers[film_of].Wt_film;
      ya.b_film = h->layers[film_of].bfilm;
      ya.film = film_dst;
    }
    if (launch_y_layer(ya, st)) return fail("y kernel: unsupported dims dy=%d dim_ffy=%d", dy, c.dim_ffy);
    return 0;
  };
  float* film_buf[2] = {w.film, w.film + (size_t)B * FL};
  bool nodelin_done = false;     // the fused node kernel of layer l already produced layer l+1's q | k | v | x_e_mul
  if (L_run > 0) TRY(y_round(0, false, film_buf[0]));
  for (int l = 0; l < L_run; ++l) {
    const Layer3& L = h->layers[l];
    const std::string p = "tf_layers." + std::to_string(l) + ".", sa = p + "self_attn.";
    const bool need_x = l != c.n_layers - 1 || eq || tap;      // the last layer's X feeds nothing in Graph3DTransformer
    const float* film = film_buf[l & 1];
    if (!nodelin_done) TRY(rowlin(L.nl, L.Wnl, L.bnl, w.X, dx, w.nodelin, NL, NL, dx, false));
    nodelin_done = false;
    {
      Block3DArgs a;
      a.E_in = Ecur; a.E_out = Enext; a.mask = mask; a.nodelin = w.nodelin; a.ldn = NL; a.film = film; a.ldf = FL;
      a.pos4 = w.pos4 + (size_t)l * M * 4; a.img = L.img; a.attn_part = w.attn; a.stat_part = w.stat;
      a.units = w.units; a.unit_count = w.unit_count; a.B = B; a.n = n; a.n_head = H; a.need_attn = need_x ? 3 : 0;   // bit 0: attention partials, bit 1: statistics partials
      { static const char* dbg = getenv("SCGG_B3D_PARTS"); if (dbg && need_x) a.need_attn = atoi(dbg); }   // timing experiments only
      a.inv_sqrt_de = 1.0f / sqrtf((float)de);
      if (launch_block3d_edge(a, dx, de, c.dim_ffE, st)) return fail("3D edge kernel: unsupported dims");
    }
    if (need_x) {
      RowMerge3DArgs r{w.attn, w.stat, w.chunk_cnt, film, FL, w.wvcat, w.rowstat, w.rowagg, w.rows, w.row_count, n, H, dx, de};
      TRY(launch_block3d_row_merge(r, M, st));
      TRY(rowlin_named(L.out, sa + "out", w.wvcat, H * dx, w.t_out, dx, dx, H * dx, false));
      {   // the two EtoX Linears (same K = 4 de, same rows) in one launch when both have their bf16x3 images
        int rc = -1;
        if (L.ex.W3 != nullptr && L.px.W3 != nullptr) {
          NodeLinearArgs a0{w.rowstat, nullptr, L.ex.b, w.t_ex, w.rows, w.row_count, dx, 4 * de, (dx + 31) / 32, 8 * de, dx, 1, 0};
          NodeLinearArgs a1{w.rowstatD + (size_t)l * M * 4 * de, nullptr, L.px.b, w.t_px, w.rows, w.row_count, dx, 4 * de,
                            (dx + 31) / 32, 4 * de, dx, 1, 0};
          rc = launch_node_linear_x3_pair(a0, L.ex.W3, a1, L.px.W3, M, st);
          if (rc != 0 && rc != -1) return fail("row-list Linear pair launch failed");
        }
        if (rc != 0) {
          TRY(rowlin_named(L.ex, sa + "e_x_mul.lin", w.rowstat, 8 * de, w.t_ex, dx, dx, 4 * de, false));
          TRY(rowlin_named(L.px, sa + "pos_x_mul.lin", w.rowstatD + (size_t)l * M * 4 * de, 4 * de, w.t_px, dx, dx, 4 * de, false));
        }
      }
      TRY(launch_block3d_newx(w.t_out, w.t_ex, w.t_px, mask, film, FL, 2 * de, w.newX, M, n, dx, st));
      if (L.predicts_y) {   // y(l+1) and the FiLM vectors of layer l+1 (into the other buffer: film(l) is still being read)
        if (!y_pool_on) {
          GraphMerge3DArgs g{w.rowagg, w.rowaggD + (size_t)l * M * 4 * de, mask, w.newX, w.zcat, w.etoy, kcat, n, dx, de, dy};
          TRY(launch_block3d_graph_merge(g, B, st));
        }
        TRY(y_round(l, true, film_buf[(l + 1) & 1]));
      }
      // x_out, normX1, linX1 / linX2, normX2 and the next layer's q | k | v | x_e_mul1 | x_e_mul2: one fused launch of the 2D
      // engine's node kernel (fed with the finished newX) when its shape is built, else the separate launches
      bool fused = false;
      if (node_layer_supported(dx, c.dim_ffX) && node_layer_x3_enabled() && L.x_out.W3 && L.linX1.W3 && L.linX2.W3) {
        const Param3 *g1 = find3(h, p + "normX1.weight", dx), *b1 = find3(h, p + "normX1.bias", dx),
                     *g2 = find3(h, p + "normX2.weight", dx), *b2 = find3(h, p + "normX2.bias", dx);
        if (!g1 || !b1 || !g2 || !b2) return -1;
        const bool next = l + 1 < c.n_layers && h->layers[l + 1].nl.W3 != nullptr && NL % 32 == 0;
        if (next || l + 1 >= c.n_layers) {
          NodeLayerArgs na{};
          na.mask = mask; na.X = w.X; na.rows = w.rows; na.row_count = w.row_count; na.n = n;
          na.b_xout = L.x_out.b; na.ln1_g = g1->p; na.ln1_b = b1->p; na.b_lin1 = L.linX1.b; na.b_lin2 = L.linX2.b;
          na.ln2_g = g2->p; na.ln2_b = b2->p;
          na.W3_xout = L.x_out.W3; na.W3_lin1 = L.linX1.W3; na.W3_lin2 = L.linX2.W3;
          na.X_out = w.X;                        // in place: a workgroup reads its rows' residual before it writes them
          na.newX_in = w.newX;
          if (next) {
            na.W3_qkv = h->layers[l + 1].nl.W3; na.b_qkv = h->layers[l + 1].nl.b; na.qkv = w.nodelin;
            na.tail_pairs = NL / 32; na.tail_ld = NL; na.tail_relu = 0;
          }
          if (launch_node_layer(na, dx, c.dim_ffX, M, st) == 0) {
            fused = true;
            nodelin_done = next;
          }
        }
      }
      if (!fused) {
        TRY(rowlin_named(L.x_out, sa + "x_out", w.newX, dx, w.t_out, dx, dx, dx, false));
        TRY(ln_rows(w.X, w.t_out, p + "normX1", w.X1));
        TRY(rowlin_named(L.linX1, p + "linX1", w.X1, dx, w.t_ff, c.dim_ffX, c.dim_ffX, dx, true));
        TRY(rowlin_named(L.linX2, p + "linX2", w.t_ff, c.dim_ffX, w.t_out, dx, dx, c.dim_ffX, false));
        TRY(ln_rows(w.X1, w.t_out, p + "normX2", w.X));
      }
    }
    float* t = Ecur; Ecur = Enext; Enext = t;
  }
  if (tap) {
    if (o1) (void)hipMemcpyAsync(o1, w.X, (size_t)M * dx * 4, hipMemcpyDeviceToDevice, st);
    if (o2) {
      const long long total = (long long)M * n * de;
      hipLaunchKernelGGL(sym_copy3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Ecur, mask, o2, B, n, de);
    }
    if (o3) (void)hipMemcpyAsync(o3, w.y, (size_t)B * dy * 4, hipMemcpyDeviceToDevice, st);
    return check3("denoiser3d tap");
  }
  if (eq && c.out_X > 0) TRY(x_head(w.X));
  // ---- edge heads (:468-487): logits with the residual to the input classes, radius without one
  if (hipMemsetAsync(o2, 0, (size_t)M * n * c.out_E * 4, st) != hipSuccess) return fail("memset failed");
  {
    EdgeMlpUnitArgs a{Ecur, E, mask, h->out_E_img, o2, w.units, w.unit_count, B, n, c.in_E, c.out_E};
    if (launch_edge_mlp_out(a, de, c.hid_E, c.out_E, st)) return fail("edge output MLP: unsupported dims");
  }
  if (o3) {
    if (hipMemsetAsync(o3, 0, (size_t)M * n * 4, st) != hipSuccess) return fail("memset failed");
    EdgeMlpUnitArgs a{Ecur, nullptr, mask, h->rad_img, o3, w.units, w.unit_count, B, n, c.in_E, 1};
    if (launch_edge_mlp_out(a, de, c.hid_E, 1, st)) return fail("radius MLP: unsupported dims");
  }
  return check3("denoiser3d forward");
}

}  // namespace

extern "C" scgg_denoiser3d* scgg_denoiser3d_create(const scgg_denoiser3d_config* cfg) {
  if (!cfg) { fail("null config"); return nullptr; }
  const scgg_denoiser3d_config& c = *cfg;
  if (c.n_layers < 1 || c.n_head < 1 || c.n_head > 16 || c.dx % c.n_head != 0) { fail("bad n_layers / n_head"); return nullptr; }
  if (c.in_X < 1 || c.in_E < 1 || c.in_y < 1 || c.out_E < 1 || c.out_E > c.in_E || c.out_X > c.in_X || c.out_X < 0) {
    fail("bad input / output widths"); return nullptr;
  }
  if (block3d_img_f4(c.dx, c.de, c.dim_ffE) < 0) {
    fail("3D edge kernel has no build for dx=%d de=%d dim_ffE=%d (built: 128/64/128, 256/64/128, 32/32/64)", c.dx, c.de, c.dim_ffE);
    return nullptr;
  }
  if (edge_mlp_img_f4(c.in_E, c.hid_E, c.de) < 0 || edge_mlp_img_f4(c.de, c.hid_E, c.out_E) < 0) {
    fail("edge MLP kernels have no build for in_E=%d hid_E=%d de=%d out_E=%d", c.in_E, c.hid_E, c.de, c.out_E);
    return nullptr;
  }
  if (c.dy % 4 != 0 || c.dx % 4 != 0) { fail("dx, dy must be multiples of 4"); return nullptr; }
  if (!y_layer_supported(c.dy, c.dim_ffy, c.dy + 4 * c.dx + 8 * c.de)) {
    fail("y kernel has no build for dy=%d dim_ffy=%d", c.dy, c.dim_ffy); return nullptr;
  }
  scgg_denoiser3d* h = new scgg_denoiser3d();
  h->c = c;
  return h;
}

extern "C" void scgg_denoiser3d_destroy(scgg_denoiser3d* h) {
  if (!h) return;
  if (h->arena) (void)hipFree(h->arena);
  delete h;
}

extern "C" int scgg_denoiser3d_set_param(scgg_denoiser3d* h, const char* name, const float* data, int64_t numel) {
  if (!h || !name || !data) return fail("set_param: null argument");
  h->params[name] = Param3{data, numel};
  return 0;
}

extern "C" int scgg_denoiser3d_finalize(scgg_denoiser3d* h, void* stream) {
  if (!h) return fail("finalize: null handle");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (h->arena == nullptr) {
    if (build3(h, false, st)) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&h->arena), h->arena_bytes) != hipSuccess) return fail("arena allocation failed");
    if (hipMemsetAsync(h->arena, 0, h->arena_bytes, st) != hipSuccess) return fail("arena memset failed");
  }
  if (build3(h, true, st)) return -1;
  h->finalized = true;
  return 0;
}

extern "C" size_t scgg_denoiser3d_workspace_bytes(const scgg_denoiser3d* h, int32_t B, int32_t n) {
  if (!h || B <= 0 || n <= 0) return 0;
  return carve3(h->c, B, n, nullptr).bytes;
}

extern "C" int scgg_denoiser3d_forward(scgg_denoiser3d* h, const float* X, const float* E, const float* y, const float* pos,
                                       const uint8_t* node_mask, int32_t B, int32_t n, float* out_X, float* out_E,
                                       float* out_radius, void* workspace, size_t workspace_bytes, void* stream) {
  if (!out_E || (h && h->c.out_X > 0 && !out_X)) return fail("denoiser3d_forward: null output");
  return run3(h, X, E, y, pos, node_mask, B, n, 0, false, out_X, out_E, out_radius, workspace, workspace_bytes,
              static_cast<hipStream_t>(stream));
}

extern "C" int scgg_denoiser3d_forward_hidden(scgg_denoiser3d* h, const float* X, const float* E, const float* y, const float* pos,
                                              const uint8_t* node_mask, int32_t B, int32_t n, int32_t n_run_layers, float* hid_X,
                                              float* hid_E, float* hid_y, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h) return fail("null handle");
  if (n_run_layers < 0 || n_run_layers > h->c.n_layers) return fail("n_run_layers out of range");
  return run3(h, X, E, y, pos, node_mask, B, n, n_run_layers, true, hid_X, hid_E, hid_y, workspace, workspace_bytes,
              static_cast<hipStream_t>(stream));
}
