// Internal kernel argument blocks and launcher prototypes (not part of the C ABI; see include/scgg.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scgg {

int fail(const char* fmt, ...);      // records the thread's last error, returns -1
const char* last_error();

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---- per-DEVICE launch set-up caches (a process may drive more than one GPU: nothing below is per process)
constexpr int MAX_DEVICES = 64;
inline int current_device() {
  int d = -1;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < MAX_DEVICES) ? d : -1;
}
inline int device_cu_count() {   // CUs of the current device (-1 on failure)
  static int n_cu[MAX_DEVICES] = {};
  const int d = current_device();
  if (d < 0) return -1;
  if (n_cu[d] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess) return -1;
    n_cu[d] = prop.multiProcessorCount;
  }
  return n_cu[d];
}
struct LdsAttrOnce { bool done[MAX_DEVICES] = {}; };   // one per kernel instantiation (a function-local static)
inline int set_max_dynamic_lds(LdsAttrOnce& once, const void* kern, size_t lds) {
  const int d = current_device();
  if (d < 0) return -2;
  if (!once.done[d]) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    once.done[d] = true;
  }
  return 0;
}

struct EdgeMlpPackArgs {  // raw (torch layout) two-layer MLP -> LDS image of edge_mlp_unit_kernel
  const float* W1;        // [N1, K1]
  const float* b1;
  const float* W2;        // [N2, N1]
  const float* b2;
  f32x4_t* img;           // [edge_mlp_img_f4(K1, N1, N2)]
  int K1, N1, N2;
};

struct EdgeMlpUnitArgs {  // mlp_in_E / mlp_out_E over the 16-key units
  const float* src;       // in: model input E [B,n,n,dE_in]; out: hidden E [B,n,n,de] (unsymmetrised)
  const float* E_in;      // out only: model input E (residual uses channels [0,C)); null: no residual
  const uint8_t* mask;
  const f32x4_t* img;
  float* dst;             // in: hidden E [B,n,n,de]; out: logits [B,n,n,C] (zero-filled by the caller beforehand)
  const int* units;
  const int* unit_count;
  int B, n, dE_in, C;
};

struct EdgeLayerArgs {
  const float* E_in;      // [B,n,n,de]  (unsymmetrised producer output)
  float* E_out;           // [B,n,n,de]
  const uint8_t* mask;    // [B,n]
  const float* Q;         // [B,n,dx] masked
  const float* K;
  const float* V;
  int ldqkv;              // row stride of Q/K/V (they are column slices of one [B*n, 3*dx] buffer)
  const float* film;      // [B,4*dx] = [y_e_add | y_e_mul+1 | y_x_add | y_x_mul+1]
  const f32x4_t* phase_img;  // per-phase weight + bias images built by launch_edge_unit_pack
  const float* b_out;     // e_out.bias
  const float* ln1_g;
  const float* ln1_b;
  const float* ln2_g;
  const float* ln2_b;
  float* attn_part;       // [B*n*nch, 3, dx] per unit and channel: softmax max | sum p | sum p V
  const int* units;       // row * nch + jc of every unit, built by launch_build_unit_list (nch = ceil(n/16))
  const int* unit_count;  // device scalar
  float inv_sqrt_df;
  int B, n;
  int xcd_map = 1;          // x3 kernel: workgroups of one XCD take consecutive stretches of the unit list (0: plain order)
};

struct EdgeUnitPackArgs {   // raw (torch layout) weights of one layer's edge path -> phase images for edge_unit_layer_kernel
  const float* W_mul;     // [dx, de]
  const float* b_mul;
  const float* W_add;     // [dx, de]
  const float* b_add;
  const float* W_out;     // [de, dx]
  const float* W_ff1;     // [ffE, de]
  const float* b_ff1;
  const float* W_ff2;     // [de, ffE]
  const float* b_ff2;
  f32x4_t* img;           // [edge_phase_img_f4(dx,de,ffe)]
};

struct AttnFinishArgs {   // merge the units of every valid row: newX_pre = y_x_add + (y_x_mul+1) * weighted_V
  const float* attn_part;
  const uint8_t* mask;
  const float* film;      // [B, 4*dx]
  float* newX_pre;        // [B,n,dx]
  const int* rows;        // valid-row list + device count
  const int* row_count;
  int n, dx;
};

int launch_edge_mlp_in(const EdgeMlpUnitArgs& a, int dE_in, int hid, int de, hipStream_t st);
long long edge_mlp_img_f4(int K1, int N1, int N2);      // -1 if unsupported
int launch_edge_mlp_pack(const EdgeMlpPackArgs& a, hipStream_t st);
int launch_edge_layer(const EdgeLayerArgs& a, int dx, int de, int ffe, hipStream_t st);
long long edge_phase_img_f4(int dx, int de, int ffe);   // float4 count of one layer's phase images, -1 if unsupported
int launch_edge_unit_pack(const EdgeUnitPackArgs& a, int dx, int de, int ffe, hipStream_t st);
// valid-row list and 16-key unit list of a node mask; goff: 2*B ints of scratch (per-graph offsets)
int launch_build_lists(const uint8_t* mask, int B, int n, int* goff, int* rows, int* row_count, int* units,
                       int* unit_count, hipStream_t st, unsigned* chunk_bits = nullptr);
int launch_attn_finish(const AttnFinishArgs& a, int M, hipStream_t st);
int launch_edge_mlp_out(const EdgeMlpUnitArgs& a, int de, int hid, int C, hipStream_t st);

struct EdgePoolArgs {     // Etoy pooling straight from a hidden (unsymmetrised) E, n <= 512
  const float* E;         // [B,n,n,de]
  const uint8_t* mask;
  float* z;
  int B, n, de, z_stride, z_offset;
};
int launch_edge_pool_stats(const EdgePoolArgs& a, hipStream_t st);
// W workgroups per graph; parts: [B][W][5][de] floats of scratch
int launch_edge_pool_stats_sliced(const EdgePoolArgs& a, float* parts, int W, hipStream_t st);
constexpr int POOL_SLICED_MIN_N = 64;   // graphs beyond this many nodes: several workgroups per graph for the Etoy statistics
constexpr int POOL_SLICED_W = 32;

// ---- node / global path -------------------------------------------------------------------------
enum GemmFlags : int { GEMM_RELU = 1, GEMM_ROWMASK = 2, GEMM_ADD_ONE_COLS = 4 };

struct GemmArgs {
  const float* A;         // [M,K] row-major, lda
  const float* W;         // [N,K] torch Linear layout, ldw
  const float* bias;      // [N] or null
  float* C;               // [M,N], ldc
  const uint8_t* rowmask; // [M] (GEMM_ROWMASK: zero masked rows)
  int M, N, K, lda, ldw, ldc;
  int flags;
  int one_lo, one_hi;     // GEMM_ADD_ONE_COLS: columns in [one_lo,one_hi) and [one_lo2,one_hi2) get +1
  int one_lo2, one_hi2;
};
int launch_gemm(const GemmArgs& a, hipStream_t st);

struct AddLnArgs {        // out = LN(x + r) * rowmask
  const float* x;
  const float* r;
  const float* gamma;
  const float* beta;
  const uint8_t* rowmask; // or null
  float* out;
  int M, D;
  const int* rows;        // optional valid-row list (then only those rows are normalised) + device count
  const int* row_count;
};
int launch_add_layernorm(const AddLnArgs& a, hipStream_t st);

struct NodeLinearArgs {   // row-list Linear on pre-packed weights (node path)
  const float* A;         // [M,K] lda
  const f32x4_t* Wp;      // packed [N x K]
  const float* bias;      // padded to NB*32
  float* C;               // [M,N] ldc
  const int* rows;
  const int* row_count;
  int N, K, NB, lda, ldc, rbw, relu;   // relu (x3 kernel): 0 none, 1 ReLU, 2 tanh-GELU, 3 accumulate into C (no activation)
  // Optional prologue of the x3 kernel (the SiT blocks; K <= 256): the operand row is not read from A but formed from the
  // residual stream: h[row] (+ gate (.) o[row], stored back to h), LayerNorm without affine (eps ln_eps), (1 + scale), + shift.
  // gate / shift / scale are rows of per-timestep tables (already offset to their columns, row stride ln_ld); the row's
  // timestep is ln_t[row / ln_n] or ln_tc.
  float* ln_h = nullptr;
  const float* ln_o = nullptr;
  const float* ln_gate = nullptr;
  const float* ln_shift = nullptr;
  const float* ln_scale = nullptr;
  const int32_t* ln_t = nullptr;
  int ln_ld = 0, ln_n = 1, ln_tc = 0;
  float ln_eps = 0.f;
  // with ln_x: the residual row is formed as ln_be + ln_We . ln_x[row] (a [K, ln_cin] embedding of the raw input row, ln_cin
  // small) and stored to ln_h; ln_o is null then
  const float* ln_x = nullptr;
  const float* ln_We = nullptr;
  const float* ln_be = nullptr;
  int ln_cin = 0;
};
int launch_node_linear(const NodeLinearArgs& a, int M, hipStream_t st);
// the same Linear on the bf16x3 scheme (node_layer.hip; W3 from launch_pack_linear_x3); -1: shape not covered, use the above
int launch_node_linear_x3(const NodeLinearArgs& a, const f32x4_t* W3, int M, hipStream_t st);
int launch_node_linear_x3_pair(const NodeLinearArgs& a0, const f32x4_t* W0, const NodeLinearArgs& a1, const f32x4_t* W1, int M,
                               hipStream_t st);   // two Linears of the same K in one launch; -1: not covered

struct SitBlockArgs {     // one SiT block between two attention launches (sit_block.hip); weights as bf16x3 images
  const float* att;       // [M,H] attention output of this block
  float* h;               // [M,H] residual stream, updated in place
  float* o;               // [M,H] last block only: the MLP output (gate_mlp is then applied by the caller)
  const f32x4_t* W3_proj; // [H x H]
  const f32x4_t* W3_fc1;  // [4H x H]
  const f32x4_t* W3_fc2;  // nslices images [H x 256], fc2_stride f32x4 apart: K slice s of fc2
  const f32x4_t* W3_qkv;  // the NEXT block's [3H x H], or null after the last block
  const float* b_proj;    // biases padded to a multiple of 32
  const float* b_fc1;
  const float* b_fc2;
  const float* b_qkv;
  const float* tab;       // this block's modulation table [T][6H]: shift_msa | scale_msa | gate_msa | shift_mlp | scale_mlp | gate_mlp
  const float* tab_next;  // the next block's
  float* qkv;             // [M,3H] the next block's q | k | v
  const int* rows;
  const int* row_count;
  const int32_t* t;       // [B] timestep per graph, or null: t_const
  int t_const, n, nslices;
  long long fc2_stride;
};
int launch_sit_block(const SitBlockArgs& a, int H, int M, hipStream_t st);

struct NodeLayerArgs {    // fused node part of one layer (node_layer.hip); packed weights as for NodeLinearArgs
  const float* attn_part; // [B*n*nch, 3, dx] per-unit attention partials of this layer's edge kernel
  const uint8_t* mask;
  const float* film;      // [B, 4*dx]
  float* X;               // [B,n,dx] residual stream, updated in place on the valid rows
  const int* rows;
  const int* row_count;
  const f32x4_t* W_xout;  const float* b_xout;
  const float* ln1_g;     const float* ln1_b;
  const f32x4_t* W_lin1;  const float* b_lin1;
  const f32x4_t* W_lin2;  const float* b_lin2;
  const float* ln2_g;     const float* ln2_b;
  const f32x4_t* W_qkv;   const float* b_qkv;   // next layer's q|k|v (null: not computed)
  float* qkv;             // [B,n,3*dx]
  int n;
  // the same four weight matrices as three-term bf16 images (launch_pack_linear_x3) for the x3 kernel
  const f32x4_t* W3_xout; const f32x4_t* W3_lin1; const f32x4_t* W3_lin2; const f32x4_t* W3_qkv;
  float* X_out;           // where the updated residual stream goes (X itself: in place)
  // x3 kernel: shape of the trailing Linear on the new X (W3_qkv / b_qkv / qkv): pairs of 16-channel output blocks, row
  // stride of the destination, ReLU.  0 pairs = the next layer's q|k|v (3 dx wide, no activation); the last layer passes
  // mlp_out_X.0 here (transformer_model.py:371), so that the output head needs no separate launch over X.
  int tail_pairs, tail_ld, tail_relu;
  // x3 kernel only: the block's node output before x_out, already complete ([B,n,dx]; the position-aware block, whose attention
  // merge and EtoX FiLMs happen in its own kernels).  Non-null: attn_part / film are not read.
  const float* newX_in;
  const unsigned* chunk_bits;   // [B] bit jc: chunk jc of the graph holds a valid key (launch_build_lists), or null: from the mask
};
bool node_layer_supported(int dx, int ffx);
bool node_layer_x3_enabled();   // the x3 kernels are selected (SCGG_NODE_MODE is not f32)
int launch_node_layer(const NodeLayerArgs& a, int dx, int ffx, int M, hipStream_t st);

struct YLayerArgs {       // fused global-feature part of one layer (y_layer.hip); weights TRANSPOSED to [K][N]
  float* y;               // [B,dy] in / out
  const float* ycat;      // [B,kcat]: columns [dy, dy+4dx) = Xtoy stats (the other columns are not read)
  const float* etoy;      // [B,4de] Etoy stats of the layer input (edge_pool_stats_kernel)
  const float* Wt_cat;  const float* b_cat;   // [kcat,dy] = [y_y | x_y.lin | e_y.lin]^T, summed biases
  const float* Wt_o0;   const float* b_o0;    // y_out.0
  const float* Wt_o2;   const float* b_o2;    // y_out.2
  const float* ln1_g;   const float* ln1_b;
  const float* Wt_l1;   const float* b_l1;    // lin_y1^T [dy,ffy]
  const float* Wt_l2;   const float* b_l2;    // lin_y2^T [ffy,dy]
  const float* ln2_g;   const float* ln2_b;
  const float* Wt_film; const float* b_film;  // next layer's FiLM weights^T [dy,4dx] (null: none)
  float* film;            // [B,4dx] destination of the FiLM vectors
  int B, dy, ffy, kcat, dx;
  int update;             // 0: only the FiLM vectors of the current y
  int film_n;             // 0: 4 dx FiLM outputs with +1 on the "mul" quarters (2D block); > 0: film_n raw outputs
  // Optional (update = 1): the workgroup takes the pooled statistics of its graph itself instead of reading ycat / etoy
  // (pool_stats.h; the forward on one stream: nothing runs beside the y round that could have produced them earlier)
  const float* pool_X;    // [B,n,dx] the layer's input X, or null: Xtoy from ycat
  const float* pool_E;    // [B,n,n,de] the layer's input E (hidden, unsymmetrised), or null: Etoy from etoy
  const uint8_t* mask;    // [B,n]
  int n, de;
  // Position-aware block: the graph merge of block3d (Xtoy inputs of the unmasked newX, Etoy inputs of Y and dist1 from the
  // row aggregates) taken by the y workgroup of the graph itself; kcat = dy + 4 dx + 8 de.  null: from ycat / etoy.
  const float* gm_rowagg;   // [B*n, 8 de]
  const float* gm_rowaggD;  // [B*n, 4 de]
  const float* gm_newX;     // [B,n,dx]
  int pair;                 // 1: two graphs per workgroup (y_layer_kernel<2,512>; the 2D engine sets it for every y launch so that
                            // its schedule switches stay bit-identical to one another); 0: one (the position-aware engine)
};

bool y_layer_supported(int dy, int ffy, int kcat);
int launch_y_layer(const YLayerArgs& a, hipStream_t st);

struct NarrowLinearArgs { // row-list Linear with a narrow input (K <= 64), torch-layout weights
  const float* A;         // [M,K] lda
  const float* W;         // [N,K]
  const float* bias;      // [N]
  float* C;               // [M,N] ldc
  const int* rows;
  const int* row_count;
  int N, K, lda, ldc, relu;
};
int launch_narrow_linear(const NarrowLinearArgs& a, int M, hipStream_t st);
int launch_zero_masked_rows(float* X, const uint8_t* mask, int M, int D, hipStream_t st);

struct NodeStatsArgs {    // Xtoy pooling: z[b] = [mean|min|max|var] over valid nodes
  const float* X;         // [B,n,dx]
  const uint8_t* mask;
  float* z;
  int B, n, dx, z_stride, z_offset;
};
int launch_node_stats(const NodeStatsArgs& a, hipStream_t st);

struct PackArgs {
  const float* W;         // [N,K]
  f32x4_t* Wp;
  int N, K, NB, KB;
};
int launch_pack_linear(const PackArgs& a, hipStream_t st);
// three-term bf16 image of W for node_layer_x3_kernel: [2 NB][K/32][term][64 lanes] x 16 bytes (Wp holds NB*(K/16)*192 slots)
int launch_pack_linear_x3(const PackArgs& a, hipStream_t st);
int launch_pad_vec(const float* src, float* dst, int n, int n_pad, float add, hipStream_t st);
int launch_copy_cols(const float* src, int lds, float* dst, int ldd, int M, int ncols, hipStream_t st);
int launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t st);   // dst[c][r] = src[r][c]
int launch_row_head(const float* hid, int ldh, const float* W, const float* b, const float* X_in, int dX_in, const int* rows,
                    const int* row_count, float* out, int M, int N, int K, hipStream_t st);   // -1: shape not covered
int launch_x_out_residual(const float* mlp, const float* X_in, int dX_in, const uint8_t* mask, float* out,
                          int M, int nout, hipStream_t st);

// padded combinatorial Laplacian for any n (adjacency_tiled.hip); scgg_laplacian uses it beyond the 32-node LDS kernel
int launch_laplacian_rows(const int8_t* labels, const uint8_t* mask, int B, int n, float* lap_out, hipStream_t st);

// ---- position-aware block of Graph3DTransformer (block3d.hip) ---------------------------------------------
struct Block3DPackArgs {  // raw (torch layout) weights of one 3D layer's edge path -> LDS image of block3d_edge_kernel
  const float *W_in, *b_in, *W_dadd, *b_dadd, *W_dmul, *b_dmul, *W_eout, *b_eout;   // in_E, dist_add_e, dist_mul_e, e_out [de,de]
  const float *W_ff1, *b_ff1, *W_ff2, *b_ff2;                                       // linE1 [ffE,de], linE2 [de,ffE]
  const float *W_eatt, *b_eatt, *W_patt, *b_patt;                                   // e_att_mul, pos_att_mul [n_head,de]
  const float* W_a;                                                                 // a [n_head,dx], no bias
  const float *ln1g, *ln1b, *ln2g, *ln2b;                                           // normE1, normE2
  const float *W_dist, *b_dist, *W_n1, *b_n1, *W_n2, *b_n2;                         // lin_dist1 [de,2], lin_norm_pos1/2 [de,1]
  f32x4_t* img;
  int n_head;
};
struct Block3DArgs {
  const float* E_in;      // [B,n,n,de] unsymmetrised producer output
  float* E_out;
  const uint8_t* mask;
  const float* nodelin;   // [B*n, ldn] masked rows: q | k | v | x_e_mul1 | x_e_mul2
  int ldn;
  const float* film;      // [B, ldf]: y_e_add | y_e_mul | y_x_add | y_x_mul
  int ldf;
  const float* pos4;      // [B*n, 4]: masked, re-centred positions of this layer and their norms
  const f32x4_t* img;
  float* attn_part;       // [B*n*nch][n_head*dx + 32]: sum p V per head | max (16) | sum p (16)
  float* stat_part;       // [B*n*nch][8*de], first half used: mean | M2 | min | max of Y over the unit's valid keys
  const int* units;
  const int* unit_count;
  int B, n, n_head, need_attn;   // need_attn: bit 0 = attention partials, bit 1 = statistics partials
  float inv_sqrt_de;
};
struct RowMerge3DArgs {
  const float* attn_part;
  const float* stat_part;
  const int* chunk_cnt;   // [B, nch] valid keys per chunk
  const float* film;
  int ldf;
  float* wvcat;           // [B*n, n_head*dx]
  float* rowstat;         // [B*n, 8*de], first half used: EtoX input of E1 ([mean|min|max|var])
  float* rowagg;          // [B*n, 8*de], first half used: (mean, M2, min, max) of Y over the row's valid keys
  const int* rows;
  const int* row_count;
  int n, n_head, dx, de;
};
struct GraphMerge3DArgs {
  const float* rowagg;    // [B*n, 8*de], first half used: (mean, M2, min, max) of Y
  const float* rowaggD;   // [B*n, 4*de]: the same of dist1 (block3d_dist_stats_kernel, this layer's slice)
  const uint8_t* mask;
  const float* newX;      // [B,n,dx], padded rows included
  float* zcat;            // [B, ldz]: columns [dy, dy+4dx) <- Xtoy inputs
  float* etoy;            // [B, 8de]: Etoy inputs of Y | of dist1
  int ldz, n, dx, de, dy;
};
long long block3d_img_f4(int dx, int de, int ffe);      // -1 if unsupported
int launch_block3d_pack(const Block3DPackArgs& a, int dx, int de, int ffe, hipStream_t st);
int launch_block3d_edge(const Block3DArgs& a, int dx, int de, int ffe, hipStream_t st);
int launch_block3d_row_merge(const RowMerge3DArgs& a, int M, hipStream_t st);
int launch_block3d_newx(const float* outv, const float* ex, const float* px, const uint8_t* mask, const float* film, int ldf,
                        int film_off, float* newX, int M, int n, int dx, hipStream_t st);
int launch_block3d_graph_merge(const GraphMerge3DArgs& a, int B, hipStream_t st);
int launch_block3d_pos(const float* pos, const uint8_t* mask, int B, int n, int n_layers, float* pos4, hipStream_t st);
int launch_block3d_dist_stats(const float* pos4, const uint8_t* mask, const float* dvec, const int* rows, const int* row_count,
                              float* rowstatD, float* rowaggD, int B, int n, int de, int n_layers, hipStream_t st);
int launch_block3d_chunk_count(const uint8_t* mask, int B, int n, int* cnt, hipStream_t st);

}  // namespace scgg
