/* scgg.h -- C ABI of the MI355X (gfx950) discrete edge-diffusion hot path.
 *
 * The reference (chinmay5/semantically_consistent_graph_generation) is pure Python and has no FFI layer; its
 * boundary for this path is four Python call sites in src/ConStruct/diffusion_model_discrete_3d.py
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it replaces.  The Python modules
 * of semantically_consistent_graph_generation_amd/ bind these symbols with ctypes (INTEGRATION.md shows the stub)
 * and keep the reference's class / method names on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP, current device) unless its name ends in _host;
 *   - tensors are dense row-major fp32 with the reference's axis order: X[B,n,d], E[B,n,n,d], y[B,d];
 *   - node_mask is uint8 [B,n] (torch.bool storage);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work and never
 *     synchronise; after scgg_denoiser_finalize none of the compute calls allocates;
 *   - return value 0 = ok, negative = error; scgg_last_error() gives the message of the calling thread's last
 *     failure.  Nothing here falls back to a CPU path.
 */
#ifndef SCGG_H
#define SCGG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int scgg_version(void);
const char* scgg_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Denoiser: GraphTransformer  (src/ConStruct/models/transformer_model.py:251-389; layers :15-248;
 * pooling src/ConStruct/models/layers.py:5-51; masking src/ConStruct/utils.py:134-167).
 * Constructed at diffusion_model_discrete_3d.py:111-132 and called at :201.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct scgg_denoiser scgg_denoiser;

typedef struct scgg_denoiser_config {
  int32_t n_layers;
  int32_t in_X, in_E, in_y;            /* input_dims.X (+charges), .E, .y                      (:282,:291,:299) */
  int32_t out_X, out_E, out_y;         /* output_dims.X + output_dims.charges, .E, .y          (:326,:333,:341) */
  int32_t hid_X, hid_E, hid_y;         /* hidden_mlp_dims                                                    */
  int32_t dx, de, dy, n_head;          /* hidden_dims                                                        */
  int32_t dim_ffX, dim_ffE, dim_ffy;   /* dim_ffy: the layer's real lin_y1 width (2048 in the reference, :34) */
} scgg_denoiser_config;

/* replaces GraphTransformer.__init__ (:257-342).  Returns NULL on unsupported dims (scgg_last_error says why). */
scgg_denoiser* scgg_denoiser_create(const scgg_denoiser_config* cfg);
void scgg_denoiser_destroy(scgg_denoiser* h);

/* Borrow one parameter tensor by its reference state_dict name ("tf_layers.0.self_attn.q.weight", ...;
 * SURVEY.md section 8b lists them).  The pointer must stay valid; the data is read at finalize / refresh time. */
int scgg_denoiser_set_param(scgg_denoiser* h, const char* name, const float* data, int64_t numel);

/* Checks that every parameter is present, allocates the packed-weight arena (the only allocation) and packs.
 * scgg_denoiser_refresh re-packs from the same borrowed pointers after the caller changed the weights. */
int scgg_denoiser_finalize(scgg_denoiser* h, void* stream);
int scgg_denoiser_refresh(scgg_denoiser* h, void* stream);

size_t scgg_denoiser_workspace_bytes(const scgg_denoiser* h, int32_t B, int32_t n);

/* replaces GraphTransformer.forward (:344-389), eval mode.
 *   X[B,n,in_X] E[B,n,n,in_E] y[B,in_y] node_mask[B,n]
 *   out_X[B,n,out_X] (X and charges concatenated, masked), out_E[B,n,n,out_E] (symmetric, zero diagonal, masked),
 *   out_y[B,out_y] (may be NULL when out_y == 0).  Inputs are not modified.  workspace: >= workspace_bytes(B,n),
 *   256-byte aligned, contents irrelevant. */
int scgg_denoiser_forward(scgg_denoiser* h, const float* X, const float* E, const float* y, const uint8_t* node_mask,
                          int32_t B, int32_t n, float* out_X, float* out_E, float* out_y, void* workspace,
                          size_t workspace_bytes, void* stream);

/* The same forward for callers that cannot allocate and have no workspace to pass (tight serving loops, harnesses that
 * forbid hipMalloc, code that prepares a HIP graph -- the call itself is NOT capture-safe: it records an event on the
 * engine's own second stream, which a stream under capture may not wait for; run it outside the capture).  No counterpart in the reference.  Scratch = a file-scope
 * `static __device__` array of scgg_denoiser_static_scratch_bytes() = 32 MiB per GPU (held until the process ends; its
 * per-device address is looked up with hipGetSymbolAddress in scgg_denoiser_finalize, else on the first call on a device);
 * shapes whose workspace_bytes(B, n) exceed it are rejected with an error, nothing is allocated or freed by the call.
 * Every call on one GPU shares that GPU's array: calls that overlap on one GPU (two streams, two handles) race; calls
 * ordered on one stream do not.  The buffer keeps the previous call's contents, which no kernel of the forward reads
 * before writing (same contract as `workspace` above: contents irrelevant).  Results are bit-identical to
 * scgg_denoiser_forward. */
int scgg_denoiser_forward_static(scgg_denoiser* h, const float* X, const float* E, const float* y,
                                 const uint8_t* node_mask, int32_t B, int32_t n, float* out_X, float* out_E,
                                 float* out_y, void* stream);
size_t scgg_denoiser_static_scratch_bytes(void);

/* The engine's second stream (a hipStream_t, owned by the handle; NULL + scgg_last_error on failure).  The forward itself
 * runs entirely on the caller's stream; this stream only carries what the CALLER enqueues on it.  The forward joins it
 * (event wait, no host sync) in front of the first kernel that reads its X or y arguments -- behind the work lists, the
 * zero fills and the edge input MLP, which need E and node_mask only -- so work that PRODUCES X or y may be enqueued here
 * before scgg_denoiser_forward is called (after making this stream wait for whatever that work reads): it then runs beside
 * the edge input MLP, the long pole of the prologue.  The sampling loop (diffusion_model_discrete.py:653-663) runs the
 * eigen features, which feed only X and y (extra_features.py:41-60), this way.  E and node_mask must be complete on the
 * forward's own stream as usual.  No counterpart in the reference (PyTorch serialises the whole step on one stream). */
void* scgg_denoiser_side_stream(scgg_denoiser* h);

/* Debug / test tap: runs the input MLPs and the first `n_run_layers` layers of the same network and returns the
 * hidden state as the reference holds it after that layer (transformer_model.py:360-369): hid_X[B,n,dx],
 * hid_E[B,n,n,de] (symmetrised, masked), hid_y[B,dy].  Any output pointer may be NULL. */
int scgg_denoiser_forward_hidden(scgg_denoiser* h, const float* X, const float* E, const float* y,
                                 const uint8_t* node_mask, int32_t B, int32_t n, int32_t n_run_layers, float* hid_X,
                                 float* hid_E, float* hid_y, void* workspace, size_t workspace_bytes, void* stream);

/* Measurement aid (no counterpart in the reference): time the dominant kernel (the fused edge-layer kernel) with HIP
 * events recorded on the launch stream around its launches in every `every_n_forwards`-th forward (0 = off).  At most
 * max_launches launches are timed between two reads.  scgg_denoiser_profile allocates the events (call it outside
 * the timed region); scgg_denoiser_profile_read needs the stream synchronised and returns count and summed ms. */
int scgg_denoiser_profile(scgg_denoiser* h, int32_t every_n_forwards, int32_t max_launches);
int scgg_denoiser_profile_read(scgg_denoiser* h, int32_t* n_launches, double* total_ms);

/* ------------------------------------------------------------------------------------------------------------
 * Position-aware denoiser: Graph3DTransformer (src/ConStruct/models/pos_transformer_model.py:338-493; block :33-209,
 * layer :212-303; EtoX src/ConStruct/models/layers.py:55-71) and, with `equivariant` set, EquivariantGraph3DTransformer
 * (src/ConStruct/models/equivariant_pos_transformer_model.py:334-486: same block, node features from mlp_in_X and the
 * node head on the hidden X).  Constructed at diffusion_model_discrete_3d.py:111-132 and called at :201; it is the model
 * the shipped driver instantiates.  Same life cycle as scgg_denoiser: create, set_param per state_dict name, finalize.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct scgg_denoiser3d scgg_denoiser3d;

typedef struct scgg_denoiser3d_config {
  int32_t n_layers;
  int32_t in_X, in_E, in_y;            /* input_dims.X + input_dims.charges, .E, .y                                */
  int32_t out_X, out_E;                /* output_dims.X + output_dims.charges, .E  (output_dims.y must be 0: neither
                                          class has an mlp_out_y, pos_transformer_model.py:479-481)                */
  int32_t hid_X, hid_E, hid_y;         /* hidden_mlp_dims                                                          */
  int32_t dx, de, dy, n_head;          /* hidden_dims                                                              */
  int32_t dim_ffX, dim_ffE, dim_ffy;   /* dim_ffy: the layer's real lin_y1 width (2048: :223-225 default, not passed) */
  int32_t equivariant;                 /* 0: Graph3DTransformer, 1: EquivariantGraph3DTransformer                  */
} scgg_denoiser3d_config;

scgg_denoiser3d* scgg_denoiser3d_create(const scgg_denoiser3d_config* cfg);   /* NULL on unsupported dims */
void scgg_denoiser3d_destroy(scgg_denoiser3d* h);
int scgg_denoiser3d_set_param(scgg_denoiser3d* h, const char* name, const float* data, int64_t numel);
int scgg_denoiser3d_finalize(scgg_denoiser3d* h, void* stream);   /* also re-packs after the weights changed */
size_t scgg_denoiser3d_workspace_bytes(const scgg_denoiser3d* h, int32_t B, int32_t n);

/* replaces Graph3DTransformer.forward (:433-493) / EquivariantGraph3DTransformer.forward (:428-486), eval mode.
 *   X[B,n,in_X] E[B,n,n,in_E] y[B,in_y] pos[B,n,3] (raw: masking and the re-centring of PlaceHolder.mask,
 *   utils.py:160-164, once on entry and once after every layer, happen inside) node_mask[B,n]
 *   out_X[B,n,out_X] (X and charges, masked; may be NULL when out_X == 0), out_E[B,n,n,out_E] (symmetric, zero
 *   diagonal, masked), out_radius[B,n,n] (same; NULL = the radius head is not run, as when model_input.radius is None).
 *   Distances are taken from coordinate differences at every n (the reference's torch.cdist switches to the expanded
 *   form above 25 points, where the distance of a node to itself is rounding noise). */
int scgg_denoiser3d_forward(scgg_denoiser3d* h, const float* X, const float* E, const float* y, const float* pos,
                            const uint8_t* node_mask, int32_t B, int32_t n, float* out_X, float* out_E,
                            float* out_radius, void* workspace, size_t workspace_bytes, void* stream);

/* Debug / test tap: hidden state after the input MLPs and the first n_run_layers layers, as the reference holds it:
 * hid_X[B,n,dx], hid_E[B,n,n,de] (symmetrised, masked), hid_y[B,dy].  Any output pointer may be NULL. */
int scgg_denoiser3d_forward_hidden(scgg_denoiser3d* h, const float* X, const float* E, const float* y, const float* pos,
                                   const uint8_t* node_mask, int32_t B, int32_t n, int32_t n_run_layers, float* hid_X,
                                   float* hid_E, float* hid_y, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Node-coordinate DDPM: the SiT-T denoiser (src/bootstrap/node_coordinate_diffusion/sit_model.py:157-247, blocks :112-134,
 * AttentionWithPads :80-103, TimestepEmbedder :41-77) and the ancestral sampler NodeCoordsDiffusionModel.sample
 * (cow_diff_nodes_with_3d_checks.py:63-81, schedule :22-46), which sample_limit_dist calls once per sampled batch
 * (src/ConStruct/diffusion/noise_model.py:323-325).  Same life cycle as scgg_denoiser: create, set_param per state_dict
 * name ("blocks.0.attn.qkv.weight", ...), finalize (also builds the per-timestep modulation tables: the only allocation).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct scgg_sit scgg_sit;
typedef struct scgg_sit_config {
  int32_t in_channels;                 /* 3 */
  int32_t hidden, n_blocks, n_heads;   /* SiT-T: 192, 6, 6 (sit_model.py:271) */
  int32_t mlp_hidden;                  /* int(hidden * mlp_ratio) = 768 */
  int32_t noise_steps;                 /* 1000 */
  float beta_start, beta_end;          /* 1e-4, 0.02: torch.linspace schedule (cow_diff_nodes_with_3d_checks.py:22-46) */
} scgg_sit_config;

scgg_sit* scgg_sit_create(const scgg_sit_config* cfg);          /* NULL on unsupported dims */
void scgg_sit_destroy(scgg_sit* h);
int scgg_sit_set_param(scgg_sit* h, const char* name, const float* data, int64_t numel);
int scgg_sit_finalize(scgg_sit* h, void* stream);
size_t scgg_sit_workspace_bytes(const scgg_sit* h, int32_t B, int32_t n);

/* replaces SiT.forward(x, t, padding_mask) (sit_model.py:228-247): x[B,n,3], t[B] int32 timesteps in [0, noise_steps),
 * node_mask[B,n] -> eps_out[B,n,3] on the valid rows, 0 on padded ones (the reference lets padded tokens attend uniformly to
 * every slot; nothing downstream reads them). */
int scgg_sit_forward(scgg_sit* h, const float* x, const int32_t* t, const uint8_t* node_mask, int32_t B, int32_t n,
                     float* eps_out, void* workspace, size_t workspace_bytes, void* stream);

/* replaces NodeCoordsDiffusionModel.sample(pos, node_mask) (:63-81): starts from x_T[B,n,3] (the reference draws it with
 * torch.randn inside; here the caller passes it) and runs i = noise_steps-1 .. 1 (`steps` >= 0: only the first `steps`
 * iterations).  noise: [iterations][B,n,3] normal draws in iteration order (tests), or NULL: Philox4x32-10 + Box-Muller
 * keyed by (seed, graph_id_base + b, node, i).  x_out[B,n,3]: valid rows, zero on padded slots. */
int scgg_sit_sample(scgg_sit* h, const float* x_T, const uint8_t* node_mask, int32_t B, int32_t n, int32_t steps,
                    const float* noise, uint64_t seed, uint32_t graph_id_base, float* x_out, void* workspace,
                    size_t workspace_bytes, void* stream);

/* replaces the torch.randn start coordinates of sample_limit_dist (noise_model.py:313-325: `pos = torch.randn(...)`, and the
 * randn x_T inside NodeCoordsDiffusionModel.sample, cow_diff_nodes_with_3d_checks.py:66): out[B,n,3] standard normals keyed
 * by (seed, graph_id_base + b, node, step word) -- the GLOBAL graph id, so a graph's start does not depend on the shard it
 * is sampled in.  `step`: a word no iteration of the chain uses (the SiT sampler uses 1 .. noise_steps-1). */
int scgg_philox_normal_rows(float* out, int32_t B, int32_t n, uint64_t seed, uint32_t graph_id_base, uint32_t step, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Reverse step: NoiseModel.sample_zs_from_zt_and_pred (src/ConStruct/diffusion/noise_model.py:342-447) with
 * sample_discrete_features (src/ConStruct/diffusion/diffusion_utils.py:200-285).  Called at
 * diffusion_model_discrete_3d.py:749.
 *
 * Edge state is carried as int8 labels [B,n,n]: class index, or -1 where the reference's one-hot row is the zero
 * vector (diagonal and padded pairs, utils.py:154).  Randomness: either explicit uniforms (u0[B,n,n] for the single
 * draw, uk[B,n,n,top_k] for the top-k draw) or, when u0 == NULL, Philox4x32-10 keyed by (seed; i, j,
 * graph_id_base + b, step).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct scgg_rng {
  const float* u0;
  const float* uk;
  uint64_t seed;
  uint32_t step;
  uint32_t graph_id_base;
} scgg_rng;

/* E branch of sample_zs_from_zt_and_pred for every transition of the form Q = a I + (1-a) 1 m^T
 * (uniform, marginal, absorbing, absorbing_edges):  beta_tab / abar_tab are the [T+1] fp32 tables of the E branch
 * (get_Qt :101-125 / :524-537, get_Qt_bar :127-154 / :539-552), marginals = limit distribution m[C].
 * t_int / s_int are int32 [B].  prob_out (optional) receives prob_E as handed to the sampler. */
int scgg_reverse_step_edges(const float* logits_E, const int8_t* labels_t, const uint8_t* node_mask,
                            const int32_t* t_int, const int32_t* s_int, const float* beta_tab, const float* abar_tab,
                            const float* marginals, int32_t B, int32_t n, int32_t C, int32_t top_k,
                            const scgg_rng* rng, int8_t* labels_s, int8_t* cands_s, float* prob_out, void* stream);

/* E branch of sample_discrete_features alone (probabilities given): used by apply_noise (noise_model.py:207)
 * and by the bit-exactness tests.  top_k == 0 -> no candidates (diffusion_utils.py:275-277). */
int scgg_sample_edges_from_probs(const float* prob_E, const uint8_t* node_mask, int32_t B, int32_t n, int32_t C,
                                 int32_t top_k, const scgg_rng* rng, int8_t* labels, int8_t* cands, void* stream);

/* conversions between the int8 state and the reference's tensors: one-hot float E (zero rows where label < 0) and
 * int64 extra_info (noise_model.py:412,:445; utils.py:154,:166) */
int scgg_labels_to_onehot(const int8_t* labels, const int8_t* cands, int64_t n_pairs, int32_t C, int32_t top_k,
                          float* E_onehot, int64_t* extra_info, void* stream);
int scgg_onehot_to_labels(const float* E_onehot, int64_t n_pairs, int32_t C, int8_t* labels, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Extra features: the AdjacencyFeatures part of ExtraFeatures.__call__
 * (src/ConStruct/diffusion/extra_features.py:60-77, :198-385) plus the Laplacian that EigenFeatures hands to
 * eigh (:88-94).  Called at diffusion_model_discrete_3d.py:186.  Works on the int8 edge state.
 *   labels int8 [B,n,n] (adjacency = label > 0), node_labels int8 [B,n] or NULL (single node class), node_mask.
 *   x_out: node (b,i) gets its 3 features (3/4/5-cycle counts, clamp(.,0,5)/5, masked) at x_out[(b*n+i)*x_ld + x_off ..]
 *   y_out: graph b gets 4 + (num_degree+2) + Cx + C features (cycles per graph | degree hist | node hist | edge
 *          hist) at y_out[b*y_ld + y_off ..]
 *   e_out: pair p gets its max_degree+1 features (reachability A^1..A^max_degree, Adamic-Adar index) at
 *          e_out[p*e_ld + e_off ..]; with onehot_out != NULL the masked one-hot of the label goes to
 *          onehot_out[p*e_ld + 0..C)  (so the denoiser's inputs [state | extra] are assembled in one pass)
 *   lap_out[B,n,n] (optional)                         padded combinatorial Laplacian
 * scgg_adjacency_features: one workgroup per graph, everything in LDS, n <= 32, no workspace.
 * scgg_adjacency_features_tiled: the same outputs for any n <= 4096 (and for n <= 32 too): the K-1 adjacency products
 *   (calculate_kpowers :256-268) and the w w^T product of local_neighbors (:348) run as batched fp32 MFMA GEMMs over
 *   [B,np,np] planes (np = n rounded up to 64) in `workspace` (>= scgg_adjacency_features_workspace_bytes(B, n), 256-byte
 *   aligned, contents irrelevant).  Integer-valued outputs (cycle counts, reachability, histograms, Laplacian) are
 *   identical to the LDS kernel's; the Adamic-Adar channel differs by fp32 summation order only.  onehot_out must be NULL
 *   or e_out.
 * ---------------------------------------------------------------------------------------------------------- */
int scgg_adjacency_features(const int8_t* labels, const int8_t* node_labels, const uint8_t* node_mask, int32_t B,
                            int32_t n, int32_t C, int32_t Cx, int32_t num_degree, int32_t max_degree,
                            int32_t cycle_features, float* x_out, int32_t x_ld, int32_t x_off, float* y_out,
                            int32_t y_ld, int32_t y_off, float* e_out, int32_t e_ld, int32_t e_off, float* onehot_out,
                            float* lap_out, void* stream);
size_t scgg_adjacency_features_workspace_bytes(int32_t B, int32_t n);
int scgg_adjacency_features_tiled(const int8_t* labels, const int8_t* node_labels, const uint8_t* node_mask, int32_t B,
                                  int32_t n, int32_t C, int32_t Cx, int32_t num_degree, int32_t max_degree,
                                  int32_t cycle_features, float* x_out, int32_t x_ld, int32_t x_off, float* y_out,
                                  int32_t y_ld, int32_t y_off, float* e_out, int32_t e_ld, int32_t e_off,
                                  float* onehot_out, float* lap_out, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* The padded combinatorial Laplacian alone: replaces EigenFeatures.compute_laplacian
 * (src/ConStruct/diffusion/extra_features.py:121-127 with the padding of :88-94), bit-identical to the lap_out of
 * scgg_adjacency_features.  Separate so that scgg_eigen_features can start (on another stream) while the remaining
 * structural features are computed.  labels int8 [B,n,n] (> 0 = edge), lap_out [B,n,n].  Any n (one workgroup per
 * graph up to 32 nodes, one wave per row beyond). */
int scgg_laplacian(const int8_t* labels, const uint8_t* node_mask, int32_t B, int32_t n, float* lap_out, void* stream);

/* EigenFeatures.compute_features after the Laplacian (extra_features.py:96-112, :138-195), including the
 * eigendecomposition the reference delegates to torch.linalg.eigh (:96): batched Jacobi, one wave per graph.
 *   eval_feat: graph b gets [#connected components | first num_eigenvalues non-zero eigenvalues / n_valid (padded
 *              with 2)] at eval_feat[b*ld_ev + off_ev ..]
 *   evec_feat: node (b,i) gets [not-in-largest-component indicator | first num_eigenvectors eigenvectors] at
 *              evec_feat[(b*n+i)*ld_vec + off_vec ..]
 *   w_out[B,n] ascending eigenvalues, v_out[B,n,n] eigenvectors in columns (both optional, as torch.linalg.eigh).
 * Eigenvectors inside a degenerate eigenspace are basis-dependent in the reference too; see DESIGN.md.  n <= 32. */
int scgg_eigen_features(const float* laplacian, const uint8_t* node_mask, int32_t B, int32_t n,
                        int32_t num_eigenvalues, int32_t num_eigenvectors, float* eval_feat, int32_t ld_ev,
                        int32_t off_ev, float* evec_feat, int32_t ld_vec, int32_t off_vec, float* w_out, float* v_out,
                        void* stream);

/* The same features for graphs of up to 256 nodes (vessel graphs), where the reference's torch.linalg.eigh (:96) is the
 * slowest call of a step: Householder tridiagonalisation in LDS, all eigenvalues by Sturm-count multisection, and ONLY
 * the eigenvector columns the features read (column 0 and columns ncc .. ncc + num_eigenvectors - 1, :159-195) by inverse
 * iteration.  One 1024-thread workgroup per graph.  eval_feat / evec_feat as above; w_out[B,n] ascending eigenvalues;
 * vsel_out[B,n,1+num_eigenvectors] the selected eigenvector columns (rows of masked nodes zero) before the reference's
 * column mask (both optional; the whole spectrum is only computed when w_out is given).  n <= 256, num_eigenvalues <= 15,
 * num_eigenvectors <= 7. */
int scgg_eigen_features_tridiag(const float* laplacian, const uint8_t* node_mask, int32_t B, int32_t n,
                                int32_t num_eigenvalues, int32_t num_eigenvectors, float* eval_feat, int32_t ld_ev,
                                int32_t off_ev, float* evec_feat, int32_t ld_vec, int32_t off_vec, float* w_out,
                                float* vsel_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Validity projector of the reverse loop (src/ConStruct/projector/projector_3d_utils.py; created at
 * diffusion_model_discrete_3d.py:636-650 and called once per reverse step at :661).
 *
 * Device half.  scgg_new_edges replaces the reference's `(argmax(z_s.E) - z_t_adj).nonzero()` and its per-graph
 * full-tensor copies to the host (projector_3d_utils.py:259-279): labels_s / labels_t are the int8 states [B,n,n]
 * (-1 on padding and diagonal), cands int8 [B,n,n,top_k] (NULL with top_k = 0).  An edge is new where
 * labels_s > labels_t (binary != 0: where it is > 0 and labels_t is not), upper triangle only.  offsets [B+1] int32
 * receives the exclusive prefix of the per-graph counts (offsets[B] = total m); rec receives m records of
 * 4 + top_k int32 (graph, i, j, label, candidates...) in (graph, i, j) order -- the order of Tensor.nonzero() --
 * at most `cap` records are written; status[0] is cleared and then gets bit 0 set when an edge of labels_t is missing
 * or lowered in labels_s (the reference asserts on that, :261).
 * scgg_apply_edge_edits writes the host's decisions back: edits int32 [n_edits,4] = (graph, i, j, class), both
 * (i,j) and (j,i) are set (the reference's in-place z_s.E[...] = one_hot writes, :288-293, :323-324, :340-355). */
int scgg_new_edges(const int8_t* labels_s, const int8_t* labels_t, const int8_t* cands, int32_t B, int32_t n, int32_t top_k,
                   int32_t binary, int32_t* offsets, int32_t* rec, int64_t cap, int32_t* status, void* stream);
int scgg_apply_edge_edits(int8_t* labels, const int32_t* edits, int64_t n_edits, int32_t B, int32_t n, void* stream);

/* Host half (all pointers are HOST pointers; no GPU work).  One object per batch, as the reference's projector
 * objects: it owns every sample's graph and its blocked-edge table.
 *   kind: SCGG_PROJ_COW = CoWProjector (:213-387), SCGG_PROJ_COW_VANILLA = CoWVanillaProjector (:464-598),
 *         SCGG_PROJ_ATM = ATMTreeProjector (:662-840), SCGG_PROJ_ATM_VANILLA = ATMVanillaTreeProjector (:917-1052),
 *         SCGG_PROJ_TREE = TreeProjector (:157-164 over AbstractProjector :16-147).
 * A reverse step is begin + finish because the reference shuffles the new edges of every graph that falls back to
 * one-by-one insertion with np.random.shuffle (:330, :555, :112), i.e. from a random stream the CALLER owns:
 *   scgg_projector_begin   rec = the m records of scgg_new_edges (copied to the host).  Deletes edges blocked in
 *                          earlier steps, tries all new edges of a graph at once, and lists in fb_graph / fb_count
 *                          (capacity B each) the graphs that need the fallback and how many edges each has;
 *   scgg_projector_finish  perm = for each listed graph, in order, a permutation of 0..fb_count-1 (what
 *                          np.random.shuffle(arange(fb_count)) gives; position k takes edge perm[k] but keeps the
 *                          k-th label and candidates, as the reference does), n_perm = sum of fb_count.  Writes the
 *                          step's edits (graph, i, j, class; class 0 = edge deleted; one entry per pair, the last
 *                          write of the reference's sequence) to edits_out (capacity edits_cap rows; m is enough).
 * scgg_projector_export copies graph b's class matrix [n_b,n_b] (the reference's nx_graphs_list[b]) and its
 * blocked table [n_b,n_b] (blocked_edges[b], upper triangle); scgg_projector_counter is
 * top_cow_validity_check_and_fix_obj.counter (:411-413). */
typedef struct scgg_projector scgg_projector;
enum { SCGG_PROJ_COW = 0, SCGG_PROJ_COW_VANILLA = 1, SCGG_PROJ_ATM = 2, SCGG_PROJ_ATM_VANILLA = 3, SCGG_PROJ_TREE = 4 };
int scgg_projector_create(int32_t kind, int32_t B, const int32_t* n_nodes_host, scgg_projector** out);
void scgg_projector_destroy(scgg_projector* p);
int scgg_projector_begin(scgg_projector* p, int64_t m, const int32_t* rec_host, int32_t top_k, int32_t* fb_graph_host,
                         int32_t* fb_count_host, int32_t* n_fb_host);
int scgg_projector_finish(scgg_projector* p, const int32_t* perm_host, int64_t n_perm, int32_t* edits_out_host,
                          int64_t edits_cap, int64_t* n_edits_host);
int scgg_projector_export(const scgg_projector* p, int32_t b, int8_t* graph_labels_host, uint8_t* blocked_host);
int64_t scgg_projector_counter(const scgg_projector* p);

/* ------------------------------------------------------------------------------------------------------------
 * Training pieces (row a11): the dropout of the train-mode forward and the loss of training_step
 * (src/ConStruct/diffusion_model_discrete_3d.py:215-239).
 *
 * scgg_dropout replaces nn.Dropout at the reference's dropout sites (transformer_model.py:88-105 layers, :284-340 in /
 * out MLPs).  The reference draws from torch's global generator; here the mask of `site` is a pure function of
 * (seed, site, flat index): Philox4x32-10, counter ((index >> 2) low, (index >> 2) high, site, 0xD0), key = seed, output
 * word (index & 3), kept iff u >= (float)p, out = x / (float)(1 - p) (p is a double so that 1 - p rounds as the reference's Python float does).  The backward pass calls it again on the incoming gradient (same seed and site).
 * x == out is allowed.
 *
 * scgg_masked_cross_entropy replaces CrossEntropyMetric (src/ConStruct/metrics/abstract_metrics.py:77-93) inside
 * TrainLoss.forward (src/ConStruct/metrics/train_metrics.py:25-85): rows of C logits, int8 target per row (< 0: row
 * masked out, as the reference's boolean row selection :40-47).  partial (optional, scgg_masked_cross_entropy_blocks(rows)
 * floats) receives per-block sums of the cross entropies (their sum / number of rows that take part = the reference's
 * loss); grad (optional, [rows, C]) receives (softmax - onehot) * grad_scale[0] on the rows that take part and 0
 * elsewhere (grad_scale is a device scalar: lambda / count * upstream gradient, so no host sync is needed).
 * ---------------------------------------------------------------------------------------------------------- */
int scgg_dropout(const float* x, float* out, int64_t numel, double p, uint64_t seed, uint32_t site, void* stream);
/* the same with the key (seed & 0xffffffff, seed >> 32) read from device memory when the kernel runs: the form a captured
 * HIP graph needs, whose kernel arguments are frozen at capture time */
/* out[b,i,j,:] = 0.5 (R[b,i,j,:] + R[b,j,i,:]) on pairs of valid nodes i != j, 0 elsewhere (transformer_model.py:108-112
 * followed by PlaceHolder.mask, utils.py:160-164, with the zero diagonal of :389): one pass over [B,n,n,C] fp32; out != R.
 * Linear and self-adjoint: the backward of the training path is the same call on the gradient. */
int scgg_sym_mask(const float* R, const uint8_t* mask, int32_t B, int32_t n, int32_t C, float* out, void* stream);
/* The global-feature (y) path of one XEyTransformerLayer in training mode (transformer_model.py:229-246 and :100-105): under
 * autograd eleven Linears on [B, dy] vectors with LayerNorms, ReLUs and three dropout sites, ~120 launches per layer and step;
 * here a forward kernel, a backward kernel (data gradients; recomputes the chain from the inputs) and ONE kernel for all
 * weight gradients.  hy [B,dy], xstats [B,dx4] (Xtoy, dx4 = 4 dx), estats [B,de4] (Etoy); weights row-major [out,in] as
 * nn.Linear holds them; out = the layer's new y.  Dropout: p, key (k0, k1) or key_dev, sites of y1 / y2-inner / y2-outer (the
 * stream of scgg_dropout on [B, width] tensors).  Backward: dout [B,dy] -> dhy, dxstats, destats and `save`
 * [B, scgg_train_y_save_width(dy, ffy)] for scgg_train_y_wgrad, which writes the 18 parameter gradients in the order of the
 * parameters below.  dy <= 256. */
typedef struct scgg_train_y_args {
  const float *hy, *xstats, *estats;
  const float *W_yy, *b_yy, *W_xy, *b_xy, *W_ey, *b_ey, *W_o0, *b_o0, *W_o2, *b_o2, *g1, *be1, *W_1, *b_1, *W_2, *b_2, *g2, *be2;
  float* out;
  const float* dout;
  float *dhy, *dxstats, *destats, *save;
  int32_t B, dy, dx4, de4, ffy;
  float p;
  uint32_t k0, k1, site1, site2, site3;
  const uint32_t* key_dev;
  float* fsave;   /* optional [B, scgg_train_y_keep_width(dy, ffy)]: the forward leaves its intermediates there and a backward
                     given the same buffer reads them instead of recomputing the chain */
} scgg_train_y_args;
int64_t scgg_train_y_save_width(int32_t dy, int32_t ffy);
int64_t scgg_train_y_keep_width(int32_t dy, int32_t ffy);
int scgg_train_y_forward(const scgg_train_y_args* a, void* stream);
int scgg_train_y_backward(const scgg_train_y_args* a, void* stream);
int scgg_train_y_wgrad(const scgg_train_y_args* a, float* const* grads, void* stream);
/* Xtoy's pooling (layers.py:12-21) under autograd, one launch each way: out [B, 4 dx] = mean | min | max | var exactly as the
 * reference forms them (sums over all n rows / number of valid nodes; +-1e5 on masked rows for the extrema); aux [B, 3 dx] and
 * arg [B, 2 dx] (rows of the extrema) are what the backward needs; dX [B,n,dx] from g [B, 4 dx]. */
int scgg_train_node_stats(const float* X, const uint8_t* mask, int32_t B, int32_t n, int32_t dx, float* out, float* aux, int32_t* arg,
                          void* stream);
int scgg_train_node_stats_backward(const float* X, const uint8_t* mask, int32_t B, int32_t n, int32_t dx, const float* g,
                                   const float* aux, const int32_t* arg, float* dX, void* stream);
int scgg_dropout_devkey(const float* x, float* out, int64_t numel, double p, const uint32_t* key_dev, uint32_t site, void* stream);
int64_t scgg_masked_cross_entropy_blocks(int64_t rows);
int scgg_masked_cross_entropy(const float* logits, const int8_t* target, int64_t rows, int32_t C, float* partial,
                              float* grad, const float* grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Training step, edge path of one XEyTransformerLayer as fused kernels with a hand-written backward (row a11):
 * replaces, for one layer, NodeEdgeBlock.forward's E part + attention (transformer_model.py:165-227) and the layer's E
 * path with its three dropout sites (:94-98), together with what autograd does for them in training_step
 * (diffusion_model_discrete_3d.py:215-239).  Nothing of size [B,n,n,dx] is kept between forward and backward: the
 * backward recomputes the chain per 16-key unit from the layer's inputs and the saved (lse, wV).
 *
 * All tensors fp32, contiguous.  E [B,n,n,de] (the layer's input: symmetric, masked); Q / K / V [B,n,dx] (masked rows);
 * yea / yem1 [B,dx] = y_e_add(y), y_e_mul(y) + 1; mask [B,n].  F_*: fragment images of the weights from
 * scgg_train_edge_pack (transpose = 0), F_*T of their transposes (transpose = 1; backward only).  Dropout: p, seed =
 * (k1 << 32 | k0), sites of newE / the FFN's hidden layer / its output (the stream of scgg_dropout); p = 0: none.
 * forward:  E_out [B,n,n,de] (zero-filled by the caller; written on pairs of valid nodes, NOT yet symmetrised or masked),
 *           part [B*n*nch, 3, dx] (scratch), wV_out / lse_out [B,n,dx] (attention output, log-sum-exp per channel).
 * backward: dOut = dL/dE_out, dwV = dL/dwV, wV / lse from the forward -> dE [B,n,n,de] and, per DENSE pair slot
 *           s = ((b n + i) nch + jc) 16 + (j & 15) (all slots written, zeros on non-pairs), the factors of the weight
 *           gradients: G_e [S,de], G_da [S,2dx] (d e_mul out | d e_add out), G_z [S,dx] (input of e_out), G_dnewE [S,de],
 *           G_h1 [S,de] (input of linE1), G_df [S,ffe], G_f [S,ffe] (input of linE2), G_do [S,de], G_dkv [S,2dx] (summands
 *           of dK | dV over i), upart [B*n*nch, 2dx + 4de] = dQ row partial (before 1/sqrt(df)) | sum dz*Y (d yem1) |
 *           d gamma2 | d beta2 | d gamma1 | d beta1.  The caller forms dW = G_g^T G_x (one GEMM each) and the sums.
 * Built for (de, dim_ffE) = (32, 64) and (64, 128), dx a multiple of 16.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct scgg_train_edge_args {
  const float *E, *Q, *K, *V, *yea, *yem1;
  const uint8_t* mask;
  const float *F_mul, *F_add, *F_out, *F_ff1, *F_ff2;
  const float *F_mulT, *F_addT, *F_outT, *F_ff1T, *F_ff2T;
  const float *b_mul, *b_add, *b_out, *g1, *b1, *b_ff1, *b_ff2, *g2, *b2;
  float *E_out, *part, *wV_out, *lse_out;
  const float *dOut, *dwV, *wV, *lse;
  float *dE, *G_e, *G_da, *G_z, *G_dnewE, *G_h1, *G_df, *G_f, *G_do, *G_dkv, *upart;
  /* backward, optional (all NULL: none): the gradient of Etoy's pooled statistics of E (scgg_train_edge_stats) is added to
   * dE: st_mean [B,de]; st_cm = g_mean / count; st_cv = 2 g_var / count; st_gmi / st_gma = g_min / g_max; st_imn / st_imx =
   * pair index i n + j of the extrema */
  const float *st_mean, *st_cm, *st_cv, *st_gmi, *st_gma;
  const int32_t *st_imn, *st_imx;
  int32_t B, n, nch, dx, de, ffe;
  int32_t b0;   /* the call covers graphs [b0, b0 + B) of a larger batch (all pointers already offset): only the dropout stream sees it */
  float inv_sqrt_df, p;
  uint32_t k0, k1, site1, site2, site3;
  const uint32_t* key_dev;   /* optional: (k0, k1) read from device memory when the kernel RUNS (a captured HIP graph then replays
                                with whatever key the caller has written there); NULL: k0 / k1 above */
} scgg_train_edge_args;
int scgg_train_edge_pack(const float* W, int32_t rows, int32_t cols, int32_t transpose, float* out, void* stream);
/* the same for up to ten matrices in one launch (a layer's five weights, plain and transposed) */
typedef struct scgg_train_pack_list {
  const float* W[10];
  float* out[10];
  int32_t rows[10], cols[10], transpose[10];
  int32_t count;
} scgg_train_pack_list;
int scgg_train_edge_pack_multi(const scgg_train_pack_list* l, void* stream);
int scgg_train_edge_forward(const scgg_train_edge_args* a, void* stream);
int scgg_train_edge_backward(const scgg_train_edge_args* a, void* stream);
/* the weight gradient of one Linear from the backward's factors (what autograd's `grad_out.T @ input` and
 * `grad_out.sum(0)` do for nn.Linear): G [S,M], X [S,N] -> partial rows r = 0 .. scgg_train_wgrad_slabs(S, slab) - 1 at
 * partial + r * ld_partial (0: M*N + M), each [M*N | M] = G^T X and the column sums of G over a slab of `slab` slots; the
 * caller adds the rows (every element of every row is written).  A row stride wider than M*N + M lets the partials of
 * several products share one buffer and one final sum.  M, N, slab multiples of 16. */
/* Etoy's pooling of the layer input for the y update (layers.py:34-41), one read of E: per graph b and slice s (rows i = s,
 * s + slices, ...) out[b][s] = [count | mean | M2 | min | max] x de and idx[b][s] = [argmin | argmax] x de (pair index i n + j);
 * scgg_train_edge_stats_finish merges the slices in index order (pairwise variance update; ties of an extremum: the smaller
 * pair index) into stats [B, 4 de] = mean | min | max | var (Etoy's output) and mean / count / imn / imx [B, de] (what
 * scgg_train_edge_backward takes as st_*). */
int scgg_train_edge_stats(const float* E, const uint8_t* mask, int32_t B, int32_t n, int32_t de, int32_t slices, float* out,
                          int32_t* idx, void* stream);
int scgg_train_edge_stats_finish(const float* out, const int32_t* idx, int32_t B, int32_t slices, int32_t de, float* stats,
                                 float* mean, float* count, int32_t* imn, int32_t* imx, void* stream);
int64_t scgg_train_wgrad_slabs(int64_t S, int32_t slab);
int scgg_train_wgrad(const float* G, int32_t M, const float* X, int32_t N, int64_t S, int32_t slab, float* partial,
                     int64_t ld_partial, void* stream);
/* Sum over the middle index of src [A,R,C] (fp32, C % 4 == 0): out[split * ld_split + a * C + c] = sum of the rows of slice
 * `split` (splits = scgg_train_sum_rows_splits(A, R, C) slices of ceil(R / splits) rows; ld_split 0: A * C; wider: the
 * call fills graphs [b0, b0 + A) of a [splits, B, C] buffer); the caller adds the few slices.  The
 * reductions autograd would run over the backward kernel's factor tensors (dK | dV over the query rows, per-graph sums). */
int64_t scgg_train_sum_rows_splits(int64_t A, int64_t R, int32_t C);
int scgg_train_sum_rows(const float* src, int64_t A, int64_t R, int32_t C, int32_t splits, float* out, int64_t ld_split,
                        void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Stress configuration (BASELINE.json configs[4]): NodeEdgeBlock.forward (transformer_model.py:165-248, without the y
 * update) on a bf16 edge tensor, de = 128 (or 64), dx = 256, n <= 1024.
 *   E_in / newE: bf16 [B,n,n,de] (raw 16-bit words, torch.bfloat16 storage).  Q / K / V: fp32 rows with stride ldqkv,
 *   already masked (:181-183).  film: fp32 [B, 4 dx] = [y_e_add | y_e_mul + 1 | y_x_add | y_x_mul + 1](y).
 *   image: scgg_edge_block_bf16_image_bytes(dx, de) bytes filled by scgg_edge_block_bf16_pack from e_mul / e_add / e_out
 *   (weights rounded to bf16, biases fp32).  units / unit_count: row * ceil(n/16) + chunk of every (valid row, 16-key chunk
 *   with a valid key) and their number (device scalar).  newE gets e_out(...) * m_i m_j on those rows (rows of padded nodes
 *   are not written: zero the buffer first); attn_part [B*n*ceil(n/16), 3, dx] gets the per-unit softmax partials, which
 *   scgg_attn_merge turns into newX_pre = y_x_add + (y_x_mul + 1) * weighted V for the rows listed in rows / row_count.
 * Arithmetic: one bf16 MFMA per product, fp32 accumulation: results agree with the fp32 reference to bf16 rounding of
 * E, the weights and the FiLM'ed Y (tests/test_gpu_stress_bf16.py states the bound).
 * ---------------------------------------------------------------------------------------------------------- */
size_t scgg_edge_block_bf16_image_bytes(int32_t dx, int32_t de);
int scgg_edge_block_bf16_pack(const float* W_mul, const float* b_mul, const float* W_add, const float* b_add,
                              const float* W_out, int32_t dx, int32_t de, void* image, void* stream);
int scgg_edge_block_bf16(const uint16_t* E_in, uint16_t* newE, const uint8_t* node_mask, const float* Q, const float* K,
                         const float* V, int32_t ldqkv, const float* film, const void* image, const float* b_out,
                         float* attn_part, const int32_t* units, const int32_t* unit_count, int32_t B, int32_t n,
                         int32_t dx, int32_t de, int32_t n_head, void* stream);
int scgg_attn_merge(const float* attn_part, const uint8_t* node_mask, const float* film, float* newX_pre,
                    const int32_t* rows, const int32_t* row_count, int32_t max_rows, int32_t n, int32_t dx, void* stream);

/* A Linear layer over a list of rows (torch.nn.functional.linear on the rows listed; the other rows of C are not written),
 * fp32 results from three-term bf16 products on the matrix pipe.  Replaces the q / k / v / x_out Linears of
 * NodeEdgeBlock.forward (transformer_model.py:178-183, :236-238) for callers outside the engines (stress.py).
 *   image: scgg_linear_rows_image_bytes(N, K) bytes, 16-byte aligned, filled once per weight by scgg_linear_rows_pack from
 *   W [N,K] (torch layout) and b [N].  K in {32, 64, 128, 256}.  rows / row_count: flat row indices and their number (device
 *   scalar), max_rows an upper bound of it.  relu: 0 none, 1 ReLU, 2 tanh-GELU. */
size_t scgg_linear_rows_image_bytes(int32_t N, int32_t K);
int scgg_linear_rows_pack(const float* W, const float* b, int32_t N, int32_t K, void* image, void* stream);
int scgg_linear_rows(const float* A, int32_t lda, const void* image, int32_t N, int32_t K, float* C, int32_t ldc,
                     const int32_t* rows, const int32_t* row_count, int32_t max_rows, int32_t relu, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCGG_H */
