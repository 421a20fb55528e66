/*
 * neurad_hip.h -- C ABI of libneurad_hip.so: NeuRAD's volumetric ray-marching hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary for the path named by BASELINE.json:north_star.  Every entry point
 * replaces one reference interface (cited as file:line relative to the neurad-studio tree); the
 * reference-side binding (ctypes) a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - All data pointers are DEVICE pointers (HBM) unless the comment says "host".  Buffers are owned by
 *    the caller (PyTorch allocates them); the library never allocates, frees or retains device memory.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every call only enqueues
 *    kernels on that stream; nothing synchronises the device.
 *  - Return value: 0 = NRHIP_OK, otherwise an NRHIP_ERR_* code; nrhip_last_error() returns a
 *    thread-local human-readable message for the last failing call on this thread.
 *  - Floating point is fp32 end to end (the parity target is the reference's implementation="torch"
 *    path).  Hash tables may be stored as fp32 (param_dtype 0) or fp16 (param_dtype 1, BASELINE config 5).
 *  - Row-major, innermost dimension last.  R = rays, S = samples per ray, N = R*S, L = levels,
 *    F = features per level, T = 2^log2_table_size entries per level.
 */
#ifndef NEURAD_HIP_H_
#define NEURAD_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRHIP_OK 0
#define NRHIP_ERR_INVALID_ARG 1  /* bad size / NULL pointer / unsupported configuration */
#define NRHIP_ERR_UNSUPPORTED 2  /* configuration outside what the kernels are instantiated for */
#define NRHIP_ERR_LAUNCH 3       /* hipLaunchKernel / hipGetLastError failure */

#define NRHIP_MAX_LEVELS 32
#define NRHIP_MAX_LAYERS 8

/* ---- descriptors (plain C structs, passed by pointer from host memory) ------------------------- */

/* Multi-resolution hash grid == HashEncoding(implementation="torch")
 * (nerfstudio/field_components/encodings.py:326-384).  scalings[l] = floor(min_res * g^l) is computed
 * by the host exactly as encodings.py:348-350 does, so the device never re-derives it.           */
typedef struct {
  int32_t num_levels;       /* L  */
  int32_t n_features;       /* F in {1,2,4,8} */
  int32_t log2_table_size;  /* T = 1 << log2_table_size (per level) */
  int32_t param_dtype;      /* 0 = fp32 table, 1 = fp16 table */
  float scalings[NRHIP_MAX_LEVELS];
} nrhip_grid;

/* MLP == MLP.pytorch_fwd (nerfstudio/field_components/mlp.py:142-178): Linear(+bias)+ReLU ... Linear.
 * weight[k] is nn.Linear.weight of layer k ([out_k][in_k] row-major), bias[k] may be NULL.         */
typedef struct {
  int32_t in_dim;
  int32_t hidden_dim;
  int32_t out_dim;
  int32_t num_layers;       /* number of Linear layers (>= 1) */
  const float* weight[NRHIP_MAX_LAYERS];
  const float* bias[NRHIP_MAX_LAYERS];
} nrhip_mlp;

/* A ray batch == the per-ray part of RayBundle/Frustums (nerfstudio/cameras/rays.py:33-59,251-357).
 * Samples are described by per-ray origin/direction and per-sample [start,end]; the materialised
 * [R,S,3] broadcast views of rays.py:336-355 are never built.                                       */
typedef struct {
  int64_t n_rays;           /* R */
  int32_t n_samples;        /* S */
  const float* origins;     /* [R,3] */
  const float* directions;  /* [R,3] */
  const float* pixel_area;  /* [R]   */
  const float* starts;      /* [R,S] euclidean bin starts */
  const float* ends;        /* [R,S] euclidean bin ends   */
  int32_t sample_stride;    /* row stride (floats) of starts/ends; 0 = S.  With bin EDGES e[R,S+1] pass
                               starts = e, ends = e + 1, sample_stride = S + 1 (no copies).            */
  const int32_t* order;     /* optional [R] permutation (NULL = batch order): the ORDER in which the fused kernels
                               (nrhip_field_fwd*, nrhip_render_fwd*) walk the rays -- a cache-locality hint from
                               nrhip_ray_order; every output stays indexed by the ray's own position in the batch,
                               results do not depend on it.  Other entry points ignore it.                      */
} nrhip_rays;

/* NeuRADField (nerfstudio/fields/neurad_field.py:78-152), static scene part.
 *   geo:  L*F -> hidden -> 1 + geo_feat_dim   (neurad_field.py:98-106)
 *   feat: geo_feat_dim + 16 (SH deg 4) -> hidden ... -> geo_feat_dim  (neurad_field.py:109-117)   */
typedef struct {
  nrhip_grid grid;
  const void* table;        /* [L*T, F] level-major rows (encodings.py:382-384) */
  float static_scale;       /* ScaledSceneContraction scale (neurad_encoding.py:99) */
  nrhip_mlp geo;
  nrhip_mlp feat;
  int32_t use_sdf;          /* 1: ALPHA = sigmoid(-sdf*beta) ; 0: DENSITY = exp(geo_out) */
  float beta;               /* |beta| + beta_min already applied (model_components/utils.py:38-41) */
  /* Optional, inference only (both NULL = off): the table re-laid out by nrhip_eval_layout_build and its description
   * from nrhip_eval_layout_plan (HOST array [L][4]).  nrhip_render_fwd* then reads the coarse levels from their shadow
   * copies; outputs are bit-identical to the plain table's.  A cache of the caller: rebuild after a parameter change. */
  const void* eval_table;
  const uint32_t* eval_layout;
} nrhip_field;

/* NeuRADProposalField (nerfstudio/fields/neurad_field.py:182-216): grid -> Linear(L*F,1,no bias) -> exp */
typedef struct {
  nrhip_grid grid;
  const void* table;
  float static_scale;
  const float* decoder_weight; /* [L*F] */
} nrhip_proposal;

/* ---- library ---------------------------------------------------------------------------------- */
const char* nrhip_last_error(void);
int nrhip_version(void);
/* number of CUs / XCDs of the current device (host out-pointers) */
int nrhip_device_info(int32_t* n_cus, int32_t* n_xcds, int64_t* hbm_bytes);
/* The library's A/B switches (NRHIP_* environment variables, csrc/common.h: struct Tuning) are read once, when the library is
 * loaded; a process that changes one afterwards calls this to have them read again.  Every switch selects between two
 * implementations of the same result. */
int nrhip_tuning_reload(void);

/* ---- eval-time layout of the coarse levels (csrc/eval_layout.hip) ----------------------------------------------
 * Levels whose lattice (0 .. ceil(scalings[l]))^3, padded to 2^s per axis, fits the table size are copied into a shadow
 * region indexed ix | iy << s | iz << 2s (8 corners in 4 cache lines, neighbours share lines); the others keep the
 * reference hash (encodings.py:419-444).  plan: layout [L][4] = {mulY, mulZ, mask, row0} per level and the eval table's
 * row count (host outputs, no GPU needed); build: fills eval_table [rows, F] (table's dtype) from the table.        */
int nrhip_eval_layout_plan(const nrhip_grid* g, uint32_t* layout /*host [L*4]*/, int64_t* rows /*host*/);
int nrhip_eval_layout_build(const nrhip_grid* g, const void* table, const uint32_t* layout /*host*/, void* eval_table,
                            void* stream);

/* ---- H1: hash grid (replaces HashEncoding.pytorch_fwd, encodings.py:425-466; tcnn.Encoding
 *          {otype:"HashGrid"} call sites encodings.py:362-373,468-471) --------------------------- */
int nrhip_hashgrid_fwd(const nrhip_grid* g, const void* table, const float* x /*[N,3] in [0,1]*/, int64_t n,
                       float* out /*[N,L*F]*/, void* stream);
/* grad_table [L*T,F] fp32 is ACCUMULATED into (caller zeroes it): autograd of encodings.py:446-464 */
int nrhip_hashgrid_bwd(const nrhip_grid* g, const float* x, const float* grad_out /*[N,L*F]*/, int64_t n,
                       float* grad_table, void* stream);

/* dL/dx of the lookup (fp32 table): only actor-hit samples need it */
int nrhip_hashgrid_bwd_input(const nrhip_grid* g, const void* table, const float* x, const float* grad_out, int64_t n,
                             float* grad_x /*[N,3]*/, void* stream);

/* Several grids of one shape in one launch -- the per-actor grids of NeuRADHashEncoding (the reference loops over
 * actor ids, `_get_actor_features_slow`, neurad_encoding.py:270-295).  tables / grad_tables: DEVICE arrays of n_grids
 * pointers to [L*T,F] fp32 tables; grid_id [N] int32 selects the grid of each sample.  grad_tables are ACCUMULATED
 * into; a sample whose grid_id is outside [0, n_grids) or whose grad_tables entry is NULL sends nothing (ABI 511; the
 * forward and _bwd_input need valid ids). */
int nrhip_hashgrid_multi_fwd(const nrhip_grid* g, const void* const* tables, int32_t n_grids, const int32_t* grid_id,
                             const float* x /*[N,3]*/, int64_t n, float* out /*[N,L*F]*/, void* stream);
int nrhip_hashgrid_multi_bwd(const nrhip_grid* g, int32_t n_grids, const int32_t* grid_id, const float* x,
                             const float* grad_out, int64_t n, float* const* grad_tables, void* stream);
/* The same table gradients WITHOUT memory-side atomics (ABI 511): the radix partition of nrhip_encode_bwd_binned over
 * (slot, level, slice) -- the gradients of the n_slots grids that have samples form ONE block [n_slots][L*T][F] (fp32, or
 * fp16 for fp16-storage grids: block_dtype 1, at most 2^23 samples), EVERY element of which is written (no zero-fill by
 * the caller).  slot_of [n_grids] int32 DEVICE: the grid's position in the block, < 0 = no gradient wanted (its samples
 * send nothing, like samples with grid_id outside [0, n_grids)).  Same sums as nrhip_hashgrid_multi_bwd up to the order of
 * the fp32 additions (this one is bit-reproducible).  _workspace gives 0 bytes when the shape cannot be partitioned
 * (n_slots x T / slice length > 2048 columns per level): use the atomic entry point then. */
int nrhip_hashgrid_multi_bwd_binned_workspace(const nrhip_grid* g, int32_t n_slots, int64_t n, int64_t* bytes /*host*/);
int nrhip_hashgrid_multi_bwd_binned(const nrhip_grid* g, int32_t n_grids, const int32_t* grid_id, const int32_t* slot_of,
                                    int32_t n_slots, const float* x /*[N,3]*/, const float* grad_out /*[N,L*F]*/, int64_t n,
                                    void* grad_block, int32_t block_dtype, void* workspace, int64_t workspace_bytes,
                                    void* stream);
int nrhip_hashgrid_multi_bwd_input(const nrhip_grid* g, const void* const* tables, int32_t n_grids,
                                   const int32_t* grid_id, const float* x, const float* grad_out, int64_t n,
                                   float* grad_x /*[N,3]*/, void* stream);

/* ---- H2+H3+H1+H4: NeuRADHashEncoding static path (neurad_encoding.py:164-169,265-268,297-304;
 *      cameras/rays.py:109-124; spatial_distortions.py:103-141) --------------------------------- */
int nrhip_encode_fwd(const nrhip_grid* g, const void* table, float static_scale, const nrhip_rays* rays,
                     float* out /*[N,L*F]*/, void* stream);
int nrhip_encode_bwd(const nrhip_grid* g, float static_scale, const nrhip_rays* rays,
                     const float* grad_out /*[N,L*F]*/, float* grad_table, void* stream);
/* dL/d(origins) [R,3] and dL/d(directions) [R,3] (both WRITTEN) of the same path given grad_out [N,L*F] = dL/d(rescaled
 * features): what autograd hands a camera optimizer that moves the rays (cameras/camera_optimizers.py:173-182,
 * apply_to_raybundle; the `*-scaleopt` methods, configs/method_configs.py:438-447) -- mean = o + d t (cameras/rays.py:119,
 * t constant: bins are detached, ray_samplers.py:363-364) -> contraction of mean AND std (spatial_distortions.py:126-141)
 * -> trilinear offsets (encodings.py:425-464) and the std-dependent rescale (neurad_encoding.py:297-304).  No atomics. */
int nrhip_encode_bwd_rays(const nrhip_grid* g, const void* table, float static_scale, const nrhip_rays* rays,
                          const float* grad_out /*[N,L*F]*/, float* grad_origins, float* grad_directions, void* stream);
/* Same result without memory-side atomics (every table entry gets one owning workgroup; see
 * csrc/encode_bwd_binned.hip).  Needs scratch: ask _workspace for the size (0 = this grid can not be binned, use
 * nrhip_encode_bwd), hand in a 16-byte aligned device buffer of at least that many bytes.  overwrite = 0: grad_table
 * is ACCUMULATED into, as above; overwrite = 1: every element of grad_table is WRITTEN (no zero-fill needed before,
 * and the pass over the table is a store instead of a read-modify-write). */
/* grad_table of the _binned entry points is ALWAYS fp32 [L*T,F]; g->param_dtype keeps its meaning (the TABLE's storage type)
 * and is not looked at (the partition never reads the table).  (ABI 500 read it as the gradient's type; 510 does not: a caller
 * that passes the descriptor of its fp16-storage table together with an fp32 grad_table gets fp32 again.)
 * nrhip_encode_bwd_binned_f16 writes the gradient as fp16 [L*T,F] instead -- the dtype autograd wants for an fp16-storage
 * table, without an fp32 image and a cast pass: every element is WRITTEN (overwrite semantics), at most 2^23 samples (one
 * round), NRHIP_ERR_UNSUPPORTED otherwise. */
int nrhip_encode_bwd_binned_workspace(const nrhip_grid* g, int64_t n_samples, int64_t* bytes);
/* The same scratch size serves the two other table gradients below (it depends on the grid and the sample count
 * only): nrhip_hashgrid_bwd_binned == nrhip_hashgrid_bwd, nrhip_proposal_density_bwd_binned ==
 * nrhip_proposal_density_bwd, without memory-side atomics. */
int nrhip_hashgrid_bwd_binned(const nrhip_grid* g, const float* x, const float* grad_out /*[N,L*F]*/, int64_t n,
                              float* grad_table, int32_t overwrite, void* workspace, int64_t workspace_bytes,
                              void* stream);
int nrhip_encode_bwd_binned(const nrhip_grid* g, float static_scale, const nrhip_rays* rays,
                            const float* grad_out /*[N,L*F]*/, float* grad_table, int32_t overwrite, void* workspace,
                            int64_t workspace_bytes, void* stream);
int nrhip_encode_bwd_binned_f16(const nrhip_grid* g, float static_scale, const nrhip_rays* rays,
                                const float* grad_out /*[N,L*F]*/, void* grad_table_fp16, void* workspace,
                                int64_t workspace_bytes, void* stream);

/* ---- F3: SHEncoding(levels=4) (encodings.py:797-805 -> utils/math.py:31-94) -------------------- */
int nrhip_sh4_fwd(const float* dirs /*[N,3]*/, int64_t n, float* out /*[N,16]*/, void* stream);

/* ---- F2: MLP (mlp.py:142-183; tcnn.Network call sites mlp.py:102-113,180-183) ------------------
 * hidden (optional, may be NULL): [N, (num_layers-1)*hidden_dim] post-ReLU activations, saved for bwd */
int nrhip_mlp_fwd(const nrhip_mlp* m, const float* x /*[N,in]*/, int64_t n, float* y /*[N,out]*/,
                  float* hidden, void* stream);
/* Backward.  grad_x may be NULL.  grad_weight[k] ([out_k][in_k]) and grad_bias[k] (may be NULL) are
 * ACCUMULATED into.  workspace: at least N*(num_layers-1)*hidden_dim floats (dZ of the hidden layers); with the
 * nrhip_mlp_bwd_workspace() amount NeuRAD's own MLP shapes get their weight gradients from the same pass as the
 * data gradient (no second read of the activations, no atomics). */
int nrhip_mlp_bwd_workspace(const nrhip_mlp* m, int64_t n, int64_t* floats);
int nrhip_mlp_bwd(const nrhip_mlp* m, const float* x, const float* hidden, const float* grad_y, int64_t n,
                  float* grad_x, float* const* grad_weight /*host array*/, float* const* grad_bias /*host array*/,
                  float* workspace, int64_t workspace_floats, void* stream);
/* Backward of NeuRADField's feature head, feature = geo[:, 1:] + mlp_feature([geo[:, 1:] | sh]) (neurad_field.py:146-152),
 * in one pass: the feature MLP's weight / bias gradients as nrhip_mlp_bwd forms them (x [N,48] = (embedding | sh), hidden,
 * workspace as there, sized by nrhip_mlp_bwd_workspace), and -- instead of grad_x -- the geometry MLP's complete output
 * gradient grad_geo [N,33]: column 0 = grad_geo0[n] (the sdf / density logit's gradient), columns 1..32 = grad_feature +
 * the embedding columns of grad_x (the residual connection).  The sh columns' gradient has no consumer and is not formed.
 * Covers the feature head's shapes 48 -> {32,64} -> {32,64} -> 32; pointers 16-byte aligned. */
int nrhip_field_feature_bwd(const nrhip_mlp* m, const float* x, const float* hidden, const float* grad_feature /*[N,32]*/,
                            const float* grad_geo0 /*[N]*/, int64_t n, float* grad_geo /*[N,33]*/,
                            float* const* grad_weight, float* const* grad_bias, float* workspace, int64_t workspace_floats,
                            void* stream);

/* ---- F1+F4: NeuRADField.forward, per-sample outputs (neurad_field.py:128-152) ------------------
 * feature [R,S,C], sdf_or_raw [R,S] (sdf when use_sdf else the pre-exp geo output), alpha_or_density [R,S] */
int nrhip_field_fwd(const nrhip_field* f, const nrhip_rays* rays, float* feature, float* sdf, float* alpha,
                    void* stream);
/* Training forward: the same kernel, additionally storing what nrhip_mlp_bwd / nrhip_encode_bwd* need, in their
 * layouts (N = R*S, H = hidden width, all 16-byte aligned):  save_enc [N,32] rescaled grid features (geometry MLP
 * input), save_geo_hidden [N,H], save_feat_in [N,48] = geometry embedding | SH(direction) (feature MLP input),
 * save_feat_hidden [N,2H] = layer 0 | layer 1.  Replaces encode_fwd + 2x mlp_fwd + sh4 + concat of the operator path. */
int nrhip_field_fwd_train(const nrhip_field* f, const nrhip_rays* rays, float* feature, float* sdf, float* alpha,
                          float* save_enc, float* save_geo_hidden, float* save_feat_in, float* save_feat_hidden,
                          void* stream);
/* The same with ROW OVERRIDES -- the training forward of a scene with dynamic actors (neurad_encoding.py:150-187): a
 * sample with ovr_row[i] = p >= 0 (i = ray * S + sample) takes its encoding row from ovr_rows [P,32] (the actor grid's
 * rescaled features, zero-padded, computed by the differentiable actor branch for the few samples inside a box) and the
 * view direction of its SH inputs from ovr_dirs [P,3] (box frame, neurad_encoding.py:203-208) instead of the static
 * lookup and the ray direction; ovr_row[i] = -1: the static scene.  save_enc then holds the overriding rows, so the
 * backward's d/d enc of those samples is the gradient of ovr_rows.  fp32 and fp16 static tables. */
int nrhip_field_fwd_train_ovr(const nrhip_field* f, const nrhip_rays* rays, const int32_t* ovr_row, const float* ovr_rows,
                              const float* ovr_dirs, float* feature, float* sdf, float* alpha, float* save_enc,
                              float* save_geo_hidden, float* save_feat_in, float* save_feat_hidden, void* stream);

/* ---- C1: nerfacc 0.5.2 dense-mode (call sites models/neurad.py:716-723,734; renderers.py:88,345) */
int nrhip_render_weight_from_alpha(const float* alphas /*[R,S]*/, int64_t r, int32_t s, float* weights,
                                   float* trans, void* stream);
int nrhip_render_weight_from_alpha_bwd(const float* alphas, const float* grad_w, const float* grad_t, int64_t r,
                                       int32_t s, float* grad_alphas, void* stream);
int nrhip_render_weight_from_density(const float* t_starts, const float* t_ends, const float* sigmas, int64_t r,
                                     int32_t s, float* weights, float* trans, float* alphas, void* stream);
int nrhip_render_weight_from_density_bwd(const float* t_starts, const float* t_ends, const float* sigmas,
                                         const float* grad_w, int64_t r, int32_t s, float* grad_sigmas,
                                         void* stream);
/* values may be NULL (accumulation only, C=1) */
int nrhip_accumulate_along_rays(const float* weights /*[R,S]*/, const float* values /*[R,S,C]*/, int64_t r,
                                int32_t s, int32_t c, float* out /*[R,C]*/, void* stream);
/* its autograd (values given): grad_weights [R,S] = sum_c g[r,c] v[r,s,c], grad_values [R,S,C] = w[r,s] g[r,c];
 * either output may be NULL */
int nrhip_accumulate_along_rays_bwd(const float* weights, const float* values, const float* g_out /*[R,C]*/, int64_t r,
                                    int32_t s, int32_t c, float* grad_weights, float* grad_values, void* stream);

/* ---- C4 / §8(f) row 2: lidar carving.  is_close [R,S] (uint8) = NeuRADModel._compute_is_close_to_lidar
 * (models/neurad.py:677-700): lidar sample whose midpoint lies within carving_epsilon of the measured return, or -- no
 * return -- anywhere closer than non_return_lidar_distance; with `weights` also the proposal carving term of
 * models/neurad.py:399-408 per ray, loss_per_ray [R] = sum_s (w * (is_lidar & ~close))^2, and its gradient
 * grad_weights [R,S] = 2 w (is_lidar & ~close).  is_lidar / did_return (may be NULL = all returned) are uint8 [R];
 * any of the three outputs may be NULL.                                                                             */
int nrhip_lidar_carving(const float* starts, const float* ends, int32_t sample_stride /*0 = S*/, const float* weights,
                        const uint8_t* is_lidar, const uint8_t* did_return, const float* distance /*[R]*/,
                        float carving_epsilon, float non_return_lidar_distance, int64_t r, int32_t s, uint8_t* is_close,
                        float* loss_per_ray, float* grad_weights, void* stream);

/* ---- C3: appearance embedding (models/neurad.py:423-441): out[r,:] = E[idx_lo[r]] * (1 - frac[r]) + E[idx_hi[r]] * frac[r]
 * (idx_hi == frac == NULL: the plain lookup of use_temporal_appearance = False).  E [n_embed, dim], indices int64.
 * bwd ACCUMULATES the table gradient into grad_weight [n_embed, dim] (caller zeroes): nn.Embedding's backward.     */
int nrhip_embedding_lerp_fwd(const float* weight, const int64_t* idx_lo, const int64_t* idx_hi, const float* frac,
                             int64_t r, int32_t n_embed, int32_t dim, float* out /*[R,dim]*/, void* stream);
int nrhip_embedding_lerp_bwd(const float* g_out /*[R,dim]*/, const int64_t* idx_lo, const int64_t* idx_hi,
                             const float* frac, int64_t r, int32_t n_embed, int32_t dim, float* grad_weight,
                             void* stream);

/* ---- C2: get_nff_outputs compositing (models/neurad.py:377-395,727-734) ------------------------
 * weights [R,S] come from C1; the residual 1-acc goes on the last (sky) sample; depth drops it.     */
int nrhip_composite_fwd(const float* weights, const float* features /*[R,S,C]*/, const float* starts,
                        const float* ends, int64_t r, int32_t s, int32_t c, float* out_features /*[R,C]*/,
                        float* out_depth /*[R]*/, float* out_acc /*[R]*/, void* stream);
int nrhip_composite_bwd(const float* weights, const float* features, const float* starts, const float* ends,
                        const float* g_features, const float* g_depth, const float* g_acc, int64_t r, int32_t s,
                        int32_t c, float* grad_weights /*[R,S]*/, float* grad_features /*[R,S,C]*/, void* stream);

/* ---- F1+C1+C2 fused: the headline kernel -- hash lookup + MLPs + compositing per ray -----------
 * (NeuRADModel.get_nff_outputs minus sampling, models/neurad.py:373-395).  Optional per-sample
 * outputs (may be NULL): weights [R,S] (pre sky residual, as returned by C1).
 * The MLPs' product sums are fp32 sums: formed from fp16 pairs on the matrix cores (x = fp16(x) + fp16(x - fp16(x)), fp32
 * accumulation; inputs that do not fit a pair take the fp32 MFMA tile by tile), 1e-7 from the fp32 MFMA's, which the
 * environment variable NRHIP_MLP_PAIRS=0 selects everywhere.                                        */
int nrhip_render_fwd(const nrhip_field* f, const nrhip_rays* rays, float* out_features /*[R,C]*/,
                     float* out_depth /*[R]*/, float* out_acc /*[R]*/, float* out_weights /*[R,S] or NULL*/,
                     void* stream);
/* The same with options.  early_stop_eps > 0 (eval only; 0 = exact): a ray stops marching once the transmittance
 * entering a 16-sample tile has fallen below it -- the samples behind that tile carry less than early_stop_eps of
 * weight in total and are skipped (wave-uniform test; weights of skipped samples are written as 0). */
int nrhip_render_fwd_ex(const nrhip_field* f, const nrhip_rays* rays, float* out_features, float* out_depth,
                        float* out_acc, float* out_weights, float early_stop_eps, void* stream);

/* ---- ray generation on the device (the step before the path) -------------------------------------
 * Sensor tables are device arrays indexed by camera / lidar; one call turns R (sensor index, pixel | point) pairs into
 * the per-ray part of a RayBundle.  Outputs: origins [R,3], directions [R,3] (unit), pixel_area [R], times [R]. */
typedef struct {
  const float* camera_to_worlds;     /* [C,3,4] row major (Cameras.camera_to_worlds) */
  const float* fx; const float* fy;  /* [C] */
  const float* cx; const float* cy;  /* [C] */
  const float* times;                /* [C] or NULL */
  int32_t rolling_shutter;           /* 0 none; 1 rows (PandaSet, top to bottom); 2 columns; 3 columns reversed
                                        (metadata["rs_direction"], cameras.py:941-953) */
  const float* rolling_shutter_time; /* [C] metadata["rolling_shutter_time"]   (NULL when rolling_shutter == 0) */
  const float* time_to_center_pixel; /* [C] metadata["time_to_center_pixel"] */
  const float* velocities;           /* [C,3] metadata["velocities"] */
  const float* shutter_extent;       /* [C] image height (mode 1) or width (modes 2, 3) as float */
} nrhip_camera_table;
/* == Cameras._generate_rays_from_coords (cameras.py:560-968) for PERSPECTIVE cameras without lens distortion (the
 * caller checks camera_type / distortion_params and keeps the reference's generator for anything else).  coords [R,2]
 * = (y, x) pixel-centre coordinates as in Cameras.get_image_coords.  directions_norm [R] = the pre-normalisation norm
 * (metadata["directions_norm"]).  times may be NULL when the table has no times. */
int nrhip_camera_rays(const nrhip_camera_table* cams, const int64_t* camera_indices /*[R]*/, const float* coords /*[R,2]*/,
                      int64_t n_rays, float* origins, float* directions, float* pixel_area, float* directions_norm,
                      float* times, void* stream);

typedef struct {
  const float* lidar_to_worlds;             /* [Ln,3,4] */
  const float* times;                       /* [Ln] or NULL */
  const float* velocities;                  /* [Ln,3] metadata["velocities"] or NULL */
  const float* horizontal_beam_divergence;  /* [Ln] */
  const float* vertical_beam_divergence;    /* [Ln] */
  int32_t assume_ego_compensated;           /* Lidars.assume_ego_compensated */
  float valid_lidar_distance_threshold;     /* did_return = distance < threshold (lidars.py:447) */
} nrhip_lidar_table;
/* == Lidars._generate_rays_from_points (lidars.py:399-460).  points [R,point_dim]: xyz in the lidar frame, intensity,
 * time offset within the sweep (point_dim >= 5 for the motion / time terms).  distance [R] = range of the point
 * (metadata["directions_norm"]), did_return [R] uint8. */
int nrhip_lidar_rays(const nrhip_lidar_table* lidars, const int64_t* lidar_indices /*[R]*/, const float* points,
                     int32_t point_dim, int64_t n_rays, float* origins, float* directions, float* pixel_area,
                     float* distance, uint8_t* did_return, float* times, void* stream);

/* == ScaledPatchSampler.collate_image_dataset_batch (data/pixel_samplers.py:618-742) on a device-resident image batch
 * images [n_images, height, width, channels] (image_dtype 0 = fp32, 1 = uint8; may be NULL together with patches):
 * patch centres -> ray_indices [P * patch_size^2, 3] int64 = (image_idx[c] (or c when image_idx is NULL), y, x), the
 * ground-truth patches [P, K, K, channels] with K = patch_size * patch_scale, and (optional, may be NULL) coords
 * [P * patch_size^2, 2] = (y + 0.5, x + 0.5), the image_coords[y, x] row RayGenerator.forward feeds to
 * nrhip_camera_rays (model_components/ray_generators.py:41-55).  Exactly one of
 *   uniforms [P,3] fp32  -- the torch.rand((P,3)) draws of PixelSampler.sample_method (:100-103); centre =
 *                           (u * float(n_images, H-K+1, W-K+1)).long() + (0, K/2, K/2)  (:722-726), and
 *   centers  [P,3] int64 -- (image, y, x) given directly (the sampling-weights branch, :728-742, clips them itself)
 * is non-NULL.  Integer results are bit-identical to the reference's; patches are copies.  Image reads are clamped into
 * the image (a centre closer than K/2 to the border would raise in the reference). */
int nrhip_patch_sample(const float* uniforms, const int64_t* centers, int64_t n_patches, int32_t n_images, int32_t height,
                       int32_t width, int32_t channels, int32_t patch_size, int32_t patch_scale, const int64_t* image_idx,
                       const void* images, int32_t image_dtype, int64_t* ray_indices, float* coords, void* patches,
                       void* stream);

/* == LidarPointSampler.collate_image_dataset_batch (data/pixel_samplers.py:538-583) on packed point clouds
 * lidar [sum(points_per_lidar), point_dim] fp32 (lidar_packed_collate, image_lidar_datamanager.py:60-74).  The caller
 * hands over torch's own draws -- shuffle [n_lidars] int64 = torch.randperm (:540) and draws [n_lidars, rays_per_lidar]
 * fp64 = torch.rand(..., dtype=float64) (:552), rays_per_lidar = ceil(n_rays / n_lidars) (:542).  Output row r < n_rays
 * belongs to scan l = shuffle[r / rays_per_lidar] and is its point p = floor(draws[l, r % rays_per_lidar] *
 * points_per_lidar[l]): indices [n_rays,2] = (lidar_idx[l] (or l when NULL), p), points [n_rays, point_dim] = the
 * gathered rows -- exactly what LidarRayGenerator / nrhip_lidar_rays consume.  n_lidars <= 2048. */
int nrhip_lidar_point_sample(const int64_t* shuffle, const double* draws, const int64_t* points_per_lidar,
                             const int64_t* lidar_idx, const float* lidar, int32_t n_lidars, int32_t rays_per_lidar,
                             int32_t point_dim, int64_t n_rays, int64_t* indices, float* points, void* stream);

/* ---- optimizer step of a hash table == torch.optim.Adam / AdamW on one fp32 tensor
 * (engine/optimizers.py:168-181; hashgrids group: lr 1e-2, eps 1e-15, configs/method_configs.py:423-426).  In place on
 * param / exp_avg / exp_avg_sq [n], 16-byte aligned; step = 1 for the first update; weight_decay is decoupled (AdamW),
 * 0 = plain Adam; grad_scale multiplies the gradient first (1 = off).  Hyper-parameters are doubles: torch derives 1 - beta,
 * lr / (1 - beta1^t) ... from Python floats and rounds once (1.f - 0.999f would be 1.3e-5 off).  Rows with grad = exp_avg = exp_avg_sq = 0 are
 * skipped: their update is exactly zero. */
int nrhip_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step, double lr,
                    double beta1, double beta2, double eps, double weight_decay, double grad_scale, void* stream);

/* The same update for MANY tensors in one launch per 24 tensors (the per-actor grids: 66 launches -> 3 on a 32-actor
 * scene), with two extras per tensor: grad_dtype 1 = the gradient is fp16 (what autograd hands an fp16-storage table),
 * converted in registers; image_fp16 != NULL = `param` is the fp32 MASTER copy of an fp16-storage table and the rounded
 * new value is written to image_fp16 in the same pass (no .float() / copy_ passes over the table).  step is per tensor
 * (a table no ray touched keeps its count: optim.py skips it, as torch.optim.Adam skips parameters without a gradient). */
typedef struct nrhip_adam_tensor {
  float* param;        /* [n] fp32: the tensor the update runs on                    */
  const void* grad;    /* [n] fp32 or fp16 (grad_dtype)                               */
  float* exp_avg;      /* [n] fp32                                                    */
  float* exp_avg_sq;   /* [n] fp32                                                    */
  void* image_fp16;    /* [n] fp16 or NULL                                            */
  int64_t n;
  int64_t step;        /* >= 1, this tensor's step count AFTER the increment          */
  int32_t grad_dtype;  /* 0 fp32, 1 fp16                                              */
  int32_t reserved;
} nrhip_adam_tensor;
int nrhip_adam_step_many(const nrhip_adam_tensor* tensors /* HOST array */, int32_t n_tensors, double lr, double beta1,
                         double beta2, double eps, double weight_decay, double grad_scale, void* stream);

/* The device-controlled form of the same update: torch.amp.GradScaler's optimizer protocol (engine/trainer.py:550-576 steps
 * every optimizer through GradScaler.step; optimizers with `_step_supports_amp_scaling` get the scale and the found-inf flag
 * as DEVICE scalars, torch/amp/grad_scaler.py) and the form a captured step (HIP graph) replays.
 *   step       DEVICE fp32 scalar per tensor (torch's capturable=True state layout), advanced by the launch itself;
 *   found_inf  DEVICE fp32 scalar or NULL: != 0 -> NOTHING is touched (parameters, moments, images, step counts);
 *   grad_scale DEVICE fp32 scalar or NULL: GradScaler's scale S, gradients are divided by it inside the update (no unscale
 *              pass over the table gradients); host_grad_scale multiplies on top (1 = off);
 *   lr_dev     DEVICE fp32 scalar or NULL: overrides lr (a scheduler fills it between graph replays);
 *   workspace  DEVICE, nrhip_adam_step_many_workspace(n_tensors) bytes: the per-tensor arguments a one-thread-per-tensor
 *              kernel derives (in double, rounded once -- the host form's arithmetic) for the streaming kernel.
 * No host read anywhere; every argument of a replay is behind a pointer. */
typedef struct nrhip_adam_tensor_dev {
  float* param;
  const void* grad;
  float* exp_avg;
  float* exp_avg_sq;
  void* image_fp16;
  int64_t n;
  float* step;         /* DEVICE fp32 scalar: the count BEFORE this update                */
  int32_t grad_dtype;  /* 0 fp32, 1 fp16                                                  */
  int32_t reserved;
} nrhip_adam_tensor_dev;
int nrhip_adam_step_many_workspace(int32_t n_tensors, int64_t* bytes);
int nrhip_adam_step_many_dev(const nrhip_adam_tensor_dev* tensors /* HOST array */, int32_t n_tensors, double lr,
                             const float* lr_dev, double beta1, double beta2, double eps, double weight_decay,
                             double host_grad_scale, const float* grad_scale, const float* found_inf, void* workspace,
                             void* stream);

/* GradScaler's inf check over gradients, READ-ONLY (ABI 511).  torch.amp.GradScaler.step runs `_check_inf_per_device` on an
 * optimizer that consumes the scale itself (engine/optimizers.py:168-181 steps every group through it): torch's
 * `_amp_foreach_non_finite_check_and_unscale_` with a scale of 1, which reads and re-writes every element.  This entry point
 * only reads: *found_inf (DEVICE fp32 scalar) is set to 1 when any element of any tensor is +-inf or NaN and is left alone
 * otherwise (the caller zeroes it; several calls may accumulate into one flag).  dtype 0 = fp32, 1 = fp16. */
typedef struct nrhip_check_tensor {
  const void* data;  /* [n], 16-byte aligned */
  int64_t n;
  int32_t dtype;
  int32_t reserved;
} nrhip_check_tensor;
int nrhip_nonfinite_check_many(const nrhip_check_tensor* tensors /* HOST array */, int32_t n_tensors, float* found_inf,
                               void* stream);

/* Processing order for cache locality (fills nrhip_rays.order): a permutation that groups rays looking at the same
 * region -- counting sort by the Morton code of the contracted position (ScaledSceneContraction, static_scale as in
 * nrhip_field) of the point origin + direction * t_ref, t_ref = a representative sample distance (the sampler's
 * median).  The reference has no counterpart: its rays arrive in data-loader order (camera patches + random lidar
 * points, data/datamanagers/image_lidar_datamanager.py:150-169).  order [R] int32 device buffer. */
int nrhip_ray_order(const float* origins /*[R,3]*/, const float* directions /*[R,3]*/, int64_t n_rays, float t_ref,
                    float static_scale, int32_t key_bits /* per axis: 0 = default (4), 3..5 */, int32_t* order,
                    void* stream);

/* The same permutation property (a counting sort by the same key; the order inside a bucket is unspecified in both) over
 * many workgroups, for eval chunks of tens of thousands of rays where the single-workgroup pass (~2 us per 1024 rays) would
 * cost more than the ordered kernels return.  workspace: (n_rays + 8^key_bits) uint32, see _workspace. */
int nrhip_ray_order_workspace(int64_t n_rays, int32_t key_bits, int64_t* bytes /*host*/);
int nrhip_ray_order_large(const float* origins, const float* directions, int64_t n_rays, float t_ref, float static_scale,
                          int32_t key_bits, void* workspace, int64_t workspace_bytes, int32_t* order, void* stream);

/* ---- S2: NeuRADProposalField.get_density (neurad_field.py:208-213) ----------------------------- */
/* level_features (may be NULL): LEVEL-MAJOR [L, R*S] rescaled per-level features, saved for the decoder gradient;
 * when requested (training) the lookups run level-partitioned over the XCDs (csrc/hashgrid.hip) */
int nrhip_proposal_density_fwd(const nrhip_proposal* p, const nrhip_rays* rays, float* density /*[R,S]*/,
                               float* level_features, void* stream);
/* grad_table / grad_decoder are accumulated into */
int nrhip_proposal_density_bwd(const nrhip_proposal* p, const nrhip_rays* rays, const float* density,
                               const float* grad_density, float* grad_table, float* grad_decoder, void* stream);
/* level_features: what the forward saved, or NULL (the interpolated features are then recomputed) */
int nrhip_proposal_density_bwd_binned(const nrhip_proposal* p, const nrhip_rays* rays, const float* density,
                                      const float* level_features, const float* grad_density, float* grad_table,
                                      float* grad_decoder, int32_t overwrite /*grad_table only*/, void* workspace,
                                      int64_t workspace_bytes, void* stream);

/* ---- S3: RaySamples.get_weights (cameras/rays.py:188-210) -------------------------------------- */
int nrhip_weights_from_density(const float* deltas, const float* densities, int64_t r, int32_t s, float* weights,
                               void* stream);
int nrhip_weights_from_density_bwd(const float* deltas, const float* densities, const float* grad_w, int64_t r,
                                   int32_t s, float* grad_densities, void* stream);

/* ---- S1: PowerSampler / SpacedSampler (ray_samplers.py:80-132,838-852; utils/math.py:541-579) ---
 * t_rand [R,S+1] = injected stratified jitter (training) or NULL (eval).  Outputs are bin EDGES.
 * last_edge > 0: the last euclidean edge is set to it -- the model's sky stretch (models/neurad.py:451-455) when
 * the PowerSampler bins go straight to the field; <= 0: off.                                          */
int nrhip_power_sampler(const float* nears /*[R]*/, const float* fars /*[R]*/, int64_t r, int32_t s, float lam,
                        float scaling, const float* t_rand, float last_edge, float* spacing_bins /*[R,S+1]*/,
                        float* euclid_bins /*[R,S+1]*/, void* stream);
/* nrhip_power_sampler + nrhip_ray_order in ONE launch (same arguments, same results as the two calls): workgroup 0 runs the
 * single-workgroup ordering pass while the others fill bins, so that the processing order of an eval chunk costs no launch
 * of its own in front of the render kernel. */
int nrhip_power_sampler_ordered(const float* nears, const float* fars, int64_t r, int32_t s, float lam, float scaling,
                                const float* t_rand, float last_edge, float* spacing_bins, float* euclid_bins,
                                const float* origins /*[R,3]*/, const float* directions /*[R,3]*/, float t_ref,
                                float static_scale, int32_t key_bits, int32_t* order /*[R]*/, void* stream);

/* ---- S4: PDFSampler (ray_samplers.py:280-376), include_original=False -------------------------
 * rand: NULL (eval) or [R] (single_jitter) / [R,S_new+1] (rand_stride = 1 / S_new+1)               */
int nrhip_pdf_sample(const float* weights /*[R,S_prev]*/, const float* spacing_bins /*[R,S_prev+1]*/,
                     const float* nears, const float* fars, int64_t r, int32_t s_prev, int32_t s_new, float lam,
                     float scaling, float histogram_padding, const float* rand, int32_t rand_stride,
                     float* new_spacing_bins /*[R,S_new+1]*/, float* new_euclid_bins /*[R,S_new+1]*/,
                     void* stream);

/* ---- H5: dynamic actors (NeuRADHashEncoding._split_static_vs_actors / _get_actor_indices /
 *      _get_actor_features_slow, field_components/neurad_encoding.py:189-295; DynamicActors.get_boxes2world,
 *      model_components/dynamic_actors.py:251-268; interpolate_trajectories_6d, utils/poses.py:90-150).
 *      Torch-path semantics: one 3-D hash grid per actor; when boxes overlap the highest actor index wins
 *      (the CPU index_put order of neurad_encoding.py:184-185); eval mode (no random flip).              */
typedef struct {
  int32_t n_actors;              /* A */
  int32_t n_times;               /* Tn */
  const float* timestamps;       /* [Tn]      unique_timestamps */
  const float* positions;        /* [Tn,A,3]  actor_positions */
  const float* rotations_6d;     /* [Tn,A,6]  actor_rotations_6d */
  const uint8_t* present;        /* [Tn,A]    actor_present_at_time */
  const float* bounds;           /* [A,3]     actor_sizes/2 + actor_padding (dynamic_actors.py:106-107) */
  nrhip_grid grid;               /* geometry shared by all actor grids (ActorSettings) */
  const void* const* tables;     /* DEVICE array of A table pointers, indexed by actor_to_id[actor] */
  float actor_scale;             /* actor contraction scale, 10 m (neurad_encoding.py:52,100) */
  int32_t max_candidates;        /* K: row length of the per-ray candidate lists below; 0 = NRHIP_DEFAULT_ACTOR_CANDIDATES.
                                    With K = n_actors no ray can overflow (the reference has no limit,
                                    neurad_encoding.py:225-263) and the lists cost R*K*52 bytes of HBM.          */
} nrhip_actors;

#define NRHIP_DEFAULT_ACTOR_CANDIDATES 8
#define NRHIP_MAX_SAMPLE_CONTAINMENTS 8 /* boxes that can contain ONE sample and still all receive gradients */
/* Per ray: interpolate every actor's pose at the ray's time, cull by distance to the ray's first->last sample line,
 * and compact the survivors: cand_count [R] int32, cand_actor [R,K] int32, cand_w2b [R,K,12] (3x4 world->box),
 * K = actors->max_candidates.  overflow (device int32, caller zeroes; may be NULL) is set if a ray has more than K. */
int nrhip_actor_prepare(const nrhip_actors* a, const nrhip_rays* rays, const float* times /*[R]*/,
                        int32_t* cand_count, int32_t* cand_actor, float* cand_w2b, int32_t* overflow, void* stream);
/* The same with the EVAL-time actor edit of DynamicActors.edit_boxes2world (model_components/dynamic_actors.py:181-249,
 * applied by get_boxes2world when the module is not training, :261-265; set by the viewer sliders :58-104 and by the
 * actor-shift FID evaluation, pipelines/ad_pipeline.py:476-480): after the pose interpolation the boxes of `index` (-1:
 * all; clamped to the last actor) are moved by (lateral, longitudinal, height) in their own frame and yawed by `rotation`
 * (radians, pre-multiplied: the translation stays).  Everything downstream -- line cull, in-box test, box-frame positions
 * and directions -- sees the edited pose.  As in the reference nothing happens unless lateral, longitudinal or rotation is
 * non-zero (a height-only edit is ignored, :182-187).  edit = NULL: nrhip_actor_prepare.                                    */
typedef struct {
  float lateral, longitudinal, height, rotation;
  int32_t index;
} nrhip_actor_edit;
int nrhip_actor_prepare_edited(const nrhip_actors* a, const nrhip_rays* rays, const float* times /*[R]*/,
                               const nrhip_actor_edit* edit, int32_t* cand_count, int32_t* cand_actor, float* cand_w2b,
                               int32_t* overflow, void* stream);
/* Field features: for every sample inside an actor box, OVERWRITE its feature row [out_dim] with the actor grid's
 * rescaled features zero-padded to out_dim, and write the per-sample direction (box frame, renormalised) -- ray
 * direction elsewhere.  hit [N] int32 = index of the actor whose grid was used, -1 elsewhere.                                         */
int nrhip_actor_encode(const nrhip_actors* a, const nrhip_rays* rays, const int32_t* cand_count,
                       const int32_t* cand_actor, const float* cand_w2b, int32_t out_dim, float* features /*[N,out_dim]*/,
                       float* directions /*[N,3]*/, int32_t* hit /*[N] actor index or -1*/,
                       const float* ray_flip /*[R] +-1 (training x-flip, neurad_encoding.py:212-219) or NULL*/,
                       void* stream);
/* All containments: hits [N, NRHIP_MAX_SAMPLE_CONTAINMENTS] = the actors whose boxes contain the sample, compacted in
 * ascending actor order and padded with -1 (the last non-negative entry is the one nrhip_actor_encode used; if more
 * boxes than that overlap at one point, the lowest indices drop out).  The reference's index_put backward
 * gives EVERY duplicate (ray, sample) row the upstream gradient, so training needs the whole list.            */
int nrhip_actor_hits(const nrhip_actors* a, const nrhip_rays* rays, const int32_t* cand_count,
                     const int32_t* cand_actor, const float* cand_w2b, int32_t* hits /*[N,8]*/, void* stream);
/* Proposal density: density = exp(decoder . padded actor features) for samples inside an actor box
 * (fields/neurad_field.py:208-213 with the actor branch of neurad_encoding.py:170-185).                      */
int nrhip_actor_density(const nrhip_actors* a, const nrhip_rays* rays, const int32_t* cand_count,
                        const int32_t* cand_actor, const float* cand_w2b, const float* decoder_weight, int32_t n_dec,
                        float* density /*[R,S] overwritten where hit*/, int32_t* hit /*actor index or -1*/, const float* ray_flip,
                        void* stream);

/* Training path of the in-box samples: box-frame, contracted position of P (sample, actor) pairs -- the chain
 * interpolate_trajectories_6d (utils/poses.py:90-150) -> rotation_6d_to_matrix (cameras/camera_utils.py:422-443) ->
 * pose inverse (utils/poses.py:42-55) -> transform_points_pairwise (cameras/lidars.py:550-564) -> training x-flip
 * (neurad_encoding.py:212-219) -> ScaledSceneContraction(inf) (spatial_distortions.py:103-141) -- and its backward into
 * the trajectory parameters (what autograd does for `require_actor_grad`, neurad_encoding.py:174-176).
 * sample_idx [P] int64 = flat sample index ray*S + s; actor_idx [P] int32; x01 [P,3] in [0,1]^3, cstd [P].
 * bwd ACCUMULATES into grad_positions [Tn,A,3] and grad_rotations_6d [Tn,A,6] (caller zeroes).                    */
int nrhip_actor_pair_positions_fwd(const nrhip_actors* a, const nrhip_rays* rays, const float* times,
                                   const int64_t* sample_idx, const int32_t* actor_idx, const float* ray_flip /*[R] or NULL*/,
                                   int64_t n_pairs, float* x01, float* cstd, void* stream);
int nrhip_actor_pair_positions_bwd(const nrhip_actors* a, const nrhip_rays* rays, const float* times,
                                   const int64_t* sample_idx, const int32_t* actor_idx, const float* ray_flip,
                                   int64_t n_pairs, const float* grad_x01, const float* grad_cstd,
                                   float* grad_positions, float* grad_rotations_6d, void* stream);
/* The same backward, additionally ACCUMULATING dL/d(origins) [R,3] and dL/d(directions) [R,3] of the pairs' rays (caller
 * zeroes or pre-fills): the in-box samples' share of the pose gradient of a camera optimizer that moves the rays
 * (cameras/camera_optimizers.py:173-182; `require_actor_grad`, neurad_encoding.py:174-176 keeps positions.mean in the graph). */
int nrhip_actor_pair_positions_bwd_rays(const nrhip_actors* a, const nrhip_rays* rays, const float* times,
                                        const int64_t* sample_idx, const int32_t* actor_idx, const float* ray_flip,
                                        int64_t n_pairs, const float* grad_x01, const float* grad_cstd,
                                        float* grad_positions, float* grad_rotations_6d, float* grad_origins,
                                        float* grad_directions, void* stream);

/* The (sample, actor) pairs of a hits table [N, NRHIP_MAX_SAMPLE_CONTAINMENTS] in (sample, slot) order -- what
 * `(hits >= 0).nonzero()` + a gather produce in the reference's style of indexing (neurad_encoding.py:221-263 returns such index
 * lists from torch.nonzero), as two streaming passes.  count: block_offsets [ceil(N / 1024)] u32 (scratch, exclusive prefix
 * on return) and *total (DEVICE int64) = number of pairs; the caller reads total, allocates sample_idx int64 [P] / actor_idx
 * int32 [P], and calls write with the same block_offsets and total (device pointers). */
int nrhip_actor_pairs_count(const int32_t* hits, int64_t n_samples, uint32_t* block_offsets, int64_t* total, void* stream);
int nrhip_actor_pairs_write(const int32_t* hits, int64_t n_samples, const uint32_t* block_offsets, const int64_t* total,
                            int64_t* sample_idx, int32_t* actor_idx, void* stream);

/* Proposal density of the in-box samples in TRAINING (fields/neurad_field.py:208-213 over neurad_encoding.py:150-187,
 * index_put order :184-185): rows [P, row_dim] = rescaled actor features of the P (sample, actor) pairs, decoder_weight
 * [row_dim], sample_idx [P] int64 flat sample index, winner [P] u8 (the pair whose actor the forward uses: the highest index
 * containing the sample; exactly one per hit sample).
 * fwd: density [N] IN/OUT -- the winners' trunc_exp(rows . w) overwrite the static density of their samples; logit [P] out.
 * bwd: grad_density [N] IN/OUT = the caller's copy of grad_out, zeroed here at the hit samples; grad_rows [P, row_dim] out
 *      (winners: through trunc_exp's backward exp(clamp(logit, +-15)), field_components/activations.py:37-41; shadowed pairs:
 *      grad_out * merged density * w, the duplicate-index gradient of the reference's index_put); grad_decoder [row_dim]
 *      ACCUMULATES (caller zeroes) and comes from the winners only.                                                        */
int nrhip_actor_density_splice_fwd(const float* rows, int32_t row_dim, const float* decoder_weight, const int64_t* sample_idx,
                                   const uint8_t* winner, int64_t n_pairs, float* density, float* logit, void* stream);
int nrhip_actor_density_splice_bwd(const float* rows, int32_t row_dim, const float* decoder_weight, const int64_t* sample_idx,
                                   const uint8_t* winner, const float* logit, const float* density_out, const float* grad_out,
                                   int64_t n_pairs, float* grad_density, float* grad_rows, float* grad_decoder, void* stream);

/* F1+C1+C2 with dynamic actors in ONE kernel (eval): nrhip_render_fwd_ex where a sample inside an actor's box reads
 * that actor's grid at its box-frame position and uses the box-frame view direction -- NeuRADHashEncoding.forward
 * (field_components/neurad_encoding.py:150-187, 203-208) + NeuRADField.forward (fields/neurad_field.py:128-152) +
 * compositing (models/neurad.py:373-395).  cand_*: the per-ray candidate lists of nrhip_actor_prepare (row length
 * a->max_candidates).  NRHIP_ERR_UNSUPPORTED unless both grids share one storage type (fp32 or fp16), the actor grid has the static grid's
 * features per level and at most its number of levels (the reference's defaults: static 8x4, actors 4x4).
 * Three launches, no host round trip: a device-side split of the processing order into rays without / with candidate
 * actors, the plain static kernel over the first slice, the actor-aware instantiation over the second.
 * workspace: device scratch of (n_rays + 4) int32.                                                                  */
int nrhip_render_fwd_actors(const nrhip_field* f, const nrhip_actors* a, const nrhip_rays* rays,
                            const int32_t* cand_count, const int32_t* cand_actor, const float* cand_w2b,
                            float* out_features, float* out_depth, float* out_acc, float* out_weights /*or NULL*/,
                            float early_stop_eps, int32_t* workspace, void* stream);

/* ---- S6: occupancy-grid ray march (VolumetricSampler.forward -> nerfacc OccGridEstimator.sampling,
 *      model_components/ray_samplers.py:483-566).  nerfacc is un-vendored and nothing in neurad-studio instantiates
 *      VolumetricSampler: the marching rule is this library's own statement (csrc/occgrid.hip header), parity
 *      unpinned.  Two passes: offsets == NULL -> per-ray counts; then, with the exclusive prefix sum of the counts as
 *      offsets, the packed (ray_indices, t_starts, t_ends) are written.                                        */
typedef struct {
  float aabb[6];            /* min xyz, max xyz */
  int32_t resolution;       /* res^3 cells, one level; cell index = (ix*res + iy)*res + iz */
  const uint8_t* binaries;  /* [res,res,res] occupancy */
} nrhip_occgrid;
int nrhip_occgrid_march(const nrhip_occgrid* grid, const float* origins, const float* directions,
                        const float* t_min /*[R] or NULL*/, const float* t_max /*[R] or NULL*/,
                        const float* t_rand /*[R] stratified offset in [0,1) or NULL*/, int64_t r,
                        float render_step_size, float near_plane, float far_plane, float cone_angle,
                        int32_t max_candidates, int32_t* counts /*[R], counting pass*/,
                        const int64_t* offsets /*[R], write pass*/, int64_t* ray_indices, float* t_starts,
                        float* t_ends, void* stream);
/* nerfacc render_visibility_from_alpha, packed: segments [R+1] delimit each ray's samples */
int nrhip_packed_visibility_from_alpha(const float* alphas, const int64_t* segments, int64_t r, float early_stop_eps,
                                       float alpha_thre, uint8_t* mask, void* stream);

/* ---- S5+M1 fused: ProposalNetworkSampler as driven by NeuRADModel._get_ray_samples
 *      (ray_samplers.py:623-666, models/neurad.py:443-459).  One wave marches one ray through
 *      power bins -> (density -> weights -> pdf resample) x n_rounds, entirely on chip.
 *      props[i] is the field evaluated in round i (the caller reproduces the late-binding quirk of
 *      models/neurad.py:248 by passing the same field twice).
 *      Outputs: final bins [R,S_final+1] (spacing + euclid, NOT sky-stretched -- M1's stretch is applied
 *      by the host mirror) and per-round weights / bins for the proposal losses.                     */
typedef struct {
  int32_t n_rounds;                 /* <= 2 */
  int32_t n_samples[3];             /* e.g. {128, 64, 32} */
  float lam, scaling;               /* PowerSampler(lambda_, scaling) */
  float histogram_padding;          /* 0.01 */
  float sky_distance;               /* fars are clamped to this (neurad.py:445-446) */
} nrhip_sampler_cfg;
int nrhip_proposal_sampler_fwd(const nrhip_sampler_cfg* cfg, const nrhip_proposal* props /*host array[n_rounds]*/,
                               const float* origins, const float* directions, const float* pixel_area,
                               const float* nears, const float* fars, int64_t r,
                               float* const* round_weights /*host array[n_rounds] of [R,S_i]*/,
                               float* const* round_spacing /*host array[n_rounds+1] of [R,S_i+1]*/,
                               float* const* round_euclid  /*host array[n_rounds+1] of [R,S_i+1]*/,
                               void* stream);
/* The same for a scene with dynamic actors: a proposal sample inside an actor's box takes its density from that actor's
 * grid of the proposal field (fields/neurad_field.py:208-213 over neurad_encoding.py:150-187).  actors[i] = the actor
 * grids of props[i] (host array[n_rounds]; all rounds share the actor set: bounds, max_candidates); cand_* = the per-ray
 * candidate lists of nrhip_actor_prepare -- they depend on the ray's line only, so one call on any two samples of the
 * ray serves every round and the field.  NRHIP_ERR_UNSUPPORTED unless the actor grids have 1 feature per level, fp32
 * tables and at most the static grid's levels (the reference's proposal defaults).                                  */
int nrhip_proposal_sampler_fwd_actors(const nrhip_sampler_cfg* cfg, const nrhip_proposal* props,
                                      const nrhip_actors* actors /*host array[n_rounds]*/, const int32_t* cand_count,
                                      const int32_t* cand_actor, const float* cand_w2b, const float* origins,
                                      const float* directions, const float* pixel_area, const float* nears,
                                      const float* fars, int64_t r, float* const* round_weights,
                                      float* const* round_spacing, float* const* round_euclid, void* stream);

/* ---- losses on the sampler outputs (model_components/losses.py) ---------------
 * One wavefront per ray; spacing-space bin edges c [R,S+1] in [0,1], weights [R,S].
 * zipnerf_interlevel_loss (losses.py:645-705), ONE proposal level per call: c/w = the fine (field) histogram
 * (detached in the reference), cp/wp = the proposal level, pulse_width = 0.03 / 0.003 for level 0 / 1.
 * loss_per_ray [R] = sum_s relu(w_s - wp)^2 / (wp + 1e-5); grad_wp [R,n_prop] (may be NULL) = d loss_per_ray / d wp.
 * The reference's value is the mean over rays, summed over levels. */
int nrhip_interlevel_loss(const float* c, const float* w, int32_t n_fine, const float* cp, const float* wp,
                          int32_t n_prop, float pulse_width, int64_t r, float* loss_per_ray, float* grad_wp,
                          void* stream);
/* distortion_loss / lossfun_distortion (losses.py:137-156): loss_per_ray [R], grad_w [R,S] (may be NULL) */
int nrhip_distortion_loss(const float* c, const float* w, int32_t n_samples, int64_t r, float* loss_per_ray,
                          float* grad_w, void* stream);

/* ---- the training step's glue as kernels (csrc/train_fused.hip) ---------------------------------------------------
 * What the reference spreads over dozens of elementwise torch ops per step between the field / sampler kernels.
 * `edges` is a bin-edge tensor e[R, >= S+1] with row stride edge_stride (floats): sample s of a ray spans
 * [e[s], e[s+1]] -- the [R,S,1] starts / ends / deltas views of cameras/rays.py:313-357 are never materialised.   */

/* RaySamples.get_weights (cameras/rays.py:188-210) + render_depth_simple of the round (models/neurad.py:396,727-734):
 * weights [R,S] = nan_to_num((1 - exp(-delta*dens)) * exp(-exclusive_cumsum(delta*dens))), depth [R] (may be NULL)
 * = sum_s weights * (e[s] + e[s+1]) / 2.                                                                            */
int nrhip_prop_weights_fwd(const float* edges, int32_t edge_stride, const float* densities /*[R,S]*/, int64_t r,
                           int32_t s, float* weights, float* depth, void* stream);
/* its autograd: grad_densities [R,S] from grad_weights [R,S] and / or grad_depth [R] (either may be NULL) */
int nrhip_prop_weights_bwd(const float* edges, int32_t edge_stride, const float* densities, const float* grad_weights,
                           const float* grad_depth, int64_t r, int32_t s, float* grad_densities, void* stream);

/* SigmoidDensity (model_components/utils.py:21-41; `beta` = DEVICE pointer to the raw learnable parameter, the kernel
 * applies |beta| + beta_min) -> nerfacc.render_weight_from_alpha -> accumulation -> sky residual on the last sample ->
 * features over all S samples, depth over the first S-1 (models/neurad.py:377-395).  alpha [R,S] (saved for the
 * backward), weights_ns [R,S-1] = the weights of the non-sky samples (`weights[..., :-1, :]` of models/neurad.py:388),
 * out_features rows of out_stride floats (>= C: room for the appearance embedding beside them), depth / acc [R].   */
int nrhip_sdf_render_fwd(const float* sdf /*[R,S]*/, const float* beta, float beta_min, const float* features /*[R,S,C]*/,
                         const float* edges, int32_t edge_stride, int64_t r, int32_t s, int32_t c, float* alpha,
                         float* weights_ns, float* out_features, int32_t out_stride, float* out_depth, float* out_acc,
                         void* stream);
/* floats of scratch nrhip_sdf_render_bwd needs (host out-pointer) */
int nrhip_sdf_render_bwd_workspace(int64_t r, int64_t* floats);
/* backward of the above: g_features rows of g_stride floats, g_depth / g_acc [R] and g_weights_ns [R,S-1] may be NULL;
 * -> grad_features [R,S,C], grad_sdf [R,S], grad_beta [1] (w.r.t. the raw parameter, sign(beta) applied; summed in a
 * fixed order: reproducible)                                                                                          */
int nrhip_sdf_render_bwd(const float* sdf, const float* beta, float beta_min, const float* alpha, const float* features,
                         const float* edges, int32_t edge_stride, const float* g_features, int32_t g_stride,
                         const float* g_depth, const float* g_acc, const float* g_weights_ns, int64_t r, int32_t s,
                         int32_t c, float* grad_features, float* grad_sdf, float* grad_beta, float* workspace,
                         void* stream);

/* NeuRADModel._get_appearance_embedding (models/neurad.py:423-441) with the slot arithmetic in the kernel:
 * t = times / duration * n_per_sensor, lo = clamp(floor(t)), hi = clamp(lo + 1), out = E[lo + sensor*n] (1 - (t - lo))
 * + E[hi + sensor*n] (t - lo); temporal = 0 (or times NULL): out = E[sensor].  sensor_idx may be NULL (= 0).  Rows of
 * `out` are out_stride floats apart.  Indices are clamped into the table.                                            */
int nrhip_appearance_fwd(const float* weight /*[E,D]*/, const int64_t* sensor_idx /*[R]*/, const float* times /*[R]*/,
                         float duration, int32_t n_per_sensor, int32_t temporal, int64_t r, int32_t n_embed, int32_t dim,
                         float* out, int32_t out_stride, void* stream);
/* gradient to the embedding table [E,D] (written, not accumulated) */
int nrhip_appearance_bwd(const float* g_out, int32_t g_stride, const int64_t* sensor_idx, const float* times,
                         float duration, int32_t n_per_sensor, int32_t temporal, int64_t r, int32_t n_embed, int32_t dim,
                         float* grad_weight, void* stream);

/* rows [n_out] (in order; the first n_out set positions) of a uint8 ray mask [R] + inverse [R] (slot of the row or -1)
 * + count [1] (all may be NULL): what `x[mask]` needs a nonzero + a device->host read for (models/neurad.py:478,485). */
int nrhip_mask_compact(const uint8_t* mask, int64_t r, int64_t* rows, int64_t n_out, int32_t* inverse, int32_t* count,
                       void* stream);

/* The lidar terms of NeuRADModel.get_metrics_dict (models/neurad.py:485-521) over the n lidar rays of a batch:
 * depths: HOST array of n_levels DEVICE pointers [R] (level 0 = the field's depth, 1.. = prop_depth_i), read at
 * lidar_rows [n].  metrics [2 + n_levels] = depth_loss (mean over the rays below the `quantile` of the per-ray error,
 * torch.quantile's linear interpolation), intensity_loss (same rays & returned), ray_drop_loss (BCE with logits, target
 * = no return), depth_loss_0, ...  unit_grads [(n_levels + 2), n] and scratch (nrhip_lidar_losses_workspace floats) carry
 * what the backward needs.  A per-ray pass + one single-workgroup pass; the quantile is a radix select, nothing is sorted. */
int nrhip_lidar_losses_workspace(int64_t n, int64_t* floats /*host*/);
int nrhip_lidar_losses(const float* const* depths, int32_t n_levels, const int64_t* lidar_rows, const float* distance,
                       const uint8_t* did_return, const float* intensity, const float* intensity_target,
                       const float* ray_drop_logits, int64_t n, float non_return_distance, float non_return_mult,
                       float quantile, float* metrics, float* unit_grads, float* scratch, void* stream);
/* d metrics / d predictions x upstream [2 + n_levels] (device) scattered to the batch: grad_depths = HOST array of
 * n_levels DEVICE pointers [R] (0 for camera rays; entries may be NULL), grad_intensity / grad_logits [n] (may be NULL);
 * unit_grads / scratch / did_return as given to (and filled by) nrhip_lidar_losses, inverse [R] from nrhip_mask_compact */
int nrhip_lidar_losses_bwd(const float* unit_grads, const float* scratch, const uint8_t* did_return, const int32_t* inverse,
                           const float* upstream, int32_t n_levels, int64_t r, int64_t n, float* const* grad_depths,
                           float* grad_intensity, float* grad_logits, void* stream);

/* ---- the level-sparse gradient exchange (opt-in; the reference's DDP all-reduce, pipelines/base_pipeline.py:304-307,
 *      sends every level of a hash table densely).  grad [n_levels * rows_per_level, f] fp32, 16-byte aligned.
 * count  : block_counts [n_levels, ceil(rows_per_level / NRHIP_GRAD_ROWS_PER_BLOCK)] = non-zero rows per block, turned into their
 *          exclusive prefix per level before the call returns; level_counts [n_levels] int64 = non-zero rows per level.
 * compact: the levels `levels[i]` (i < n_list_levels <= 32) as ordered lists: entries [sum caps[<i], + caps[i]) of rows (int32,
 *          level-local, ascending; the CALLER pre-fills -1: padding up to the capacity agreed between the ranks) and of vals
 *          [., f] = grad * scale.  block_offsets = count's block_counts.
 * apply  : mode 0: grad[level, row] = 0; mode 1: grad[level, row] += vals -- for the entries with row >= 0 of a list laid out
 *          the same way.  The rows of one list are distinct: no atomics, and lists applied in a fixed order give every rank
 *          bit-identical sums.                                                                                              */
#define NRHIP_GRAD_ROWS_PER_BLOCK 256 /* blocks per level = ceil(rows_per_level / 256) */
int nrhip_grad_rows_count(const float* grad, int32_t n_levels, int64_t rows_per_level, int32_t f, uint32_t* block_counts,
                          int64_t* level_counts, void* stream);
int nrhip_grad_rows_compact(const float* grad, int32_t n_levels, int64_t rows_per_level, int32_t f,
                            const uint32_t* block_offsets, const int32_t* levels /*host*/, const int64_t* caps /*host*/,
                            int32_t n_list_levels, float scale, int32_t* rows, float* vals, void* stream);
int nrhip_grad_rows_apply(float* grad, int32_t n_levels, int64_t rows_per_level, int32_t f, const int32_t* levels /*host*/,
                          const int64_t* caps /*host*/, int32_t n_list_levels, const int32_t* rows, const float* vals,
                          int32_t mode, void* stream);

/* ---- (f)-1: RGB CNN decoder (models/neurad.py:198-216,359-366; model_components/cnns.py:20-46) on fp16 operands with
 *          fp32 accumulation -- the arithmetic of the reference's mixed-precision trainer.  Activations are NHWC fp16:
 *          [B, H, W, 32].                                                                                              */
/* torch Conv2d weight [32][32][7][7] fp32 -> the kernel's fragment order, 49*2*64*16 bytes.  mode 0: forward (replaces
 * the four BasicBlocks' Conv2d.forward, cnns.py:38-44); mode 1: the same convolution's input gradient (flipped taps,
 * channels swapped), i.e. nrhip_conv7x7 on a packed mode-1 weight IS conv2d_backward w.r.t. the input.               */
int nrhip_conv7x7_pack(const float* weight, int32_t mode, void* wfrag, void* stream);
/* the same for up to 8 convolutions in one launch: weights = HOST array of n device pointers; wfrag [n][2][49*2*64*16 bytes]
 * receives mode 0 and mode 1 of every weight (the decoder packs once per step).                                       */
int nrhip_conv7x7_pack_many(const float* const* weights, int32_t n, void* wfrag, void* stream);
/* workgroups per image for an h x w image (a workgroup covers 4*rows_per_wave rows x 32 columns): stats_partial below is
 * [b * tiles, 64] floats.                                                                                             */
int nrhip_conv7x7_tiles(int32_t h, int32_t w, int32_t rows_per_wave, int32_t* tiles);
/* out = conv7x7(in, padding 3) + bias (bias may be NULL).  stats_partial (optional): per workgroup the sum and the sum
 * of squares of the ROUNDED outputs per channel (BatchNorm2d's batch statistics, cnns.py:40,43, in a fixed order).
 * rows_per_wave in {1, 2, 4}.                                                                                          */
int nrhip_conv7x7(const void* in, const void* wfrag, const float* bias, void* out, float* stats_partial, int32_t b,
                  int32_t h, int32_t w, int32_t rows_per_wave, void* stream);
/* Weight and bias gradient of the same convolution (conv2d_backward w.r.t. weight / bias): x = the convolution's input,
 * grad_out = the gradient of its output, both NHWC fp16.  grad_weight [32][32][7][7] and grad_bias [32] (optional) are fp32
 * and ACCUMULATED into.  workspace: nrhip_conv7x7_wgrad_workspace floats (per-workgroup partial sums, reduced in a fixed
 * order: bit-reproducible).                                                                                            */
int nrhip_conv7x7_wgrad_workspace(int32_t b, int32_t h, int32_t w, int64_t* floats);
int nrhip_conv7x7_wgrad(const void* x, const void* grad_out, float* workspace, float* grad_weight, float* grad_bias,
                        const float* grad_scale, int32_t b, int32_t h, int32_t w, void* stream);

/* The decoder's other layers.  All reductions over pixels go through per-workgroup partial sums in `workspace` that are
 * added in a fixed order; gradients of parameters are ACCUMULATED into (caller zeroes), fp32, in torch's layouts.
 * grad_scale (optional, every backward entry point): device {S, 1/S} from nrhip_dec_grad_scale.  The fp16 gradient
 * tensors inside the decoder carry the factor S -- the per-call equivalent of the loss scale the reference's
 * mixed-precision trainer applies (engine/trainer.py:189,553), without which gradients of order 1e-7 fall into fp16's
 * subnormals -- and every fp32 result (parameter gradients, grad_features) is multiplied by 1/S.                       */
/* scale [3 floats: S, 1/S, scratch]: S = the power of two that brings max |grad| into [0.5, 1)                        */
int nrhip_dec_grad_scale(const float* grad, int64_t n, float* scale, void* stream);
/* BatchNorm2d, training mode (cnns.py:40,43; functional.batch_norm): batch statistics from nrhip_conv7x7's stats_partial
 * [n_partial, 64]; coef [4][32] = scale, shift, mean, rstd; running_mean / running_var (both or neither) are updated as
 * torch does (momentum, unbiased variance).                                                                            */
int nrhip_dec_bn_finalize(const float* stats_partial, int32_t n_partial, int64_t count, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                          float* coef, void* stream);
/* out = relu(c * scale + shift [+ skip]) with the reference's fp16 roundings (BasicBlock.forward, cnns.py:31,38-44)   */
int nrhip_dec_bn_act(const void* c, const float* coef, const void* skip, void* out, int64_t n_pixels, void* stream);
/* backward of the above w.r.t. c (and gamma, beta): grad_c = batch_norm_backward(grad_out * (act > 0)), act = the
 * forward's output.  workspace: nrhip_dec_bn_bwd_workspace floats.                                                     */
int nrhip_dec_bn_bwd_workspace(int64_t n_pixels, int64_t* floats);
int nrhip_dec_bn_bwd(const void* grad_out, const void* act, const void* c, const float* gamma, const float* coef,
                     float* workspace, float* grad_gamma, float* grad_beta, const float* grad_scale, void* grad_c,
                     int64_t n_pixels, void* stream);
/* out = a + grad_out * (act > 0): a residual block's input gradient (convolution path + skip path)                    */
int nrhip_dec_add_masked(const void* a, const void* grad_out, const void* act, void* out, int64_t n_pixels, void* stream);
/* Conv2d(cin, 32, 1) + ReLU on the rendered feature rows (models/neurad.py:201-203, 361-364): features [n, cin] fp32 ->
 * h [n, 32] fp16; and its backward (grad_features [n, cin] fp32 is written, not accumulated).                          */
int nrhip_dec_conv1x1_in_fwd(const float* features, const float* weight /*[32,cin]*/, const float* bias, void* out,
                             int64_t n, int32_t cin, void* stream);
int nrhip_dec_conv1x1_in_bwd_workspace(int64_t n, int32_t cin, int64_t* floats);
int nrhip_dec_conv1x1_in_bwd(const float* features, const void* h, const void* grad_h, const float* weight,
                             float* workspace, float* grad_features, float* grad_weight, float* grad_bias,
                             const float* grad_scale, int64_t n, int32_t cin, void* stream);
/* ConvTranspose2d(32, 32, kernel_size = stride = 3) (models/neurad.py:206-211): [b,h,w,32] -> [b,3h,3w,32] as nine
 * [pixels,32] x [32,32] products on the matrix cores.  pack: weight [32 in][32 out][3][3] fp32 -> wup, 2*9*2*64*16 bytes
 * (forward and input-gradient fragment orders).                                                                        */
int nrhip_dec_upsample_pack(const float* weight, void* wup, void* stream);
int nrhip_dec_upsample_fwd(const void* h, const void* wup, const float* bias, void* out, int32_t b, int32_t hh, int32_t w,
                           void* stream);
int nrhip_dec_upsample_bwd_workspace(int32_t b, int32_t hh, int32_t w, int64_t* floats);
int nrhip_dec_upsample_bwd(const void* h, const void* grad_out, const void* wup, float* workspace, void* grad_h,
                           float* grad_weight, float* grad_bias, const float* grad_scale, int32_t b, int32_t hh, int32_t w,
                           void* stream);
/* Conv2d(32, 3, 1) + Sigmoid (models/neurad.py:214-215): rgb [n_pixels, 3] fp32, and its backward                    */
int nrhip_dec_rgb_fwd(const void* h, const float* weight /*[3,32]*/, const float* bias, float* rgb, int64_t n_pixels,
                      void* stream);
int nrhip_dec_rgb_bwd_workspace(int64_t n_pixels, int64_t* floats);
int nrhip_dec_rgb_bwd(const void* h, const float* rgb, const float* grad_rgb, const float* weight, float* workspace,
                      void* grad_h, float* grad_weight, float* grad_bias, const float* grad_scale, int64_t n_pixels,
                      void* stream);

/* The whole decoder behind one entry point per direction (what a binding needs; the per-layer entry points above are its
 * parts and stay callable).  Parameters in torch's layouts, fp32; features [n_patches * patch_h * patch_w, cin] fp32 in
 * patch order (models/neurad.py:361-362); rgb [n_patches, 3 patch_h, 3 patch_w, 3] fp32.  conv_w/conv_b/bn_*[2k + j] =
 * convolution j (0: first, 1: second) of BasicBlock k (0, 1 before the upsampling, 2, 3 after).                        */
typedef struct {
  int32_t n_patches, patch_h, patch_w, cin;
  int32_t training; /* BatchNorm2d: 1 = batch statistics (running statistics are updated), 0 = running statistics      */
  const float* conv_in_w; /* [32, cin] */
  const float* conv_in_b;
  const float* conv_w[8]; /* [32, 32, 7, 7] */
  const float* conv_b[8];
  const float* bn_gamma[8];
  const float* bn_beta[8];
  float* bn_running_mean[8];
  float* bn_running_var[8];
  float bn_eps[8];
  float bn_momentum[8];
  const float* up_w; /* [32 in, 32 out, 3, 3] */
  const float* up_b;
  const float* out_w; /* [3, 32] */
  const float* out_b;
} nrhip_rgb_decoder;
/* bytes of `saved` (activations kept for the backward) and `workspace` (scratch), floats of grad_params                */
int nrhip_rgb_decoder_sizes(const nrhip_rgb_decoder* d, int64_t* saved_bytes, int64_t* workspace_bytes,
                            int64_t* grad_param_floats);
int nrhip_rgb_decoder_fwd(const nrhip_rgb_decoder* d, const float* features, void* saved, void* workspace, float* rgb,
                          void* stream);
/* grad_params (written, not accumulated): conv_in (w, b), then per convolution i = 0..7 (w, b, gamma, beta), up (w, b),
 * out (w, b).  training == 0: BatchNorm normalised with its running statistics in the forward, its backward is then the
 * affine map gamma * rstd * g (round 4).                                                                               */
int nrhip_rgb_decoder_bwd(const nrhip_rgb_decoder* d, const float* features, const void* saved, const float* rgb,
                          const float* grad_rgb, void* workspace, float* grad_features, float* grad_params, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEURAD_HIP_H_ */
