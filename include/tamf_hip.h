/*
 * tamf_hip.h - C-ABI of the MI355X (gfx950) MF-MDM denoiser + DDPM sampler library (libtamf_hip.so).
 *
 * The reference (oakink/OakInk2-TaMF) is pure Python/PyTorch and has no FFI layer; the drop-in boundary
 * of this path is its two Python call contracts.  Each entry point below names the
 * reference interface it stands under (paths relative to src/oakink2_tamf/ of the reference):
 *
 *   model(x, ts, batch=...)            model/interaction_segment_mdm.py:134-174   -> tamf_set_cond + tamf_denoise
 *   diffusion.p_sample_loop(...)       model/diffusion/gaussian_diffusion.py:506-640 -> tamf_sample_loop
 *   GaussianDiffusion.p_sample         model/diffusion/gaussian_diffusion.py:412-460 -> tamf_ddpm_step
 *   create_gaussian_diffusion tables   model/diffusion_util.py:5-31, gaussian_diffusion.py:116-161 -> tamf_set_schedule
 *   load_state_dict(torch.load(ckpt))  launch/sample.py:190-192, util/state_util.py:22-39 -> tamf_load_weight / tamf_finalize_weights
 *   SegmentRefineModel.forward trunk   model/segment_refine_model.py:175-217      -> tamf_refine
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative tamf_status; the message of the
 *     last failure is available from tamf_last_error(ctx) (or tamf_last_error(NULL) for ctx-less failures).
 *     Nothing throws across the ABI.
 *   - "dev" pointers are device (HBM) pointers owned by the caller (e.g. torch tensors' data_ptr());
 *     "host" pointers are host memory.  The library owns weights, workspaces and hipGraphs.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the call returns without
 *     synchronising (exceptions are stated per function).  A context is not thread-safe (one thread per context); distinct contexts
 *     are independent, also when driven from different threads: libtamf_hip.so takes no process-wide lock in its entry points (rounds
 *     3 - 5 did, for the process-global kernel-selection word of tamf_set_gemm_tuning - which now exists only in the test build,
 *     include/tamf_hip_test.h; there the lock is kept).  Only tamf_ctx_create is serialised across threads.
 *   - tensors use the reference's layouts: x / x0 / noise are (B, input_dim, 1, T) float32 contiguous.
 */
#ifndef TAMF_HIP_H
#define TAMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tamf_ctx tamf_ctx;

typedef enum tamf_status {
  TAMF_OK = 0,
  TAMF_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
  TAMF_ERR_STATE = -2,     /* call order violated (weights not finalised, cond not set, ...) */
  TAMF_ERR_HIP = -3,       /* a HIP runtime call failed */
  TAMF_ERR_MISSING = -4,   /* a required checkpoint tensor was not loaded */
  TAMF_ERR_NOMEM = -5,
  TAMF_ERR_RANGE = -6      /* a weight does not fit the operand format of the context's precision (f16x3: a non-finite weight) */
} tamf_status;

/* bits of tamf_get_status_flags */
typedef enum tamf_status_flag {
  TAMF_STATUS_F16_RANGE = 1, /* f16x3 only: an activation beyond +-65504 (or +-inf) was stored as a split-fp16 operand since the
                                last clear; results computed since then may differ from the reference's fp32 arithmetic */
  TAMF_STATUS_F16_WEIGHT_RANGE = 2 /* f16x3 only, set by tamf_finalize_weights and never cleared: more than 1 % of the non-zero weights
                                      of some tensor are below 2^-17.5 of that tensor's largest magnitude (one power-of-two scale per
                                      tensor: an outlier dominates it) and are stored with fewer than 22 significand bits;
                                      tamf_last_error names the tensor after the call.  The Python modules fall back to f32. */
} tamf_status_flag;

/* arithmetic mode of the MFMA contractions (everything else - residual stream, LayerNorm, softmax,
 * DDPM state - is float32 in every mode) */
typedef enum tamf_precision {
  TAMF_PREC_F32 = 0,    /* v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate (parity mode) */
  TAMF_PREC_BF16 = 1,   /* v_mfma_f32_16x16x32_bf16: bf16 operands, fp32 accumulate */
  TAMF_PREC_BF16X3 = 2, /* split-bf16 (hi+lo) operands, 3 bf16 MFMAs per product: ~2^-17 relative operand error */
  TAMF_PREC_F16X3 = 3   /* split-fp16 (hi+lo) operands, 3 f16 MFMAs per product: ~2^-22 relative operand error (fp32: 2^-24),
                           operands limited to |v| <= 65504 */
} tamf_precision;

typedef enum tamf_model_kind {
  TAMF_KIND_G = 0, /* InterationSegmentMDM (5 prefix tokens: t, text, side, shape, obj) */
  TAMF_KIND_R = 1  /* SegmentRefineModel trunk (3 prefix tokens; h2o distance feature; residual output) */
} tamf_model_kind;

/* keys of config/arch_*.yml `model:` (launch/param/model.py:14-87) + the two constants of the modules */
typedef struct tamf_arch {
  int32_t input_dim;      /* 99  */
  int32_t obj_input_dim;  /* 9   */
  int32_t hand_shape_dim; /* 10  */
  int32_t obj_embed_dim;  /* 768 */
  int32_t latent_dim;     /* 256 (arch_mdm, arch_refine) / 512 (arch_mdm_l); supported: 128, 256, 512 */
  int32_t ff_size;        /* 1024 / 2048; multiple of 128 */
  int32_t num_layers;     /* 8 */
  int32_t num_heads;      /* latent_dim / num_heads must be 64 or 128 */
  int32_t clip_dim;       /* 512 (interaction_segment_mdm.py:25) */
  int32_t h2o_dim;        /* 778 (segment_refine_model.py:267-279); R only */
  int32_t kind;           /* tamf_model_kind */
} tamf_arch;

/* Create a context on `device` able to run up to max_batch clips of up to max_frames frames. */
int tamf_ctx_create(const tamf_arch* arch, int32_t max_batch, int32_t max_frames, int32_t precision,
                    int32_t device, tamf_ctx** out);
/* Re-dimension the context for up to max_batch clips of up to max_frames frames: the workspaces (sampler state, token rows, operand
 * planes, scratch) are freed and allocated anew; weights, tables and the schedule stay, so no checkpoint is uploaded or repacked again
 * (a launcher whose clip source changes shape, launch/sample.py:204-215 / launch/sample_refine.py:224-236, pays milliseconds).
 * Synchronises the device, drops the captured hipGraph and the conditioning (tamf_set_cond must be called again).  The context's
 * sticky status word is carried over (a range flag not yet read survives).  Transactional: the new workspaces are allocated beside
 * the old ones, which are freed only when every allocation has succeeded; on failure (TAMF_ERR_NOMEM: the larger batch does not fit)
 * the context keeps its old dimensions, workspaces and conditioning and stays usable. */
int tamf_ctx_resize(tamf_ctx* ctx, int32_t max_batch, int32_t max_frames);
void tamf_ctx_destroy(tamf_ctx* ctx);
const char* tamf_last_error(const tamf_ctx* ctx);

/* One call per checkpoint tensor (SURVEY.md A.2 key set; host float32, row-major, `shape[ndim]`).
 * Unknown names are ignored (strict=False semantics of launch/sample.py:190-192; returns 0),
 * a known name with a wrong shape is TAMF_ERR_INVALID. */
int tamf_load_weight(tamf_ctx* ctx, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
/* Repack into kernel layouts (operand precision, fused input weights, timestep-embedding table for
 * t in [0, max_timesteps)).  Synchronises `stream`.  TAMF_ERR_MISSING names the first absent tensor;
 * TAMF_ERR_RANGE (f16x3 only) the first tensor holding a non-finite weight (finite weights of any magnitude are stored scaled by
 * a per-tensor power of two that the GEMM epilogue takes out again, exactly; see tamf_get_status_flags). */
int tamf_finalize_weights(tamf_ctx* ctx, int32_t max_timesteps, void* stream);

/* float64 tables of GaussianDiffusion.__init__ for the n_steps-step process; they are cast to float32
 * exactly where the reference casts them (gaussian_diffusion.py:1275). */
int tamf_set_schedule(tamf_ctx* ctx, int32_t n_steps, const double* posterior_mean_coef1,
                      const double* posterior_mean_coef2, const double* posterior_log_variance_clipped);
/* Respaced sampling (reference model/diffusion/respace.py:60-119: SpacedDiffusion keeps a subset of the base process' timesteps and its
 * _WrappedModel evaluates the denoiser at timestep_map[t]).  Call after tamf_set_schedule with the SAME n_steps: step i of the loop
 * (i = n_steps - 1 .. 0, coefficients i of the schedule) evaluates the denoiser at timestep map_host[i] of the base process - strictly
 * increasing entries inside the timestep table (max_timesteps of tamf_finalize_weights).  The library gathers the rows of its
 * timestep-embedding table once; the captured loop is step-agnostic as before.  NULL restores the identity map, and so does every
 * later tamf_set_schedule.  tamf_denoise is not affected (the caller's timesteps index the table directly, as _WrappedModel's do). */
int tamf_set_timestep_map(tamf_ctx* ctx, int32_t n_steps, const int32_t* map_host);

/* Conditioning of one batch (the `batch` dict of model.forward, CLIP output supplied as text_embedding):
 *   text_emb_dev  (B, clip_dim) f32 [G only, NULL for R]      hand_side_host (B,) uint8: 0 = "rh", 1 = "lh"
 *   shape_dev     (B, T, hand_shape_dim) f32                  obj_emb_dev    (B, nobj, obj_embed_dim) f32
 *   obj_traj_dev  (B, nobj, T, obj_input_dim) f32
 * Runs the step-invariant precompute (prefix tokens 1..4, object half of input_merge.0) on `stream`.
 * A hand_side value other than 0/1 is TAMF_ERR_INVALID (the reference raises ValueError, :284). */
int tamf_set_cond(tamf_ctx* ctx, int32_t B, int32_t T, int32_t nobj, const float* text_emb_dev,
                  const uint8_t* hand_side_host, const float* shape_dev, const float* obj_emb_dev,
                  const float* obj_traj_dev, void* stream);

/* The same with per-clip object counts (host, (B,) int32, each in [1, nobj]; NULL = tamf_set_cond): the two object means of the
 * forward (interaction_segment_mdm.py:233-263) run over clip b's first obj_num[b] objects instead of all nobj rows of the zero-padded
 * batch.  That is what the reference's launchers compute - they call the model one clip at a time (launch/sample.py:206,
 * launch/sample_refine.py:228), so a clip never sees another clip's padding - and what a batched launcher must pass to reproduce them
 * (dataset/collate.py:43-56 pads `obj_traj` / `obj_embedding` to the batch maximum and carries `obj_num`). */
int tamf_set_cond_ragged(tamf_ctx* ctx, int32_t B, int32_t T, int32_t nobj, const int32_t* obj_num_host, const float* text_emb_dev,
                         const uint8_t* hand_side_host, const float* shape_dev, const float* obj_emb_dev,
                         const float* obj_traj_dev, void* stream);

/* x0_hat = G(x_t, t | cond).  x_dev, x0_out_dev: (B, input_dim, 1, T) f32; t_dev: (B,) int64 device
 * (values in [0, max_timesteps)). */
int tamf_denoise(tamf_ctx* ctx, const float* x_dev, const int64_t* t_dev, float* x0_out_dev, void* stream);

/* One reverse step x_{t-1} = coef1[t] x0 + coef2[t] x_t + [t != 0] exp(0.5 logvar[t]) eps, elementwise over
 * n float32 values.  noise_dev may be NULL only when t == 0. */
int tamf_ddpm_step(tamf_ctx* ctx, const float* x_t_dev, const float* x0_dev, int32_t t, const float* noise_dev,
                   float* x_out_dev, int64_t n, void* stream);

/* The full reverse loop x_T -> x_0 (n_steps of the schedule, t = n_steps-1 .. 0), hipGraph-replayed.
 *   noise_dev != NULL : (n_steps + 1, B, input_dim, 1, T) f32 - draw 0 is x_T, draw k the eps of the k-th
 *                       step, i.e. the reference's th.randn / th.randn_like call order
 *                       (gaussian_diffusion.py:604,448)  [parity mode]
 *   noise_dev == NULL : Philox4x32-10 + Box-Muller on device, keyed (seed, clip_id_base + b, draw, element),
 *                       so a clip's noise does not depend on how a batch is sharded  [throughput mode]
 * dump_dev (optional): (n_steps, B, input_dim, 1, T) f32 receives x after every step (dump_steps of the reference).
 * use_graph = 0 issues plain launches (debugging). */
int tamf_sample_loop(tamf_ctx* ctx, const float* noise_dev, uint64_t seed, int64_t clip_id_base,
                     float* x0_out_dev, float* dump_dev, int32_t use_graph, void* stream);

/* R trunk: refine = x_in + head(encoder(...)).  sample_pose_repr_dev, out_dev: (B, T, input_dim) f32;
 * h2o_dist_dev: (B, T, h2o_dim) f32. */
int tamf_refine(tamf_ctx* ctx, const float* sample_pose_repr_dev, const float* h2o_dist_dev, float* out_dev,
                void* stream);

/* ---- geometry either side of the trunks ---------------------------- */
/* Pose decode of launch/sample_refine.py:254-260 / model/segment_refine_model.py:117-124:
 * pose_repr (n_frames, 3 + 6*n_joints) f32 -> tsl (n_frames, 3) [may be NULL] and unit quaternions
 * (n_frames, n_joints, 4) in (w, x, y, z) order with w >= 0 (dev_fn/transform/rotation.py:446-467,167-213,24-35). */
int tamf_pose_decode(const float* pose_repr_dev, int64_t n_frames, int32_t n_joints, float* tsl_out_dev,
                     float* quat_out_dev, void* stream);
/* Hand->object distance feature of SegmentRefineModel.multi_object_h2o_dist (model/segment_refine_model.py:142-168,
 * model/loss/chamfer_distance.py:4-64 with y_normals = None; replaces the external chamfer_distance CUDA extension):
 *   h2o[b,t,v] = min_{o < obj_num[b], j < P} || hand_verts[b,t,v] - (R(b,o,t) obj_points[b,o,j] + tsl(b,o,t)) ||_2
 * hand_verts (B,T,V,3), obj_traj (B,nobj,T,9) = [tsl | rot6d], obj_points (B,nobj,P,3) in the object frame,
 * obj_num (B,) int32 device or NULL (= nobj for every clip), out (B,T,V); V <= 1024. */
int tamf_h2o_dist(const float* hand_verts_dev, const float* obj_traj_dev, const float* obj_points_dev,
                  const int32_t* obj_num_dev, int32_t B, int32_t T, int32_t V, int32_t nobj, int32_t P,
                  float* h2o_out_dev, void* stream);
/* Per-frame contact distance of the Contact-Ratio score (script/compute_score/compute_score_cr.py:122-149,282-283:
 * transf_merge_obj_pointcloud + torch.cdist(hand_verts, merged_points).min per frame; a frame is "in contact" when
 * the value is < 0.005 m):  min_dist[b,t] = min_v h2o[b,t,v] with h2o as in tamf_h2o_dist.  Same argument layout;
 * min_dist_out (B,T). */
int tamf_contact_min_dist(const float* hand_verts_dev, const float* obj_traj_dev, const float* obj_points_dev,
                          const int32_t* obj_num_dev, int32_t B, int32_t T, int32_t V, int32_t nobj, int32_t P,
                          float* min_dist_out_dev, void* stream);

/* Object point clouds moved along their trajectories: out[o,t,j] = R(o,t) points[o,j] + tsl(o,t), (tsl | rot6d) = traj[o,t]
 * (dev_fn/transform/transform_np.py:169-175 tslrot6d_to_transf_np + :36-53 transf_point_array_np, as called by
 * compute_score_cr.py:122-137 and compute_score_siv.py:146).  traj (n_obj,T,9), points (n_obj,P,3), out (n_obj,T,P,3);
 * is_f64: 0 = float32 buffers, 1 = float64 buffers. */
int tamf_transform_points(const void* obj_traj_dev, const void* obj_points_dev, int32_t n_obj, int32_t T, int32_t P,
                          int32_t is_f64, void* out_dev, void* stream);
/* Vertex normals of a mesh sequence (model/segment_refine_model.py:131-133: pytorch3d Meshes(verts, faces).verts_normals_packed()
 * of the MANO hand; pytorch3d 0.7.2 _compute_vertex_normals: area-weighted face normals summed per vertex, x / max(|x|, 1e-6)).
 * verts (n_mesh, V, 3) f32; the incidence list of the (fixed) topology in CSR form, built by the caller from faces (F,3):
 * csr_off (V + 1) int32, csr_ent (2 * 3F) int32 = for every corner (face f, corner c) of vertex faces[f][c] the pair
 * (faces[f][(c+1)%3], faces[f][(c+2)%3]), ordered corner 1 / corner 2 / corner 0, faces ascending (oakink2_tamf_amd.geometry
 * .vertex_normals builds it); normals_out (n_mesh, V, 3) f32. */
int tamf_vertex_normals(const float* verts_dev, int64_t n_mesh, int32_t V, const int32_t* csr_off_dev, const int32_t* csr_ent_dev,
                        float* normals_out_dev, void* stream);
/* Point-in-closed-mesh test of the Solid-Intersection-Volume score (script/compute_score/compute_score_siv.py:128-153 ->
 * dev_fn/external/libmesh/inside_mesh.py:8-149 check_mesh_contains, whose Cython TriangleHash is an acceleration structure
 * only): float64, the reference's operation order, no fused multiply-adds - the result is bit-identical to numpy's.
 * verts (V,3) f64, faces (F,3) int32, points (N,3) f64, all device; scale3 / translate3: HOST arrays of the reference's
 * rescaling to the hash grid, scale = (resolution - 1) / (bbox_max - bbox_min), translate = 0.5 - scale * bbox_min over the
 * vertices referenced by faces (inside_mesh.py:21-26); tri_workspace: 16 * F doubles (device scratch);
 * contains_out (N,) uint8: 1 = inside. */
int tamf_mesh_contains(const double* verts_dev, const int32_t* faces_dev, int32_t n_faces, const double* points_dev,
                       int64_t n_points, const double* scale3, const double* translate3, int32_t resolution,
                       double* tri_workspace_dev, uint8_t* contains_out_dev, void* stream);

/* Range guard of the split-fp16 mode.  The reference computes in fp32 (launch/sample.py:173); TAMF_PREC_F16X3 stores every
 * MFMA operand as two fp16 planes, so an ACTIVATION beyond +-65504 cannot be represented (weights are pre-scaled per tensor by a
 * power of two at tamf_finalize_weights - max |w| lands in [2^14, 2^15) - and only a non-finite weight is refused with
 * TAMF_ERR_RANGE, naming the tensor).  Activations are checked by the kernels that split them, which raise a sticky bit in the
 * status word of THEIR CONTEXT (a device allocation owned by the context; contexts on one device neither see nor clear each
 * other's bits, and creating a context clears nothing).  This call synchronises `stream`, returns the bits raised by this
 * context's launches since its last clear in *flags and, with clear != 0, resets them.  The Python module calls it after every
 * forward / sampling loop and re-runs the call in TAMF_PREC_F32 when the bit is set
 * (oakink2_tamf_amd/model/interaction_segment_mdm.py). */
int tamf_get_status_flags(tamf_ctx* ctx, uint32_t* flags, int32_t clear, void* stream);

/* Introspection for bench / profiles: number of kernels one denoiser step launches. */
int tamf_step_kernel_count(const tamf_ctx* ctx);
/* hipGraph bookkeeping of the sampling loop: how often a step sequence has been captured + instantiated on this context
 * (once per (B, T, n_steps): seeds, clip ranges and noise tensors replay the same executable graph) and how many graph
 * launches the last tamf_sample_loop call made (n_steps / steps per graph).  Either pointer may be NULL. */
int tamf_loop_stats(const tamf_ctx* ctx, int32_t* graph_captures, int32_t* graph_launches_last_loop);
/* Runs ONE denoiser step (DDPM update at t = n_steps/2, Philox noise; the sampler state is advanced by it) kernel by
 * kernel with hipEvents recorded on `stream` after every launch, synchronises, and reports per launch: elapsed
 * milliseconds, the algorithmic FLOPs of the reference work it stands for, and a name
 * (names_host: max_n x 48 chars).  Returns the number of launches (>= 0) or a negative tamf_status. */
int tamf_step_profile(tamf_ctx* ctx, int32_t max_n, float* ms_host, double* flops_host, char* names_host, void* stream);
/* The same for ONE tamf_refine call of an R context (arguments of tamf_refine, then those of tamf_step_profile); the first
 * interval also holds the input-packing kernel of the call. */
int tamf_refine_profile(tamf_ctx* ctx, const float* sample_pose_repr_dev, const float* h2o_dist_dev, float* out_dev, int32_t max_n,
                        float* ms_host, double* flops_host, char* names_host, void* stream);

/* The kernel-level test hooks, the guard-band / failure-injection switches and the kernel benchmarks (tamf_test_*, tamf_bench_*,
 * tamf_set_gemm_tuning) are NOT part of this surface: they are declared in include/tamf_hip_test.h and exist only in
 * libtamf_hip_hooks.so, the -DTAMF_TEST_HOOKS build of the same sources that tests/ and tools/ load.  libtamf_hip.so exports exactly
 * the functions declared above. */

#ifdef __cplusplus
}
#endif
#endif /* TAMF_HIP_H */
