/*
 * dgdm_hip.h - C-ABI of the MI355X (gfx950) guided-sampling library, libdgdm_hip.so.
 *
 * The reference (real-stanford/dgdm @ 2024_08_07) has no FFI layer: its hot path is Python
 * calling torch.nn modules.  This header is the boundary the replacement sits behind.  Each
 * entry point names the reference interface it stands in for (paths relative to the
 * reference tree); INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.  Plain C types only: device pointers, sizes, a hipStream_t passed as void*.
 *
 * Conventions
 *  - every function returns 0 on success, a negative DGDM_E* code otherwise;
 *    dgdm_last_error() gives the message (thread-local).  The Python shim turns the codes into
 *    the reference's exceptions (ValueError('opt obj not supported'), ...).
 *  - "dev" pointers are HIP device memory owned by the caller; "host" pointers are ordinary
 *    memory.  Model handles own their packed weights and workspaces in device memory.
 *  - all floating point is IEEE binary32; timesteps / point indices are int64 on the host side
 *    (as in the reference) and int32 on the device side.
 *  - launches go to the stream given; nothing here synchronises unless documented.
 */
#ifndef DGDM_HIP_H
#define DGDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGDM_OK            0
#define DGDM_EINVAL       -1   /* bad argument / unsupported shape                          */
#define DGDM_EKEY         -2   /* state_dict key missing or wrong size                      */
#define DGDM_EHIP         -3   /* HIP runtime error (message carries hipGetErrorString)     */
#define DGDM_EOBJECTIVE   -4   /* 'opt obj not supported'   (generator/diffusion.py:470)    */
#define DGDM_EMODE        -5   /* 'model type not supported' (generator/diffusion.py:502)   */
#define DGDM_ENODEVICE    -6   /* no gfx950 device visible                                  */

/* One host tensor of a reference-format state_dict (torch layout, contiguous). */
/* contraction dtypes of dgdm_unet1d_set_contraction_dtype / dgdm_guidance_set_contraction_dtype */
enum { DGDM_DTYPE_F32 = 0, DGDM_DTYPE_BF16 = 1, DGDM_DTYPE_F32_MFMA = 2, DGDM_DTYPE_F32_F16X3 = 3 };      /* (4, the six-bf16-product form of rounds 3-5, is retired: DGDM_EINVAL) */

typedef struct DgdmTensor {
    const char *name;     /* e.g. "linears.3.weight", "module."-prefix already stripped     */
    const void *data;     /* host memory                                                    */
    int64_t     numel;
    int32_t     dtype;    /* 0 = float32, 1 = int64                                         */
} DgdmTensor;

typedef struct DgdmUnet1d   DgdmUnet1d;    /* generator/diffusion_utils.py:123 ConditionalUnet1D        */
typedef struct DgdmDynamics DgdmDynamics;  /* dynamics/profile_forward_{2d,3d}.py ProfileForward{2,3}DModel */
typedef struct DgdmGuidance DgdmGuidance;  /* state of Diffusion.cond_fn for a batch of chains           */

/* Objective of Diffusion.deltas_to_objective (generator/diffusion.py:430-471) in gradient form:
 *   d objective / d delta_j = lin[j] + 2*quad[j]*delta_j                     (all but 'convergence')
 * 'convergence' sets use_rowcoef: d objective / d delta_0 of reference row r is rowcoef[r],
 * the signed multiplicity of r in the slicer() windows (dynamics/metrics.py:32-38), built by
 * dgdm_convergence_rowcoef().                                                                   */
typedef struct DgdmObjective {
    float   lin[3];
    float   quad[3];
    int32_t use_rowcoef;
    int32_t object;       /* index into the object bank given to dgdm_guidance_set_objects      */
} DgdmObjective;

/* ------------------------------------------------------------------ library */
int         dgdm_version(void);
const char *dgdm_last_error(void);
/* 0 when device `ordinal` exists and is gfx950; selects it for this thread. */
int         dgdm_device_init(int ordinal);
/* Fills *out (lin/quad/use_rowcoef) for a reference objective name; DGDM_EOBJECTIVE otherwise. */
int         dgdm_objective_from_name(const char *opt_obj, DgdmObjective *out);

/* ------------------------------------------------------------------ a7: noise-prediction net
 * ConditionalUnet1D(input_dim=1, global_cond_dim=0, down_dims, diffusion_step_embed_dim,
 * kernel_size=5, n_groups=8)  (generator/diffusion_utils.py:124-236).                          */
int dgdm_unet1d_create(DgdmUnet1d **out, const DgdmTensor *state_dict, int n_tensors,
                       const int32_t *down_dims, int n_down, int step_embed_dim, int kernel_size, int n_groups);
void dgdm_unet1d_destroy(DgdmUnet1d *m);
/* ConditionalUnet1D.forward (diffusion_utils.py:238-285): sample_dev [B][L] (input_dim 1),
 * timestep_dev [B] int32 (one per sample), eps_dev [B][L].                                      */
int dgdm_unet1d_forward(DgdmUnet1d *m, const float *sample_dev, const int32_t *timestep_dev, float *eps_dev,
                        int B, int L, void *stream);
/* Arithmetic of the convolutions with more than one input and output channel.  DGDM_DTYPE_F32 (default, the parity path):
 * float32-grade on the f16 matrix pipe - every float32 product as three f16 MFMA products on operands scaled by exact powers of two
 * (per convolution's weights, per sample and convolution input) and split in two f16 pieces, float32 accumulation (DESIGN_HISTORY.md 4.2;
 * closer to float64 than the float32 MFMA chain).  DGDM_DTYPE_F32_F16X3 selects the same form; DGDM_DTYPE_F32_MFMA the float32 MFMA
 * chain (also used where the split form's LDS slabs do not fit: L = 44, 46);
 * DGDM_DTYPE_BF16: weights and activations entering those convolutions rounded to bf16, float32 accumulation (BASELINE configs[4]).
 * GroupNorm, Mish, FiLM, residual adds, the Linear layers and the single-channel first/last convolutions are float32 in every mode.   */
int dgdm_unet1d_set_contraction_dtype(DgdmUnet1d *m, int dtype);
/* What a dgdm_unet1d_forward call with B samples of L control points actually runs: the DGDM_DTYPE_* of its convolutions
 * (DGDM_DTYPE_F32_F16X3, DGDM_DTYPE_F32_MFMA - also where the split form's slabs do not fit the LDS - or DGDM_DTYPE_BF16), + 16 when the
 * layer-by-layer batched form is used (large batches; the same bits as the per-sample kernel).  Negative: bad argument.               */
int dgdm_unet1d_effective_form(const DgdmUnet1d *m, int B, int L);

/* ------------------------------------------------------------------ a13/a14: scheduler step
 * noise_pred - sqrt(1-abar_t)*grad*scale  (generator/diffusion.py:575,645) followed by
 * DDIMScheduler.step(eta=0, clip_sample=True).prev_sample (diffusers 0.11.1; call sites
 * diffusion.py:201,256,576,647).  grad_dev may be NULL (unguided loops :193-201, :249-256).
 * grad_dev holds n_grad stacked gradients [n_grad][n]; their mean is used
 * (guided_sample_multi_object :640-644).  The four square roots are float32 values computed
 * by the host scheduler exactly as the reference computes them.                                 */
int dgdm_ddim_guided_step(const float *x_dev, const float *eps_dev, const float *grad_dev, int n_grad,
                          float *x_next_dev, int64_t n, float sqrt_abar_t, float sqrt_1m_abar_t,
                          float sqrt_abar_prev, float sqrt_1m_abar_prev, float guidance_scale, void *stream);
/* DDIMScheduler.add_noise (diffusion.py:145,185): out = sqrt_abar*x0 + sqrt_1m_abar*noise */
int dgdm_ddim_add_noise(const float *x0_dev, const float *noise_dev, float *out_dev, int64_t n,
                        float sqrt_abar, float sqrt_1m_abar, void *stream);

/* ------------------------------------------------------------------ a8-a12: dynamics models
 * kind 2: ProfileForward2DModel(W=256, params_ch, object_ch) (dynamics/profile_forward_2d.py:78-135)
 * kind 3: ProfileForward3DModel(W=256, params_ch)            (dynamics/profile_forward_3d.py:13-65)
 * BatchNorm layers are folded with their running statistics (the reference runs the classifier
 * in eval mode with frozen parameters: generator/train.py:91-92).                 */
int dgdm_dynamics_create(DgdmDynamics **out, int kind, const DgdmTensor *state_dict, int n_tensors,
                         int params_ch, int object_ch);
void dgdm_dynamics_destroy(DgdmDynamics *m);

/* ProfileForward2DModel.forward (profile_forward_2d.py:137-156) on `rows` arbitrary rows:
 * x_ctrl [rows][params_ch], x_ori [rows][1], x_pos [rows][2], t [rows] (already divided by
 * num_train_timesteps, as cond_fn passes it), object [rows][object_ch] -> logits [rows][3].     */
int dgdm_dyn2d_forward(DgdmDynamics *m, const float *x_ctrl_dev, const float *x_ori_dev, const float *x_pos_dev,
                       const float *t_dev, const float *object_dev, float *logits_dev, int rows, void *stream);

/* PointNet2.forward (dynamics/models/pointnet2.py:21-32): xyz_dev [rows][3][N] (channel-major, as
 * the reference passes it), FPS start indices of sa1 and sa2 for every row (the values
 * torch.randint draws at pointnet2_utils.py:83) in host memory -> emb_dev [rows][256].
 * Rows holding bit-identical clouds share the per-cloud work.                                   */
int dgdm_pointnet2_forward(DgdmDynamics *m, const float *xyz_dev, const int64_t *start_sa1_host,
                           const int64_t *start_sa2_host, float *emb_dev, int rows, int N, void *stream);

/* ProfileForward3DModel.forward (profile_forward_3d.py:67-86): x_ctrl [rows][3][params_ch]
 * (only channel 1 is read, :78), object xyz [rows][3][N], FPS starts as above.                  */
int dgdm_dyn3d_forward(DgdmDynamics *m, const float *x_ctrl_dev, const float *x_ori_dev, const float *x_pos_dev,
                       const float *t_dev, const float *xyz_dev, const int64_t *start_sa1_host,
                       const int64_t *start_sa2_host, float *logits_dev, int rows, int N, void *stream);

/* The index functions of dynamics/models/pointnet2_utils.py on their own, for arbitrary batches of clouds (the guided path reads
 * per-object tables built by the same device code instead; these exist so that the reference's names run on the device and can be
 * compared one to one).  Indices are int32 on the device.
 *   farthest_point_sample(xyz, npoint)        :71-92   xyz_dev [B][N][3] (N <= 1024); start_host [B] = the torch.randint draw of :83
 *   query_ball_point(radius, nsample, xyz, new_xyz) :95-115   radius_squared = float32(radius ** 2); out [B][S][nsample], first
 *                                             in-radius indices in ascending order, padded with the first (N if the ball is empty)
 *   square_distance(src, dst)                 :27-48   out [B][S][N], the expanded form in the reference's operation order
 *   index_points(points, idx)                 :51-68   points [B][N][C], idx [B][M] -> out [B][M][C]; an index outside [0, N) (where
 *                                                      the reference's indexing raises) yields NaN; empty inputs give empty results  */
int dgdm_farthest_point_sample(const float *xyz_dev, const int64_t *start_host, int B, int N, int npoint, int32_t *out_dev, void *stream);
int dgdm_query_ball_point(float radius_squared, int nsample, const float *xyz_dev, const float *new_xyz_dev, int B, int N, int S,
                          int32_t *out_dev, void *stream);
int dgdm_square_distance(const float *src_dev, const float *dst_dev, int B, int S, int N, float *out_dev, void *stream);
int dgdm_index_points(const float *points_dev, const int32_t *idx_dev, int B, int N, int M, int C, float *out_dev, void *stream);
/* PointNetSetAbstraction.forward on its own (pointnet2_utils.py:184-210, eval mode): after sample_and_group (the functions above) the shared
 * MLP is applied to the grouped rows layer by layer - y = act(x W^T + b) with the eval-mode BatchNorm folded into W, b by the caller;
 * wt_dev [K][N] (W transposed), x_dev [rows][ldx], y_dev [rows][ldy], act 0 none / 1 ReLU / 2 SiLU - and the result is reduced with the
 * max over each group's nsample rows (:206): x_dev [groups][nsample][C] -> out_dev [groups][C].                                       */
int dgdm_linear_act(const float *x_dev, int ldx, const float *wt_dev, const float *bias_dev, float *y_dev, int ldy, int rows, int K, int N,
                    int act, void *stream);
int dgdm_group_max(const float *x_dev, int64_t groups, int nsample, int C, float *out_dev, void *stream);

/* ------------------------------------------------------------------ a4-a6: guidance gradient
 * One DgdmGuidance serves up to max_chains independent chains (object x objective pairs) that
 * share B fingers, the (grid_size, num_pos, ori_range) pose grid and the timestep; each chain
 * has its own x [B][L].  This is Diffusion.cond_fn (generator/diffusion.py:473-504) with the
 * R = B*grid_size*num_pos^2 replicated rows evaluated without materialising them.              */
typedef struct DgdmGuidanceConfig {
    int32_t batch;            /* B fingers per chain                                            */
    int32_t grid_size;        /* G  (--grid_size)                                               */
    int32_t num_pos;          /* P  (--num_pos)                                                 */
    float   ori_lo, ori_hi;   /* ori_range                                                      */
    int32_t max_chains;
    int32_t num_train_timesteps;
    int32_t sub_batch_size;   /* 3-D: --sub_bs; it fixes which rows share a torch.randint call  */
    int32_t num_object_points;/* 3-D: N points per object cloud; 2-D: vertices per contour      */
    int32_t max_objects;
} DgdmGuidanceConfig;

int  dgdm_guidance_create(DgdmGuidance **out, DgdmDynamics *model, const DgdmGuidanceConfig *cfg);
void dgdm_guidance_destroy(DgdmGuidance *g);
/* Arithmetic of the trunk contractions inside dgdm_dyn{2,3}d_guidance_grad.  DGDM_DTYPE_F32 (default, the parity path) =
 * DGDM_DTYPE_F32_F16X3: float32 operands as two f16 pieces each after exact power-of-two scaling (per weight matrix, per tile row),
 * three f16 MFMAs per product with float32 accumulation (csrc/trunk_f16l.hip: 1.9e-7 rms of a 256-term contraction vs float64;
 * DESIGN_HISTORY.md 4.12).  DGDM_DTYPE_F32_MFMA: the k-ordered
 * float32 fma chain itself (v_mfma_f32_32x32x2_f32; 2.0e-7).  DGDM_DTYPE_BF16 (BASELINE configs[4]: "bf16 contractions, f32
 * accumulate"): weights and the activations/gradients entering a contraction rounded to bf16 (nearest even), float32 accumulation;
 * first-layer tables, biases, objective and row sums stay float32.  The reference has no such switch (it calls
 * torch.set_float32_matmul_precision('high'), generator/diffusion.py:102, which is a no-op on its CPU path).          */
int  dgdm_guidance_set_contraction_dtype(DgdmGuidance *g, int dtype);
/* Objects the chains refer to.  2-D: objects_dev [n][num_vertices][2] (flattened to object_ch as
 * cond_fn does, diffusion.py:485).  3-D: objects_dev [n][N][3]; builds the per-object PointNet++
 * tables (DESIGN_HISTORY.md §4) on `stream`.                                                            */
int  dgdm_guidance_set_objects(DgdmGuidance *g, const float *objects_dev, int n_objects, void *stream);
/* Rows of the pose grid per chain: R = B * G * P * P (reference row r = cell*B + b). */
int64_t dgdm_guidance_rows(const DgdmGuidance *g);
/* 3-D only: number of int64 FPS start indices one cond_fn call consumes per chain (= 2*R:
 * for every sub-batch, sa1's draw then sa2's, pointnet2_utils.py:83 via diffusion.py:495-498). */
int64_t dgdm_guidance_starts_per_call(const DgdmGuidance *g);

/* Diffusion.cond_fn for n_chains chains at once.
 *   x_dev        [n_chains][B][L]       current samples
 *   timestep     the (shared) integer diffusion timestep t; the model sees t/num_train_timesteps
 *   objectives   [n_chains] host array (objective + object index per chain)
 *   rowcoef_dev  [n_chains][R] or NULL  (only read for chains with use_rowcoef)
 *   starts_host  3-D: [n_chains][starts_per_call] int64 in the order the reference draws them;
 *                2-D: NULL
 *   grad_dev     [n_chains][B][L]       d sum(objective) / d x                                 */
int dgdm_dyn2d_guidance_grad(DgdmGuidance *g, const float *x_dev, int timestep, const DgdmObjective *objectives,
                             const float *rowcoef_dev, int n_chains, float *grad_dev, void *stream);
int dgdm_dyn3d_guidance_grad(DgdmGuidance *g, const float *x_dev, int timestep, const DgdmObjective *objectives,
                             const float *rowcoef_dev, const int64_t *starts_host, int n_chains, float *grad_dev,
                             void *stream);

/* The whole guided denoise loop in one call: Diffusion.guided_sample's loop body (generator/diffusion.py:570-576) for n_chains chains, or
 * guided_sample_multi_object's (:637-647) for n_chains chains that each average n_grad gradients - n_steps x [eps-net; cond_fn; guidance
 * combine; scheduler step] without returning to the host language in between.
 *   noise_dev    [B][L]             the start sample every chain begins from (:570)
 *   objectives   [n_grad * n_chains] host array, object-major: gradient j of chain k is entry j * n_chains + k (n_grad = 1: one per chain)
 *   rowcoef_dev  [n_grad * n_chains][R] or NULL
 *   starts_host  3-D: [n_steps][n_grad * n_chains][starts_per_call] int64 in the reference's draw order per gradient chain; 2-D: NULL
 *   timesteps    [n_steps] host; coef [n_steps][4] host = sqrt(abar_t), sqrt(1 - abar_t), sqrt(abar_prev), sqrt(1 - abar_prev) per step
 *   scales       [n_chains] host: the classifier scale of every chain (:549-560)
 *   x_out_dev    [n_chains][B][L]   the final samples
 * The first eps-net call is evaluated once for all chains (they all hold the start sample).  Equivalent, bit for bit, to calling
 * dgdm_unet1d_forward / dgdm_dyn{2,3}d_guidance_grad / dgdm_ddim_guided_step step by step.                                             */
int dgdm_guided_chains_run(DgdmUnet1d *unet, DgdmGuidance *g, const float *noise_dev, int n_chains, int n_grad,
                           const DgdmObjective *objectives, const float *rowcoef_dev, const int64_t *starts_host,
                           const int32_t *timesteps, const float *coef, const float *scales, int n_steps, float *x_out_dev, void *stream);

/* Forward-only sweep of Diffusion.get_convergence_centers (diffusion.py:506-531): G orientations,
 * pos = 0, t = 0, rows r = g*B + b.  logits_dev [n_chains][B*G][3].  starts_host (3-D) holds
 * 2*B*G indices per chain in draw order with the sub_batch_size partition of :524-526.           */
int dgdm_guidance_orientation_sweep(DgdmGuidance *g, const float *x_dev, const int32_t *object_of_chain,
                                    const int64_t *starts_host, int n_chains, float *logits_dev, void *stream);

/* Host helper for 'convergence': rowcoef_host[r] for one chain from its centers [B]
 * (deltas_to_objective :445-452 + slicer; `rows_in_call` is R for 2-D and the sub-batch
 * partition is applied for 3-D exactly as cond_fn :494-499 does).                               */
int dgdm_convergence_rowcoef(const int64_t *centers_host, int n_centers, int grid_size, int num_pos,
                             int64_t total_rows, int64_t sub_batch_size /* 0 = no sub-batching */, float *rowcoef_host);

/* ------------------------------------------------------------------ the path's random input: FPS start draws
 * farthest_point_sample draws its start index with torch.randint(0, N, (rows,)) on torch's CPU default generator
 * (dynamics/models/pointnet2_utils.py:83), twice per classifier call and sub-batch.  These host functions replay that generator
 * (at::mt19937; randint = output % range) on its own state blob - `state` is what torch.get_rng_state() / Generator.get_state()
 * returns (5056 bytes) - and advance the blob exactly as the torch calls would, so that the draws can be made (or skipped) from a
 * worker thread ahead of the launches and the blob handed back with torch.set_rng_state().                                    */
int dgdm_torch_rng_seed(uint8_t *state, int64_t state_bytes, uint64_t seed);             /* torch.Generator().manual_seed(seed)        */
int dgdm_torch_rng_randint(uint8_t *state, int64_t state_bytes, uint32_t high, int64_t n,
                           int64_t *out_host /* NULL: the n draws are skipped */);      /* torch.randint(0, high, (n,))               */
/* The draws of n_calls consecutive classifier calls over `rows` rows each (diffusion.py:495-498): per call and sub-batch of n rows,
 * sa1's n draws in [0, num_points) then sa2's n draws in [0, 512) - the starts_host layout of dgdm_dyn3d_guidance_grad.          */
int dgdm_torch_rng_fps_starts(uint8_t *state, int64_t state_bytes, int num_points, int64_t sub_batch_size, int64_t rows,
                              int64_t n_calls, int64_t *out_host /* NULL: skipped */,
                              int64_t out_call_stride /* elements between the outputs of consecutive calls; 0 = 2*rows */);

/* ------------------------------------------------------------------ (f) next: finger-geometry decode
 * What the reference does on the host, gripper by gripper, between the sampler and the simulator.  Both maps are linear in
 * the control values, so the library applies one constant matrix (built in double precision) to the whole batch on the device.
 *
 * 2-D  sim_test_batch (dynamics/sim_test_mj.py:257-262): y = 0.03 p - 0.015 over x = linspace(-0.12, 0.12, num_ctrl/2) for the
 *      left finger (first half of the control values) and the right one; generate_gripper / generate_finger_shape
 *      (assets/finger_sampler.py:7-12,39-51): scipy CubicSpline (not-a-knot) on linspace(x_0, x_last, num_points).
 *      samples_dev [batch][num_ctrl] -> curve_dev [batch][2 fingers][num_points][2] = (x, y) in metres.
 * 3-D  sim_test_batch_3d (dynamics/sim_test_mj_3d.py:236-237): y = 0.05 p - 0.05 on the 7 x 3 control net of
 *      generate_3d_ctrlpts (assets/finger_3d.py:77-81: x = linspace(-0.12, 0.12, 7), z = linspace(0, 0.12, 3), control point
 *      (i, j) = y[3 i + j]); generate_3d_finger_vertices (assets/finger_3d.py:60-68): geomdl BSpline.Surface of degree (3, 2),
 *      generate_knot_vector (clamped, uniform), sample_size x sample_size evaluation points in u-major order.
 *      samples_dev [batch][42] -> surface_dev [batch][2 fingers][sample_size^2][3] = (x, y, z) in metres.
 * The affine map from sampler units to metres is an argument (y = scale * p + offset): DGDM_DECODE_2D_* / DGDM_DECODE_3D_* are
 * the reference's constants; scale 1, offset 0 decodes control values that are already in metres.                              */
#define DGDM_DECODE_2D_SCALE 0.03f
#define DGDM_DECODE_2D_OFFSET (-0.015f)
#define DGDM_DECODE_3D_SCALE 0.05f
#define DGDM_DECODE_3D_OFFSET (-0.05f)
int dgdm_finger_decode_2d(const float *samples_dev, int batch, int num_ctrl, int num_points, float scale, float offset,
                          float *curve_dev, void *stream);
int dgdm_finger_decode_3d(const float *samples_dev, int batch, int num_ctrl, int sample_size, float scale, float offset,
                          float *surface_dev, void *stream);

/* ------------------------------------------------------------------ measurement hooks
 * When enabled, the launches of every stage of the path are bracketed by hipEvents on the stream they are launched on.
 * dgdm_prof_read_stage synchronises those events and returns, for one stage, the number of bracketed regions, their total
 * milliseconds and the algorithmic work they covered (FLOPs for the MFMA-bound stages, 0 where the host cannot know it).
 * dgdm_prof_read = dgdm_prof_read_stage(DGDM_STAGE_TRUNK): the dominant kernel (DESIGN_HISTORY.md §5).                               */
enum {
    DGDM_STAGE_TRUNK = 0,   /* trunk_kernel / trunk_bf16_kernel: fused dynamics trunk forward + backward (work = FLOPs, real rows only) */
    DGDM_STAGE_UNET,        /* unet_kernel: one eps-net forward (work = useful FLOPs)                                          */
    DGDM_STAGE_XOBJ,        /* 3-D: FPS-start upload + per-row PointNet++ embedding gather (xobj kernels)                      */
    DGDM_STAGE_TABLES,      /* dgdm_guidance_set_objects: per-object PointNet++ tables (3-D) / object encoder (2-D)            */
    DGDM_STAGE_GUIDE_MISC,  /* small kernels of cond_fn around the trunk: encoders, first-layer tables, partial-sum fold       */
    DGDM_STAGE_DDIM,        /* guidance combine + scheduler step                                                               */
    DGDM_STAGE_COUNT
};
int dgdm_prof_enable(int on);
/* Test hook (3-D): where a reference row's PointNet++ embedding comes from (DESIGN_HISTORY.md §4.3) - mode 0: default: one workgroup per
 * (chain, s1) group with the variant's crowded-centre rows staged in LDS, until the objects of the last dgdm_guidance_set_objects have
 * served more than 5 guidance calls; from then on the per-object embedding table X[s1][start point] (built at that moment) and no gather
 * kernel at all; 5: the NEXT set_objects builds the embedding tables right away; 3: always the group kernel; 2: the per-row table kernel;
 * 1: every row runs its own FPS(128) instead of reading the per-object table of order-independent sequences; 4: like 0, and the NEXT
 * set_objects builds the crowded centres' sa2 features with per-(variant, centre) global gathers instead of the LDS-staged kernel.
 * Results must be identical.  Also reports, per object of the bank, whether the table of FPS(128) sequences is admissible
 * (out_fast_ok[n_objects], may be NULL).                                                                                                */
int dgdm_guidance_debug_fps_path(DgdmGuidance *g, int mode, int32_t *out_fast_ok);
/* Test hook: the per-tile partial sums of d objective / d z1 (z1 = the first trunk layer's pre-activation, BatchNorm folded) that the
 * last dgdm_dyn{2,3}d_guidance_grad call produced: out_dev [n_chains * B * tiles_per_finger][width], tile index
 * (chain * B + b) * tiles_per_finger + t = cells 32 t .. 32 t + 31 of finger b.  A ReLU whose float32 pre-activation has the other sign
 * than in exact arithmetic changes exactly one tile, which is what tests/test_gpu_fullgrid.py looks at.  out_dev may be NULL (sizes only). */
int dgdm_guidance_debug_partials(DgdmGuidance *g, int n_chains, float *out_dev, int32_t *tiles_per_finger, int32_t *width, void *stream);
int dgdm_prof_read(int64_t *launches, double *total_ms, double *total_flops);
int dgdm_prof_read_stage(int stage, int64_t *launches, double *total_ms, double *total_work);
/* Unit-test hook for the register-resident MFMA chain (csrc/mfma_chain.h): one 32-row tile through one 256 -> 256 layer,
 * Y = X W^T + bias.  W_host [256][256] row-major, bias_host [256] (host memory); X_dev, Y_dev [32][256] (device).  Synchronises. */
int dgdm_debug_chain_layer(const float *W_host, const float *bias_host, const float *X_dev, float *Y_dev, void *stream);

/* Test hook for the index work of PointNet++ on ONE cloud xyz_dev [N][3] (128 <= N <= 1024), through the same device code as the
 * production kernels; every output is int32 device memory; synchronises.
 *   fps512_dev [N][512], fps128_dev [N][128]: farthest_point_sample (dynamics/models/pointnet2_utils.py:71-92) from every start index
 *                                             (row v = the sequence torch.randint's draw v would give);
 *   fps128_flags_dev [N]: 1 = that 128-sequence met an exact distance tie between different coordinates (it is then order-dependent
 *                                             and the production path re-runs FPS per row instead of using the table);
 *   ball1_dev [N][32]: query_ball_point(0.2, 32, xyz, centre = point p) (pointnet2_utils.py:95-115), padded with the first index;
 *   ball2_dev [N][64] + ball2_count_dev [N]: query_ball_point(0.4, 64, ...) for centre point c when the candidates are scanned in the
 *                                             order perm_dev[0..perm_len) (sa2 sees the cloud re-ordered by sa1's FPS); entries past the
 *                                             count are -1 (the reference pads with the first);
 *   crowded_dev [N]: 1 = more than 64 points in that ball (DESIGN_HISTORY.md 4.3).                                                          */
int dgdm_debug_pointnet_indices(DgdmDynamics *m, const float *xyz_dev, int N, const int32_t *perm_dev, int perm_len,
                                int32_t *fps512_dev, int32_t *fps128_dev, int32_t *fps128_flags_dev, int32_t *ball1_dev,
                                int32_t *ball2_dev, int32_t *ball2_count_dev, int32_t *crowded_dev, void *stream);

/* ------------------------------------------------------------------ (f) rank 4: training the 2-D dynamics model
 * Trainer (dynamics/trainer.py:16-106) for ProfileForward2DModel: parameters, gradients and torch.optim.Adam state
 * (trainer.py:46: betas (0.9, 0.95), weight_decay) live on the device.  state_dict: the model's tensors ("module."-prefix
 * stripped), BatchNorm running statistics included.                                                                        */
typedef struct DgdmTrainer2d DgdmTrainer2d;
int dgdm_trainer2d_create(DgdmTrainer2d **out, const DgdmTensor *state_dict, int n_tensors, int params_ch, int object_ch,
                          float beta1, float beta2, float eps, float weight_decay);
void dgdm_trainer2d_destroy(DgdmTrainer2d *m);
/* Trainer.step (trainer.py:53-103, the branch without sub-batches) on `rows` rows when train != 0: noisy control points
 * sqrt_abar[r]*ctrl + sqrt_1m_abar[r]*noise (DDIMScheduler.add_noise, :75-79; noise_dev null = ctrl as given), model forward in
 * training mode (BatchNorm1d batch statistics; running statistics updated), loss = nn.MSELoss()(pred, score), backward, one Adam
 * update with learning rate lr.  train == 0 is Trainer.inference (:108-146): eval-mode forward (running statistics) and the loss,
 * nothing is updated.  t_dev [rows] = timesteps / num_train_timesteps (:80).  pred_dev [rows][3]; loss_host (optional) receives
 * the loss and makes the call synchronous, like loss.item().  Deterministic: same inputs, same bits.                              */
int dgdm_trainer2d_step(DgdmTrainer2d *m, const float *ctrl_dev, const float *noise_dev, const float *sqrt_abar_dev,
                        const float *sqrt_1m_abar_dev, const float *t_dev, const float *ori_dev, const float *pos_dev,
                        const float *object_dev, const float *score_dev, int64_t rows, float lr, int train, float *pred_dev,
                        float *loss_host, void *stream);
/* Optional hints for the NEXT dgdm_trainer2d_step / _forward_backward call only: which encoder inputs repeat over the rows.  The
 * result is the same function of the inputs (only the float32 summation order of the two encoders' gradients changes); the time and
 * object encoders then run on their distinct inputs instead of on every row (1.15 M rows -> 15 and 128 at the shipped configuration).
 *   t_index_dev [rows] + t_values_dev [n_t], n_t <= 32: row r's t is t_values[t_index[r]] (t_dev of the step call is then ignored);
 *   rows_per_object > 1: object_dev's rows come in runs of that many identical rows (dynamics/main.py:34 builds them so); rows must
 *                        be a multiple of it, otherwise the hint is ignored.                                                        */
typedef struct DgdmTrainGroups {
    const int32_t *t_index_dev;
    const float   *t_values_dev;
    int32_t        n_t;
    int32_t        rows_per_object;
} DgdmTrainGroups;
int dgdm_trainer2d_set_groups(DgdmTrainer2d *m, const DgdmTrainGroups *g);

/* Data-parallel training, one process per GPU (replaces nn.DataParallel around the model, dynamics/trainer.py:41-43, whose replicas
 * each normalise with the batch statistics of THEIR chunk and whose gradients add up on the source device):
 *   dgdm_trainer2d_forward_backward  this rank's `rows` of a batch of `total_rows`: forward in training mode on these rows' statistics,
 *                                    d loss / d pred = 2 (pred - score) / (3 total_rows), backward; no update.  *loss_host = this rank's
 *                                    share of the loss (the shares add up to the batch loss).
 *   dgdm_trainer2d_gradients         the flat gradient buffer (dgdm_trainer2d_gradient_count floats, layout private to the library:
 *                                    only elementwise reductions are meaningful) device -> flat_dev, or flat_dev -> device (to_trainer);
 *                                    between the two calls: all-reduce(sum) over RCCL.
 *   dgdm_trainer2d_apply             one Adam update from the gradient buffer.
 * dgdm_trainer2d_step == forward_backward(rows, rows) + apply.                                                                      */
int dgdm_trainer2d_forward_backward(DgdmTrainer2d *m, const float *ctrl_dev, const float *noise_dev, const float *sqrt_abar_dev,
                                    const float *sqrt_1m_abar_dev, const float *t_dev, const float *ori_dev, const float *pos_dev,
                                    const float *object_dev, const float *score_dev, int64_t rows, int64_t total_rows, float *pred_dev,
                                    float *loss_host, void *stream);
int64_t dgdm_trainer2d_gradient_count(const DgdmTrainer2d *m);
int dgdm_trainer2d_gradients(DgdmTrainer2d *m, float *flat_dev, int64_t numel, int to_trainer, void *stream);
int dgdm_trainer2d_apply(DgdmTrainer2d *m, float lr, void *stream);
/* BatchNorm running statistics, 8 x [mean | var] x 256 floats, device -> flat_dev or flat_dev -> device (to_trainer): every rank updates
 * them from its own chunk; nn.DataParallel keeps replica 0's, so rank 0's are broadcast to the others after each data-parallel step.   */
int dgdm_trainer2d_running_stats(DgdmTrainer2d *m, float *flat_dev, int64_t numel, int to_trainer, void *stream);
/* Copies into the host buffers of `tensors` (matched by name; data is written despite the const): which = 0 the state_dict
 * (parameters + running statistics: Trainer.save_checkpoint, trainer.py:105-106), 1 the gradients of the last step, 2 / 3 Adam's
 * exp_avg / exp_avg_sq.                                                                                                           */
int dgdm_trainer2d_export(DgdmTrainer2d *m, int which, DgdmTensor *tensors, int n_tensors);
int64_t dgdm_trainer2d_steps(const DgdmTrainer2d *m);      /* training steps taken = BatchNorm's num_batches_tracked increment */

/* ------------------------------------------------------------------ (f) rank 4: training the eps-net
 * Diffusion.get_stats / training_step (generator/diffusion.py:126-177) for the ConditionalUnet1D of generator/train.py:80
 * (generator/diffusion_utils.py:123-285; input_dim 1, down_dims [128, 256], kernel 5, 8 groups), torch.optim.Adam
 * (diffusion.py:711-714) and the EMA copy of diffusers' EMAModel (diffusion.py:716-724).  Parameters, gradients, both Adam moments and the
 * EMA copy live on the device in the state_dict's own tensor layouts (keys as ConditionalUnet1D.state_dict() gives them).          */
typedef struct DgdmUnetTrainer DgdmUnetTrainer;
int dgdm_unet_trainer_create(DgdmUnetTrainer **out, const DgdmTensor *state_dict, int n_tensors, int num_points, const int32_t *down_dims,
                             int n_down, int diffusion_step_embed_dim, int kernel_size, int n_groups, float beta1, float beta2, float eps,
                             float weight_decay);
void dgdm_unet_trainer_destroy(DgdmUnetTrainer *m);
/* One training step on `samples` samples: noisy = sqrt_abar[s] * x0 + sqrt_1m_abar[s] * noise (DDIMScheduler.add_noise, :144-148),
 * noise_pred = eps-net(noisy, timesteps) (:151-160), loss = F.mse_loss(noise_pred, noise) (:164), backward, one Adam update with
 * learning rate lr.  x0_dev, noise_dev [samples][num_points]; sqrt_abar_dev, sqrt_1m_abar_dev [samples]; timesteps_dev [samples]
 * int64; pred_dev (optional) [samples][num_points] receives noise_pred; loss_host (optional) receives the loss and makes the call
 * synchronous, like loss.item().  Deterministic: same inputs, same bits.                                                           */
int dgdm_unet_trainer_step(DgdmUnetTrainer *m, const float *x0_dev, const float *noise_dev, const float *sqrt_abar_dev,
                           const float *sqrt_1m_abar_dev, const int64_t *timesteps_dev, int samples, float lr, float *pred_dev,
                           float *loss_host, void *stream);
/* The same without the update: forward (+ backward when backward != 0) on this rank's `samples` of a batch of `total_samples`
 * (d loss / d pred = 2 (pred - noise) / (total_samples num_points); *loss_host = this rank's share of the batch loss).  With
 * dgdm_unet_trainer_gradients (flat gradient buffer device -> flat_dev, or flat_dev -> device scaled by `scale` when to_trainer) and
 * dgdm_unet_trainer_apply (one Adam update) this is data-parallel training, one process per GPU, with an RCCL all-reduce in between
 * (Lightning's DDP averages the ranks' gradients: total_samples = samples, scale = 1 / world).                                      */
int dgdm_unet_trainer_forward_backward(DgdmUnetTrainer *m, const float *x0_dev, const float *noise_dev, const float *sqrt_abar_dev,
                                       const float *sqrt_1m_abar_dev, const int64_t *timesteps_dev, int samples, int64_t total_samples,
                                       int backward, float *pred_dev, float *loss_host, void *stream);
int64_t dgdm_unet_trainer_gradient_count(const DgdmUnetTrainer *m);
int dgdm_unet_trainer_gradients(DgdmUnetTrainer *m, float *flat_dev, int64_t numel, int to_trainer, float scale, void *stream);
int dgdm_unet_trainer_apply(DgdmUnetTrainer *m, float lr, void *stream);
/* EMAModel.step (diffusers 0.11.1, called from on_train_batch_end, diffusion.py:716-720): ema = ema * decay + one_minus_decay * param
 * for every parameter; the decay schedule (power, update_after_step) is the caller's.                                              */
int dgdm_unet_trainer_ema_step(DgdmUnetTrainer *m, float decay, float one_minus_decay, void *stream);
/* which = 0 parameters, 1 gradients of the last backward, 2 / 3 Adam's exp_avg / exp_avg_sq, 4 the EMA copy; tensors are matched by
 * name and written (export: data is written despite the const) or read (import; adam_steps >= 0 also sets Adam's step count -
 * resuming from a checkpoint).                                                                                                      */
int dgdm_unet_trainer_export(DgdmUnetTrainer *m, int which, DgdmTensor *tensors, int n_tensors);
int dgdm_unet_trainer_import(DgdmUnetTrainer *m, int which, const DgdmTensor *tensors, int n_tensors, int64_t adam_steps);
int64_t dgdm_unet_trainer_steps(const DgdmUnetTrainer *m);      /* Adam updates taken */

/* ------------------------------------------------------------------ (f) rank 4: training the 3-D dynamics model
 * Trainer.step / Trainer.inference (dynamics/trainer.py:53-146) for ProfileForward3DModel (dynamics/profile_forward_3d.py:13-86):
 * PointNet++ (dynamics/models/pointnet2.py:11-32, pointnet2_utils.py:169-210) with BatchNorm2d in TRAINING mode, the trunk with
 * BatchNorm1d batch statistics, nn.MSELoss, backward with every weight gradient, torch.optim.Adam (trainer.py:46) - evaluated as
 * written, one row = one (control points, pose, cloud) item.  state_dict: the model's tensors ("module."-prefix stripped),
 * running statistics included; time_encoder (constructed, never called by forward) is carried unchanged.                           */
typedef struct DgdmTrainer3d DgdmTrainer3d;
int dgdm_trainer3d_create(DgdmTrainer3d **out, const DgdmTensor *state_dict, int n_tensors, int params_ch, int num_object_points,
                          float beta1, float beta2, float eps, float weight_decay);
void dgdm_trainer3d_destroy(DgdmTrainer3d *m);
/* One forward / backward / optimizer.step() of the loop body of trainer.py:83-92 (one slice of --use_sub_batch; the whole batch when
 * it is not sub-batched) when train != 0; train == 0: Trainer.inference's eval-mode forward + loss (:108-146).
 *   ctrl1_dev [rows][params_ch]   channel 1 of the control points, the only one the model reads (profile_forward_3d.py:77) and the
 *                                 only one trainer.py:68 adds noise to: noisy = sqrt_abar[r] * ctrl1 + sqrt_1m_abar[r] * noise (null: as given)
 *   t_dev [rows] = timesteps / num_train_timesteps (:80); ori_dev [rows][1]; pos_dev [rows][2]; score_dev [rows][3]
 *   xyz_dev [rows][num_object_points][3]  the clouds, point-major (the model's (rows, 3, N) input transposed)
 *   start_sa1_host / start_sa2_host [rows] int64: the FPS start draws of pointnet2_utils.py:83 for sa1 and sa2, in the order the
 *                                 reference makes them (torch.randint on the CPU generator, sa1 then sa2, per forward)
 * pred_dev (optional) [rows][3]; loss_host (optional) receives the loss and makes the call synchronous.  Deterministic.
 * Memory: 68 MB of activations and gradients per row (PointNet++ as written): rows <= 4096.                                        */
int dgdm_trainer3d_step(DgdmTrainer3d *m, const float *ctrl1_dev, const float *noise_dev, const float *sqrt_abar_dev,
                        const float *sqrt_1m_abar_dev, const float *t_dev, const float *ori_dev, const float *pos_dev, const float *xyz_dev,
                        const int64_t *start_sa1_host, const int64_t *start_sa2_host, const float *score_dev, int64_t rows, float lr,
                        int train, float *pred_dev, float *loss_host, void *stream);
/* Data-parallel training, one process per GPU, with nn.DataParallel's semantics (dynamics/trainer.py:41-43 wraps the 3-D model as well):
 *   dgdm_trainer3d_forward_backward  this rank's `rows` of a batch (or --use_sub_batch slice) of `total_rows`: forward in training mode on
 *                                    these rows' batch statistics (running statistics updated), loss share sum / (3 total_rows), backward;
 *                                    no update
 *   dgdm_trainer3d_gradients         the flat gradient buffer (dgdm_trainer3d_gradient_count floats, layout private to the library) read
 *                                    (to_trainer = 0) or written back after the all-reduce (1)
 *   dgdm_trainer3d_apply             one Adam update from the gradient buffer
 *   dgdm_trainer3d_running_stats     BatchNorm running means / variances of all 13 layers as one flat buffer
 *                                    (dgdm_trainer3d_running_stats_count floats): rank 0's are broadcast after every step
 * dgdm_trainer3d_step(train = 1) == forward_backward(rows, rows) + apply.                                                              */
int dgdm_trainer3d_forward_backward(DgdmTrainer3d *m, const float *ctrl1_dev, const float *noise_dev, const float *sqrt_abar_dev,
                                    const float *sqrt_1m_abar_dev, const float *t_dev, const float *ori_dev, const float *pos_dev,
                                    const float *xyz_dev, const int64_t *start_sa1_host, const int64_t *start_sa2_host, const float *score_dev,
                                    int64_t rows, int64_t total_rows, float *pred_dev, float *loss_host, void *stream);
int64_t dgdm_trainer3d_gradient_count(const DgdmTrainer3d *m);
int dgdm_trainer3d_gradients(DgdmTrainer3d *m, float *flat_dev, int64_t numel, int to_trainer, void *stream);
int dgdm_trainer3d_apply(DgdmTrainer3d *m, float lr, void *stream);
int64_t dgdm_trainer3d_running_stats_count(const DgdmTrainer3d *m);
int dgdm_trainer3d_running_stats(DgdmTrainer3d *m, float *flat_dev, int64_t numel, int to_trainer, void *stream);
/* which = 0 the state_dict (parameters + running statistics), 1 gradients of the last step, 2 / 3 Adam's exp_avg / exp_avg_sq */
int dgdm_trainer3d_export(DgdmTrainer3d *m, int which, DgdmTensor *tensors, int n_tensors);
int64_t dgdm_trainer3d_steps(const DgdmTrainer3d *m);      /* training steps taken = num_batches_tracked increment */
/* Test hook: an intermediate tensor of the last call, `count` 32-bit words device -> out_dev.  which: 0 trunk input [rows][800] =
 * [object embedding 256 | gripper 256 | pose 27 | time 256 | 0 x 5], 1 sa1 output [rows*512][128], 2 sa2 output [rows*128][256],
 * 3 sa1 grouped coordinates [rows*512*32][4], 4 sa1 first conv [..][64], 5 sa1 centres [rows][512][3], 6 sa1 FPS indices (int32),
 * 7 sa1 ball lists (int32), 8 gradient of the trunk input [rows][800], 9 predictions [rows][4].                                    */
int dgdm_trainer3d_debug_read(DgdmTrainer3d *m, int which, void *out_dev, int64_t count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DGDM_HIP_H */
