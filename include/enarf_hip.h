/*
 * enarf_hip.h - C ABI of libenarf_hip.so: the MI355X (gfx950) implementation of ENARF-GAN's per-ray
 * rendering hot path. Plain pointers and sizes only; every pointer named "device" is a HIP device
 * pointer to contiguous fp32 (unless stated), every call is asynchronous on `stream` (a hipStream_t
 * passed as void*, NULL = the null stream) and returns 0 on success, a negative ENARF_ERR_* for an
 * argument it rejects, or a positive hipError_t. enarf_last_error() gives the message (thread local).
 *
 * Each entry point names the reference interface (nogu-atsu/ENARF-GAN, file:line) it replaces; the
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef ENARF_HIP_H
#define ENARF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ENARF_ABI_VERSION 4

#define ENARF_ERR_ARG          (-1)   /* null pointer / non-positive size / bad enum */
#define ENARF_ERR_UNSUPPORTED  (-2)   /* valid in the reference but not implemented here (message says what) */

#define ENARF_FEAT_DIM   32           /* models/narf.py:22 */
#define ENARF_HIDDEN     64           /* models/narf.py:77 StyledMLP(32, 64, 4) */
#define ENARF_MAX_PARTS  32           /* validity bit masks are uint32; the reference uses 23 or 24 */
#define ENARF_MAX_JOINTS 32

/* interpolation / padding enums of cuda_extension/triplane_sampler.py:7-16 */
#define ENARF_INTERP_BILINEAR 0
#define ENARF_INTERP_NEAREST  1
#define ENARF_PAD_ZEROS       0
#define ENARF_PAD_BORDER      1
#define ENARF_PAD_REFLECTION  2

/* origin_location of libraries/NARF/pose_utils.py:129-148 */
#define ENARF_ORIGIN_CENTER       0
#define ENARF_ORIGIN_CENTER_FIXED 1
#define ENARF_ORIGIN_CENTER_HEAD  2

/* arithmetic of the density/colour MLP (libraries/NeRF/net.py:10-27) */
#define ENARF_MARCH_AUTO   0   /* by shape, from measurements (enarf_render.hip): the default */
#define ENARF_MARCH_RAY    1   /* one 4-wave workgroup marches one ray at a time (3 workgroups per CU) */
#define ENARF_MARCH_TASK   2   /* one 12-wave workgroup per CU, several rays in flight, 16-sample tiles claimed as tasks */

#define ENARF_MLP_F32     0   /* v_mfma_f32_16x16x4_f32: exact fp32 products (bitwise an fmaf chain) */
#define ENARF_MLP_BF16X3  1   /* 3-term split-bf16 on v_mfma_f32_16x16x32_bf16: ~1e-5 relative */
#define ENARF_MLP_BF16    2   /* plain bf16 operands, fp32 accumulate: ~4e-3 relative */
#define ENARF_MLP_F16X3   3   /* 3-term split-fp16 on v_mfma_f32_16x16x32_f16: ~1e-6 relative, same MFMA count as
                                 BF16X3; operands saturate at +-65504 per half (|x| < 1.3e5 stays finite) */

typedef void *enarf_stream_t;

int         enarf_abi_version(void);
int         enarf_version(void);          /* the same number under the name SURVEY.md 8(b) lists */
const char *enarf_last_error(void);

/* Sticky status of the CURRENT device, written by kernels that had to give up on their work (the entry points themselves
 * are asynchronous and have returned 0 long before). One word of pinned host memory per device: reading it costs no
 * synchronisation and shows every launch that has completed; synchronise the stream first to cover a particular launch.
 * clear != 0 resets the word after reading. Returns 0 and *flags (0 = nothing to report), or a hipError_t when the
 * status word could not be set up. The reference has no counterpart: its kernels cannot give up (kernel.cu:245-246). */
#define ENARF_STATUS_MARCH_WATCHDOG 1u   /* enarf_render_fwd / _step_fwd with the task march (ENARF_MARCH_TASK, or AUTO for
                                            Nc / Nf > 64 at B == 1): a wave found no work for ~0.3 s although rays were in
                                            flight and ended the launch; that launch's outputs are INCOMPLETE */
int enarf_device_status(unsigned int *flags, int clear);

/* ---------------------------------------------------------------------------------------------
 * a1. The TriplaneSampler operator.
 * Replaces triplane_sampler_cuda.triplane_sampler_forward / _backward
 * (cuda_extension/TriplaneSampler.cpp:15-24, :26-52; kernels TriplaneSampler_kernel.cu:13-92, :94-229).
 *   out[b,c,i] = sum_{p in xy,yz,zx} sample(input[b, p*C + c], (grid[b,i,p], grid[b,i,(p+1)%3]))
 * input (B, 3C, H, W) NCHW, grid (B, n_pts, 3) with n_pts = h*w, out (B, C, n_pts).
 * Nearest mode keeps the reference's overwrite semantics (kernel.cu:84-86): the last plane (zx) wins.
 * `workspace` (device, enarf_triplane_sample_workspace_bytes() bytes, may be NULL) lets the operator
 * re-lay the planes channel-last once per call; without it a slower direct-NCHW kernel runs.
 * Backward: grad_input (same shape as input) must be zero-filled by the caller, as in the reference
 * (TriplaneSampler.cpp:33-39); either grad pointer may be NULL (= output_mask false). With a workspace of
 * enarf_triplane_sample_bwd_workspace_bytes() bytes (non-zero for C = 32, used in bilinear mode) the gradient is
 * accumulated channel-last with whole-line atomics and folded back into grad_input; without it the direct kernel
 * scatters 4-byte atomics into the NCHW planes (several times slower).
 * --------------------------------------------------------------------------------------------- */
size_t enarf_triplane_sample_workspace_bytes(int B, int C, int H, int W);
size_t enarf_triplane_sample_bwd_workspace_bytes(int B, int C, int H, int W);
int enarf_triplane_sample_fwd(const float *input, const float *grid, float *out,
                              int B, int C, int H, int W, long long n_pts,
                              int interpolation_mode, int padding_mode, int align_corners,
                              void *workspace, enarf_stream_t stream);
int enarf_triplane_sample_bwd(const float *grad_out, const float *input, const float *grid,
                              float *grad_input, float *grad_grid,
                              int B, int C, int H, int W, long long n_pts,
                              int interpolation_mode, int padding_mode, int align_corners,
                              void *workspace, enarf_stream_t stream);

/* The same gather for the rest of libraries/triplane/sampling.py's API surface (sample_feature with reduction "prod",
 * with batch_idx; sample_triplane_part_prob; sample_weighted_feature_v2), which the fused kernels make unnecessary on the
 * render path but callers of the reference API may use stand-alone:
 *   reduction   ENARF_PLANES_SUM: out (B, C, n) as above; ENARF_PLANES_SEPARATE: out (B, 3, C, n), the planes' samples
 *               side by side (the caller applies sigmoid / product / clamp, sampling.py:43-48);
 *   point_image NULL, or device (n_pts,) int32 with a grid of batch 1: point i samples image point_image[i] of `input`
 *               (n_images of them) - sample_feature's batch_idx (sampling.py:34-38) without the side-by-side plane copy.
 * The backward is the true gradient (grad_input zero-filled by the caller, shape of input; either may be NULL). */
#define ENARF_PLANES_SUM      0
#define ENARF_PLANES_SEPARATE 1
int enarf_triplane_sample_ex_fwd(const float *input, const float *grid, float *out, int B, int C, int H, int W,
                                 long long n_pts, int interpolation_mode, int padding_mode, int align_corners,
                                 int reduction, const int *point_image, int n_images, enarf_stream_t stream);
int enarf_triplane_sample_ex_bwd(const float *grad_out, const float *input, const float *grid, float *grad_input,
                                 float *grad_grid, int B, int C, int H, int W, long long n_pts, int interpolation_mode,
                                 int padding_mode, int align_corners, int reduction, const int *point_image, int n_images,
                                 enarf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Tri-plane re-layout. The renderer reads the 96 feature channels channel-last,
 *   feat_cl[b][plane][y][x][32]   (one texel = 128 B = one cache line),
 * and the 3P one-channel part-probability planes in place from the NCHW tensor (channels 96..).
 * Replaces the F.pad + permute copy of libraries/triplane/sampling.py:96-97.
 * tri_nchw (B, 96 + 3P, H, W); feat_cl (B, 3, H, W, 32).
 * --------------------------------------------------------------------------------------------- */
int enarf_triplane_pack(const float *tri_nchw, float *feat_cl, int B, int channels_total, int H, int W,
                        enarf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-call preparation (one small launch): 24 joints -> P part frames, canonical scale, and the
 * per-image style-modulated, demodulated MLP weights packed in MFMA operand order.
 * Replaces transform_pose (libraries/NARF/pose_utils.py:129-148), the translation scaling of
 * render (libraries/NeRF/rendering.py:258-260), canonical_scale (models/narf.py:165) and
 * ModulatedConv1d's weight path (libraries/custom_stylegan2/net.py:233-243) for the 3 layers of
 * StyledMLP(32, 64, 4).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int B;                      /* images */
    int num_joints;             /* 24 */
    int origin_location;        /* ENARF_ORIGIN_* ; P = num_joints - 1 (+1 for CENTER_HEAD) */
    int style_dim;              /* z_rend width */
    float coordinate_scale;     /* nerf_params.coordinate_scale */
    int parents[ENARF_MAX_JOINTS];        /* parent joint ids, parents[0] = -1 (host values) */
    const float *pose_to_camera;          /* device (B, J, 4, 4) */
    const float *bone_length;             /* device (B, J-1) */
    const float *canonical_bone_length;   /* device (P,)  buffer of models/narf.py:119 */
    const float *z_rend;                  /* device (B, style_dim) */
    /* StyledMLP parameters, device, in the reference's state-dict shapes:           */
    const float *conv_weight[3];          /* layers.i.conv.weight            (1, out, in, 1)        */
    const float *mod_weight[3];           /* layers.i.conv.modulation.weight (in, style_dim)        */
    const float *mod_bias[3];             /* layers.i.conv.modulation.bias   (in,)                  */
    const float *bias[3];                 /* layers.i.bias                   (1, out, 1)            */
    /* outputs */
    float *parts;                         /* device (B, P, 16): R row-major 9, t*scale 3, can_scale, pad 3 */
    void  *mlp_pack;                      /* device (B, enarf_mlp_pack_bytes()) */
} enarf_prepare_args;

size_t enarf_mlp_pack_bytes(void);
int enarf_prepare(const enarf_prepare_args *args, enarf_stream_t stream);

/* Debug/interop helper: unpack the fp32 section of one image's mlp_pack into dense row-major
 * W1 (64,32), W2 (64,64), W3 (4,64), b1 (64), b2 (64), b3 (4) - 6724 floats (host or device memory
 * that the host can write is NOT required: `dense` is a device pointer). */
int enarf_mlp_unpack(const void *mlp_pack_one_image, float *dense, enarf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a9. The density/colour query on a point cloud.
 * Replaces TriPlaneNARF.calc_density_and_color_from_camera_coord_v2 (models/narf.py:176-211) with
 * backbone(mode="weight_feature") (:213-275): bone inverse transform, validity, part probability,
 * weighted tri-plane feature, StyledMLP, tanh colour, MyReLU*10 density, density *= any_valid.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int B; long long N;                   /* points per image */
    int P;                                /* parts (<= ENARF_MAX_PARTS) */
    int H, W;                             /* plane resolution (256) */
    int mlp_mode;                         /* ENARF_MLP_* */
    int multiply_density_with_weight;     /* nerf_params.multiply_density_with_triplane_wieght */
    const float *points;                  /* device (B, 3, N) camera coords in the scaled space */
    const float *parts;                   /* device (B, P, 16) from enarf_prepare */
    const float *canonical_pose;          /* device (P, 4, 4) buffer of models/narf.py:120 */
    const float *feat_cl;                 /* device (B or 1, 3, H, W, 32) from enarf_triplane_pack */
    long long feat_batch_stride;          /* floats between images; 0 = one tri-plane shared by all */
    const float *mask_planes;             /* device: &tri_nchw[0][96][0][0]; (P*3, H, W) per image */
    long long mask_batch_stride;          /* floats between images; 0 = shared */
    const void *mlp_pack;                 /* device (B, enarf_mlp_pack_bytes()) */
    float *density;                       /* device (B, 1, N) */
    float *color;                         /* device (B, 3, N), may be NULL */
    uint32_t *valid_bits;                 /* device (B, N), bit k = part k valid; may be NULL */
    float *dbg_canonical;                 /* device (B, P, 3, N) canonical coords; may be NULL */
    float *dbg_weight;                    /* device (B, P, N) part probability (0.125 where invalid); may be NULL */
    /* Lattice mode (points == NULL, grid_D > 0, N == grid_D^3, the same lattice for every image): point i = (ix, iy, iz)
     * = (i / D^2, i / D % D, i % D) sits at ((ix - c) / c + center_x, ...) * grid_scale with c = (D - 1) / 2 - the grid of
     * create_mesh (libraries/NARF/mesh_rendering.py:58-59: arange(-c, c + 1) / c, + center, * coordinate_scale),
     * generated in the kernel instead of streamed from memory. */
    int grid_D;
    float grid_center[3];
    float grid_scale;
    /* part-probability variants of libraries/triplane/sampling.py:43-76 / models/narf.py:133-134 */
    int clamp_mask;                       /* nerf_params.clamp_mask: plane samples clamped to [-2, 5] before the sigmoid */
    int uniform_part_weight;              /* nerf_params.no_selector: every part weighs 1 / P (no part-probability planes read) */
} enarf_query_args;

int enarf_query_fwd(const enarf_query_args *args, enarf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a13. The fused ray march: a small ray set-up launch (depth ranges, live-ray list) then the march itself,
 * persistent workgroups of 4 wavefronts taking one ray at a time from a queue.
 * Replaces render (libraries/NeRF/rendering.py:227-359) = decide_frustrum_range (:10-79) +
 * coarse_sample (:82-135) + coarse_to_fine_sample (:138-224) + two density/colour queries + alpha
 * compositing (:307-335), including the batch-global near/far planes (:15-17) and, when
 * drop_invalid_rays, the B == 1 removal of rays that hit no cube (:107-110, :337-350).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int B, n;                             /* images, rays per image */
    int P, Nc, Nf;                        /* parts, coarse / fine samples per ray (2 <= Nc, Nf <= 128) */
    int H, W;
    int mlp_mode;
    int multiply_density_with_weight;
    int drop_invalid_rays;                /* the reference does this iff batchsize == 1 */
    float render_scale;
    float early_stop_eps;                 /* 0 = exact (default). > 0: a quarter of the fine samples is skipped when the
                                             coarse pass puts the transmittance in front of it below eps */
    const float *image_coord;             /* device (B, 3, n) homogeneous pixel coords */
    const float *inv_intrinsics;          /* device (B, 3, 3) */
    const float *parts;                   /* device (B, P, 16) */
    const float *canonical_pose;          /* device (P, 4, 4) */
    const float *feat_cl; long long feat_batch_stride;
    const float *mask_planes; long long mask_batch_stride;
    const void *mlp_pack;
    const float *bins;                    /* device (B, n, Nf) sorted importance samples in [0,1); NULL = draw
                                             them in-kernel (Philox, `seed`), as rendering.py:192-197 */
    uint64_t seed;
    /* outputs */
    float *color;                         /* device (B, 3, n) */
    float *mask;                          /* device (B, n) */
    float *disparity;                     /* device (B, n) */
    float *fine_weights;                  /* device (B, n, Nf-1) or NULL   (buffers_tensors["fine_weights"]) */
    float *fine_depth;                    /* device (B, n, Nf)   or NULL   (buffers_tensors["fine_depth"]) */
    /* parity taps, NULL in production */
    float *dbg_depth_min, *dbg_depth_max; /* (B, n) */
    uint8_t *dbg_ray_valid;               /* (B, n) */
    float *dbg_coarse_density;            /* (B, n, Nc) */
    float *dbg_fine_density;              /* (B, n, Nf) */
    float *dbg_fine_color;                /* (B, 3, n, Nf) */
    uint32_t *dbg_fine_valid;             /* (B, n, Nf) bit masks */
    float *dbg_bins;                      /* (B, n, Nf) the bins actually used */
    unsigned long long *counters;         /* 8 x u64, atomically accumulated (zero them before the call); NULL = not counted.
                                             [0] valid (part,point) pairs sampled, [1] MLP tiles of 16 points run,
                                             [2] rays marched (rays that miss every cube included, dropped rays not),
                                             [3] gather rounds (wave-level), [4] fine tiles skipped by early_stop_eps,
                                             [5], [6] unused by the product library, [7] != 0: the task march's
                                             scheduler watchdog fired and the outputs are incomplete (never expected) */
    void *workspace;                      /* device, >= enarf_render_workspace_bytes(B, n): two queue headers, per-ray
                                             records (depth range, candidate parts, direction) and the ray lists; one
                                             workspace must not be shared by launches that can overlap. */
    int clamp_mask, uniform_part_weight;  /* as in enarf_query_args */
    int march;                            /* ENARF_MARCH_*: which of the two march kernels runs (same results, bit for bit) */
    int ws_epoch;                         /* 0: the call clears the queue headers itself (one extra fill launch on
                                             `stream`) - always safe. k > 0: the caller promises that the previous call
                                             that used this workspace had ws_epoch k - 1 (any of enarf_render_fwd /
                                             enarf_render_step_fwd; enarf_render_bwd counts as epoch 0) and completed
                                             or is ordered before this one on the stream: that call left the header this
                                             one uses clean, so no fill is launched. k > 0 also requires the same (B, n,
                                             group_frames) as that previous call: a batch rendered in groups keeps one
                                             queue header pair per group inside the workspace. */
    /* ---- a batch rendered in several calls, or in several launches of one call ---- */
    const float *near_far;                /* device [2] or NULL. The near / far planes are a reduction over the WHOLE batch
                                             (rendering.py:15-17): NULL = reduce over this call's B frames; else the two
                                             floats enarf_near_far wrote for the whole batch - for a caller that renders a
                                             batch in pieces (ranks of a data-parallel job, groups of frames) */
    unsigned long long ray_id_base;       /* added to b * n + ray in the in-kernel sampler's counter: first frame of this
                                             call x n, so that a frame draws the same samples whichever call renders it */
    int group_frames;                     /* frames per march launch when every frame has its own tri-plane
                                             (feat_batch_stride != 0): 0 = the library's choice (8: one frame per XCD,
                                             from measurements - 16 frames in one launch march 14 % slower than 2 x 8, the
                                             XCDs' L2s then serve two frames' texels alternately), k > 0 = k, >= B = one
                                             launch. Same results bit for bit for every value. */
} enarf_render_args;

/* near / far planes of a batch: out[0] = max(min_z - sqrt(3), 0.3), out[1] = max(max_z + sqrt(3), 5) over the z of every part
 * centre of parts (B, P, 16) - the reduction decide_frustrum_range makes over the whole batch (rendering.py:15-17), in the
 * same arithmetic as the ray set-up. out: device [2]. */
int enarf_near_far(const float *parts, int B, int P, float *out, enarf_stream_t stream);

size_t enarf_render_workspace_bytes(int B, int n);
int enarf_render_fwd(const enarf_render_args *args, enarf_stream_t stream);

/* The whole forward step in two launches: one "pre-march" launch whose blocks do enarf_triplane_pack,
 * enarf_prepare and the ray set-up side by side (the three are independent: set-up blocks derive their part frames
 * from the raw poses themselves, bit-identically), then the march. Same results as calling the three entry points
 * in sequence. `render->parts` must be `prep->parts` and `render->mlp_pack` must be `prep->mlp_pack`.
 * tri_nchw == NULL skips the re-layout (feat_cl of a constant tri-plane is already up to date).
 * phases: ENARF_STEP_PRE | ENARF_STEP_MARCH; a caller that wants to time the march alone issues the two phases as two
 * calls on the same stream (the march phase needs the pre-march phase of the same arguments before it). */
#define ENARF_STEP_PRE 1
#define ENARF_STEP_MARCH 2
#define ENARF_STEP_ALL 3
int enarf_render_step_fwd(const enarf_prepare_args *prep, const float *tri_nchw, float *feat_cl, int tri_B,
                          int channels_total, const enarf_render_args *render, int phases, enarf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of the fused renderer: what `loss_gen.backward()` computes through render
 * (libraries/NeRF/rendering.py:283-335), the fine-pass query (models/narf.py:176-275), the StyledMLP and MyReLU's
 * custom backward (libraries/NeRF/activation.py:12-16). Nf <= 128. As in the reference, gradients flow through the FINE pass
 * only (the importance samples are not differentiable) and not into poses.
 *   enarf_render_bwd   d loss / d tri-plane (atomically accumulated into caller-zeroed buffers) and, per valid
 *                      16-sample tile, 16 compact rows (x: the 32 gathered features, dz3: dL/d the 4 pre-activation
 *                      outputs of the last layer) from which enarf_weight_grad forms the weight gradients
 *                      dW'_l = dZ_l^T H_{l-1} per image by re-running the MLP forward and backward on them
 *                      (144 B per sample; round 2 exported all six activation rows: 1 168 B per sample).
 *   enarf_prepare_bwd  d loss / d (conv.weight, modulation.weight, modulation.bias, z_rend) from dW' (the backward of
 *                      ModulatedConv1d's modulate + F.normalize), per image; the caller sums the shared parameters.
 *   enarf_triplane_unpack_add   grad_tri[:, :96] += channel-last gradient (inverse of enarf_triplane_pack).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int B, n, P, Nf, H, W;
    int drop_invalid_rays;
    float render_scale;
    const float *image_coord, *inv_intrinsics, *parts, *canonical_pose;
    const float *feat_cl; long long feat_batch_stride;
    const float *mask_planes; long long mask_batch_stride;
    const void *mlp_pack;
    const float *bins;                    /* device (B, n, Nf): the bins the forward used (its dbg_bins output) */
    const float *g_color, *g_mask, *g_disparity;   /* upstream gradients (B,3,n), (B,n), (B,n); NULL = zero */
    float *grad_feat_cl; long long grad_feat_batch_stride;      /* (B|1, 3, H, W, 32), zero-filled by the caller */
    float *grad_mask_planes; long long grad_mask_batch_stride;  /* &grad_tri[0][96][0][0], zero-filled by the caller */
    float *rows_x, *rows_dz3;             /* (B, rows_per_image, 32) and (B, rows_per_image, 4) */
    long long rows_per_image;             /* >= enarf_render_bwd_rows_per_image(n, Nf) */
    unsigned int *row_blocks;             /* device (B,): 16-row blocks written per image (zeroed by the call) */
    void *workspace;                      /* as enarf_render_fwd */
    const float *near_far;                /* as enarf_render_args: NULL = reduce over this call's B frames */
    int group_frames;                     /* as enarf_render_args (per-frame tri-planes: frames per launch; 0 = 8) */
    unsigned long long *counters;         /* optional device [8], zeroed by the caller: [0] valid (part, fine sample) pairs,
                                             [1] 16-sample tiles taken through the MLP backward, [2] rays, [3] 128-B lines
                                             added into the feature-plane gradient (float atomics, after on-chip merging),
                                             [4] 4-byte adds into the part-probability gradient planes, [5] gather rounds */
    int clamp_mask, uniform_part_weight;  /* as in enarf_query_args (clamp_mask: straight-through gradient, sampling.py:46-47) */
    int multiply_density_with_weight;     /* nerf_params.multiply_density_with_triplane_wieght: the gradient also reaches the
                                             part probability that attains the maximum (models/narf.py:271-272) */
} enarf_render_bwd_args;

long long enarf_render_bwd_rows_per_image(int n, int Nf);
int enarf_render_bwd(const enarf_render_bwd_args *args, enarf_stream_t stream);

/* Backward of enarf_query_fwd (a9: TriPlaneNARF.calc_density_and_color_from_camera_coord_v2, models/narf.py:176-275)
 * w.r.t. the tri-plane and the per-image demodulated MLP weights: given dL/d density (B,1,N) and dL/d color (B,3,N)
 * (either may be NULL) it adds into grad_feat_cl / grad_mask_planes and exports the rows enarf_weight_grad consumes.
 * The points carry no gradient (the reference detaches nothing here, but every caller of the query samples points from
 * non-differentiable poses). multiply_density_with_triplane_wieght is not differentiable here (as in enarf_render_bwd). */
typedef struct {
    int B, P, H, W;
    long long N;
    const float *points;                  /* (B, 3, N) */
    const float *parts, *canonical_pose;  /* as enarf_query_args */
    const float *feat_cl; long long feat_batch_stride;
    const float *mask_planes; long long mask_batch_stride;
    const void *mlp_pack;
    const float *g_density;               /* (B, 1, N) or NULL */
    const float *g_color;                 /* (B, 3, N) or NULL */
    float *grad_feat_cl; long long grad_feat_batch_stride;      /* (B|1, 3, H, W, 32), zero-filled by the caller */
    float *grad_mask_planes; long long grad_mask_batch_stride;  /* &grad_tri[0][96][0][0], zero-filled by the caller */
    float *rows_x, *rows_dz3;             /* (B, rows_per_image, 32) and (B, rows_per_image, 4), as enarf_render_bwd_args */
    long long rows_per_image;             /* >= enarf_query_bwd_rows_per_image(N) */
    unsigned int *row_blocks;             /* device (B,): 16-row blocks written per image (zeroed by the call) */
    int clamp_mask, uniform_part_weight, multiply_density_with_weight;
} enarf_query_bwd_args;
long long enarf_query_bwd_rows_per_image(long long N);
int enarf_query_bwd(const enarf_query_bwd_args *args, enarf_stream_t stream);

/* Weight gradients of the per-image (demodulated) StyledMLP from the compact rows enarf_render_bwd / enarf_query_bwd
 * exported: every 16-row tile is taken through the exact-fp32 MLP forward (X -> H1, H2) and backward (dZ3 -> dZ2, dZ1)
 * again - the same code on the same operands as the kernel that exported it, so the same bits - and
 *   dW1 (B,64,32) = dZ1^T X, dW2 (B,64,64) = dZ2^T H1, dW3 (B,4,64) = dZ3^T H2, db_l (B, out) = column sums of dZ_l
 * accumulate in registers over the first 16 * row_blocks[b] rows of image b. Replaces autograd through
 * conv1d(groups=B) (libraries/custom_stylegan2/net.py:240-243) for the rendered samples. Deterministic (no atomics).
 * workspace: enarf_weight_grad_workspace_bytes(B, rows_per_image) bytes of device memory. */
typedef struct {
    int B;
    const float *rows_x, *rows_dz3;       /* as enarf_render_bwd_args */
    const void *mlp_pack;                 /* the pack of the forward (enarf_prepare), B images */
    long long rows_per_image;
    const unsigned int *row_blocks;
    float *dW1, *dW2, *dW3, *db1, *db2, *db3;
    void *workspace;
} enarf_weight_grad_args;
size_t enarf_weight_grad_workspace_bytes(int B, long long rows_per_image);
int enarf_weight_grad(const enarf_weight_grad_args *args, enarf_stream_t stream);

typedef struct {
    int B, style_dim;
    const float *z_rend;                  /* (B, style_dim) */
    const float *conv_weight[3], *mod_weight[3], *mod_bias[3];   /* as enarf_prepare_args */
    const float *dW[3];                   /* (B, out, in) dense row-major gradients of the demodulated weights */
    float *d_conv_weight[3];              /* (B, out, in)       per image */
    float *d_mod_weight[3];               /* (B, in, style_dim) per image */
    float *d_mod_bias[3];                 /* (B, in)            per image */
    float *d_z_rend;                      /* (B, 3, style_dim)  per image and layer */
} enarf_prepare_bwd_args;
int enarf_prepare_bwd(const enarf_prepare_bwd_args *args, enarf_stream_t stream);

int enarf_triplane_unpack_add(const float *grad_feat_cl, float *grad_tri_nchw, int B, int channels_total, int H, int W,
                              enarf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Deformation-field tri-plane producer (models/narf.py:40-58, nerf_params.deformation_field): the constant feature
 * planes warped by a per-image flow, out[b, p, y, x, :] = bilinear(src[p], x + flow[b, 2p, y, x], y + flow[b, 2p+1, y, x])
 * (F.grid_sample: bilinear, zeros padding, align_corners False, on the grid (pixel centre + flow) / (W/2) - 1), written
 * CHANNEL-LAST, ready for enarf_render_fwd (feat_cl per image, part-probability planes shared, mask_batch_stride 0).
 * src_cl (3, H, W, 32) = enarf_triplane_pack of the constant tri-plane; flow (B, 6, H, W) NCHW; out_cl (B, 3, H, W, 32).
 * Backward: g_src_cl (3, H, W, 32) is accumulated into (zero-fill it; fold it back with enarf_triplane_unpack_add),
 * g_flow (B, 6, H, W) is written; either may be NULL.
 * --------------------------------------------------------------------------------------------- */
int enarf_triplane_warp_fwd(const float *src_cl, const float *flow, float *out_cl, int B, int H, int W,
                            enarf_stream_t stream);
int enarf_triplane_warp_bwd(const float *g_out_cl, const float *src_cl, const float *flow, float *g_src_cl,
                            float *g_flow, int B, int H, int W, enarf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a16. mask_based_sampler on the device (libraries/NeRF/ray_sampler.py:7-39): per image the k largest values of
 *   max_pool2d(mask, 2 radius + 1, stride 1, padding radius) + noise   (the reference: radius 64, noise ~ U[0, 1))
 * as flat pixel ids y * w + x, unordered (torch.topk(sorted=False)); scores equal to the k-th largest are taken in
 * ascending pixel order. mask (B, h, w) fp32, noise (B, h * w) fp32, out_idx (B, k) int64; workspace:
 * enarf_mask_topk_workspace_bytes(B, h, w) bytes of device memory. Separable window maximum + one radix select per image.
 * --------------------------------------------------------------------------------------------- */
size_t enarf_mask_topk_workspace_bytes(int B, int h, int w);
int enarf_mask_dilate_topk(const float *mask, const float *noise, long long *out_idx, int B, int h, int w, int k, int radius,
                           void *workspace, enarf_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * the two element-level ops of the reference's 2-D GAN networks (discriminator, background
 * generator: libraries/custom_stylegan2/net.py:346-676). The reference imports them from an un-vendored submodule
 * (net.py:12-14: FusedLeakyReLU / fused_leaky_relu, Blur / Upsample of rosinality/stylegan2-pytorch, CUDA extensions
 * `fused_bias_act` and `upfirdn2d`); these entry points are what that import would bind instead.
 *
 * enarf_bias_act: x, out contiguous (outer, C, inner) fp32.
 *   ref == NULL:  out = gain * leaky_relu(x + bias[c], negative_slope)       (bias may be NULL)
 *   ref != NULL:  out = x * gain * (ref > 0 ? 1 : negative_slope)            (ref = the forward OUTPUT; the derivative of the
 *                 line above with respect to x applied to x - and, being linear in x, its own derivative too)
 * enarf_upfirdn2d: x (planes, H, W) -> out (planes, OH, OW): insert up - 1 zeros after every sample, pad by
 *   (pad_*0, pad_*1) (negative = crop), convolve with the kh x kw filter `kernel_host` (HOST memory, row-major, <= 8 x 8; a
 *   true convolution: the filter is flipped), keep every down-th sample. OH = enarf_upfirdn2d_out_size(H, kh, up, down,
 *   pad_y0, pad_y1) = (H * up + pad_y0 + pad_y1 - kh) / down + 1. Built for up / down in {1/1, 2/1, 1/2}.
 * --------------------------------------------------------------------------------------------- */
int enarf_bias_act(const float *x, const float *bias, const float *ref, float *out, long long outer, int C, long long inner,
                   float negative_slope, float gain, enarf_stream_t stream);
int enarf_upfirdn2d_out_size(int in_size, int taps, int up, int down, int pad0, int pad1);
int enarf_upfirdn2d(const float *x, float *out, long long planes, int H, int W, const float *kernel_host, int kh, int kw,
                    int up, int down, int pad_x0, int pad_x1, int pad_y0, int pad_y1, enarf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ENARF_HIP_H */
