/*
 * gsr.h -- C ABI of libgsr_hip.so: MI355X (gfx950) differentiable 2D-Gaussian-surfel rasterizer
 * and 3-nearest-neighbour mean squared distance.
 *
 * What this replaces in alevalve/gaussmart (paths relative to the reference checkout):
 *   - the native extension `diff_surfel_rasterization._C` (submodules/diff-surfel-rasterization,
 *     an EMPTY un-pinned submodule: .gitmodules:1-3), reached through
 *     `GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp, opacities,
 *     scales, rotations, cov3D_precomp)`            gaussian_renderer/__init__.py:14,37-53,97-106
 *   - `simple_knn._C.distCUDA2(points)` (submodules/simple-knn, EMPTY: .gitmodules:4-6)
 *                                                   scene/gaussian_model.py:22,261
 * The reference constrains only the Python operator surface; this C layer is what a binding for
 * that surface calls (see INTEGRATION.md for the ctypes / pybind11 stubs).
 *
 * Conventions
 *   - every pointer marked `device` is HIP device memory owned by the CALLER; the library never
 *     frees or retains it and keeps no global mutable state besides the opt-in profiler and one
 *     64-byte pinned host slot per calling thread (the read-back below);
 *   - all work is enqueued on `stream`; gsr_forward performs exactly one stream synchronisation
 *     (to learn the number of tile instances) before it asks for the instance-sized buffers;
 *     gsr_backward synchronises only for scenes whose worst-case gradient rows exceed
 *     GSR_EXACT_ROWS_BYTES (environment, default 8 GiB), to size that buffer exactly;
 *   - matrices are row-major 4x4 in the reference's transposed / row-vector convention
 *     (viewmatrix = W2C^T, projmatrix = viewmatrix @ P^T; scene/cameras.py:56-58);
 *   - return value 0 = ok, negative = GSR_E_*; gsr_last_error() gives the thread's last message.
 */
#ifndef GSR_H_
#define GSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 6   /* 5: BINNING carries row_count / slot_off / the scan workspace, 72-byte gradient rows, GsrRowScanJob;
                                6: gsr_surface_maps_forward / _backward */
#define GSR_MAX_CHANNELS 64   /* widest per-pixel payload of gsr_forward / gsr_backward */

typedef void* gsr_stream_t; /* hipStream_t */

enum {
    GSR_OK = 0,
    GSR_E_INVALID = -1,     /* bad argument combination (mirrors the operator's Python checks) */
    GSR_E_HIP = -2,         /* a HIP runtime call failed                                        */
    GSR_E_ALLOC = -3,       /* the caller's allocator returned NULL                             */
    GSR_E_UNSUPPORTED = -4  /* e.g. channels not in {3, 4..64 step 4}, sh_degree > 3            */
};

/* GsrView.flags: reproduce the two places where the recalled upstream backward is not the
 * derivative of its forward.  GSR_FLAGS_UPSTREAM is what the Python operator passes. */
enum {
    GSR_FLAG_CLAMP_PASSTHROUGH = 1, /* gradient flows through alpha = min(0.99, o*G) when clamped */
    GSR_FLAG_FILTER_DEPTH_GRAD = 2, /* low-pass branch: dL/dz also added to dL/dTw.x,.y times s   */
    GSR_FLAGS_UPSTREAM = 3,
    GSR_FLAG_AABB_GRAD_CUTOFF1 = 512, /* third recalled non-derivative (a reviewer's recollection of upstream backward.cu, as
                                       unverifiable here as the two above): the gradient of the screen-space centre
                                       (dL/dmean2D of the low-pass branch -> T) is chained with the weights (1, 1, -1)
                                       although the forward's centre uses (cutoff^2, cutoff^2, -1) = (9, 9, -1).  Not
                                       part of GSR_FLAGS_UPSTREAM; kernels and oracle implement both settings */
    GSR_FLAG_DEBUG_NO_CULL = 4,     /* test aid: ignore the per-wave cull rect (results are bit-identical) */
    GSR_FLAG_DEBUG_RECT_CULL_ONLY = 64, /* test / measurement aid: cull with the rect only, skip the ellipse test (same results) */
    GSR_FLAG_DEFER_COLOR = 16,      /* enqueue the SH colour pass as LATE as possible -- after binning, right before the
                                       compositing -- and announce it through the allocator (GSR_BUF_SYNC_SH): the caller
                                       may make the stream wait there for SH parameters that are still being updated on
                                       another stream (pipelined data-parallel step); the geometry inputs (means, scales,
                                       rotations, opacities) must be final when gsr_forward is called */
    GSR_FLAG_RAW_PARAMS = 8,        /* opacities are logits, scales are log-scales, rotations un-normalised:
                                       the activations of scene/gaussian_model.py:37-43 (sigmoid, exp,
                                       normalize) run inside the kernels and the gradients returned are
                                       w.r.t. the raw parameters */
    GSR_FLAG_FORWARD_ONLY = 128,    /* gsr_forward for inference (render.py / view.py under torch.no_grad();
                                       utils/mesh_utils.py:100-123): colour, allmap and radii are produced as usual, but
                                       nothing is kept for a backward -- the IMAGE buffer is not requested (out->image =
                                       NULL), the touch words of BINNING are not written and gsr_backward must not be
                                       called with these buffers.  Honoured for 3-channel output; ignored for wide payloads */
    GSR_FLAG_COLOR_CACHED = 256,    /* `shs` AND `colors_precomp` given, channels = 3: colors_precomp is a COLOUR CACHE, f32[13 N] =
                                       [ rgb [N,3] | clamp bits u32 [N] | d(rgb)/d(dir) [N,9] ], holding the SH colour of THIS view
                                       for THESE coefficients and positions as gsr_adam_sh_factored_next left it.  gsr_forward
                                       copies rgb / clamp bits of the visible Gaussians into its records where the SH colour
                                       pass would run (same position in the stream, same GSR_FLAG_DEFER_COLOR handling) and
                                       never reads the coefficients; gsr_backward (pass the same pointer and flag) takes
                                       d(rgb)/d(dir) from the cache.  The caller guarantees that the cache matches */
    GSR_FLAG_NO_DIST_MEDIAN = 4096, /* gsr_forward + gsr_backward (RGB payload): the caller consumes neither the distortion (allmap
                                       channel 6) nor the median depth (channel 5) -- the reference's default training configuration,
                                       lambda_dist = 0 and depth_ratio = 0 (arguments/__init__.py:72,87).  Both channels come back
                                       as zeros (not accumulated: the distortion arithmetic is 10 % of the forward), the other five
                                       and the colour are those of the general forward bit for bit; the backward treats the two
                                       channels as constants */
    GSR_FLAG_COLOR_ONLY = 2048,     /* gsr_forward (RGB payload, not with GSR_FLAG_FORWARD_ONLY): the caller does not consume allmap -- a
                                       trainer while no regularizer is active.  out->allmap is NOT written, the kept image state
                                       holds T and the last contributor only; colour, radii and everything the backward walks are
                                       those of the general forward bit for bit.  The backward of such a forward must carry
                                       GSR_FLAG_NO_SURFACE_GRAD (gsr_backward rejects it otherwise) */
    GSR_FLAG_NO_SURFACE_GRAD = 1024, /* gsr_backward only: the caller promises that dL_dallmap is identically zero (it must still
                                       point at W*H*7 zeros) -- the reference's evaluation flags (scripts/dtu_eval.py:45:
                                       --lambda_normal 0 --lambda_dist 0) and the first 7,000 iterations of every run
                                       (train.py:132-133).  The compositing backward then runs without the surface terms on a
                                       16-float staged record; same gradients as without the flag, bit for bit */
    GSR_FLAG_FACTORED_SH_GRAD = 32  /* gsr_backward with `shs`: the SH gradient of one view is the outer product
                                       basis_k(dir) x g_c of the 16 basis values of the view direction and the
                                       clamp-masked colour gradient g = dL/drgb (utils/sh_utils.py:57-112 is linear in
                                       the coefficients).  With this flag the [N,M,3] arrays dL_dshs / dL_dshs_rest are
                                       NOT written (may be NULL); dL_dcolors, f32[3N + 4], receives g as [N,3] (zeros for
                                       culled Gaussians) followed by the camera position (x, y, z, 0) -- the whole
                                       factored gradient of the view in one contiguous record -- and
                                       gsr_adam_sh_factored rebuilds the gradient inside the optimiser step.  48 -> 3 floats per Gaussian written, re-read and -- in the view-parallel
                                       step -- exchanged between GPUs */
};

/* GaussianRasterizationSettings (gaussian_renderer/__init__.py:37-51) */
typedef struct GsrView {
    int32_t width, height;
    float tanfovx, tanfovy;
    float scale_modifier;
    int32_t sh_degree;       /* active degree, 0..3                                  */
    int32_t sh_coeffs;       /* coefficients stored per Gaussian (shs is [N, M, 3])  */
    int32_t channels;        /* 3 (RGB), or 4..GSR_MAX_CHANNELS (multiple of 4) with
                                colors_precomp [N,channels]: wide per-pixel payload  */
    uint32_t flags;          /* GSR_FLAG_*                                           */
    const float* bg;         /* device f32[channels]                                 */
    const float* viewmatrix; /* device f32[16]                                       */
    const float* projmatrix; /* device f32[16]                                       */
    const float* campos;     /* device f32[3]                                        */
} GsrView;

/* Operator inputs (gaussian_renderer/__init__.py:97-106).  Exactly one of shs / colors_precomp
 * and exactly one of (scales, rotations) / transmat_precomp must be non-NULL. */
typedef struct GsrGaussians {
    int32_t count;                 /* N                                              */
    const float* means3D;          /* device [N,3]                                   */
    const float* shs;              /* device [N,M,3] or NULL                         */
    const float* colors_precomp;   /* device [N,channels] or NULL (16-byte aligned when channels != 3) */
    const float* opacities;        /* device [N] (post-sigmoid)                      */
    const float* scales;           /* device [N,2] (post-exp) or NULL                */
    const float* rotations;        /* device [N,4] (w,x,y,z) or NULL                 */
    const float* transmat_precomp; /* device [N,9] rows Tu,Tv,Tw, or NULL            */
    const float* shs_rest;         /* device [N,M-1,3] or NULL.  When non-NULL, `shs` holds only
                                      the DC coefficient [N,1,3] (split storage = the model's two
                                      feature parameters, no concatenation needed)               */
} GsrGaussians;

/* Buffers the forward hands to the backward.  The caller allocates them through the callback
 * (two-phase: the instance-sized ones are requested after the scan) and keeps them alive until
 * the backward has run.  GSR_BUF_SCRATCH may be released as soon as the call returns
 * (stream-ordered). */
enum { GSR_BUF_GEOM = 0, GSR_BUF_BINNING = 1, GSR_BUF_IMAGE = 2, GSR_BUF_SCRATCH = 3,
       GSR_BUF_SCRATCH2 = 4, GSR_BUF_COUNT = 5,
       /* not a buffer: with GSR_FLAG_DEFER_COLOR the allocator is called once with this kind and 0 bytes right before
          the SH colour pass is enqueued (the last moment the SH parameters may still be in flight on another
          stream: the callback may enqueue a stream wait); any non-NULL return value means "go on" */
       GSR_BUF_SYNC_SH = 100,
       /* not a buffer either: with GSR_FLAG_DEFER_COLOR the allocator is first asked, with this kind and 0 bytes right
          after the geometry pass, for a SECOND stream (hipStream_t) on which the SH parameters will be ready (e.g. the
          stream that is still updating them).  Non-NULL: the colour pass is enqueued on that stream at once, ordered
          after the geometry pass by an event, and only the compositing waits for it -- sorting and binning overlap the
          parameter update AND the colour pass; GSR_BUF_SYNC_SH is then not announced.  NULL: the colour pass stays on
          the call's stream at the late position described above */
       GSR_BUF_COLOR_STREAM = 101 };

/* Must return device memory of >= bytes, 256-byte aligned, or NULL. */
typedef void* (*gsr_alloc_fn)(void* ctx, int32_t which, size_t bytes);

typedef struct GsrForwardOut {
    float* out_color;      /* device [channels,H,W]                                   */
    float* out_allmap;     /* device [7,H,W]: depth, alpha, normal xyz (view space), median
                              depth, distortion (gaussian_renderer/__init__.py:117-141) */
    int32_t* radii;        /* device [N]                                              */
    int32_t num_rendered;  /* out: number of (Gaussian, tile) instances D             */
    void* geom;            /* out: the pointers the allocator returned                */
    void* binning;
    void* image;
} GsrForwardOut;

typedef struct GsrGrads {
    float* dL_dmeans3D;   /* device [N,3]                                              */
    float* dL_dmeans2D;   /* device [N,3]: densification statistic, .z = 0
                             (consumer scene/gaussian_model.py:551-553)                */
    float* dL_dopacity;   /* device [N]                                                */
    float* dL_dshs;       /* device [N,M,3] or NULL                                    */
    float* dL_dcolors;    /* device [N,channels] or NULL (when colors_precomp was given; with
                             GSR_FLAG_FACTORED_SH_GRAD: f32[3N + 4], see the flag)  */
    float* dL_dscales;    /* device [N,2] or NULL                                      */
    float* dL_drotations; /* device [N,4] or NULL                                      */
    float* dL_dtransmat;  /* device [N,9] or NULL (when transmat_precomp was given)    */
    float* dL_dshs_rest;  /* device [N,M-1,3] or NULL (when shs_rest was given)        */
} GsrGrads;

int32_t gsr_abi_version(void);
const char* gsr_last_error(void);

/* Forward: preprocess -> depth sort -> scan -> instance emit -> tile sort -> ranges -> composite. */
int32_t gsr_forward(const GsrView* view, const GsrGaussians* g, GsrForwardOut* out,
                    gsr_alloc_fn alloc, void* alloc_ctx, gsr_stream_t stream);

/* Backward: back-to-front replay per 8x8 pixel quad (one independent list walk per 4x4 block) writing one dense
 * gradient row per (instance, block) the forward blended, then a per-Gaussian reduction + chain rule.  Every element of every non-NULL GsrGrads array is
 * written (no pre-zeroing needed).  Deterministic: no floating-point atomics. */
int32_t gsr_backward(const GsrView* view, const GsrGaussians* g, int32_t num_rendered,
                     const int32_t* radii, const void* geom, const void* binning,
                     const void* image, const float* dL_dcolor, const float* dL_dallmap,
                     GsrGrads* grads, gsr_alloc_fn alloc, void* alloc_ctx, gsr_stream_t stream);

/* The backward starts with an exclusive scan of the per-instance gradient-row counts the forward left in BINNING (two
 * small launches).  Nothing between the forward and the backward depends on it, so kernels that run in between can carry it
 * as a side job: gsr_row_scan_job describes it (pointers into BINNING), gsr_loss_forward_job / gsr_loss_backward_finish
 * take the description and run its two halves in extra workgroups of their own launches (recording that in `stage`), and
 * gsr_backward_with_job skips the scan when it finds both halves done for exactly its buffers.  Entirely optional: with
 * stage < 2 (or no job) the backward scans itself. */
typedef struct GsrRowScanJob {
    const void* counts;      /* device u8[n]   (BINNING "row_count") */
    void* slot_off;          /* device u32[n + 1] */
    void* workspace;         /* device, scan partials */
    int64_t n;               /* instances (num_rendered) */
    int32_t stage;           /* host-side: 0 = nothing enqueued, 1 = first half enqueued, 2 = both */
} GsrRowScanJob;
int32_t gsr_row_scan_job(const void* binning, int32_t num_rendered, int32_t width, int32_t height, GsrRowScanJob* job);
int32_t gsr_backward_with_job(const GsrView* view, const GsrGaussians* g, int32_t num_rendered,
                              const int32_t* radii, const void* geom, const void* binning,
                              const void* image, const float* dL_dcolor, const float* dL_dallmap,
                              GsrGrads* grads, const GsrRowScanJob* job, gsr_alloc_fn alloc, void* alloc_ctx,
                              gsr_stream_t stream);

/* Introspection of the saved buffers, for parity tests.  Writes byte offset and size of a named
 * field inside buffer `which` for a problem of N Gaussians, D instances, W x H pixels.
 * Names: GEOM: "splat" f32[N,20], "clamped" u32[N], "tiles_touched" u32[N], "depth_key" u32[N],
 *        "order" u32[N] (depth rank -> Gaussian id), "offs" u32[N+1] (depth rank -> first emission index);
 *        BINNING: "point_list" u32[D], "inst_row" u32[D] (emission index of each list entry), "ranges" u32[tiles,2],
 *                 "covered" u32[tiles,4] (list entries each 8x8 quad staged), "touch" u32[D] (per list entry: one byte per
 *                 quad, one bit per 4x4 block it was blended into; a quad's byte is defined below its `covered`),
 *                 "row_count" u8[D] (gradient rows per instance, by emission index: left by the forward for the backward);
 *        IMAGE: "final_T" f32[3,H,W], "n_contrib" u32[2,H,W]. */
int32_t gsr_buffer_field(int32_t which, const char* name, int32_t N, int32_t D, int32_t W,
                         int32_t H, size_t* offset, size_t* bytes);

/* distCUDA2 (scene/gaussian_model.py:261): mean squared distance to the 3 nearest other points. */
size_t gsr_knn3_workspace_bytes(int32_t n);
int32_t gsr_knn3(const float* xyz, int32_t n, float* out_mean_sqdist, void* workspace,
                 size_t workspace_bytes, gsr_stream_t stream);

/* Stable LSD radix sort of (u32 key, u32 value) pairs on bits [begin_bit, end_bit): the sort the
 * binning uses, exposed so it can be checked bit-exactly at any size. */
size_t gsr_sort_workspace_bytes(int32_t n);
int32_t gsr_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                           uint32_t* vals_out, int32_t n, int32_t begin_bit, int32_t end_bit,
                           void* workspace, size_t workspace_bytes, gsr_stream_t stream);

/* Fused photometric loss: (1-lambda)*mean|x-y| + lambda*(1 - mean SSIM(x,y)) with
 * the reference's SSIM (utils/loss_utils.py:38-57: 11x11 Gaussian window sigma 1.5, zero padding)
 * as used at train.py:113-114.  img, gt: device f32 [C,H,W].
 * forward : writes maps f32[3,C,H,W] (partials of SSIM kept for the backward) and
 *           partials f32[gsr_loss_num_partials(H,W)] = per-block (sum SSIM, sum |x-y|) pairs which
 *           the caller adds up (fixed order => reproducible loss value).
 * backward: dimg f32[C,H,W] = grad_scale[0] * dloss/dimg (grad_scale is a DEVICE scalar, so no
 *           host sync is needed to chain it). */
int32_t gsr_loss_num_partials(int32_t H, int32_t W);
int32_t gsr_loss_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                         float* maps, float* partials, gsr_stream_t stream);
int32_t gsr_loss_backward(const float* img, const float* gt, const float* maps, int32_t C, int32_t H,
                          int32_t W, float lambda_dssim, const float* grad_scale, float* dimg,
                          gsr_stream_t stream);

/* Fused surface regularizers: from the rasterizer's allmap [7,H,W] straight to
 *   normal_loss = lambda_normal * mean(1 - rend_normal . surf_normal),  dist_loss = lambda_dist * mean(allmap[6])
 * i.e. gaussian_renderer/__init__.py:117-156 + utils/point_utils.py:9-37 + train.py:132-140 of the
 * reference.  kinv_host: HOST f32[9], row-major inverse of the pixel intrinsics the reference builds
 * at utils/point_utils.py:11-17 (ray(x,y) = kinv * [x,y,1] in camera space).
 * forward : partials f32[gsr_loss_num_partials(H,W)] = per-block (sum normal error, sum distortion).
 * backward: d_allmap f32[7,H,W] = grad_scale[0] * d(normal_loss + dist_loss)/d(allmap), every
 *           element written; grad_scale is a DEVICE scalar. */
int32_t gsr_regularizer_forward(const float* allmap, int32_t H, int32_t W, const float* kinv_host,
                                float depth_ratio, float* partials, gsr_stream_t stream);
int32_t gsr_regularizer_backward(const float* allmap, int32_t H, int32_t W, const float* kinv_host,
                                 float depth_ratio, float lambda_normal, float lambda_dist,
                                 const float* grad_scale, float* d_allmap, gsr_stream_t stream);
/* gsr_regularizer_backward that ALSO leaves the partials gsr_regularizer_forward would have written (partials != NULL): the
 * backward evaluates every pixel's surface normal anyway, so a caller that runs the backward right after the forward and
 * reads the loss value only afterwards needs no forward launch of the regularizer at all. */
int32_t gsr_regularizer_backward_partials(const float* allmap, int32_t H, int32_t W, const float* kinv_host,
                                          float depth_ratio, float lambda_normal, float lambda_dist,
                                          const float* grad_scale, float* d_allmap, float* partials,
                                          gsr_stream_t stream);

/* The maps the reference's render() derives from allmap (gaussian_renderer/__init__.py:117-156, utils/point_utils.py:9-37) as
 * TENSORS, for callers that keep the reference's own objective instead of gsr_regularizer_*:
 *   out7[0:3] = allmap[2:5] @ world_view[:3,:3].T   (rend_normal, world space)      rot_host: HOST f32[9] = world_view[:3,:3]
 *   out7[3]   = (1-r) nan_to_num(allmap[0] / allmap[1]) + r nan_to_num(allmap[5])   (surf_depth)
 *   out7[4:7] = normalize(cross of the finite differences of the back-projected depth) * allmap[1], 0 on the border
 *               (surf_normal, world space; alpha not differentiated)                rays_world_host: HOST f32[9] =
 *               c2w[:3,:3] K^-1, ray(x,y) = that * [x,y,1]
 * backward: d_allmap f32[7,H,W] from d_out7 f32[7,H,W] (every element of both read / written; rend_alpha = allmap[1] and
 * rend_dist = allmap[6] are plain slices the caller differentiates itself).  At pixels with allmap[1] = 0 torch leaves 0 / 0
 * on channels 0 and 1; this writes 0 (no splat covers such a pixel and gsr_backward never reads its gradient). */
int32_t gsr_surface_maps_forward(const float* allmap, int32_t H, int32_t W, const float* rays_world_host,
                                 const float* rot_host, float depth_ratio, float* out7, gsr_stream_t stream);
int32_t gsr_surface_maps_backward(const float* allmap, int32_t H, int32_t W, const float* rays_world_host,
                                  const float* rot_host, float depth_ratio, const float* d_out7, float* d_allmap,
                                  gsr_stream_t stream);

/* The whole training objective from the partials of gsr_loss_forward (and, when reg_partials is
 * non-NULL, gsr_regularizer_forward) in one tiny launch:
 *   out5 = { (1-l)*l1 + l*(1-ssim) + ln*normal + ld*dist,  l1,  ssim,  mean normal error,  mean dist }
 * (train.py:113-143).  Fixed summation order => reproducible loss value. */
int32_t gsr_objective_finish(const float* loss_partials, int32_t C, int32_t H, int32_t W,
                             const float* reg_partials, float lambda_dssim, float lambda_normal,
                             float lambda_dist, float* out5, gsr_stream_t stream);

/* gsr_loss_backward and gsr_objective_finish as ONE launch: the backward kernel does not depend on the five scalars, so the
 * workgroup that computes them rides along with it (one kernel boundary less between the forward and the backward of the
 * objective).  For callers that run the backward right after the forward and read the loss value only afterwards: out5 is
 * written by THIS call, not by the forward.  Same values as the two separate calls.  (GsrRowScanJob: see gsr_row_scan_job.) */
int32_t gsr_loss_backward_finish(const float* img, const float* gt, const float* maps, int32_t C, int32_t H,
                                 int32_t W, float lambda_dssim, const float* grad_scale, float* dimg,
                                 const float* loss_partials, const float* reg_partials, float lambda_normal,
                                 float lambda_dist, float* out5 /* NULL: no scalars */,
                                 GsrRowScanJob* job /* NULL, or a job at stage 1: its second half rides along */,
                                 gsr_stream_t stream);
/* gsr_loss_forward with the first half of a row-scan job (NULL, or stage 0 -> 1) in extra workgroups of its launch.
 * out5_invalidate (may be NULL): five floats the launch fills with NaN -- the objective's scalars of a caller that lets
 * gsr_loss_backward_finish write them later, so that a value read before that is visibly invalid. */
int32_t gsr_loss_forward_job(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                             float* maps, float* partials, GsrRowScanJob* job, float* out5_invalidate, gsr_stream_t stream);

/* Dense Adam step over up to 8 parameter tensors in one launch; the update of
 * torch.optim.Adam as the reference configures it (scene/gaussian_model.py:282-295).  All arrays
 * are HOST arrays of length `count`; the pointers inside are device f32 buffers of numel[i]
 * elements.  step_size[i] = lr_i / (1 - beta1^t), inv_bc2_sqrt[i] = 1 / sqrt(1 - beta2^t). */
int32_t gsr_adam_step(int32_t count, float* const* params, const float* const* grads,
                      float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                      const float* step_size, const float* inv_bc2_sqrt, double beta1, double beta2,
                      double eps, gsr_stream_t stream);
/* The same step; keep_old (may be NULL, entries may be NULL): per tensor a device f32 buffer of numel[i] elements that
 * receives the parameter's values BEFORE the update, in the same pass (the factored SH step of the next view needs the
 * positions the backward saw while the positions move: one launch and 24 B per Gaussian less than a copy beside the step). */
int32_t gsr_adam_step_keep(int32_t count, float* const* params, const float* const* grads,
                           float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                           const float* step_size, const float* inv_bc2_sqrt, double beta1, double beta2,
                           double eps, float* const* keep_old, gsr_stream_t stream);

/* Adam step of the two SH parameter tensors (features_dc [N,1,3], features_rest [N,M-1,3]) from FACTORED gradients
 * (GSR_FLAG_FACTORED_SH_GRAD), for the Gaussians [first, first + count):
 *     grad[i,k,c] = grad_scale * sum_{r < n_views} basis_k( normalize(xyz[i] - campos[r]) ) * color_grad[r,i,c]
 * with basis_k = 0 above `sh_degree` (the reference's active degree), summed in view order (reproducible).  One launch:
 * the 48 gradient values of a Gaussian are formed in registers / LDS and never touch HBM.
 *   xyz         device f32 [N,3]: the positions the backward saw (update xyz AFTER this call, or pass a snapshot)
 *   color_grad  device f32 [n_views, view_stride] floats; view r's [N,3] block starts at color_grad + r * view_stride
 *   campos      device f32 [n_views, campos_stride]; n_views <= 16
 *   *_dc / *_rest: parameter, exp_avg, exp_avg_sq of the two tensors; step sizes as for gsr_adam_step.
 * n_views = 1, grad_scale = 1 reproduces gsr_backward's dL_dshs followed by gsr_adam_step. */
int32_t gsr_adam_sh_factored(int32_t first, int32_t count, int32_t sh_coeffs, int32_t sh_degree, const float* xyz,
                             int32_t n_views, const float* color_grad, int64_t view_stride, const float* campos,
                             int32_t campos_stride, float grad_scale,
                             float* p_dc, float* m_dc, float* v_dc, float step_size_dc, float inv_bc2_sqrt_dc,
                             float* p_rest, float* m_rest, float* v_rest, float step_size_rest, float inv_bc2_sqrt_rest,
                             double beta1, double beta2, double eps, gsr_stream_t stream);

/* The same step, followed -- for the same Gaussians, from the coefficients just written, still on chip -- by the SH colour of
 * the NEXT view: color_cache f32[13 n_total] = [ rgb [n_total,3] (+0.5, clamped at 0) | clamp bits u32 [n_total] |
 * d(rgb)/d(dir) [n_total,9] ] for camera position campos_next (device f32[3]), positions xyz_next (device [n_total,3]: the
 * positions the next forward will see, i.e. AFTER their own optimiser step) and active degree sh_degree_next.  A forward /
 * backward with GSR_FLAG_COLOR_CACHED then skips the SH colour pass and its 192-byte read per Gaussian altogether. */
int32_t gsr_adam_sh_factored_next(int32_t first, int32_t count, int32_t sh_coeffs, int32_t sh_degree, const float* xyz,
                                  int32_t n_views, const float* color_grad, int64_t view_stride, const float* campos,
                                  int32_t campos_stride, float grad_scale,
                                  float* p_dc, float* m_dc, float* v_dc, float step_size_dc, float inv_bc2_sqrt_dc,
                                  float* p_rest, float* m_rest, float* v_rest, float step_size_rest, float inv_bc2_sqrt_rest,
                                  double beta1, double beta2, double eps,
                                  const float* xyz_next, const float* campos_next, int32_t sh_degree_next, int32_t n_total,
                                  float* color_cache, gsr_stream_t stream);

/* Row compaction of the per-Gaussian tensors (pruning, scene/gaussian_model.py:398-470: `tensor[mask]` for the six
 * parameters, their Adam moments and the densification statistics).  Two calls:
 *   gsr_compact_plan  : exclusive scan of the device bool mask `keep` [n_rows] into workspace memory;
 *                       *offsets_out (device u32 [n_rows + 1], inside `ws`) maps row -> new row, its last entry is
 *                       the number of rows kept -- the caller reads it to size the destination tensors;
 *   gsr_compact_apply : ONE launch moving up to 24 tensors (row sizes multiples of 4 bytes, host arrays of device
 *                       pointers) from src[i] to dst[i]. */
size_t gsr_compact_workspace_bytes(int64_t n_rows);
int32_t gsr_compact_plan(const uint8_t* keep, int64_t n_rows, void* ws, size_t ws_bytes,
                         const uint32_t** offsets_out, gsr_stream_t stream);
int32_t gsr_compact_apply(int32_t count, const void* const* src, void* const* dst, const int32_t* row_bytes,
                          int64_t n_rows, const uint8_t* keep, const uint32_t* offsets, gsr_stream_t stream);

/* Densification statistics of one iteration (train.py:199-203; scene/gaussian_model.py:551-553) in one launch, no host
 * synchronisation: for radii[i] > 0: max_radii2D[i] = max(max_radii2D[i], radii[i]); xyz_gradient_accum[i] +=
 * ||grad2d[i,:]||; denom[i] += 1.  grad2d = means2D.grad [N,3]; the three state arrays are device f32 [N]. */
int32_t gsr_densify_stats(int32_t n, const int32_t* radii, const float* grad2d, float* max_radii2D,
                          float* xyz_gradient_accum, float* denom, gsr_stream_t stream);

/* Opt-in per-kernel timing with HIP events on the launch stream (bench.py's roofline figures).
 * `mask`: bit k enables kernel k in the order of the names below (-1 = all, 0 = off); timing only
 * the few big kernels keeps the event overhead out of the measured step.
 * Kernel names: "preprocess_fwd", "sort_hist", "sort_scatter", "scan", "emit_instances",
 * "finalize_bins", "render_fwd", "render_bwd", "preprocess_bwd", "knn", "loss_fwd", "loss_bwd", "regularizer_fwd", "regularizer_bwd", "adam". */
void gsr_profile_enable(int32_t mask);
void gsr_profile_reset(void);
int32_t gsr_profile_read(const char* kernel, double* total_ms, int32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_ */
