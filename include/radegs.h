/* radegs.h -- C ABI of libradegs_hip.so, the MI355X (gfx950) differentiable Gaussian-splat
 * rasterizer behind RaDe-GS's `diff_gaussian_rasterization` operator.
 *
 * This is the drop-in boundary for the one hot path this repository accelerates.  Every entry
 * point replaces one static method of the reference's native interface
 * (DGR = /root/reference/submodules/diff-gaussian-rasterization):
 *
 *   radegs_forward       <-  CudaRasterizer::Rasterizer::forward    DGR/cuda_rasterizer/rasterizer.h:31-63
 *                            (called from RasterizeGaussiansCUDA,    DGR/rasterize_points.cu:36-133)
 *   radegs_backward      <-  CudaRasterizer::Rasterizer::backward   DGR/cuda_rasterizer/rasterizer.h:65-110
 *                            (called from RasterizeGaussiansBackwardCUDA, DGR/rasterize_points.cu:136-246)
 *   radegs_mark_visible  <-  CudaRasterizer::Rasterizer::markVisible DGR/cuda_rasterizer/rasterizer.h:24-29
 *                            (called from markVisible,               DGR/rasterize_points.cu:248-267)
 *   radegs_integrate     <-  CudaRasterizer::Rasterizer::integrate  DGR/cuda_rasterizer/rasterizer.h:112-147
 *                            (called from IntegrateGaussiansToPointsCUDA, DGR/rasterize_points.cu:269-388)
 *
 * Conventions kept from the reference:
 *   - plain device pointers + sizes; NULL means "tensor not provided" (the reference tests
 *     data_ptr()==nullptr: forward.cu:363,407, backward.cu:622,626, rasterizer_impl.cu:394,494,540);
 *   - the three scratch buffers (geometry / binning / image state) are obtained through
 *     allocator callbacks -- the C form of std::function<char*(size_t)> -- so the caller owns
 *     the memory (torch tensors in the Python binding), may keep it for backward, and hands the
 *     same pointers back.  Their internal layout is private and self-describing from (P, R, W*H);
 *   - viewmatrix/projmatrix are the transposed 4x4 float matrices the reference passes;
 *   - image outputs are CHW float32; `radii` is int32[P].
 * Differences, all deliberate:
 *   - every call takes a HIP stream (the reference launches on the legacy default stream);
 *   - errors are returned as negative codes + radegs_last_error() instead of C++ exceptions;
 *   - output images for modes that are switched off are not touched (the binding zero-fills
 *     them, as the reference's torch::full does).
 * No torch types appear here; `stream` is a hipStream_t passed as void*.
 */
#ifndef RADEGS_H_INCLUDED
#define RADEGS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RADEGS_OK 0
#define RADEGS_ERR_INVALID_ARG (-1)
#define RADEGS_ERR_ALLOC (-2)
#define RADEGS_ERR_HIP (-3)
#define RADEGS_ERR_NO_DEVICE (-4)
#define RADEGS_ERR_STATE (-5)   /* an earlier radegs_backward on this thread and device was handed an image buffer that does not hold what
                                   its forward wrote (reused or overwritten state): that call's gradients are invalid */

/* Allocator callback: return a DEVICE pointer to at least `nbytes` bytes (256-B aligned), or
 * NULL on failure.  Mirrors the resize lambdas of DGR/rasterize_points.cu:27-33. */
typedef void* (*radegs_alloc_fn)(void* user, size_t nbytes);

typedef struct RadegsFwdArgs {
  int P;              /* number of Gaussians */
  int D;              /* active SH degree */
  int M;              /* SH coefficients per Gaussian in `shs` (rows of the (P,M,3) tensor) */
  int width, height;
  const float* background;      /* [3] device */
  const float* means3D;         /* [P,3] */
  const float* shs;             /* [P,M,3] or NULL */
  const float* colors_precomp;  /* [P,3] or NULL (exactly one of shs / colors_precomp) */
  const float* opacities;       /* [P] */
  const float* scales;          /* [P,3] or NULL */
  const float* rotations;       /* [P,4] (r,x,y,z), not normalised here; or NULL */
  const float* cov3D_precomp;   /* [P,6] or NULL (exactly one of scales+rotations / cov3D) */
  const float* viewmatrix;      /* [16] device, transposed */
  const float* projmatrix;      /* [16] device, transposed */
  const float* cam_pos;         /* [3] device */
  float scale_modifier, tan_fovx, tan_fovy, kernel_size;
  int prefiltered;              /* must be 0 */
  int require_coord, require_depth, debug;
  float* out_color;             /* [3,H,W] */
  /* The five maps below are PRODUCED iff their flag is set.  A map the flags do not produce is all-zero in the reference
   * (torch::full(0), DGR/rasterize_points.cu:71-77): when its pointer is non-NULL the forward ZERO-FILLS it (every pixel, inside the
   * blend kernel -- no separate fill); pass NULL for a map you do not want touched. */
  float* out_coord;             /* [3,H,W]  produced iff require_coord */
  float* out_mcoord;            /* [3,H,W]  produced iff require_coord */
  float* out_depth;             /* [1,H,W]  produced iff require_depth */
  float* out_mdepth;            /* [1,H,W]  produced iff require_depth */
  float* out_alpha;             /* [1,H,W] */
  float* out_normal;            /* [3,H,W]  produced iff require_coord || require_depth */
  int* radii;                   /* [P] */
} RadegsFwdArgs;

/* Returns num_rendered (>= 0) or a negative RADEGS_ERR_*.
 * Host synchronisation: the FIRST call for a (device, width, height) waits for num_rendered in the middle of the forward to
 * size the binning buffer, like rasterizer_impl.cu:354.  Later calls allocate for a capacity predicted from the previous
 * counts, queue the whole forward, and wait only for the 4-byte count at the very end (a mapped host word the emission kernel
 * writes, polled -- neither an event nor a stream sync); a too small prediction is detected there and the forward is redone with
 * exact sizes.  RADEGS_SPECULATE=0 restores the first
 * behaviour for every call.  The allocators may therefore be asked for MORE than the exact state size, and -- on a redo --
 * a second time within one call.  The image-state callback is invoked after the capacity is known (its tail holds the
 * sub-tile entry streams of the blend stage when the scene's splats are small). */
int radegs_forward(const RadegsFwdArgs* args, radegs_alloc_fn geom_alloc, void* geom_user, radegs_alloc_fn binning_alloc,
                   void* binning_user, radegs_alloc_fn image_alloc, void* image_user, void* stream);

typedef struct RadegsBwdArgs {
  size_t struct_size;           /* sizeof(RadegsBwdArgs) of the header the CALLER was compiled against.  The structure has grown at its
                                   tail between releases; radegs_backward / radegs_backward_from_sums refuse a size they do not know
                                   (RADEGS_ERR_INVALID_ARG) instead of reading hooks past the end of a shorter structure */
  int P, D, M, R;               /* R = num_rendered returned by the forward */
  int width, height;
  const float* background;
  const float* means3D;
  const float* shs;
  const float* colors_precomp;
  const float* alphas;          /* out_alpha of the forward */
  const float* scales;
  const float* rotations;
  const float* cov3D_precomp;
  const float* viewmatrix;
  const float* projmatrix;
  const float* cam_pos;
  float scale_modifier, tan_fovx, tan_fovy, kernel_size;
  const int* radii;
  const float* normalmap;       /* out_normal of the forward */
  void* geom_buffer;            /* the three buffers the forward filled */
  void* binning_buffer;
  void* image_buffer;
  const float* dL_dpix;         /* [3,H,W] */
  const float* dL_dpix_coord;   /* [3,H,W] */
  const float* dL_dpix_mcoord;  /* [3,H,W] */
  const float* dL_dpix_depth;   /* [1,H,W] */
  const float* dL_dpix_mdepth;  /* [1,H,W] */
  const float* dL_dalphas;      /* [1,H,W] */
  const float* dL_dpix_normal;  /* [3,H,W] */
  /* outputs; every element is written by the call (no pre-zeroing needed) */
  float* dL_dmean2D;            /* [P,3]  x,y signed, z = abs-grad sum */
  float* dL_dcolor;             /* [P,3] */
  float* dL_dopacity;           /* [P,1] */
  float* dL_dmean3D;            /* [P,3] */
  float* dL_dcov3D;             /* [P,6] */
  float* dL_dsh;                /* [P,M,3] or NULL when M == 0 */
  float* dL_dscale;             /* [P,3] */
  float* dL_drot;               /* [P,4] */
  int require_coord, require_depth, debug;
  /* optional [P,3]: dL/dRGB of the SH colour with the clamp mask applied.  When given, dL_dsh may be NULL (it is then not
   * written): the SH gradient is the outer product basis(dir) x this vector and can be rebuilt with
   * radegs_sh_grad_from_views -- which is how the view-parallel exchange moves 12 instead of 192 bytes per Gaussian. */
  float* dL_drgb_clamped;
  /* 0 (default): the gradient the reference EXECUTES.  Rasterizer::backward passes `(float4*)dL_dconic` in the position of
   * BACKWARD::preprocess's `conic_opacity` parameter (rasterizer_impl.cu:568 vs backward.h:94), so computeCov2DCUDA's
   * `combined_opacity` (backward.cu:179-180) is the accumulated conic gradient dL_dconic[idx].w, not opacity*coef; it scales the
   * derivative of the opacity-compensation factor w.r.t. the 2D covariance (backward.cu:367-375) and from there dL_dcov3D /
   * dL_dmeans3D / dL_dscales / dL_drotations.  Negligible (~1e-6 of the conic term) at the reference's default kernel_size = 0.
   * 1: the derivative the formulas intend (combined_opacity = opacity*coef) -- what an upstream fix would compute. */
  int opacity_grad_intended;
  /* Optional hand-off for a view-parallel caller (needs dL_drgb_clamped): dL_drgb_clamped is final as soon as the blend backward has
   * run, one kernel before everything else -- it is then written by a small kernel of its own and `drgb_ready(drgb_ready_user)` is
   * called ON THE HOST once that kernel is queued on `stream`, BEFORE the per-Gaussian backward (~0.16 ms at 1M Gaussians) is queued:
   * the caller records an event there and starts its all-gather of these 12-byte rows on another stream, under that kernel. */
  void (*drgb_ready)(void* user);
  void* drgb_ready_user;
  /* Optional, for the same caller: with grad_chunks >= 2 the per-Gaussian backward is queued as that many launches over consecutive
   * ranges of Gaussians, and after each one `grads_ready(grads_ready_user, first, count)` is called ON THE HOST: once `stream` reaches
   * that point, rows [first, first + count) of dL_dmean3D / dL_dopacity / dL_dscale / dL_drot (and of every other returned gradient)
   * are final -- the caller records an event and starts that part of its all-reduce on another stream, under the launches that
   * follow.  0 or 1: one launch, no call. */
  int grad_chunks;
  void (*grads_ready)(void* user, int first, int count);
  void* grads_ready_user;
  /* Inspection (tests): 1 = after the call the accumulation scratch holds every visible Gaussian's sums in the order and units of
   * radegs_backward_from_sums' `sums`, except the constant factors listed there (1/focal on the plane sums, W/2 and H/2 on mean2D).
   * Without it the record's mean2D / conic slots may hold the blend backward's private intermediate (raw moments, csrc/rg_streams.inc). */
  int keep_sums;
  /* 1 = the caller promises that the buffer `accum_alloc` is about to hand out is ALL ZEROS, and wants it back all zeros: the call then
   * skips its fill of the scratch (64 | 128 B per Gaussian: 10 us at 1M Gaussians) and the per-Gaussian kernel clears every record it
   * consumes.  Meant for a caller that keeps ONE scratch buffer per (device, stream) across calls: zero it once, pass 1 from then on,
   * and fall back to 0 after any call that returned an error or ran with keep_sums (the buffer is then in an unknown state).
   * 0: the scratch may hold anything; it is filled with zeros first and left as the kernels leave it. */
  int acc_reuse;
} RadegsBwdArgs;

/* `accum_alloc` provides the per-Gaussian accumulation scratch (64 or 128 B per Gaussian). */
int radegs_backward(const RadegsBwdArgs* args, radegs_alloc_fn accum_alloc, void* accum_user, void* stream);
/* The SECOND half of radegs_backward on caller-supplied per-Gaussian sums (inspection / parity hook, like radegs_debug_export): the
 * per-Gaussian backward (computeCov2DCUDA + preprocessCUDA backward, DGR/cuda_rasterizer/backward.cu:145-628) runs over `sums` instead
 * of over what the blend backward accumulated.  sums: device [P][16] floats ([P][32] with require_coord) in the order
 *   dL_dcolors[3], dL_dts, dL_dray_planes[2], dL_dnormals[3], dL_dmeans2D[3], dL_dconic.{x,y,w}, dL_dopacity (the render kernel's raw sum,
 *   before backward.cu:395-403 rescales it), and with require_coord: dL_dview_points[3], dL_dcamera_planes[6], 7 unused
 * holding the values the reference's render kernel leaves in those arrays (rasterizer_impl.cu:541-555).  What summation order does
 * to the gradients is thereby taken out of a comparison: fed with the reference's own sums, every returned gradient must equal the
 * reference's (tests/test_gpu_vs_compiled_reference.py).  Of `args`, geom_buffer, radii, means3D, scales + rotations (or cov3D_precomp),
 * shs (when given), the three camera pointers and the gradient outputs are read -- all must be valid device pointers (NULL is refused);
 * `sums` must be 16-byte aligned (the records are read as 16-byte pieces).  Nothing is allocated. */
int radegs_backward_from_sums(const RadegsBwdArgs* args, const float* sums, void* stream);

/* dL_dsh[P,M,3] = scale * sum_v basis(normalize(means3D - campos[v])) (x) drgb_clamped[v]   (rows beyond (D+1)^2 zero).
 * campos: [nviews,3], drgb_clamped: [nviews,P,3] -- the all-gathered per-view outputs of radegs_backward. */
int radegs_sh_grad_from_views(int P, int D, int M, int nviews, const float* means3D, const float* campos, const float* drgb_clamped,
                              float scale, float* dL_dsh, void* stream);

/* present[i] = 1 iff Gaussian i passes the near-plane test (view z > 0.2). */
int radegs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, unsigned char* present,
                        void* stream);

typedef struct RadegsIntegrateArgs {
  int P, D, M;                  /* Gaussians, as in RadegsFwdArgs */
  int PN;                       /* number of query points */
  int width, height;
  const float* background;      /* [3] */
  const float* means3D;         /* [P,3] */
  const float* shs;             /* [P,M,3] or NULL */
  const float* colors_precomp;  /* [P,3] or NULL */
  const float* opacities;       /* [P] */
  const float* scales;          /* [P,3] or NULL */
  const float* rotations;       /* [P,4] or NULL */
  const float* cov3D_precomp;   /* [P,6] or NULL */
  const float* viewmatrix;      /* [16] transposed */
  const float* projmatrix;      /* [16] transposed */
  const float* cam_pos;         /* [3] */
  const float* points3D;        /* [PN,3] query points (world space) */
  float scale_modifier, tan_fovx, tan_fovy, kernel_size;
  int debug;
  /* outputs; every element is written by the call (initial values of rasterize_points.cu:312-320 included) */
  float* out_color;             /* [9,H,W]: rgb, expected depth, median depth, 0, max depth, alpha, #points in the pixel */
  float* out_alpha_integrated;  /* [PN]   1 for points that do not project into the image */
  float* out_color_integrated;  /* [PN,3] colour of the pixel the point falls in */
  float* out_coordinate2d;      /* [PN,2] projected position in pixels */
  float* out_sdf;               /* [PN]   median-surface depth along the ray minus point depth; -1000 if not projected */
  int* radii;                   /* [P] */
} RadegsIntegrateArgs;

/* Integrates the Gaussians' opacity along the ray of every query point (GOF-style; used by mesh extraction).
 * `point_alloc` provides the point/INTE state (the reference's point + point-binning buffers).  Returns num_rendered.
 * Documented deviations from the reference's undefined behaviour: (1) for ill-conditioned Gaussians (smallest
 * covariance eigenvalue <= 1e-8) the inverse ray-space covariance is ZERO: upstream's shadowed variable
 * (forward.cu:223) stores an uninitialised matrix there; (2) contributor ids are 32-bit (upstream truncates to uint16, wrong only for tile lists longer than
 * 65535 entries); (3) the exponent of the point opacity is clamped at +80 (a non-PSD ill-conditioned matrix would
 * otherwise overflow expf). */
int radegs_integrate(const RadegsIntegrateArgs* args, radegs_alloc_fn geom_alloc, void* geom_user, radegs_alloc_fn binning_alloc,
                     void* binning_user, radegs_alloc_fn image_alloc, void* image_user, radegs_alloc_fn point_alloc, void* point_user,
                     void* stream);

/* Sizes of the geometry / image state for (P) and (W*H) -- the binning size depends on R and is
 * requested through the callback. */
size_t radegs_geometry_bytes(int P, int require_coord);
size_t radegs_image_bytes(int width, int height);
size_t radegs_binning_bytes(int R);

/* Test/inspection hook: copy a named private array out of the state buffers into `dst`
 * (device memory, dst_bytes large enough).  Names: "point_list" u32[R], "ranges" u32[2*tiles],
 * "n_contrib" u32[2*H*W], "tiles_touched" u32[P], "splat_a" f32[P,16], "splat_b" f32[P,12],
 * "clamped" u8[P] (bits 0..2: SH channel clamped at 0; bit 3: the eigen-solver converged), "depth_key" u32[P], "blk_count" /
 * "blk_consumed" u32[8*tiles] (entry streams).  Returns bytes copied or a negative error. */
long long radegs_debug_export(const char* name, int P, int R, int width, int height, int require_coord, const void* geom_buffer,
                              const void* binning_buffer, const void* image_buffer, void* dst, size_t dst_bytes, void* stream);

/* Speculative binning (see radegs_forward): forwards that ran on a predicted capacity / those whose prediction was too small and
 * were redone with exact sizes, since the last reset. */
void radegs_binning_stats(unsigned long long* speculative_calls, unsigned long long* misses, int reset);
/* Which blend formulation the last radegs_forward of the calling thread used: 1 = sub-tile entry streams, 0 = tile-wide kernels,
 * -1 = no forward yet (bench.py reports the native decision instead of mirroring the selection rule). */
int radegs_last_forward_used_streams(void);

/* The library reads its environment switches once, at first use.  A host that changes them afterwards --
 * the test-suite does, between cases of one process -- calls this to have them read again. */
void radegs_reload_env(void);

/* Per-stage timing with HIP events recorded on the launch stream (used by bench.py for the live
 * roofline measurement).  enable(1) -> every subsequent forward/backward records an event pair per
 * stage; collect() waits for them, adds each stage's elapsed ms / launch count into the arrays
 * (n >= radegs_profile_num_stages()) and clears the log. */
void radegs_profile_enable(int on);
/* stage >= 0: record events for that stage only (each recorded stage boundary costs ~10 us of stream bubble, so a timed
 * run should select just the kernel it reports); -1: all stages. */
void radegs_profile_select(int stage);
/* With a selected stage: time only every `every`-th launch of it (an event pair costs ~10 us of stream bubble; 1 = every launch). */
void radegs_profile_stride(int every);
int radegs_profile_num_stages(void);
const char* radegs_profile_stage_name(int i);
int radegs_profile_collect(float* ms_total, int* count, int n);

/* The library remembers, by ADDRESS, which image-state buffers hold entry streams written by their forward (radegs_backward replays
 * them only then; otherwise it runs the tile-wide kernels, which are valid after either forward).  Tell it when such a buffer is
 * freed or moved, so that an unrelated buffer landing on the same address later is not mistaken for one.  (The Python binding
 * does this from the tensor's finaliser and on resize.) */
void radegs_forget_image(const void* image_buffer);

const char* radegs_last_error(void);
const char* radegs_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * The step that follows the rasterizer in every regularised training iteration: two depth (or
 * coordinate) maps -> two normal maps by central differences, and the normal-consistency loss against the rendered
 * normal map.  Replaces, with one kernel per direction, the torch-eager
 *     depths_double_to_points / point_double_to_normal / depth_double_to_normal   utils/graphics_utils.py:97-127
 *     normal_error_map / depth_normal_loss                                        train.py:152-155
 * and their autograd backward.  All maps are float32 device pointers; normal maps are [2,3,H,W] (map 1 first).
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct RadegsNormalArgs {
  int width, height;
  int points;          /* 0: map1/map2 are depth maps [1,H,W] (depth_double_to_normal); 1: coordinate maps [3,H,W] */
  double fovx, fovy;   /* view.FoVx / view.FoVy in radians (used for depth maps only) */
  const float* map1;   /* expected depth / expected coord */
  const float* map2;   /* median depth / median coord */
} RadegsNormalArgs;

int radegs_normals_forward(const RadegsNormalArgs* args, float* out_normals /* [2,3,H,W] */, void* stream);
/* grad_map1/2 have the shape of map1/2; every element is written */
int radegs_normals_backward(const RadegsNormalArgs* args, const float* grad_normals /* [2,3,H,W] */, float* grad_map1, float* grad_map2,
                            void* stream);
/* loss = (1-depth_ratio) * mean(1 - n.N_1) + depth_ratio * mean(1 - n.N_2), means over all H*W pixels (border: N = 0).
 * out_loss3 = {loss, mean error map 1, mean error map 2}; scratch: radegs_normal_loss_scratch_bytes() bytes. */
size_t radegs_normal_loss_scratch_bytes(int width, int height);
int radegs_normal_loss_forward(const RadegsNormalArgs* args, const float* rendered_normal /* [3,H,W] */, float depth_ratio, void* scratch,
                               float* out_loss3, void* stream);
/* upstream: device scalar d(objective)/d(loss) or NULL (= 1) */
int radegs_normal_loss_backward(const RadegsNormalArgs* args, const float* rendered_normal, float depth_ratio, const float* upstream,
                                float* grad_map1, float* grad_map2, float* grad_rendered_normal /* [3,H,W] */, void* stream);
const char* radegs_normals_last_error(void);

/* ---------------------------------------------------------------------------------------------------------------
 * The step that precedes the rasterizer in every render() call: parameter activations fused with
 * the 3D (mip) filter -- GaussianModel.get_scaling_n_opacity_with_3D_filter, scene/gaussian_model.py:156-166.
 *   scales[P,3] = sqrt(exp(scaling_raw)^2 + filter_3D^2);  opacity[P] = sigmoid(opacity_raw) * sqrt(det1/det2)
 * backward: grad_scales / grad_opacity may be NULL (= zero cotangent); filter_3D carries no gradient upstream either.
 * --------------------------------------------------------------------------------------------------------------- */
int radegs_filter3d_forward(int P, const float* scaling_raw /* [P,3] */, const float* opacity_raw /* [P,1] */,
                            const float* filter_3D /* [P,1] */, float* scales_out /* [P,3] */, float* opacity_out /* [P,1] */,
                            void* stream);
int radegs_filter3d_backward(int P, const float* scaling_raw, const float* opacity_raw, const float* filter_3D, const float* grad_scales,
                             const float* grad_opacity, float* grad_scaling_raw /* [P,3] */, float* grad_opacity_raw /* [P,1] */,
                             void* stream);
/* GaussianModel.compute_3D_filter (scene/gaussian_model.py:179-232) over all cameras in one pass.  cameras16: [ncam][16]
 * device floats = R (3x3 row-major as stored: p_cam = p @ R + T), T (3), focal_x, focal_y, width, height.
 * focal_length: max focal_x over the cameras (host).  scratch_distance: [P] floats, scratch_max: 1 uint32 (device).
 * filter_3D[P] = min-depth / focal_length * sqrt(0.2); Gaussians no camera sees get the largest valid depth. */
int radegs_compute_filter3d(int P, const float* xyz, int ncam, const float* cameras16, float focal_length, float* scratch_distance,
                            unsigned* scratch_max, float* filter_3D, void* stream);


/* ---------------------------------------------------------------------------------------------------------------
 * The photometric loss that closes every training iteration:
 *     (1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt))      train.py:159, utils/loss_utils.py:17-63
 * image / gt: [C,H,W] float32.  forward writes out_loss3 = {loss, l1, ssim}; when `dmaps` ([3,C,H,W]) is given it also
 * stores the per-pixel SSIM derivative maps the backward consumes.  backward: grad_image = coef2[0] * d l1/d image +
 * coef2[1] * d ssim/d image (coef2: two DEVICE floats, e.g. {g*(1-lambda), -g*lambda} for upstream gradient g).
 * --------------------------------------------------------------------------------------------------------------- */
size_t radegs_photometric_scratch_bytes(int width, int height, int channels);
int radegs_photometric_forward(int width, int height, int channels, const float* image, const float* gt, float lambda_dssim, void* scratch,
                               float* dmaps, float* out_loss3, void* stream);
int radegs_photometric_backward(int width, int height, int channels, const float* image, const float* gt, const float* dmaps,
                                const float* coef2, float* grad_image, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Multi-tensor Adam step: torch.optim.Adam(l, lr=0.0, eps=1e-15) of scene/gaussian_model.py:338-349
 * (no weight decay, no amsgrad) over up to RADEGS_ADAM_MAX_TENSORS parameter tensors in one launch.  `step` is the
 * step count AFTER this update (>= 1); `lr` the group's current learning rate.  All pointers: float32, device.
 * --------------------------------------------------------------------------------------------------------------- */
#define RADEGS_ADAM_MAX_TENSORS 16
typedef struct RadegsAdamTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  unsigned long long numel;
  float lr;
  double step;
} RadegsAdamTensor;
int radegs_adam_step(int count, const RadegsAdamTensor* tensors, double beta1, double beta2, double eps, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * simple_knn._C.distCUDA2 (scene/gaussian_model.py:315): out[i] = mean of the squared distances from points[i] to its 3
 * nearest neighbours.  The reference imports it from the un-vendored `simple-knn` submodule; restated from that library's
 * published algorithm (Morton order + boxes of 1024 points + exact rejection search).  points: [P,3] float32 device.
 * --------------------------------------------------------------------------------------------------------------- */
size_t radegs_knn_scratch_bytes(int P);
int radegs_knn_mean_dist2(int P, const float* points, void* scratch, float* out /* [P] */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RADEGS_H_INCLUDED */
