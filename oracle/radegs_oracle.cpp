// =====================================================================================
// TEST INFRASTRUCTURE ONLY.  CPU oracle for the RaDe-GS differentiable splat rasterizer.
//
// This file is a CPU restatement of the reference's CUDA rasterizer
// (/root/reference/submodules/diff-gaussian-rasterization, abbreviated DGR/ below); each
// function cites the reference file:line whose arithmetic it follows.  It exists so the
// HIP product path can be checked against something: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it.  The product (rade-gs_amd/) never links,
// imports or falls back to anything in oracle/.
//
// PARITY STATUS: PINNED to the reference's own code run here.  The reference ships no test,
// golden vector or fixture for this path and its CUDA build cannot run
// in this image, but its sources compile for the host (oracle/build_ref.py -> oracle/_ref:
// forward.cu, backward.cu, rasterizer_impl.cu, auxiliary.h unmodified, on a CUDA execution
// model made of fibers, with a restatement of the un-vendored glm subset).  This oracle
// equals that build BIT FOR BIT -- every state array, all 7 maps, all 8 gradients and their
// intermediate sums, and integrate()'s 6 outputs -- on every scene of tests/test_ref_parity.py,
// and reproduces the golden vectors that build wrote (tests/golden/g_*.npz).  What the
// comparison cannot cover is what nothing off-device can: CUDA's expf and nvcc's fma
// contraction (both sides use exp_spec and -ffp-contract=off; tests/test_oracle_exp_sensitivity.py
// and tests/test_ref_parity.py::test_fma_contraction_sensitivity measure what they move).
// Additionally: analytic known-answer cases, a float64 finite-difference and a PyTorch-autograd
// cross-check of the hand-derived backward (tests/test_oracle_*.py).
//
// Numerics: Real=float reproduces the reference's fp32 arithmetic with one rounding per
// operation (build with -ffp-contract=off), including its double-precision
// sub-expressions (auxiliary.h:57-60, forward.cu:119-124, backward.cu:215-218,367-375).
// Real=double is used only to validate derivatives.
// One documented deviation: CUDA's expf() cannot be reproduced bit-for-bit off-device,
// so the blend loop calls exp_spec() (a fully specified Cody-Waite + degree-5 polynomial
// exponential, <= 1 ulp from expf on the relevant range); the HIP kernels implement the
// same specification, which makes every thresholded decision (alpha < 1/255,
// T(1-alpha) < 1e-4, T > 0.5) reproducible between oracle and device.
// =====================================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle_eigen.h"
#include "oracle_linalg.h"

namespace orc {

static constexpr int TILE = 16;  // BLOCK_X = BLOCK_Y, DGR/cuda_rasterizer/config.h:15-16

// ---------------------------------------------------------------- specified exp ----
// exp_spec(x), x <= ~0: k = rint(x*log2e); r = x - k*ln2 (two-step Cody-Waite with
// fma); degree-5 polynomial (Cephes expf coefficients) evaluated with fma; scale by 2^k
// through the exponent field.  x < -87 returns 0 (keeps 2^k normal).  The HIP kernels
// restate this sequence operation for operation (csrc/rg_blend.h).
// g_exp_mode (oracle_set_exp_mode; tests/test_oracle_exp_sensitivity.py only): 0 = the specification; 1 = the C library's
// expf; 2 / 3 = the specification moved one ulp up / down.  The non-zero modes stand in for "a build whose expf rounds
// differently" (CUDA's expf is not reproducible off-device) to measure how many thresholded decisions that can flip.
static int g_exp_mode = 0;
// g_opacity_slip (oracle_set_opacity_slip): 1 (default) = what the reference EXECUTES.  Rasterizer::backward hands BACKWARD::preprocess
// `(float4*)dL_dconic` where the callee's parameter list has `conic_opacity` (rasterizer_impl.cu:568 against backward.h:94 /
// backward.cu:1059), so computeCov2DCUDA's `combined_opacity = conic_opacity[idx].w` (backward.cu:179-180) reads the accumulated
// conic gradient dL_dconic[idx].w instead of opacity*coef.  The value only enters the derivative of the opacity-compensation
// factor w.r.t. the 2D covariance (backward.cu:367-375 -> dL_da/db/dc): with kernel_size = 0 (the reference default) that term
// is ~1e-6 of the conic term, with kernel_size = 0.1 it is not small.  Found by running the reference's own sources on the host
// (oracle/_ref).  0 = the derivative the formulas intend (used by the finite-difference / autograd checks of the calculus).
static int g_opacity_slip = 1;
// g_ref_order (oracle_set_ref_order; fp32 only): 0 (default) = per-Gaussian sums of the blend backward accumulated in double, in
// any order (deterministic, and what the HIP path's fp32 sums are judged against).  1 = the sums are formed in fp32 in the order
// the reference's float atomics are applied when its kernel runs one block at a time with the threads of a block advancing
// batch by batch (tile, batch of 256 list entries from the back, thread rank, entry) -- the schedule of oracle/_ref's host run --
// so that the oracle's backward can be compared with the compiled reference BIT FOR BIT (tests/test_ref_parity.py).
static int g_ref_order = 0;
struct AddLogEntry { uint32_t key, slot; float v; };
struct AddLog { std::vector<AddLogEntry>* log = nullptr; const double* base = nullptr; uint32_t key = 0; };
static thread_local AddLog tl_addlog;
inline float exp_spec_impl(float x);
inline float exp_spec(float x) {
  if (g_exp_mode == 0) return exp_spec_impl(x);
  if (g_exp_mode == 1) return std::exp(x);
  const float y = exp_spec_impl(x);
  return y > 0.0f ? std::nextafter(y, g_exp_mode == 2 ? 2.0f * y : 0.0f) : y;
}
inline float exp_spec_impl(float x) {
  if (x < -87.0f) return 0.0f;
  const float kf = std::rint(x * 1.44269504088896341f);
  float r = __builtin_fmaf(kf, -0.693359375f, x);
  r = __builtin_fmaf(kf, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  const float y = __builtin_fmaf(p, r2, r) + 1.0f;
  int32_t bits;
  std::memcpy(&bits, &y, 4);
  bits += static_cast<int32_t>(kf) << 23;
  float out;
  std::memcpy(&out, &bits, 4);
  return out;
}
inline double exp_spec(double x) { return std::exp(x); }

// float -> int conversion with the device's saturating semantics (v_cvt_i32_f32):
// NaN -> 0, out of range -> INT_MIN/INT_MAX.  In-range values truncate toward zero
// exactly like the C cast the reference uses (auxiliary.h:65-70).
template <class R> inline int to_int_sat(R v) {
  if (v != v) return 0;
  if (v >= R(2147483648.0)) return 2147483647;
  if (v <= R(-2147483648.0)) return -2147483647 - 1;
  return static_cast<int>(v);
}

// getHigherMsb, DGR/cuda_rasterizer/rasterizer_impl.cu:35-50
inline uint32_t higher_msb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4, step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

// SH basis constants, DGR/cuda_rasterizer/auxiliary.h:35-52
static const float kC0 = 0.28209479177387814f;
static const float kC1 = 0.4886025119029199f;
static const float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                             -1.0925484305920792f, 0.5462742152960396f};
static const float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// auxiliary.h:74-113 (matrices are stored transposed: m[0],m[4],m[8],m[12] is math row 0)
template <class R> inline V3<R> xform_point43(const V3<R>& p, const R* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
template <class R> inline void xform_point44(const V3<R>& p, const R* m, R out[4]) {
  out[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
  out[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
  out[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
  out[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}
template <class R> inline V3<R> xform_vec43_T(const V3<R>& p, const R* m) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
          m[8] * p.x + m[9] * p.y + m[10] * p.z};
}
// dnormvdv(float3), auxiliary.h:124-134
template <class R> inline V3<R> dnormvdv(const V3<R>& v, const V3<R>& dv) {
  R sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  R invsum32 = R(1.0f) / std::sqrt(sum2 * sum2 * sum2);
  V3<R> o;
  o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return o;
}
// ndc2Pix, auxiliary.h:57-60 (evaluated in double)
template <class R> inline R ndc_to_pix(R v, int S) {
  return static_cast<R>(((static_cast<double>(v) + 1.0) * S - 1.0) * 0.5);
}
// getRect, auxiliary.h:62-72 (upper bound in the reference's left-to-right float order: ((p + r) + BLOCK) - 1)
template <class R>
inline void tile_rect(R px, R py, int max_radius, int gx, int gy, uint32_t rmin[2], uint32_t rmax[2]) {
  const R rad = static_cast<R>(max_radius);
  rmin[0] = std::min<int64_t>(gx, std::max(0, to_int_sat((px - rad) / R(TILE))));
  rmin[1] = std::min<int64_t>(gy, std::max(0, to_int_sat((py - rad) / R(TILE))));
  rmax[0] = std::min<int64_t>(gx, std::max(0, to_int_sat((((px + rad) + R(TILE)) - R(1)) / R(TILE))));
  rmax[1] = std::min<int64_t>(gy, std::max(0, to_int_sat((((py + rad) + R(TILE)) - R(1)) / R(TILE))));
}

// Shared geometry recomputation used by forward computeCov2D (forward.cu:77-264) and by
// computeCov2DCUDA (backward.cu:182-252): everything up to uvh_mn.
template <class R> struct Cov2DCommon {
  V3<R> t;            // clamped view-space mean
  R txtz, tytz;       // after clamping
  R x_grad_mul, y_grad_mul;
  M3<R> J, W, T, Vrk, cov;  // cov = T^T Vrk^T T (unfiltered)
  R det_0, det_1, coef_raw;  // coef_raw: before the "force to 0" rule
  int D;              // eigen-solver return
  V3<R> eval; M3<R> evec; unsigned min_id; bool well_conditioned;
  V3<R> evec_min; M3<R> Vrk_inv, cov_cam_inv;
  V3<R> uvh, uvh_m, uvh_mn;
};

template <class R>
inline void cov2d_common(const V3<R>& mean, R fx, R fy, R tan_fovx, R tan_fovy, R kernel_size, const R* cov3D,
                         const R* view, Cov2DCommon<R>& o) {
  V3<R> t = xform_point43(mean, view);
  const R limx = R(1.3f) * tan_fovx, limy = R(1.3f) * tan_fovy;
  R txtz = t.x / t.z, tytz = t.y / t.z;
  t.x = std::fmin(limx, std::fmax(-limx, txtz)) * t.z;
  t.y = std::fmin(limy, std::fmax(-limy, tytz)) * t.z;
  o.x_grad_mul = (txtz < -limx || txtz > limx) ? R(0) : R(1);
  o.y_grad_mul = (tytz < -limy || tytz > limy) ? R(0) : R(1);
  txtz = t.x / t.z;
  tytz = t.y / t.z;
  o.t = t; o.txtz = txtz; o.tytz = tytz;

  o.J = M3<R>(fx / t.z, R(0), -(fx * t.x) / (t.z * t.z), R(0), fy / t.z, -(fy * t.y) / (t.z * t.z), R(0), R(0), R(0));
  o.W = M3<R>(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
  o.T = o.W * o.J;
  o.Vrk = M3<R>(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  o.cov = transpose(o.T) * transpose(o.Vrk) * o.T;
  const M3<R>& c = o.cov;
  // forward.cu:119-121 / backward.cu:215-218 : max(1e-6, float expr) evaluated in double
  o.det_0 = static_cast<R>(std::fmax(1e-6, static_cast<double>(c[0][0] * c[1][1] - c[0][1] * c[0][1])));
  o.det_1 = static_cast<R>(std::fmax(
      1e-6, static_cast<double>((c[0][0] + kernel_size) * (c[1][1] + kernel_size) - c[0][1] * c[0][1])));
  o.coef_raw = static_cast<R>(std::sqrt(static_cast<double>(o.det_0) / (static_cast<double>(o.det_1) + 1e-6) + 1e-6));

  o.D = sym_eigen3(o.Vrk, o.eval, o.evec);
  const V3<R>& ev = o.eval;
  o.min_id = ev[0] > ev[1] ? (ev[1] > ev[2] ? 2 : 1) : (ev[0] > ev[2] ? 2 : 0);
  o.well_conditioned = static_cast<double>(ev[o.min_id]) > 0.00000001;  // float vs double literal, forward.cu:142
  if (o.well_conditioned) {
    M3<R> diag(1 / ev[0], R(0), R(0), R(0), 1 / ev[1], R(0), R(0), R(0), 1 / ev[2]);
    o.Vrk_inv = o.evec * diag * transpose(o.evec);
  } else {
    o.evec_min = o.evec[o.min_id];
    o.Vrk_inv = outer(o.evec_min, o.evec_min);
  }
  o.cov_cam_inv = transpose(o.W) * o.Vrk_inv * o.W;
  o.uvh = {txtz, tytz, R(1)};
  o.uvh_m = o.cov_cam_inv * o.uvh;
  o.uvh_mn = normalize(o.uvh_m);
}

template <class R> struct Oracle {
  // ---- inputs ----
  int P = 0, D = 0, M = 0, W = 0, H = 0, nthreads = 1;
  bool req_coord = false, req_depth = false;
  std::vector<R> means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp;
  bool has_sh = false, has_colors = false, has_scales = false, has_cov = false;
  R view[16], proj[16], campos[3], bg[3];
  R scale_modifier = 1, tan_fovx = 1, tan_fovy = 1, kernel_size = 0;
  // ---- derived ----
  R focal_x = 0, focal_y = 0;
  int gx = 0, gy = 0;
  // ---- geometry state (rasterizer_impl.h:29-48) ----
  std::vector<R> depths, camera_planes, ray_planes, ts, normals, means2D, view_points, cov3D, conic_opacity, rgb;
  std::vector<uint8_t> clamped;
  std::vector<int32_t> radii;
  std::vector<uint32_t> tiles_touched, point_offsets;
  // ---- binning state ----
  int num_rendered = 0;
  std::vector<uint64_t> keys_sorted;
  std::vector<uint32_t> point_list;
  std::vector<uint32_t> ranges;  // 2 per tile
  // ---- image state + outputs ----
  std::vector<uint32_t> n_contrib;  // 2*H*W
  std::vector<R> accum_coord, accum_depth, normal_length;
  std::vector<R> out_color, out_coord, out_mcoord, out_depth, out_mdepth, out_alpha, out_normal;
  // ---- backward intermediates (rasterize_points.cu:180-193) and results ----
  std::vector<R> dL_dmeans3D, dL_dview_points, dL_dmeans2D, dL_dcolors, dL_dts, dL_dcamera_planes, dL_dray_planes,
      dL_dnormals, dL_dconic, dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations;
  // render-bwd per-Gaussian accumulators before the per-Gaussian chain rule (kept for tests)
  std::vector<R> acc_dmeans2D, acc_dconic, acc_dopacity, acc_dcolors;
  // ---- integrate() path (GOF point integration) ----
  bool inte = false;                     // preprocess with the INTE template switch (forward.cu:187-235)
  int PN = 0;
  std::vector<R> invraycov, points3D, points2D, point_depths;
  std::vector<uint8_t> condition;
  std::vector<uint32_t> point_tiles, pt_list, point_ranges;
  std::vector<R> out9, final_T, out_alpha_integrated, out_color_integrated, out_coordinate2d, out_sdf;
  int64_t stat_pairs_fwd = 0;    // (pixel, list entry) pairs visited by the forward blend
  int64_t stat_blended_fwd = 0;  // pairs that passed every threshold and were blended

  std::map<std::string, std::pair<const void*, size_t>> registry;
  template <class T> void reg(const char* name, const std::vector<T>& v) {
    registry[name] = {static_cast<const void*>(v.data()), v.size() * sizeof(T)};
  }

  const R* cov3D_ptr(int idx) const { return has_cov ? &cov3D_precomp[6 * idx] : &cov3D[6 * idx]; }
  const R* feature_ptr() const { return has_colors ? colors_precomp.data() : rgb.data(); }

  // ------------------------------------------------------------------ forward ----
  // computeCov3D, forward.cu:270-304 (quaternion deliberately NOT normalised, :279)
  void cov3d_from_scale_rot(int idx, R* out) const {
    M3<R> S(R(1), R(0), R(0), R(0), R(1), R(0), R(0), R(0), R(1));
    S[0][0] = scale_modifier * scales[3 * idx + 0];
    S[1][1] = scale_modifier * scales[3 * idx + 1];
    S[2][2] = scale_modifier * scales[3 * idx + 2];
    const R r = rotations[4 * idx + 0], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2], z = rotations[4 * idx + 3];
    M3<R> Rm(R(1.f) - R(2.f) * (y * y + z * z), R(2.f) * (x * y - r * z), R(2.f) * (x * z + r * y),
             R(2.f) * (x * y + r * z), R(1.f) - R(2.f) * (x * x + z * z), R(2.f) * (y * z - r * x),
             R(2.f) * (x * z - r * y), R(2.f) * (y * z + r * x), R(1.f) - R(2.f) * (x * x + y * y));
    M3<R> Mm = S * Rm;
    M3<R> Sigma = transpose(Mm) * Mm;
    out[0] = Sigma[0][0]; out[1] = Sigma[0][1]; out[2] = Sigma[0][2];
    out[3] = Sigma[1][1]; out[4] = Sigma[1][2]; out[5] = Sigma[2][2];
  }

  // computeColorFromSH fwd, forward.cu:23-74
  void color_from_sh(int idx, R out[3]) {
    V3<R> pos{means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    V3<R> dir = pos - V3<R>{campos[0], campos[1], campos[2]};
    dir = dir / length(dir);
    auto sh = [&](int k) { return V3<R>{shs[(size_t(idx) * M + k) * 3], shs[(size_t(idx) * M + k) * 3 + 1], shs[(size_t(idx) * M + k) * 3 + 2]}; };
    V3<R> result = R(kC0) * sh(0);
    if (D > 0) {
      R x = dir.x, y = dir.y, z = dir.z;
      result = result - (R(kC1) * y) * sh(1) + (R(kC1) * z) * sh(2) - (R(kC1) * x) * sh(3);
      if (D > 1) {
        R xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        result = result + (R(kC2[0]) * xy) * sh(4) + (R(kC2[1]) * yz) * sh(5) +
                 (R(kC2[2]) * (R(2.0f) * zz - xx - yy)) * sh(6) + (R(kC2[3]) * xz) * sh(7) + (R(kC2[4]) * (xx - yy)) * sh(8);
        if (D > 2) {
          result = result + (R(kC3[0]) * y * (R(3.0f) * xx - yy)) * sh(9) + (R(kC3[1]) * xy * z) * sh(10) +
                   (R(kC3[2]) * y * (R(4.0f) * zz - xx - yy)) * sh(11) +
                   (R(kC3[3]) * z * (R(2.0f) * zz - R(3.0f) * xx - R(3.0f) * yy)) * sh(12) +
                   (R(kC3[4]) * x * (R(4.0f) * zz - xx - yy)) * sh(13) + (R(kC3[5]) * z * (xx - yy)) * sh(14) +
                   (R(kC3[6]) * x * (xx - R(3.0f) * yy)) * sh(15);
        }
      }
    }
    result = result + V3<R>{R(0.5f), R(0.5f), R(0.5f)};
    clamped[3 * idx + 0] = (result.x < 0);
    clamped[3 * idx + 1] = (result.y < 0);
    clamped[3 * idx + 2] = (result.z < 0);
    out[0] = std::fmax(result.x, R(0.0f));
    out[1] = std::fmax(result.y, R(0.0f));
    out[2] = std::fmax(result.z, R(0.0f));
  }

  // preprocessCUDA<3,false>, forward.cu:307-423 with computeCov2D<false>, forward.cu:77-264
  void preprocess_one(int idx) {
    radii[idx] = 0;
    tiles_touched[idx] = 0;
    V3<R> p_orig{means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    // in_frustum, auxiliary.h:155-180
    V3<R> p_view = xform_point43(p_orig, view);
    if (p_view.z <= R(0.2f)) return;
    R p_hom[4];
    xform_point44(p_orig, proj, p_hom);
    R p_w = R(1.0f) / (p_hom[3] + R(0.0000001f));
    R p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};

    const R* c3;
    if (has_cov) {
      c3 = &cov3D_precomp[6 * idx];
    } else {
      cov3d_from_scale_rot(idx, &cov3D[6 * idx]);
      c3 = &cov3D[6 * idx];
    }

    // ---- computeCov2D<false> ----
    Cov2DCommon<R> g;
    cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, kernel_size, c3, view, g);
    R cov2D[3] = {g.cov[0][0] + kernel_size, g.cov[0][1], g.cov[1][1] + kernel_size};
    R coef = g.coef_raw;
    if (static_cast<double>(g.det_0) <= 1e-6 || static_cast<double>(g.det_1) <= 1e-6) coef = R(0.0f);

    R* cp = &camera_planes[6 * idx];
    if (std::isnan(g.uvh_mn.x) || g.D == 0) {
      for (int ch = 0; ch < 6; ch++) cp[ch] = 0;
      normals[3 * idx] = normals[3 * idx + 1] = normals[3 * idx + 2] = 0;
      ray_planes[2 * idx] = ray_planes[2 * idx + 1] = 0;
    } else {
      const V3<R>& t = g.t;
      const R txtz = g.txtz, tytz = g.tytz;
      R u2 = txtz * txtz, v2 = tytz * tytz, uv = txtz * tytz;
      R l = std::sqrt(t.x * t.x + t.y * t.y + t.z * t.z);
      M3<R> nJ(1 / t.z, R(0), -(t.x) / (t.z * t.z), R(0), 1 / t.z, -(t.y) / (t.z * t.z), t.x / l, t.y / l, t.z / l);
      M3<R> nJ_inv(v2 + 1, -uv, R(0), -uv, u2 + 1, R(0), -txtz, -tytz, R(0));
      R vbn = dot(g.uvh_mn, g.uvh);
      R factor_normal = l / (u2 + v2 + 1);
      V3<R> plane = nJ_inv * (g.uvh_mn / std::fmax(vbn, R(0.0000001f)));
      R nl = u2 + v2 + 1;
      cp[0] = (-(v2 + 1) * t.z + plane[0] * t.x) / nl / focal_x;
      cp[1] = (uv * t.z + plane[1] * t.x) / nl / focal_y;
      cp[2] = (uv * t.z + plane[0] * t.y) / nl / focal_x;
      cp[3] = (-(u2 + 1) * t.z + plane[1] * t.y) / nl / focal_y;
      cp[4] = (t.x + plane[0] * t.z) / nl / focal_x;
      cp[5] = (t.y + plane[1] * t.z) / nl / focal_y;
      ray_planes[2 * idx] = plane[0] * l / nl / focal_x;
      ray_planes[2 * idx + 1] = plane[1] * l / nl / focal_y;
      V3<R> ray_n{-plane[0] * factor_normal, -plane[1] * factor_normal, R(-1)};
      V3<R> cam_n = nJ * ray_n;
      V3<R> nrm = normalize(cam_n);
      normals[3 * idx] = nrm.x; normals[3 * idx + 1] = nrm.y; normals[3 * idx + 2] = nrm.z;
      if (inte) {  // computeCov2D<true>, forward.cu:187-235: inverse covariance in ray space (u/f, v/f, t)
        M3<R> icr;
        if (g.well_conditioned) {
          R ltz = u2 + v2 + 1;
          M3<R> full(v2 + 1, -uv, txtz / l * ltz, -uv, u2 + 1, tytz / l * ltz, -txtz, -tytz, 1 / l * ltz);
          M3<R> nJ_inv_full = (t.z / (u2 + v2 + 1)) * full;
          M3<R> T2 = g.W * transpose(nJ_inv_full);
          icr = transpose(T2) * g.Vrk_inv * T2;
        } else {
          // Upstream declares a second, shadowing `inv_cov_ray` in this branch (forward.cu:223): what it computes there is
          // discarded and the outer, UNINITIALISED matrix is scaled and stored (undefined behaviour).  The discarded value is
          // itself meaningless (eigenvalues of a rank-1 matrix: 1/rounding-noise), so the deterministic reading is taken:
          // the matrix is zero.  In integrate_pixel such a Gaussian then acts as an opaque step at its depth plane.
          icr = M3<R>(R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0));
        }
        M3<R> sc(1 / focal_x, R(0), R(0), R(0), 1 / focal_y, R(0), R(0), R(0), R(1));
        icr = sc * icr * sc;
        R* o6 = &invraycov[6 * idx];
        o6[0] = icr[0][0]; o6[1] = icr[0][1]; o6[2] = icr[0][2]; o6[3] = icr[1][1]; o6[4] = icr[1][2]; o6[5] = icr[2][2];
      }
    }
    if (inte) condition[idx] = g.well_conditioned ? 1 : 0;

    // ---- back in preprocessCUDA, forward.cu:381-422 ----
    ts[idx] = std::sqrt(p_view.x * p_view.x + p_view.y * p_view.y + p_view.z * p_view.z);
    R det = (cov2D[0] * cov2D[2] - cov2D[1] * cov2D[1]);
    if (det == R(0.0f)) return;
    R det_inv = R(1.f) / det;
    R conic[3] = {cov2D[2] * det_inv, -cov2D[1] * det_inv, cov2D[0] * det_inv};
    R mid = R(0.5f) * (cov2D[0] + cov2D[2]);
    R lambda1 = mid + std::sqrt(std::fmax(R(0.1f), mid * mid - det));
    R lambda2 = mid - std::sqrt(std::fmax(R(0.1f), mid * mid - det));
    R my_radius = std::ceil(R(3.f) * std::sqrt(std::fmax(lambda1, lambda2)));
    R pix_x = ndc_to_pix(p_proj[0], W), pix_y = ndc_to_pix(p_proj[1], H);
    uint32_t rmin[2], rmax[2];
    tile_rect(pix_x, pix_y, to_int_sat(my_radius), gx, gy, rmin, rmax);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) return;

    if (!has_colors) {
      R c[3];
      color_from_sh(idx, c);
      rgb[3 * idx] = c[0]; rgb[3 * idx + 1] = c[1]; rgb[3 * idx + 2] = c[2];
    }
    depths[idx] = p_view.z;
    view_points[3 * idx] = p_view.x; view_points[3 * idx + 1] = p_view.y; view_points[3 * idx + 2] = p_view.z;
    radii[idx] = to_int_sat(my_radius);
    means2D[2 * idx] = pix_x; means2D[2 * idx + 1] = pix_y;
    conic_opacity[4 * idx] = conic[0]; conic_opacity[4 * idx + 1] = conic[1]; conic_opacity[4 * idx + 2] = conic[2];
    conic_opacity[4 * idx + 3] = opacities[idx] * coef;
    tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
  }

  // scan + duplicateWithKeys + SortPairs + identifyTileRanges,
  // rasterizer_impl.cu:350-390, 70-111, 151-173
  void bin_and_sort() {
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += tiles_touched[i]; point_offsets[i] = run; }
    num_rendered = P > 0 ? static_cast<int>(point_offsets[P - 1]) : 0;
    std::vector<std::pair<uint64_t, uint32_t>> kv(num_rendered);
    for (int idx = 0; idx < P; idx++) {
      if (radii[idx] > 0) {
        uint32_t off = idx == 0 ? 0 : point_offsets[idx - 1];
        uint32_t rmin[2], rmax[2];
        tile_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
        float df = static_cast<float>(depths[idx]);
        uint32_t dbits;
        std::memcpy(&dbits, &df, 4);
        for (uint32_t y = rmin[1]; y < rmax[1]; y++)
          for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
            uint64_t key = static_cast<uint64_t>(y) * gx + x;
            key <<= 32;
            key |= dbits;
            kv[off++] = {key, static_cast<uint32_t>(idx)};
          }
      }
    }
    const int bit = higher_msb(static_cast<uint32_t>(gx * gy));
    const uint64_t mask = (32 + bit) >= 64 ? ~0ull : ((1ull << (32 + bit)) - 1);
    std::stable_sort(kv.begin(), kv.end(), [mask](const auto& a, const auto& b) { return (a.first & mask) < (b.first & mask); });
    keys_sorted.resize(num_rendered);
    point_list.resize(num_rendered);
    for (int i = 0; i < num_rendered; i++) { keys_sorted[i] = kv[i].first; point_list[i] = kv[i].second; }
    std::fill(ranges.begin(), ranges.end(), 0u);
    for (int i = 0; i < num_rendered; i++) {
      uint32_t cur = keys_sorted[i] >> 32;
      if (i == 0) ranges[2 * cur] = 0;
      else {
        uint32_t prev = keys_sorted[i - 1] >> 32;
        if (cur != prev) { ranges[2 * prev + 1] = i; ranges[2 * cur] = i; }
      }
      if (i == num_rendered - 1) ranges[2 * cur + 1] = num_rendered;
    }
  }

  // renderCUDA<3,COORD,DEPTH,NORMAL> fwd for one pixel, forward.cu:428-693
  void render_pixel(uint32_t px, uint32_t py, bool COORD, bool DEPTH, bool NORMAL, int64_t& pairs, int64_t& blended) {
    const bool GEO = DEPTH || COORD || NORMAL;
    const size_t HW = size_t(H) * W;
    const uint32_t pix_id = W * py + px;
    const R pixfx = R(px), pixfy = R(py);
    const R pnx = (pixfx - W / R(2.f)) / focal_x, pny = (pixfy - H / R(2.f)) / focal_y;
    const R ln = std::sqrt(pnx * pnx + pny * pny + 1);
    const uint32_t tile = (py / TILE) * gx + (px / TILE);
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const R* feat = feature_ptr();

    R T = R(1.0f);
    uint32_t contributor = 0, last_contributor = 0, max_contributor = uint32_t(-1);
    R C[3] = {0, 0, 0}, weight = 0, Coord[3] = {0, 0, 0}, mCoord[3] = {0, 0, 0}, Depth = 0, mDepth = 0, Normal[3] = {0, 0, 0};
    for (uint32_t k = r0; k < r1; k++) {
      pairs++;
      contributor++;
      const uint32_t g = point_list[k];
      const R dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
      const R cx = conic_opacity[4 * g], cy = conic_opacity[4 * g + 1], cz = conic_opacity[4 * g + 2], co = conic_opacity[4 * g + 3];
      const R power = R(-0.5f) * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
      if (power > R(0.0f)) continue;
      const R alpha = std::fmin(R(0.99f), co * exp_spec(power));
      if (alpha < R(1.0f) / R(255.0f)) continue;
      const R test_T = T * (1 - alpha);
      if (test_T < R(0.0001f)) break;  // done = true
      const R aT = alpha * T;
      blended++;
      for (int ch = 0; ch < 3; ch++) C[ch] += feat[3 * g + ch] * aT;
      const bool before_median = T > R(0.5);
      if (COORD) {
        const R* cp = &camera_planes[6 * g];
        R coord[3] = {view_points[3 * g] + cp[0] * dx + cp[1] * dy, view_points[3 * g + 1] + cp[2] * dx + cp[3] * dy,
                      view_points[3 * g + 2] + cp[4] * dx + cp[5] * dy};
        for (int ch = 0; ch < 3; ch++) Coord[ch] += coord[ch] * aT;
        if (before_median) for (int ch = 0; ch < 3; ch++) mCoord[ch] = coord[ch];
      }
      if (DEPTH) {
        R t = ts[g] + (ray_planes[2 * g] * dx + ray_planes[2 * g + 1] * dy);
        Depth += t * aT;
        if (before_median) mDepth = t;
      }
      if (NORMAL) for (int ch = 0; ch < 3; ch++) Normal[ch] += normals[3 * g + ch] * aT;
      if (GEO && before_median) max_contributor = contributor;
      weight += aT;
      T = test_T;
      last_contributor = contributor;
    }
    n_contrib[pix_id] = last_contributor;
    n_contrib[pix_id + HW] = max_contributor;
    for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * bg[ch];
    out_alpha[pix_id] = weight;
    if (COORD) {
      for (int ch = 0; ch < 3; ch++) {
        out_coord[ch * HW + pix_id] = last_contributor ? Coord[ch] / weight : R(0);
        accum_coord[ch * HW + pix_id] = Coord[ch];
        out_mcoord[ch * HW + pix_id] = mCoord[ch];
      }
    }
    if (DEPTH) {
      R depth_ln = Depth / ln;
      accum_depth[pix_id] = depth_ln;
      out_depth[pix_id] = last_contributor ? depth_ln / weight : R(0);
      out_mdepth[pix_id] = mDepth / ln;
    }
    if (NORMAL) {
      if (last_contributor) {
        R len = std::sqrt(Normal[0] * Normal[0] + Normal[1] * Normal[1] + Normal[2] * Normal[2]);
        normal_length[pix_id] = len;
        len = std::fmax(len, R(1.0E-12F));
        for (int ch = 0; ch < 3; ch++) out_normal[ch * HW + pix_id] = Normal[ch] / len;
      } else {
        normal_length[pix_id] = 1;
        for (int ch = 0; ch < 3; ch++) out_normal[ch * HW + pix_id] = 0;
      }
    }
  }

  // Rasterizer::forward, rasterizer_impl.cu:254-425 (+ output zero-fill of rasterize_points.cu:71-78)
  int forward() {
    focal_y = H / (R(2.0f) * tan_fovy);
    focal_x = W / (R(2.0f) * tan_fovx);
    gx = (W + TILE - 1) / TILE;
    gy = (H + TILE - 1) / TILE;
    const size_t HW = size_t(H) * W;
    auto z = [&](std::vector<R>& v, size_t n) { v.assign(n, R(0)); };
    z(depths, P); z(camera_planes, 6 * size_t(P)); z(ray_planes, 2 * size_t(P)); z(ts, P); z(normals, 3 * size_t(P));
    z(means2D, 2 * size_t(P)); z(view_points, 3 * size_t(P)); z(cov3D, 6 * size_t(P)); z(conic_opacity, 4 * size_t(P)); z(rgb, 3 * size_t(P));
    clamped.assign(3 * size_t(P), 0); radii.assign(P, 0); tiles_touched.assign(P, 0); point_offsets.assign(P, 0);
    ranges.assign(2 * size_t(gx) * gy, 0);
    n_contrib.assign(2 * HW, 0);
    z(accum_coord, 3 * HW); z(accum_depth, HW); z(normal_length, HW);
    z(out_color, 3 * HW); z(out_coord, 3 * HW); z(out_mcoord, 3 * HW); z(out_depth, HW); z(out_mdepth, HW); z(out_alpha, HW); z(out_normal, 3 * HW);
    num_rendered = 0;
    if (P != 0) {  // rasterize_points.cu:90
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads)
      for (int i = 0; i < P; i++) preprocess_one(i);
      bin_and_sort();
      // template dispatch, forward.cu:732-739
      const bool COORD = req_coord, DEPTH = req_depth, NORMAL = req_coord || req_depth;
      int64_t pairs = 0, blended = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : pairs, blended) num_threads(nthreads)
      for (int tile = 0; tile < gx * gy; tile++) {
        const int ty = tile / gx, tx = tile % gx;
        for (int y = ty * TILE; y < std::min((ty + 1) * TILE, H); y++)
          for (int x = tx * TILE; x < std::min((tx + 1) * TILE, W); x++) render_pixel(x, y, COORD, DEPTH, NORMAL, pairs, blended);
      }
      stat_pairs_fwd = pairs;
      stat_blended_fwd = blended;
    } else {
      // P == 0: every output stays at its zero fill (the reference skips the whole call)
    }
    register_all();
    return num_rendered;
  }

  // ---------------------------------------------------------------- integrate ----
  // integrateCUDA for one pixel, forward.cu:938-1372.  MAXC = MAX_NUM_CONTRIBUTORS*4, MAXP = MAX_NUM_PROJECTED
  // (auxiliary.h:27-29).  Gaussian depth for the blend is ts (|p_view|, rasterizer_impl.cu:823), the sort key is z.
  void integrate_pixel(uint32_t px, uint32_t py) {
    constexpr int MAXC = 512 * 4, MAXP = 256;
    const size_t HW = size_t(H) * W;
    const uint32_t pix_id = W * py + px;
    const R pixfx = R(px) + R(0.5f), pixfy = R(py) + R(0.5f);
    const uint32_t tile = (py / TILE) * gx + (px / TILE);
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const uint32_t p0 = point_ranges[2 * tile], p1 = point_ranges[2 * tile + 1];
    const R* feat = feature_ptr();
    R T = R(1.0f);
    uint32_t contributor = 0, last_contributor = 0;
    R C[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    R mid_depth_center = 0, mid_plane[2] = {0, 0}, mid_mean2d[2] = {0, 0};
    uint32_t n_local = 0;
    std::vector<uint16_t> contributed(MAXC, 0);
    R corner_T[5] = {1, 1, 1, 1, 1};
    const R offx[5] = {R(0.0f), R(-0.5f), R(0.5f), R(-0.5f), R(0.5f)}, offy[5] = {R(0.0f), R(-0.5f), R(-0.5f), R(0.5f), R(0.5f)};
    for (uint32_t k = r0; k < r1; k++) {
      contributor++;
      const uint32_t g = point_list[k];
      const R cx = conic_opacity[4 * g], cy = conic_opacity[4 * g + 1], cz = conic_opacity[4 * g + 2], co = conic_opacity[4 * g + 3];
      const R depth_center = ts[g];
      const R dpx = ray_planes[2 * g], dpy = ray_planes[2 * g + 1];
      const R mx = means2D[2 * g], my = means2D[2 * g + 1];
      bool used = false;
      for (int c = 0; c < 5; ++c) {
        const R dx = mx - pixfx - offx[c], dy = my - pixfy - offy[c];
        const R depth = depth_center + (dpx * dx + dpy * dy);
        const R power = R(-0.5f) * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
        if (power > R(0.0f)) continue;
        const R alpha = std::fmin(R(0.99f), co * exp_spec(power));
        if (alpha < R(1.0f) / R(255.0f)) continue;
        const R test_T = corner_T[c] * (1 - alpha);
        if (test_T < R(0.0001f)) continue;
        if (c == 0) for (int ch = 0; ch < 3; ch++) C[ch] += feat[3 * g + ch] * alpha * T;
        if (depth > C[6]) C[6] = depth;
        if (c == 0) {
          C[7] += alpha * T;
          C[3] += depth * alpha * T;
          if (T > R(0.5)) { C[4] = depth; mid_depth_center = depth_center; mid_plane[0] = dpx; mid_plane[1] = dpy; mid_mean2d[0] = mx; mid_mean2d[1] = my; }
          T = test_T;
        }
        corner_T[c] = test_T;
        used = true;
      }
      if (used) {
        last_contributor = contributor;
        contributed[n_local] = static_cast<uint16_t>(contributor);
        n_local += 1;
        if (n_local >= MAXC) break;  // upstream prints an error and stops this pixel (forward.cu:1121-1125)
      }
    }
    final_T[pix_id] = T;
    n_contrib[pix_id] = last_contributor;
    for (int ch = 0; ch < 3; ch++) out9[ch * HW + pix_id] = C[ch] + T * bg[ch];
    out9[3 * HW + pix_id] = C[3];
    out9[4 * HW + pix_id] = C[4];
    out9[6 * HW + pix_id] = C[6];
    out9[7 * HW + pix_id] = C[7];

    // ---- points that project into this pixel, MAXP at a time ----
    uint32_t counter_last = 0;
    int total_projected = 0;
    while (true) {
      int num_projected = 0;
      bool exceed = false;
      uint32_t counter = 0;
      int pid[MAXP]; R pxy[MAXP][2], pdepth[MAXP];
      for (uint32_t k = p0; k < p1; k++) {
        counter++;
        if (counter <= counter_last) continue;
        const uint32_t q = pt_list[k];
        const R qx = points2D[2 * q], qy = points2D[2 * q + 1];
        if ((static_cast<double>(qx) >= (static_cast<double>(pixfx) - 0.5)) && (static_cast<double>(qx) < (static_cast<double>(pixfx) + 0.5)) &&
            (static_cast<double>(qy) >= (static_cast<double>(pixfy) - 0.5)) && (static_cast<double>(qy) < (static_cast<double>(pixfy) + 0.5))) {
          if (num_projected >= MAXP) { exceed = true; break; }
          pid[num_projected] = static_cast<int>(q);
          pxy[num_projected][0] = qx; pxy[num_projected][1] = qy;
          pdepth[num_projected] = point_depths[q];
          num_projected += 1;
        }
      }
      counter_last = counter - 1;
      total_projected += num_projected;
      R palpha[MAXP], pT[MAXP];
      for (int i = 0; i < num_projected; i++) { palpha[i] = R(0.f); pT[i] = R(1.f); }
      uint32_t num_iterated = 0;
      uint16_t nsecond = 0;
      for (uint32_t k = r0; k < r1; k++) {
        num_iterated++;
        if (num_iterated > last_contributor) break;
        if (num_iterated != static_cast<uint32_t>(contributed[nsecond])) continue;
        nsecond += 1;
        const uint32_t g = point_list[k];
        const R co = conic_opacity[4 * g + 3];
        const R depth_center = ts[g];
        const R dpx = ray_planes[2 * g], dpy = ray_planes[2 * g + 1];
        const R mx = means2D[2 * g], my = means2D[2 * g + 1];
        const R* ic = &invraycov[6 * g];
        M3<R> inv(ic[0], ic[1], ic[2], ic[1], ic[3], ic[4], ic[2], ic[4], ic[5]);
        for (int i = 0; i < num_projected; i++) {
          const R dx = mx - pxy[i][0], dy = my - pxy[i][1];
          const R depth = depth_center + (dpx * dx + dpy * dy);
          R alpha;
          if (condition[g]) {
            V3<R> du{dx, dy, depth_center - std::fmin(pdepth[i], depth)};
            R power = R(-0.5f) * (dot(du, inv * du));
            alpha = std::fmin(R(0.99f), co * exp_spec(std::fmin(power, R(80.0f))));
          } else {
            if (pdepth[i] < depth) alpha = 0;
            else {
              V3<R> du{dx, dy, depth_center};
              R power = R(-0.5f) * (dot(du, inv * du));
              alpha = std::fmin(R(0.99f), co * exp_spec(std::fmin(power, R(80.0f))));
            }
          }
          if (alpha < R(1.0f) / R(255.0f)) continue;
          const R test_T = pT[i] * (1 - alpha);
          palpha[i] += alpha * pT[i];
          pT[i] = test_T;
        }
      }
      for (int i = 0; i < num_projected; i++) {
        out_alpha_integrated[pid[i]] = palpha[i];
        for (int ch = 0; ch < 3; ch++) out_color_integrated[3 * pid[i] + ch] = C[ch] + T * bg[ch];
        out_coordinate2d[2 * pid[i]] = pxy[i][0];
        out_coordinate2d[2 * pid[i] + 1] = pxy[i][1];
        if (pdepth[i] > 0) {
          const R dx = mid_mean2d[0] - pxy[i][0], dy = mid_mean2d[1] - pxy[i][1];
          const R depth = mid_depth_center + (mid_plane[0] * dx + mid_plane[1] * dy);
          out_sdf[pid[i]] = depth - pdepth[i];
        }
      }
      if (!exceed) break;
    }
    out9[8 * HW + pix_id] = static_cast<R>(total_projected);
  }

  // Rasterizer::integrate, rasterizer_impl.cu:573-843 (+ the output fills of rasterize_points.cu:312-320)
  int integrate(const R* pts, int npts) {
    inte = true;
    PN = npts;
    points3D.assign(pts, pts + 3 * size_t(npts));
    focal_y = H / (R(2.0f) * tan_fovy);
    focal_x = W / (R(2.0f) * tan_fovx);
    gx = (W + TILE - 1) / TILE;
    gy = (H + TILE - 1) / TILE;
    const size_t HW = size_t(H) * W;
    auto z = [&](std::vector<R>& v, size_t n) { v.assign(n, R(0)); };
    z(depths, P); z(camera_planes, 6 * size_t(P)); z(ray_planes, 2 * size_t(P)); z(ts, P); z(normals, 3 * size_t(P));
    z(means2D, 2 * size_t(P)); z(view_points, 3 * size_t(P)); z(cov3D, 6 * size_t(P)); z(conic_opacity, 4 * size_t(P)); z(rgb, 3 * size_t(P));
    z(invraycov, 6 * size_t(P)); condition.assign(P, 0);
    clamped.assign(3 * size_t(P), 0); radii.assign(P, 0); tiles_touched.assign(P, 0); point_offsets.assign(P, 0);
    ranges.assign(2 * size_t(gx) * gy, 0); point_ranges.assign(2 * size_t(gx) * gy, 0);
    n_contrib.assign(2 * HW, 0);
    z(out9, 9 * HW); z(final_T, HW);
    out_alpha_integrated.assign(PN, R(1.0)); z(out_color_integrated, 3 * size_t(PN)); z(out_coordinate2d, 2 * size_t(PN));
    out_sdf.assign(PN, R(-1000.0));
    z(points2D, 2 * size_t(PN)); z(point_depths, PN); point_tiles.assign(PN, 0);
    num_rendered = 0;
    if (P != 0 && PN != 0) {
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads)
      for (int i = 0; i < P; i++) preprocess_one(i);
      bin_and_sort();
      // preprocessPointsCUDA, forward.cu:855-900
      for (int i = 0; i < PN; i++) {
        V3<R> p{points3D[3 * i], points3D[3 * i + 1], points3D[3 * i + 2]};
        V3<R> pv = xform_point43(p, view);
        if (pv.z <= R(0.2f)) continue;
        const R ix = static_cast<R>(static_cast<double>(focal_x * pv.x / (pv.z + R(0.0000001f))) + W / 2.);
        const R iy = static_cast<R>(static_cast<double>(focal_y * pv.y / (pv.z + R(0.0000001f))) + H / 2.);
        if (ix < 0 || ix >= W || iy < 0 || iy >= H) continue;
        point_depths[i] = std::sqrt(pv.x * pv.x + pv.y * pv.y + pv.z * pv.z);
        points2D[2 * i] = ix; points2D[2 * i + 1] = iy;
        point_tiles[i] = 1;
      }
      // createWithKeys + SortPairs + identifyTileRanges, rasterizer_impl.cu:114-145,784-806
      std::vector<std::pair<uint64_t, uint32_t>> kv;
      for (int i = 0; i < PN; i++) {
        if (!point_tiles[i]) continue;
        int tx = std::min(gx - 1, std::max(0, to_int_sat(points2D[2 * i] / R(TILE))));
        int ty = std::min(gy - 1, std::max(0, to_int_sat(points2D[2 * i + 1] / R(TILE))));
        float df = static_cast<float>(point_depths[i]);
        uint32_t dbits;
        std::memcpy(&dbits, &df, 4);
        kv.push_back({(static_cast<uint64_t>(ty) * gx + tx) << 32 | dbits, static_cast<uint32_t>(i)});
      }
      std::stable_sort(kv.begin(), kv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      pt_list.resize(kv.size());
      for (size_t i = 0; i < kv.size(); i++) {
        pt_list[i] = kv[i].second;
        uint32_t cur = kv[i].first >> 32;
        if (i == 0) point_ranges[2 * cur] = 0;
        else {
          uint32_t prev = kv[i - 1].first >> 32;
          if (cur != prev) { point_ranges[2 * prev + 1] = i; point_ranges[2 * cur] = i; }
        }
        if (i + 1 == kv.size()) point_ranges[2 * cur + 1] = kv.size();
      }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
      for (int tile = 0; tile < gx * gy; tile++) {
        const int ty = tile / gx, tx = tile % gx;
        for (int y = ty * TILE; y < std::min((ty + 1) * TILE, H); y++)
          for (int x = tx * TILE; x < std::min((tx + 1) * TILE, W); x++) integrate_pixel(x, y);
      }
    }
    inte = false;
    register_all();
    return num_rendered;
  }

  // ----------------------------------------------------------------- backward ----
  struct PixGrads { const R *color, *coord, *mcoord, *depth, *mdepth, *alpha, *normal; };

  static inline void atomic_add(R& dst, R v) {
#pragma omp atomic
    dst += v;
  }

  // renderCUDA bwd for one pixel, backward.cu:631-1016.  Sums into double accumulators.
  void render_pixel_bwd(uint32_t px, uint32_t py, bool COORD, bool DEPTH, bool NORMAL, const PixGrads& gin,
                        std::vector<double>& A) {
    const bool GEO = COORD || DEPTH || NORMAL;
    const size_t HW = size_t(H) * W;
    const uint32_t pix_id = W * py + px;
    const R pixfx = R(px), pixfy = R(py);
    const R pnx = (pixfx - W / R(2.f)) / focal_x, pny = (pixfy - H / R(2.f)) / focal_y;
    const R ln = std::sqrt(pnx * pnx + pny * pny + 1);
    const uint32_t tile = (py / TILE) * gx + (px / TILE);
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const R* feat = feature_ptr();

    const R T_final = 1 - out_alpha[pix_id];
    const R w_final = out_alpha[pix_id];
    R T = T_final;
    uint32_t contributor = r1 - r0;
    const int last_contributor = static_cast<int>(n_contrib[pix_id]);
    const int max_contributor = static_cast<int>(n_contrib[pix_id + HW]);

    R accum_rec[3] = {0, 0, 0}, dL_dpixel[3], accum_coord_rec[3] = {0, 0, 0}, dL_dpixel_coord[3] = {0, 0, 0};
    R accum_t_rec = 0, dL_dpixel_t = 0, dL_dpixel_mt = 0, accum_alpha_rec = 0, dL_dalpha;
    R accum_normal_rec[3] = {0, 0, 0}, dL_dpixel_normal[3] = {0, 0, 0}, dL_dpixel_mcoord[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) dL_dpixel[i] = gin.color[i * HW + pix_id];
    dL_dalpha = gin.alpha[pix_id];
    if (GEO) {
      R ww = w_final * w_final;
      if (COORD) {
        for (int i = 0; i < 3; i++) {
          R gw = gin.coord[i * HW + pix_id];
          dL_dalpha -= gw * accum_coord[i * HW + pix_id] / ww;
          dL_dpixel_coord[i] = gw / w_final;
          dL_dpixel_mcoord[i] = gin.mcoord[i * HW + pix_id];
        }
      }
      if (DEPTH) {
        R gw = gin.depth[pix_id];
        dL_dalpha -= gw * accum_depth[pix_id] / ww;
        dL_dpixel_t = gw / w_final / ln;
        dL_dpixel_mt = gin.mdepth[pix_id] / ln;
      }
      if (NORMAL) {
        V3<R> gn{gin.normal[pix_id], gin.normal[HW + pix_id], gin.normal[2 * HW + pix_id]};
        V3<R> nn{out_normal[pix_id], out_normal[HW + pix_id], out_normal[2 * HW + pix_id]};
        R nlen = normal_length[pix_id];
        V3<R> dL;
        if (nlen < R(1.0E-12F)) dL = gn / R(1.0E-12F);
        else dL = (gn - dot(gn, nn) * nn) / nlen;
        for (int i = 0; i < 3; i++) dL_dpixel_normal[i] = dL[i];
      }
    }
    R last_alpha = 0, last_color[3] = {0, 0, 0}, last_coord[3] = {0, 0, 0}, last_t = 0, last_normal[3] = {0, 0, 0};
    const R ddelx_dx = static_cast<R>(0.5 * W), ddely_dy = static_cast<R>(0.5 * H);

    for (uint32_t k = r1; k-- > r0;) {
      contributor--;
      if (contributor >= static_cast<uint32_t>(last_contributor)) continue;
      const uint32_t g = point_list[k];
      if (tl_addlog.log) {   // (batch from the back, thread rank in the 16x16 block, entry in the batch): backward.cu:796-838
        const uint32_t m = r1 - 1 - k;
        tl_addlog.key = ((m >> 8) << 16) | (((py % TILE) * TILE + (px % TILE)) << 8) | (m & 255u);
      }
      const R dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
      const R cx = conic_opacity[4 * g], cy = conic_opacity[4 * g + 1], cz = conic_opacity[4 * g + 2], co = conic_opacity[4 * g + 3];
      const R power = R(-0.5f) * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
      if (power > R(0.0f)) continue;
      const R G = exp_spec(power);
      const R alpha = std::fmin(R(0.99f), co * G);
      if (alpha < R(1.0f) / R(255.0f)) continue;
      T = T / (R(1.f) - alpha);
      const R dchannel_dcolor = alpha * T;
      R dL_dopa = R(0.0f);
      double* a = &A[size_t(g) * NACC];
      for (int ch = 0; ch < 3; ch++) {
        const R c = feat[3 * g + ch];
        accum_rec[ch] = last_alpha * last_color[ch] + (R(1.f) - last_alpha) * accum_rec[ch];
        last_color[ch] = c;
        dL_dopa += (c - accum_rec[ch]) * dL_dpixel[ch];
        add(a[A_COLOR + ch], dchannel_dcolor * dL_dpixel[ch]);
      }
      R dL_dcoords[3] = {0, 0, 0}, dL_dt = 0;
      const R* cp = &camera_planes[6 * g];
      if (COORD) {
        R coord[3] = {view_points[3 * g] + cp[0] * dx + cp[1] * dy, view_points[3 * g + 1] + cp[2] * dx + cp[3] * dy,
                      view_points[3 * g + 2] + cp[4] * dx + cp[5] * dy};
        for (int ch = 0; ch < 3; ch++) {
          const R c = coord[ch];
          accum_coord_rec[ch] = last_alpha * last_coord[ch] + (R(1.f) - last_alpha) * accum_coord_rec[ch];
          last_coord[ch] = c;
          dL_dopa += (c - accum_coord_rec[ch]) * dL_dpixel_coord[ch];
          dL_dcoords[ch] = dchannel_dcolor * dL_dpixel_coord[ch];
          if (contributor == static_cast<uint32_t>(max_contributor - 1)) dL_dcoords[ch] += dL_dpixel_mcoord[ch];
        }
        for (int ch = 0; ch < 3; ch++) {
          add(a[A_VIEWPT + ch], dL_dcoords[ch]);
          add(a[A_CAMPLANE + 2 * ch], dL_dcoords[ch] * dx / focal_x);
          add(a[A_CAMPLANE + 2 * ch + 1], dL_dcoords[ch] * dy / focal_y);
        }
      }
      if (DEPTH) {
        R t = ts[g] + (ray_planes[2 * g] * dx + ray_planes[2 * g + 1] * dy);
        accum_t_rec = last_alpha * last_t + (R(1.f) - last_alpha) * accum_t_rec;
        last_t = t;
        dL_dopa += (t - accum_t_rec) * dL_dpixel_t;
        dL_dt = dchannel_dcolor * dL_dpixel_t;
        if (contributor == static_cast<uint32_t>(max_contributor - 1)) dL_dt += dL_dpixel_mt;
        add(a[A_TS], dL_dt);
        add(a[A_RAYPLANE], dL_dt * dx / focal_x);
        add(a[A_RAYPLANE + 1], dL_dt * dy / focal_y);
      }
      if (NORMAL) {
        for (int ch = 0; ch < 3; ch++) {
          const R c = normals[3 * g + ch];
          accum_normal_rec[ch] = last_alpha * last_normal[ch] + (R(1.f) - last_alpha) * accum_normal_rec[ch];
          last_normal[ch] = c;
          dL_dopa += (c - accum_normal_rec[ch]) * dL_dpixel_normal[ch];
          add(a[A_NORMAL + ch], dchannel_dcolor * dL_dpixel_normal[ch]);
        }
      }
      accum_alpha_rec = last_alpha + (R(1.f) - last_alpha) * accum_alpha_rec;
      dL_dopa += (1 - accum_alpha_rec) * dL_dalpha;
      dL_dopa *= T;
      last_alpha = alpha;
      R bg_dot = 0;
      for (int i = 0; i < 3; i++) bg_dot += bg[i] * dL_dpixel[i];
      dL_dopa += (-T_final / (R(1.f) - alpha)) * bg_dot;

      const R dL_dG = co * dL_dopa;
      const R gdx = G * dx, gdy = G * dy;
      const R dG_ddelx = -gdx * cx - gdy * cy;
      const R dG_ddely = -gdy * cz - gdx * cy;
      R dL_ddelx = dL_dG * dG_ddelx, dL_ddely = dL_dG * dG_ddely;
      if (COORD) {
        dL_ddelx += dL_dcoords[0] * cp[0] + dL_dcoords[1] * cp[2] + dL_dcoords[2] * cp[4];
        dL_ddely += dL_dcoords[0] * cp[1] + dL_dcoords[1] * cp[3] + dL_dcoords[2] * cp[5];
      }
      if (DEPTH) {
        dL_ddelx += dL_dt * ray_planes[2 * g];
        dL_ddely += dL_dt * ray_planes[2 * g + 1];
      }
      add(a[A_MEAN2D], dL_ddelx * ddelx_dx);
      add(a[A_MEAN2D + 1], dL_ddely * ddely_dy);
      add(a[A_MEAN2D + 2], std::fabs(dL_dG * dG_ddelx * ddelx_dx) + std::fabs(dL_dG * dG_ddely * ddely_dy));
      add(a[A_CONIC], R(-0.5f) * gdx * dx * dL_dG);
      add(a[A_CONIC + 1], R(-0.5f) * gdx * dy * dL_dG);
      add(a[A_CONIC + 2], R(-0.5f) * gdy * dy * dL_dG);
      add(a[A_OPACITY], G * dL_dopa);
    }
  }
  enum { A_COLOR = 0, A_VIEWPT = 3, A_CAMPLANE = 6, A_TS = 12, A_RAYPLANE = 13, A_NORMAL = 15, A_MEAN2D = 18, A_CONIC = 21, A_OPACITY = 24, NACC = 25 };
  static inline void add(double& dst, R v) {
    if (tl_addlog.log) { tl_addlog.log->push_back({tl_addlog.key, uint32_t(&dst - tl_addlog.base), float(v)}); return; }
    const double dv = static_cast<double>(v);
#pragma omp atomic
    dst += dv;
  }

  // computeCov2DCUDA, backward.cu:145-488
  void cov2d_bwd_one(int idx) {
    if (!(radii[idx] > 0)) return;
    const R* c3 = cov3D_ptr(idx);
    V3<R> mean{means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    R dL_dconic_x = dL_dconic[4 * idx], dL_dconic_y = dL_dconic[4 * idx + 1], dL_dconic_z = dL_dconic[4 * idx + 3];
    const V3<R> dL_dnormal{dL_dnormals[3 * idx], dL_dnormals[3 * idx + 1], dL_dnormals[3 * idx + 2]};
    const R combined_opacity = g_opacity_slip ? dL_dconic[4 * idx + 3] : conic_opacity[4 * idx + 3];
    const V2<R> dcp0{dL_dcamera_planes[6 * idx], dL_dcamera_planes[6 * idx + 1]};
    const V2<R> dcp1{dL_dcamera_planes[6 * idx + 2], dL_dcamera_planes[6 * idx + 3]};
    const V2<R> dcp2{dL_dcamera_planes[6 * idx + 4], dL_dcamera_planes[6 * idx + 5]};
    const V2<R> drp{dL_dray_planes[2 * idx], dL_dray_planes[2 * idx + 1]};
    const R h_x = focal_x, h_y = focal_y;

    Cov2DCommon<R> g;
    cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, kernel_size, c3, view, g);
    const V3<R>& t = g.t;
    const R txtz = g.txtz, tytz = g.tytz;
    const M3<R>& Tm = g.T; const M3<R>& Vrk = g.Vrk; const M3<R>& W_ = g.W; const M3<R>& cov2D = g.cov;
    const R det_0 = g.det_0, det_1 = g.det_1, coef = g.coef_raw;
    const R u2 = txtz * txtz, v2 = tytz * tytz, uv = txtz * tytz;

    M3<R> dL_dVrk, dL_dnJ;
    V3<R> plane{0, 0, 0};
    R dL_du, dL_dv, dL_dl, l, nl;
    if (std::isnan(g.uvh_mn.x) || g.D == 0) {
      nl = 1; l = 1; dL_du = 0; dL_dv = 0; dL_dl = 0;
    } else {
      const V3<R>& uvh = g.uvh; const V3<R>& uvh_m = g.uvh_m; const V3<R>& uvh_mn = g.uvh_mn;
      R vb = dot(uvh_m, uvh), vbn = dot(uvh_mn, uvh);
      l = std::sqrt(t.x * t.x + t.y * t.y + t.z * t.z);
      M3<R> nJ(1 / t.z, R(0), -(t.x) / (t.z * t.z), R(0), 1 / t.z, -(t.y) / (t.z * t.z), t.x / l, t.y / l, t.z / l);
      M3<R> nJ_inv(v2 + 1, -uv, R(0), -uv, u2 + 1, R(0), -txtz, -tytz, R(0));
      R clamp_vb = std::fmax(vb, R(0.0000001f)), clamp_vbn = std::fmax(vbn, R(0.0000001f));
      nl = u2 + v2 + 1;
      R factor_normal = l / nl;
      V3<R> uvh_m_vb = uvh_mn / clamp_vbn;
      plane = nJ_inv * uvh_m_vb;
      V2<R> cpl0{(-(v2 + 1) * t.z + plane[0] * t.x) / nl, (uv * t.z + plane[1] * t.x) / nl};
      V2<R> cpl1{(uv * t.z + plane[0] * t.y) / nl, (-(u2 + 1) * t.z + plane[1] * t.y) / nl};
      V2<R> cpl2{(t.x + plane[0] * t.z) / nl, (t.y + plane[1] * t.z) / nl};
      V2<R> ray_plane{plane[0] * factor_normal, plane[1] * factor_normal};
      V3<R> ray_n{-plane[0] * factor_normal, -plane[1] * factor_normal, R(-1)};
      V3<R> cam_n = nJ * ray_n;
      V3<R> nrm = normalize(cam_n);
      R lv = length(cam_n);
      const V3<R> dn_lv = dL_dnormal / lv;
      V3<R> dL_dcam_n = dn_lv - nrm * dot(nrm, dn_lv);
      V3<R> dL_dray_n = transpose(nJ) * dL_dcam_n;
      dL_dnJ = outer(dL_dcam_n, ray_n);
      dL_dl = (-plane[0] * dL_dray_n.x - plane[1] * dL_dray_n.y + plane[0] * drp.x + plane[1] * drp.y) / nl;
      V2<R> dL_dplane{(t.x * dcp0.x + t.y * dcp1.x + t.z * dcp2.x - l * dL_dray_n[0] + drp.x * l) / nl,
                      (t.x * dcp0.y + t.y * dcp1.y + t.z * dcp2.y - l * dL_dray_n[1] + drp.y * l) / nl};
      V3<R> dL_dplane_append{dL_dplane.x, dL_dplane.y, R(0)};
      R dL_dnl = (-dcp0.x * cpl0.x - dcp0.y * cpl0.y - dcp1.x * cpl1.x - dcp1.y * cpl1.y - dcp2.x * cpl2.x - dcp2.y * cpl2.y -
                  dL_dray_n[0] * ray_n.x - dL_dray_n[1] * ray_n.y - drp.x * ray_plane.x - drp.y * ray_plane.y) / nl;
      R tmp = dL_dplane.x * plane.x + dL_dplane.y * plane.y;
      V3<R> W_uvh = W_ * uvh;
      if (g.well_conditioned) {
        dL_dVrk = -outer(g.Vrk_inv * W_uvh, (g.Vrk_inv / clamp_vb) * (W_uvh * (-tmp) + W_ * transpose(nJ_inv) * dL_dplane_append));
      } else {
        R dL_dvb = -tmp / clamp_vb;
        V3<R> nJi_dp = transpose(nJ_inv) * V3<R>{dL_dplane.x / clamp_vb, dL_dplane.y / clamp_vb, R(0)};
        M3<R> dL_dVrk_inv = outer(W_uvh, W_uvh * dL_dvb + W_ * nJi_dp);
        V3<R> dL_dvv = (dL_dVrk_inv + transpose(dL_dVrk_inv)) * g.evec_min;
        for (int j = 0; j < 3; j++) {
          if (j != static_cast<int>(g.min_id)) {
            R sc = dot(g.evec[j], dL_dvv) / std::fmin(g.eval[g.min_id] - g.eval[j], R(-0.0000001f));
            dL_dVrk = dL_dVrk + outer(g.evec[j] * sc, g.evec_min);
          }
        }
      }
      V3<R> dL_duvh = (2 * (-tmp)) * uvh_m_vb + (g.cov_cam_inv / clamp_vb) * transpose(nJ_inv) * dL_dplane_append;
      M3<R> dL_dnJ_inv = outer(dL_dplane_append, uvh_m_vb);
      dL_du = dL_dnl * 2 * txtz + dL_duvh.x + (dL_dnJ_inv[0][1] + dL_dnJ_inv[1][0]) * (-tytz) + 2 * dL_dnJ_inv[1][1] * txtz -
              dL_dnJ_inv[2][0] + (dcp0.y * t.y + dcp1.x * t.y + dcp1.y * (-2 * t.x)) / nl;
      dL_dv = dL_dnl * 2 * tytz + dL_duvh.y + (dL_dnJ_inv[0][1] + dL_dnJ_inv[1][0]) * (-txtz) + 2 * dL_dnJ_inv[0][0] * tytz -
              dL_dnJ_inv[2][1] + (dcp0.x * (-2 * t.y) + dcp0.y * t.x + dcp1.x * t.x) / nl;
    }

    // backward.cu:367-375 (double-precision sub-expressions)
    const R opacity = static_cast<R>(static_cast<double>(combined_opacity) / (static_cast<double>(coef) + 1e-6));
    const R dL_dcoef = dL_dopacity[idx] * opacity;
    const R dL_dsqrtcoef = static_cast<R>(static_cast<double>(dL_dcoef) * 0.5 * 1. / (static_cast<double>(coef) + 1e-6));
    const R dL_ddet0 = static_cast<R>(static_cast<double>(dL_dsqrtcoef) / (static_cast<double>(det_1) + 1e-6));
    const R dL_ddet1 = static_cast<R>(static_cast<double>(dL_dsqrtcoef * det_0) *
                                      (static_cast<double>(R(-1.f)) / (static_cast<double>(det_1 * det_1) + 1e-6)));
    const R dcoef_da = dL_ddet0 * cov2D[1][1] + dL_ddet1 * (cov2D[1][1] + kernel_size);
    const R dcoef_db = static_cast<R>(static_cast<double>(dL_ddet0) * (-2. * static_cast<double>(cov2D[0][1])) +
                                      static_cast<double>(dL_ddet1) * (-2. * static_cast<double>(cov2D[0][1])));
    const R dcoef_dc = dL_ddet0 * cov2D[0][0] + dL_ddet1 * (cov2D[0][0] + kernel_size);
    R a = cov2D[0][0] + kernel_size, b = cov2D[0][1], c = cov2D[1][1] + kernel_size;
    R denom = a * c - b * b;
    R dL_da = 0, dL_db = 0, dL_dc = 0;
    R denom2inv = R(1.0f) / ((denom * denom) + R(0.0000001f));
    R* dcov = &dL_dcov3D[6 * idx];
    if (denom2inv != 0) {
      dL_da = denom2inv * (-c * c * dL_dconic_x + 2 * b * c * dL_dconic_y + (denom - a * c) * dL_dconic_z);
      dL_dc = denom2inv * (-a * a * dL_dconic_z + 2 * a * b * dL_dconic_y + (denom - a * c) * dL_dconic_x);
      dL_db = denom2inv * 2 * (b * c * dL_dconic_x - (denom + 2 * b * b) * dL_dconic_y + a * b * dL_dconic_z);
      if (static_cast<double>(det_0) <= 1e-6 || static_cast<double>(det_1) <= 1e-6) {
        dL_dopacity[idx] = 0;
      } else {
        dL_da += dcoef_da; dL_dc += dcoef_dc; dL_db += dcoef_db;
        dL_dopacity[idx] = dL_dopacity[idx] * coef;
      }
      dcov[0] = (Tm[0][0] * Tm[0][0] * dL_da + Tm[0][0] * Tm[1][0] * dL_db + Tm[1][0] * Tm[1][0] * dL_dc);
      dcov[3] = (Tm[0][1] * Tm[0][1] * dL_da + Tm[0][1] * Tm[1][1] * dL_db + Tm[1][1] * Tm[1][1] * dL_dc);
      dcov[5] = (Tm[0][2] * Tm[0][2] * dL_da + Tm[0][2] * Tm[1][2] * dL_db + Tm[1][2] * Tm[1][2] * dL_dc);
      dcov[1] = 2 * Tm[0][0] * Tm[0][1] * dL_da + (Tm[0][0] * Tm[1][1] + Tm[0][1] * Tm[1][0]) * dL_db + 2 * Tm[1][0] * Tm[1][1] * dL_dc;
      dcov[2] = 2 * Tm[0][0] * Tm[0][2] * dL_da + (Tm[0][0] * Tm[1][2] + Tm[0][2] * Tm[1][0]) * dL_db + 2 * Tm[1][0] * Tm[1][2] * dL_dc;
      dcov[4] = 2 * Tm[0][2] * Tm[0][1] * dL_da + (Tm[0][1] * Tm[1][2] + Tm[0][2] * Tm[1][1]) * dL_db + 2 * Tm[1][1] * Tm[1][2] * dL_dc;
    } else {
      for (int i = 0; i < 6; i++) dcov[i] = 0;
    }
    dcov[0] += dL_dVrk[0][0];
    dcov[3] += dL_dVrk[1][1];
    dcov[5] += dL_dVrk[2][2];
    dcov[1] += dL_dVrk[0][1] + dL_dVrk[1][0];
    dcov[2] += dL_dVrk[0][2] + dL_dVrk[2][0];
    dcov[4] += dL_dVrk[1][2] + dL_dVrk[2][1];

    R dL_dT00 = 2 * (Tm[0][0] * Vrk[0][0] + Tm[0][1] * Vrk[0][1] + Tm[0][2] * Vrk[0][2]) * dL_da + (Tm[1][0] * Vrk[0][0] + Tm[1][1] * Vrk[0][1] + Tm[1][2] * Vrk[0][2]) * dL_db;
    R dL_dT01 = 2 * (Tm[0][0] * Vrk[1][0] + Tm[0][1] * Vrk[1][1] + Tm[0][2] * Vrk[1][2]) * dL_da + (Tm[1][0] * Vrk[1][0] + Tm[1][1] * Vrk[1][1] + Tm[1][2] * Vrk[1][2]) * dL_db;
    R dL_dT02 = 2 * (Tm[0][0] * Vrk[2][0] + Tm[0][1] * Vrk[2][1] + Tm[0][2] * Vrk[2][2]) * dL_da + (Tm[1][0] * Vrk[2][0] + Tm[1][1] * Vrk[2][1] + Tm[1][2] * Vrk[2][2]) * dL_db;
    R dL_dT10 = 2 * (Tm[1][0] * Vrk[0][0] + Tm[1][1] * Vrk[0][1] + Tm[1][2] * Vrk[0][2]) * dL_dc + (Tm[0][0] * Vrk[0][0] + Tm[0][1] * Vrk[0][1] + Tm[0][2] * Vrk[0][2]) * dL_db;
    R dL_dT11 = 2 * (Tm[1][0] * Vrk[1][0] + Tm[1][1] * Vrk[1][1] + Tm[1][2] * Vrk[1][2]) * dL_dc + (Tm[0][0] * Vrk[1][0] + Tm[0][1] * Vrk[1][1] + Tm[0][2] * Vrk[1][2]) * dL_db;
    R dL_dT12 = 2 * (Tm[1][0] * Vrk[2][0] + Tm[1][1] * Vrk[2][1] + Tm[1][2] * Vrk[2][2]) * dL_dc + (Tm[0][0] * Vrk[2][0] + Tm[0][1] * Vrk[2][1] + Tm[0][2] * Vrk[2][2]) * dL_db;
    R dL_dJ00 = W_[0][0] * dL_dT00 + W_[0][1] * dL_dT01 + W_[0][2] * dL_dT02;
    R dL_dJ02 = W_[2][0] * dL_dT00 + W_[2][1] * dL_dT01 + W_[2][2] * dL_dT02;
    R dL_dJ11 = W_[1][0] * dL_dT10 + W_[1][1] * dL_dT11 + W_[1][2] * dL_dT12;
    R dL_dJ12 = W_[2][0] * dL_dT10 + W_[2][1] * dL_dT11 + W_[2][2] * dL_dT12;
    R tz = R(1.f) / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    R l3 = l * l * l;
    R dL_dtx = g.x_grad_mul * (-h_x * tz2 * dL_dJ02 + dL_du * tz - dL_dnJ[0][2] * tz2 + dL_dnJ[2][0] * (1 / l - t.x * t.x / l3) +
                               dL_dnJ[2][1] * (-t.x * t.y / l3) + dL_dnJ[2][2] * (-t.x * t.z / l3) +
                               (dcp0.x * plane[0] + dcp0.y * plane[1] + dcp2.x) / nl + dL_dl * t.x / l);
    R dL_dty = g.y_grad_mul * (-h_y * tz2 * dL_dJ12 + dL_dv * tz - dL_dnJ[1][2] * tz2 + dL_dnJ[2][0] * (-t.x * t.y / l3) +
                               dL_dnJ[2][1] * (1 / l - t.y * t.y / l3) + dL_dnJ[2][2] * (-t.y * t.z / l3) +
                               (dcp1.x * plane[0] + dcp1.y * plane[1] + dcp2.y) / nl + dL_dl * t.y / l);
    R dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12 -
               (dL_du * t.x + dL_dv * t.y) * tz2 + (dL_dnJ[0][0] + dL_dnJ[1][1]) * (-tz2) + dL_dnJ[0][2] * (2 * t.x * tz3) +
               dL_dnJ[1][2] * (2 * t.y * tz3) + (dL_dnJ[2][0] * t.x + dL_dnJ[2][1] * t.y) * (-t.z / l3) +
               dL_dnJ[2][2] * (1 / l - t.z * t.z / l3) +
               (dcp0.x * (-(v2 + 1)) + dcp0.y * uv + dcp1.x * uv + dcp1.y * (-(u2 + 1)) + dcp2.x * plane[0] + dcp2.y * plane[1]) / nl +
               dL_dl * t.z / l;
    V3<R> dm = xform_vec43_T(V3<R>{dL_dtx, dL_dty, dL_dtz}, view);
    dL_dmeans3D[3 * idx] = dm.x; dL_dmeans3D[3 * idx + 1] = dm.y; dL_dmeans3D[3 * idx + 2] = dm.z;
  }

  // computeColorFromSH bwd, backward.cu:21-140
  void sh_bwd_one(int idx) {
    V3<R> pos{means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    V3<R> dir_orig = pos - V3<R>{campos[0], campos[1], campos[2]};
    V3<R> dir = dir_orig / length(dir_orig);
    auto sh = [&](int k) { return V3<R>{shs[(size_t(idx) * M + k) * 3], shs[(size_t(idx) * M + k) * 3 + 1], shs[(size_t(idx) * M + k) * 3 + 2]}; };
    V3<R> dRGB{dL_dcolors[3 * idx], dL_dcolors[3 * idx + 1], dL_dcolors[3 * idx + 2]};
    dRGB.x *= clamped[3 * idx + 0] ? 0 : 1;
    dRGB.y *= clamped[3 * idx + 1] ? 0 : 1;
    dRGB.z *= clamped[3 * idx + 2] ? 0 : 1;
    V3<R> dRGBdx{0, 0, 0}, dRGBdy{0, 0, 0}, dRGBdz{0, 0, 0};
    R x = dir.x, y = dir.y, z = dir.z;
    auto put = [&](int k, R w) {
      V3<R> v = w * dRGB;
      R* o = &dL_dsh[(size_t(idx) * M + k) * 3];
      o[0] = v.x; o[1] = v.y; o[2] = v.z;
    };
    put(0, R(kC0));
    if (D > 0) {
      put(1, -R(kC1) * y); put(2, R(kC1) * z); put(3, -R(kC1) * x);
      dRGBdx = -R(kC1) * sh(3);
      dRGBdy = -R(kC1) * sh(1);
      dRGBdz = R(kC1) * sh(2);
      if (D > 1) {
        R xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        put(4, R(kC2[0]) * xy); put(5, R(kC2[1]) * yz); put(6, R(kC2[2]) * (R(2.f) * zz - xx - yy));
        put(7, R(kC2[3]) * xz); put(8, R(kC2[4]) * (xx - yy));
        dRGBdx = dRGBdx + ((R(kC2[0]) * y) * sh(4) + (R(kC2[2]) * R(2.f) * -x) * sh(6) + (R(kC2[3]) * z) * sh(7) + (R(kC2[4]) * R(2.f) * x) * sh(8));
        dRGBdy = dRGBdy + ((R(kC2[0]) * x) * sh(4) + (R(kC2[1]) * z) * sh(5) + (R(kC2[2]) * R(2.f) * -y) * sh(6) + (R(kC2[4]) * R(2.f) * -y) * sh(8));
        dRGBdz = dRGBdz + ((R(kC2[1]) * y) * sh(5) + (R(kC2[2]) * R(2.f) * R(2.f) * z) * sh(6) + (R(kC2[3]) * x) * sh(7));
        if (D > 2) {
          put(9, R(kC3[0]) * y * (R(3.f) * xx - yy)); put(10, R(kC3[1]) * xy * z); put(11, R(kC3[2]) * y * (R(4.f) * zz - xx - yy));
          put(12, R(kC3[3]) * z * (R(2.f) * zz - R(3.f) * xx - R(3.f) * yy)); put(13, R(kC3[4]) * x * (R(4.f) * zz - xx - yy));
          put(14, R(kC3[5]) * z * (xx - yy)); put(15, R(kC3[6]) * x * (xx - R(3.f) * yy));
          // glm: scalar * vec3 first, then further scalars multiply the vec3 left to right
          dRGBdx = dRGBdx + ((((R(kC3[0]) * sh(9)) * R(3.f)) * R(2.f)) * xy + (R(kC3[1]) * sh(10)) * yz + ((R(kC3[2]) * sh(11)) * R(-2.f)) * xy +
                   (((R(kC3[3]) * sh(12)) * R(-3.f)) * R(2.f)) * xz + (R(kC3[4]) * sh(13)) * (R(-3.f) * xx + R(4.f) * zz - yy) +
                   ((R(kC3[5]) * sh(14)) * R(2.f)) * xz + ((R(kC3[6]) * sh(15)) * R(3.f)) * (xx - yy));
          dRGBdy = dRGBdy + (((R(kC3[0]) * sh(9)) * R(3.f)) * (xx - yy) + (R(kC3[1]) * sh(10)) * xz + (R(kC3[2]) * sh(11)) * (R(-3.f) * yy + R(4.f) * zz - xx) +
                   (((R(kC3[3]) * sh(12)) * R(-3.f)) * R(2.f)) * yz + ((R(kC3[4]) * sh(13)) * R(-2.f)) * xy + ((R(kC3[5]) * sh(14)) * R(-2.f)) * yz +
                   (((R(kC3[6]) * sh(15)) * R(-3.f)) * R(2.f)) * xy);
          dRGBdz = dRGBdz + ((R(kC3[1]) * sh(10)) * xy + (((R(kC3[2]) * sh(11)) * R(4.f)) * R(2.f)) * yz + ((R(kC3[3]) * sh(12)) * R(3.f)) * (R(2.f) * zz - xx - yy) +
                   (((R(kC3[4]) * sh(13)) * R(4.f)) * R(2.f)) * xz + (R(kC3[5]) * sh(14)) * (xx - yy));
        }
      }
    }
    V3<R> dL_ddir{dot(dRGBdx, dRGB), dot(dRGBdy, dRGB), dot(dRGBdz, dRGB)};
    V3<R> dmean = dnormvdv(dir_orig, dL_ddir);
    dL_dmeans3D[3 * idx] += dmean.x; dL_dmeans3D[3 * idx + 1] += dmean.y; dL_dmeans3D[3 * idx + 2] += dmean.z;
  }

  // computeCov3D bwd, backward.cu:492-555 (returns raw dL_dq, no normalisation Jacobian :554)
  void cov3d_bwd_one(int idx) {
    const R r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2], z = rotations[4 * idx + 3];
    M3<R> Rm(R(1.f) - R(2.f) * (y * y + z * z), R(2.f) * (x * y - r * z), R(2.f) * (x * z + r * y),
             R(2.f) * (x * y + r * z), R(1.f) - R(2.f) * (x * x + z * z), R(2.f) * (y * z - r * x),
             R(2.f) * (x * z - r * y), R(2.f) * (y * z + r * x), R(1.f) - R(2.f) * (x * x + y * y));
    M3<R> S(R(1), R(0), R(0), R(0), R(1), R(0), R(0), R(0), R(1));
    V3<R> s{scale_modifier * scales[3 * idx], scale_modifier * scales[3 * idx + 1], scale_modifier * scales[3 * idx + 2]};
    S[0][0] = s.x; S[1][1] = s.y; S[2][2] = s.z;
    M3<R> Mm = S * Rm;
    const R* d = &dL_dcov3D[6 * idx];
    M3<R> dL_dSigma(d[0], R(0.5f) * d[1], R(0.5f) * d[2], R(0.5f) * d[1], d[3], R(0.5f) * d[4], R(0.5f) * d[2], R(0.5f) * d[4], d[5]);
    M3<R> dL_dM = R(2.0f) * Mm * dL_dSigma;
    M3<R> Rt = transpose(Rm);
    M3<R> dL_dMt = transpose(dL_dM);
    dL_dscales[3 * idx] = dot(Rt[0], dL_dMt[0]);
    dL_dscales[3 * idx + 1] = dot(Rt[1], dL_dMt[1]);
    dL_dscales[3 * idx + 2] = dot(Rt[2], dL_dMt[2]);
    dL_dMt[0] = dL_dMt[0] * s.x;
    dL_dMt[1] = dL_dMt[1] * s.y;
    dL_dMt[2] = dL_dMt[2] * s.z;
    R* q = &dL_drotations[4 * idx];
    q[0] = 2 * z * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * y * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * x * (dL_dMt[1][2] - dL_dMt[2][1]);
    q[1] = 2 * y * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * z * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * r * (dL_dMt[1][2] - dL_dMt[2][1]) - 4 * x * (dL_dMt[2][2] + dL_dMt[1][1]);
    q[2] = 2 * x * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * r * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * z * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * y * (dL_dMt[2][2] + dL_dMt[0][0]);
    q[3] = 2 * r * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * x * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * y * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * z * (dL_dMt[1][1] + dL_dMt[0][0]);
  }

  // preprocessCUDA<3> bwd, backward.cu:560-628
  void preprocess_bwd_one(int idx) {
    if (!(radii[idx] > 0)) return;
    V3<R> m{means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    R m_hom[4];
    xform_point44(m, proj, m_hom);
    R m_w = R(1.0f) / (m_hom[3] + R(0.0000001f));
    R mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    R mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    const R gx2 = dL_dmeans2D[3 * idx], gy2 = dL_dmeans2D[3 * idx + 1];
    R d1x = (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
    R d1y = (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
    R d1z = (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
    V3<R> mv = xform_point43(m, view);
    R t = std::sqrt(mv.x * mv.x + mv.y * mv.y + mv.z * mv.z);
    R dL_dt = dL_dts[idx];
    V3<R> dvp{dL_dview_points[3 * idx], dL_dview_points[3 * idx + 1], dL_dview_points[3 * idx + 2]};
    V3<R> d2 = xform_vec43_T(V3<R>{dvp.x + mv.x / t * dL_dt, dvp.y + mv.y / t * dL_dt, dvp.z + mv.z / t * dL_dt}, view);
    dL_dmeans3D[3 * idx] += d1x + d2.x;
    dL_dmeans3D[3 * idx + 1] += d1y + d2.y;
    dL_dmeans3D[3 * idx + 2] += d1z + d2.z;
    if (has_sh) sh_bwd_one(idx);
    if (has_scales) cov3d_bwd_one(idx);
  }

  // Rasterizer::backward, rasterizer_impl.cu:429-571
  void backward(const PixGrads& gin) {
    auto z = [&](std::vector<R>& v, size_t n) { v.assign(n, R(0)); };
    z(dL_dmeans3D, 3 * size_t(P)); z(dL_dview_points, 3 * size_t(P)); z(dL_dmeans2D, 3 * size_t(P)); z(dL_dcolors, 3 * size_t(P));
    z(dL_dts, P); z(dL_dcamera_planes, 6 * size_t(P)); z(dL_dray_planes, 2 * size_t(P)); z(dL_dnormals, 3 * size_t(P));
    z(dL_dconic, 4 * size_t(P)); z(dL_dopacity, P); z(dL_dcov3D, 6 * size_t(P)); z(dL_dsh, 3 * size_t(P) * M);
    z(dL_dscales, 3 * size_t(P)); z(dL_drotations, 4 * size_t(P));
    if (P != 0) {
      const bool COORD = req_coord, DEPTH = req_depth, NORMAL = req_coord || req_depth;
      std::vector<double> A(size_t(P) * NACC, 0.0);
      if (g_ref_order && sizeof(R) == 4) {
        std::vector<float> Af(size_t(P) * NACC, 0.0f);
        std::vector<AddLogEntry> log;
        for (int tile = 0; tile < gx * gy; tile++) {
          const int ty = tile / gx, tx = tile % gx;
          log.clear();
          tl_addlog.log = &log; tl_addlog.base = A.data();
          for (int y = ty * TILE; y < std::min((ty + 1) * TILE, H); y++)
            for (int x = tx * TILE; x < std::min((tx + 1) * TILE, W); x++) render_pixel_bwd(x, y, COORD, DEPTH, NORMAL, gin, A);
          tl_addlog.log = nullptr;
          std::stable_sort(log.begin(), log.end(), [](const AddLogEntry& a, const AddLogEntry& b) { return a.key < b.key; });
          for (const AddLogEntry& e : log) Af[e.slot] += e.v;
        }
        for (size_t i = 0; i < A.size(); i++) A[i] = Af[i];
      } else {
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
      for (int tile = 0; tile < gx * gy; tile++) {
        const int ty = tile / gx, tx = tile % gx;
        for (int y = ty * TILE; y < std::min((ty + 1) * TILE, H); y++)
          for (int x = tx * TILE; x < std::min((tx + 1) * TILE, W); x++) render_pixel_bwd(x, y, COORD, DEPTH, NORMAL, gin, A);
      }
      }
      for (int i = 0; i < P; i++) {
        const double* a = &A[size_t(i) * NACC];
        for (int c = 0; c < 3; c++) dL_dcolors[3 * i + c] = R(a[A_COLOR + c]);
        for (int c = 0; c < 3; c++) dL_dview_points[3 * i + c] = R(a[A_VIEWPT + c]);
        for (int c = 0; c < 6; c++) dL_dcamera_planes[6 * i + c] = R(a[A_CAMPLANE + c]);
        dL_dts[i] = R(a[A_TS]);
        for (int c = 0; c < 2; c++) dL_dray_planes[2 * i + c] = R(a[A_RAYPLANE + c]);
        for (int c = 0; c < 3; c++) dL_dnormals[3 * i + c] = R(a[A_NORMAL + c]);
        for (int c = 0; c < 3; c++) dL_dmeans2D[3 * i + c] = R(a[A_MEAN2D + c]);
        dL_dconic[4 * i] = R(a[A_CONIC]); dL_dconic[4 * i + 1] = R(a[A_CONIC + 1]); dL_dconic[4 * i + 3] = R(a[A_CONIC + 2]);
        dL_dopacity[i] = R(a[A_OPACITY]);
      }
      acc_dmeans2D = dL_dmeans2D; acc_dconic = dL_dconic; acc_dopacity = dL_dopacity; acc_dcolors = dL_dcolors;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads)
      for (int i = 0; i < P; i++) cov2d_bwd_one(i);
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads)
      for (int i = 0; i < P; i++) preprocess_bwd_one(i);
    }
    register_all();
  }

  void register_all() {
    registry.clear();
#define REG(n) reg(#n, n)
    REG(depths); REG(camera_planes); REG(ray_planes); REG(ts); REG(normals); REG(means2D); REG(view_points); REG(cov3D);
    REG(conic_opacity); REG(rgb); REG(clamped); REG(radii); REG(tiles_touched); REG(point_offsets);
    REG(keys_sorted); REG(point_list); REG(ranges); REG(n_contrib); REG(accum_coord); REG(accum_depth); REG(normal_length);
    REG(out_color); REG(out_coord); REG(out_mcoord); REG(out_depth); REG(out_mdepth); REG(out_alpha); REG(out_normal);
    REG(dL_dmeans3D); REG(dL_dview_points); REG(dL_dmeans2D); REG(dL_dcolors); REG(dL_dts); REG(dL_dcamera_planes);
    REG(dL_dray_planes); REG(dL_dnormals); REG(dL_dconic); REG(dL_dopacity); REG(dL_dcov3D); REG(dL_dsh); REG(dL_dscales);
    REG(invraycov); REG(condition); REG(points2D); REG(point_depths); REG(point_ranges); REG(out9); REG(final_T);
    REG(out_alpha_integrated); REG(out_color_integrated); REG(out_coordinate2d); REG(out_sdf); REG(pt_list);
    REG(dL_drotations); REG(acc_dmeans2D); REG(acc_dconic); REG(acc_dopacity); REG(acc_dcolors);
#undef REG
  }
};

struct Handle {
  int precision;
  Oracle<float>* f = nullptr;
  Oracle<double>* d = nullptr;
  std::vector<std::vector<char>> keep;  // converted upstream grads
};

template <class R, class S> static std::vector<R> load(const S* p, size_t n) {
  std::vector<R> v;
  if (p && n) { v.resize(n); for (size_t i = 0; i < n; i++) v[i] = static_cast<R>(p[i]); }
  return v;
}

template <class R, class S>
static Oracle<R>* build(int P, int D, int M, int W, int H, const S* bg, const S* means3D, const S* shs, const S* colors,
                        const S* opacities, const S* scales, const S* rotations, const S* cov3D, const S* view, const S* proj,
                        const S* campos, double scale_modifier, double tanfovx, double tanfovy, double kernel_size, int require_coord,
                        int require_depth, int nthreads) {
  auto* o = new Oracle<R>();
  o->P = P; o->D = D; o->M = M; o->W = W; o->H = H; o->nthreads = std::max(1, nthreads);
  o->req_coord = require_coord != 0; o->req_depth = require_depth != 0;
  o->means3D = load<R>(means3D, 3 * size_t(P));
  o->has_sh = shs != nullptr; o->shs = load<R>(shs, 3 * size_t(P) * M);
  o->has_colors = colors != nullptr; o->colors_precomp = load<R>(colors, 3 * size_t(P));
  o->opacities = load<R>(opacities, P);
  o->has_scales = scales != nullptr; o->scales = load<R>(scales, 3 * size_t(P)); o->rotations = load<R>(rotations, 4 * size_t(P));
  o->has_cov = cov3D != nullptr; o->cov3D_precomp = load<R>(cov3D, 6 * size_t(P));
  for (int i = 0; i < 16; i++) { o->view[i] = R(view[i]); o->proj[i] = R(proj[i]); }
  for (int i = 0; i < 3; i++) { o->campos[i] = R(campos[i]); o->bg[i] = R(bg[i]); }
  o->scale_modifier = R(scale_modifier); o->tan_fovx = R(tanfovx); o->tan_fovy = R(tanfovy); o->kernel_size = R(kernel_size);
  return o;
}

}  // namespace orc

using namespace orc;

extern "C" {

// precision: 32 -> inputs are float arrays, arithmetic float; 64 -> inputs are double arrays, arithmetic double.
void* oracle_create(int precision, int P, int D, int M, int W, int H, const void* bg, const void* means3D, const void* shs,
                    const void* colors, const void* opacities, const void* scales, const void* rotations, const void* cov3D,
                    const void* view, const void* proj, const void* campos, double scale_modifier, double tanfovx, double tanfovy,
                    double kernel_size, int require_coord, int require_depth, int nthreads) {
  auto* h = new Handle();
  h->precision = precision;
#define ARGS(T) (const T*)bg, (const T*)means3D, (const T*)shs, (const T*)colors, (const T*)opacities, (const T*)scales, (const T*)rotations, \
                (const T*)cov3D, (const T*)view, (const T*)proj, (const T*)campos, scale_modifier, tanfovx, tanfovy, kernel_size,             \
                require_coord, require_depth, nthreads
  if (precision == 64) h->d = build<double, double>(P, D, M, W, H, ARGS(double));
  else h->f = build<float, float>(P, D, M, W, H, ARGS(float));
#undef ARGS
  return h;
}

int oracle_forward(void* hv) {
  auto* h = static_cast<Handle*>(hv);
  return h->d ? h->d->forward() : h->f->forward();
}

// upstream grads have the element type of `precision`; layouts (3,H,W) / (1,H,W)
void oracle_backward(void* hv, const void* g_color, const void* g_coord, const void* g_mcoord, const void* g_depth,
                     const void* g_mdepth, const void* g_alpha, const void* g_normal) {
  auto* h = static_cast<Handle*>(hv);
  if (h->d) {
    Oracle<double>::PixGrads g{(const double*)g_color, (const double*)g_coord, (const double*)g_mcoord, (const double*)g_depth,
                               (const double*)g_mdepth, (const double*)g_alpha, (const double*)g_normal};
    h->d->backward(g);
  } else {
    Oracle<float>::PixGrads g{(const float*)g_color, (const float*)g_coord, (const float*)g_mcoord, (const float*)g_depth,
                              (const float*)g_mdepth, (const float*)g_alpha, (const float*)g_normal};
    h->f->backward(g);
  }
}

// Copy a named internal array. dst == NULL: just return its size in bytes. Unknown name: -1.
long long oracle_get(void* hv, const char* name, void* dst, long long nbytes) {
  auto* h = static_cast<Handle*>(hv);
  auto& reg = h->d ? h->d->registry : h->f->registry;
  auto it = reg.find(name);
  if (it == reg.end()) return -1;
  const long long sz = static_cast<long long>(it->second.second);
  if (dst) std::memcpy(dst, it->second.first, static_cast<size_t>(std::min(sz, nbytes)));
  return sz;
}

// points: element type of `precision`; returns num_rendered.  Outputs via oracle_get: out9 [9,H,W], final_T,
// out_alpha_integrated [PN], out_color_integrated [PN,3], out_coordinate2d [PN,2], out_sdf [PN], radii, invraycov, condition.
int oracle_integrate(void* hv, int PN, const void* points3D) {
  auto* h = static_cast<Handle*>(hv);
  return h->d ? h->d->integrate((const double*)points3D, PN) : h->f->integrate((const float*)points3D, PN);
}

long long oracle_stat_pairs(void* hv) {
  auto* h = static_cast<Handle*>(hv);
  return h->d ? h->d->stat_pairs_fwd : h->f->stat_pairs_fwd;
}
long long oracle_stat_blended(void* hv) {
  auto* h = static_cast<Handle*>(hv);
  return h->d ? h->d->stat_blended_fwd : h->f->stat_blended_fwd;
}

void oracle_destroy(void* hv) {
  auto* h = static_cast<Handle*>(hv);
  delete h->f;
  delete h->d;
  delete h;
}

// checkFrustum / markVisible, rasterizer_impl.cu:54-66,176-188 (near cull only: auxiliary.h:166)
void oracle_mark_visible(int P, const float* means3D, const float* view, const float* /*proj*/, unsigned char* present) {
  for (int i = 0; i < P; i++) {
    V3<float> p{means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    V3<float> pv = xform_point43(p, view);
    present[i] = !(pv.z <= 0.2f);
  }
}

float oracle_exp_spec(float x) { return exp_spec_impl(x); }
void oracle_set_exp_mode(int mode) { g_exp_mode = mode; }
void oracle_set_opacity_slip(int on) { g_opacity_slip = on; }
void oracle_set_ref_order(int on) { g_ref_order = on; }
unsigned oracle_higher_msb(unsigned n) { return higher_msb(n); }
// glm column-major KAT hook (forward.cu:126-133): mat3(1..9) * (1,1,1)
void oracle_kat_mat3(float out[3]) {
  M3<float> m(1, 2, 3, 4, 5, 6, 7, 8, 9);
  V3<float> r = m * V3<float>{1, 1, 1};
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int oracle_sym_eigen3(const float* sym6, float* evals, float* evecs9) {
  M3<float> S(sym6[0], sym6[1], sym6[2], sym6[1], sym6[3], sym6[4], sym6[2], sym6[4], sym6[5]);
  V3<float> ev; M3<float> V;
  int D = sym_eigen3(S, ev, V);
  for (int i = 0; i < 3; i++) { evals[i] = ev[i]; for (int j = 0; j < 3; j++) evecs9[3 * i + j] = V[i][j]; }
  return D;
}
}
