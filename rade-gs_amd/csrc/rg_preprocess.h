// rg_preprocess.h -- per-Gaussian forward stage (host+device, fp32, -ffp-contract=off).
//
// Computes what the reference's preprocessCUDA<3,false> computes for one Gaussian
// (DGR/cuda_rasterizer/forward.cu:307-423, with computeCov3D :270-304, computeCov2D<false>
// :77-264, computeColorFromSH :23-74, in_frustum auxiliary.h:155-180, getRect :62-72,
// ndc2Pix :57-60) and returns it in a register struct; the kernel decides where it goes.
// The shared part of computeCov2D (everything up to the ray-space plane) is also what the
// backward re-derives (backward.cu:182-252), so it lives in cov2d_common().
#pragma once
#include "rg_math.h"

namespace rg {

constexpr int kTile = 16;  // tile edge in pixels; fixed by the binning contract (config.h:15-16)

struct Camera {
  float view[16];
  float proj[16];
  float campos[3];
  float focal_x, focal_y, tan_fovx, tan_fovy;
  float kernel_size, scale_modifier;
  int W, H, gx, gy;
};

// SH constants (auxiliary.h:35-52)
#define RG_C0 0.28209479177387814f
#define RG_C1 0.4886025119029199f
#define RG_C2_0 1.0925484305920792f
#define RG_C2_1 -1.0925484305920792f
#define RG_C2_2 0.31539156525252005f
#define RG_C2_3 -1.0925484305920792f
#define RG_C2_4 0.5462742152960396f
#define RG_C3_0 -0.5900435899266435f
#define RG_C3_1 2.890611442640554f
#define RG_C3_2 -0.4570457994644658f
#define RG_C3_3 0.3731763325901154f
#define RG_C3_4 -0.4570457994644658f
#define RG_C3_5 1.445305721320277f
#define RG_C3_6 -0.5900435899266435f

// ((v+1)*S-1)/2 evaluated in double, rounded once to float (auxiliary.h:57-60)
RG_HD float ndc_to_pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

// Tile rectangle of a splat (auxiliary.h:62-72).  The upper bound is spelled as the reference evaluates it in float, left to
// right: ((p + r) + BLOCK) - 1 -- a single "+ 15" can round differently by one ulp and move a tile boundary.
RG_HD void tile_rect(float px, float py, int max_radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
  const float rad = (float)max_radius;
  x0 = imin(gx, imax(0, f2i_sat((px - rad) / (float)kTile)));
  y0 = imin(gy, imax(0, f2i_sat((py - rad) / (float)kTile)));
  x1 = imin(gx, imax(0, f2i_sat((((px + rad) + (float)kTile) - 1.0f) / (float)kTile)));
  y1 = imin(gy, imax(0, f2i_sat((((py + rad) + (float)kTile) - 1.0f) / (float)kTile)));
}

// Sigma = (S R)^T (S R) from scale and the UN-normalised quaternion (r,x,y,z); 6 unique entries.
RG_HD void cov3d_from_scale_rot(const float s3[3], float mod, const float q[4], float out[6]) {
  m3 S = mk33(1, 0, 0, 0, 1, 0, 0, 0, 1);
  S.c[0][0] = mod * s3[0];
  S.c[1][1] = mod * s3[1];
  S.c[2][2] = mod * s3[2];
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  m3 R = mk33(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
              2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
              2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
  m3 M = mul(S, R);
  m3 Sg = mul(transpose(M), M);
  out[0] = Sg.c[0][0]; out[1] = Sg.c[0][1]; out[2] = Sg.c[0][2];
  out[3] = Sg.c[1][1]; out[4] = Sg.c[1][2]; out[5] = Sg.c[2][2];
}

struct Cov2D {
  v3 t;                 // clamped view-space mean
  float txtz, tytz;     // after clamping
  float xmul, ymul;     // 0 where the clamp was active (backward only)
  m3 W, T, Vrk, cov;    // cov = T^T Vrk^T T (before the 2D filter)
  float det0, det1, coef;  // coef before the "forced to 0" rule
  int D;                // eigen-solver status
  Eig3 eig;
  int min_id;
  bool well;            // lambda_min > 1e-8
  v3 evmin;
  m3 Vinv, cam_inv;
  v3 uvh, uvh_m, uvh_mn;
};

RG_HD void cov2d_common(v3 mean, const Camera& cam, const float cov3D[6], Cov2D& o) {
  v3 t = xform43(mean, cam.view);
  const float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
  float txtz = t.x / t.z, tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  o.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  o.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  txtz = t.x / t.z;
  tytz = t.y / t.z;
  o.t = t; o.txtz = txtz; o.tytz = tytz;
  const float fx = cam.focal_x, fy = cam.focal_y;
  const float* v = cam.view;
  m3 J = mk33(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z), 0, 0, 0);
  o.W = mk33(v[0], v[4], v[8], v[1], v[5], v[9], v[2], v[6], v[10]);
  o.T = mul(o.W, J);
  o.Vrk = mk33(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  o.cov = mul(mul(transpose(o.T), transpose(o.Vrk)), o.T);
  const float c00 = o.cov.c[0][0], c01 = o.cov.c[0][1], c11 = o.cov.c[1][1], ks = cam.kernel_size;
  // fp32 determinant, double max/divide/sqrt (forward.cu:119-121, backward.cu:215-218)
  o.det0 = (float)fmax(1e-6, (double)(c00 * c11 - c01 * c01));
  o.det1 = (float)fmax(1e-6, (double)((c00 + ks) * (c11 + ks) - c01 * c01));
  o.coef = (float)sqrt((double)o.det0 / ((double)o.det1 + 1e-6) + 1e-6);

  o.D = sym_eigen3(o.Vrk.c[0][0], o.Vrk.c[1][0], o.Vrk.c[2][0], o.Vrk.c[1][1], o.Vrk.c[2][1], o.Vrk.c[2][2], o.eig);
  const float e0 = o.eig.d[0], e1 = o.eig.d[1], e2 = o.eig.d[2];
  o.min_id = e0 > e1 ? (e1 > e2 ? 2 : 1) : (e0 > e2 ? 2 : 0);
  const float emin = o.min_id == 0 ? e0 : (o.min_id == 1 ? e1 : e2);
  o.well = (double)emin > 0.00000001;
  m3 V = mk33(o.eig.a[0], o.eig.a[3], o.eig.a[6], o.eig.a[1], o.eig.a[4], o.eig.a[7], o.eig.a[2], o.eig.a[5], o.eig.a[8]);
  if (o.well) {
    m3 dg = mk33(1 / e0, 0, 0, 0, 1 / e1, 0, 0, 0, 1 / e2);
    o.Vinv = mul(mul(V, dg), transpose(V));
  } else {
    // select chain instead of a dynamic column index: keeps V in registers (no scratch / LDS promotion)
    o.evmin = o.min_id == 0 ? col(V, 0) : (o.min_id == 1 ? col(V, 1) : col(V, 2));
    o.Vinv = outer(o.evmin, o.evmin);
  }
  o.cam_inv = mul(mul(transpose(o.W), o.Vinv), o.W);
  o.uvh = mk3(txtz, tytz, 1.0f);
  o.uvh_m = mul(o.cam_inv, o.uvh);
  o.uvh_mn = normalize(o.uvh_m);
}

struct SplatFwd {
  int radius;           // 0 => invisible, nothing else is meaningful
  int tiles;            // number of tiles in the rect
  float depth;          // view-space z (sort key)
  float mx, my;         // pixel-space centre
  float cx, cy, cz;     // conic
  float op;             // opacity * coef
  float ts;             // |p_view|
  float rgb[3];
  float rp[2];          // ray-space depth gradient per pixel
  float nrm[3];
  float cp[6];          // camera-space coord gradient per pixel (3x2)
  float vp[3];          // view-space mean
  unsigned clamped;     // bit c set <=> channel c was clamped at 0
  unsigned rect;        // tile rectangle x0 | y0 << 8 | w << 16 | h << 24 (valid when the grid is at most 255x255 tiles)
  // INTE only (the integrate() path): inverse covariance in ray space (u/f, v/f, t), upper triangle, and whether the
  // 3D covariance was well conditioned (computeCov2D<true>, forward.cu:187-235)
  float icr[6];
  bool well;
};

// SH -> RGB (+0.5, clamp at 0, remember which channels clamped).  sh points at this
// Gaussian's (M,3) block; only (deg+1)^2 rows are read.
RG_HD void sh_to_rgb(int deg, const float* sh, v3 pos, const float campos[3], float rgb[3], unsigned& clamped) {
  // The coefficients are requested in batches BEFORE the first of a batch is used: degree block by degree block (rounds 1-5) the loads
  // were four dependent round trips to memory at the very end of every wave's life -- they sit behind `deg` branches the compiler cannot
  // hoist them over, and the waves of a SIMD reach this point together.  RADEGS_SH_BATCH: 1 = everything at once (48 registers live),
  // 2 = degrees 0-2 (27), then degree 3 (21).
#ifndef RADEGS_SH_BATCH
#define RADEGS_SH_BATCH 1
#endif
  float c[48];
#pragma unroll
  for (int i = 0; i < 3; i++) c[i] = sh[i];
  if (deg > 0) {
#pragma unroll
    for (int i = 3; i < 12; i++) c[i] = sh[i];
  }
  if (deg > 1) {
#pragma unroll
    for (int i = 12; i < 27; i++) c[i] = sh[i];
  }
#if RADEGS_SH_BATCH == 1
  if (deg > 2) {
#pragma unroll
    for (int i = 27; i < 48; i++) c[i] = sh[i];
  }
#endif
  v3 dir = sub(pos, mk3(campos[0], campos[1], campos[2]));
  dir = div(dir, len(dir));
#define SH(k) mk3(c[3 * (k)], c[3 * (k) + 1], c[3 * (k) + 2])
  v3 res = mul(RG_C0, SH(0));
  if (deg > 0) {
    const float x = dir.x, y = dir.y, z = dir.z;
    res = sub(add(sub(res, mul(RG_C1 * y, SH(1))), mul(RG_C1 * z, SH(2))), mul(RG_C1 * x, SH(3)));
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      res = add(add(add(add(add(res, mul(RG_C2_0 * xy, SH(4))), mul(RG_C2_1 * yz, SH(5))), mul(RG_C2_2 * (2.0f * zz - xx - yy), SH(6))),
                    mul(RG_C2_3 * xz, SH(7))), mul(RG_C2_4 * (xx - yy), SH(8)));
      if (deg > 2) {
#if RADEGS_SH_BATCH != 1
#pragma unroll
        for (int i = 27; i < 48; i++) c[i] = sh[i];
#endif
        res = add(add(add(add(add(add(add(res, mul(RG_C3_0 * y * (3.0f * xx - yy), SH(9))), mul(RG_C3_1 * xy * z, SH(10))),
                                  mul(RG_C3_2 * y * (4.0f * zz - xx - yy), SH(11))),
                              mul(RG_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), SH(12))),
                          mul(RG_C3_4 * x * (4.0f * zz - xx - yy), SH(13))),
                      mul(RG_C3_5 * z * (xx - yy), SH(14))),
                  mul(RG_C3_6 * x * (xx - 3.0f * yy), SH(15)));
      }
    }
  }
#undef SH
  res = add(res, mk3(0.5f, 0.5f, 0.5f));
  clamped = (res.x < 0 ? 1u : 0u) | (res.y < 0 ? 2u : 0u) | (res.z < 0 ? 4u : 0u);
  rgb[0] = fmaxf(res.x, 0.0f);
  rgb[1] = fmaxf(res.y, 0.0f);
  rgb[2] = fmaxf(res.z, 0.0f);
}

// One Gaussian.  cov3D_in: precomputed covariance (6) or nullptr; scale/quat used otherwise.
// sh: this Gaussian's SH block or nullptr; color_in: precomputed RGB (3) or nullptr.
template <bool INTE = false>
RG_HD void preprocess_fwd(v3 p_orig, const float* scale3, const float* quat4, const float* cov3D_in, float opacity,
                          int deg, const float* sh, const float* color_in, const Camera& cam, SplatFwd& o) {
  o.radius = 0;
  o.tiles = 0;
  o.clamped = 0;
  o.rect = 0;
  v3 p_view = xform43(p_orig, cam.view);
  if (p_view.z <= 0.2f) return;  // near cull (auxiliary.h:166); x/y frustum test is disabled upstream
  const float* pm = cam.proj;
  const float hx = pm[0] * p_orig.x + pm[4] * p_orig.y + pm[8] * p_orig.z + pm[12];
  const float hy = pm[1] * p_orig.x + pm[5] * p_orig.y + pm[9] * p_orig.z + pm[13];
  const float hw = pm[3] * p_orig.x + pm[7] * p_orig.y + pm[11] * p_orig.z + pm[15];
  const float p_w = 1.0f / (hw + 0.0000001f);
  const float projx = hx * p_w, projy = hy * p_w;

  float cov3D[6];
  if (cov3D_in) {
#pragma unroll
    for (int i = 0; i < 6; i++) cov3D[i] = cov3D_in[i];
  } else {
    cov3d_from_scale_rot(scale3, cam.scale_modifier, quat4, cov3D);
  }

  Cov2D g;
  cov2d_common(p_orig, cam, cov3D, g);
  const float ks = cam.kernel_size;
  const float cvx = g.cov.c[0][0] + ks, cvy = g.cov.c[0][1], cvz = g.cov.c[1][1] + ks;
  float coef = g.coef;
  if ((double)g.det0 <= 1e-6 || (double)g.det1 <= 1e-6) coef = 0.0f;

  if (g.uvh_mn.x != g.uvh_mn.x || g.D == 0) {  // NaN or eigen-solver failure: zero geometry (forward.cu:162-168)
#pragma unroll
    for (int i = 0; i < 6; i++) o.cp[i] = 0;
    o.nrm[0] = o.nrm[1] = o.nrm[2] = 0;
    o.rp[0] = o.rp[1] = 0;
    if constexpr (INTE) {
#pragma unroll
      for (int i = 0; i < 6; i++) o.icr[i] = 0;  // the reference leaves its zero-filled tensor untouched here
    }
  } else {
    const v3 t = g.t;
    const float txtz = g.txtz, tytz = g.tytz;
    const float u2 = txtz * txtz, v2 = tytz * tytz, uv = txtz * tytz;
    const float l = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
    m3 nJ = mk33(1 / t.z, 0.0f, -(t.x) / (t.z * t.z), 0.0f, 1 / t.z, -(t.y) / (t.z * t.z), t.x / l, t.y / l, t.z / l);
    m3 nJ_inv = mk33(v2 + 1, -uv, 0, -uv, u2 + 1, 0, -txtz, -tytz, 0);
    const float vbn = dot(g.uvh_mn, g.uvh);
    const float factor_normal = l / (u2 + v2 + 1);
    v3 plane = mul(nJ_inv, div(g.uvh_mn, fmaxf(vbn, 0.0000001f)));
    const float nl = u2 + v2 + 1;
    const float fx = cam.focal_x, fy = cam.focal_y;
    o.cp[0] = (-(v2 + 1) * t.z + plane.x * t.x) / nl / fx;
    o.cp[1] = (uv * t.z + plane.y * t.x) / nl / fy;
    o.cp[2] = (uv * t.z + plane.x * t.y) / nl / fx;
    o.cp[3] = (-(u2 + 1) * t.z + plane.y * t.y) / nl / fy;
    o.cp[4] = (t.x + plane.x * t.z) / nl / fx;
    o.cp[5] = (t.y + plane.y * t.z) / nl / fy;
    o.rp[0] = plane.x * l / nl / fx;
    o.rp[1] = plane.y * l / nl / fy;
    v3 ray_n = mk3(-plane.x * factor_normal, -plane.y * factor_normal, -1.0f);
    v3 n = normalize(mul(nJ, ray_n));
    o.nrm[0] = n.x; o.nrm[1] = n.y; o.nrm[2] = n.z;
    if constexpr (INTE) {
      m3 icr;
      if (g.well) {
        const float ltz = u2 + v2 + 1;
        m3 full = mk33(v2 + 1, -uv, txtz / l * ltz, -uv, u2 + 1, tytz / l * ltz, -txtz, -tytz, 1 / l * ltz);
        m3 T2 = mul(g.W, transpose(scale_l(t.z / (u2 + v2 + 1), full)));
        icr = mul(mul(transpose(T2), g.Vinv), T2);
      } else {
        // Upstream's shadowed `inv_cov_ray` (forward.cu:223) leaves the matrix that is stored UNINITIALISED in this
        // branch, and the value it discards is 1/rounding-noise.  Defined as zero here (include/radegs.h).
        icr = zero33();
      }
      m3 sc = mk33(1 / cam.focal_x, 0.f, 0.f, 0.f, 1 / cam.focal_y, 0.f, 0.f, 0.f, 1.f);
      icr = mul(mul(sc, icr), sc);
      o.icr[0] = icr.c[0][0]; o.icr[1] = icr.c[0][1]; o.icr[2] = icr.c[0][2];
      o.icr[3] = icr.c[1][1]; o.icr[4] = icr.c[1][2]; o.icr[5] = icr.c[2][2];
    }
  }
  if constexpr (INTE) o.well = g.well;

  o.ts = sqrtf(p_view.x * p_view.x + p_view.y * p_view.y + p_view.z * p_view.z);
  const float det = (cvx * cvz - cvy * cvy);
  if (det == 0.0f) return;
  const float det_inv = 1.f / det;
  o.cx = cvz * det_inv;
  o.cy = -cvy * det_inv;
  o.cz = cvx * det_inv;
  const float mid = 0.5f * (cvx + cvz);
  const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
  const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
  o.mx = ndc_to_pix(projx, cam.W);
  o.my = ndc_to_pix(projy, cam.H);
  int x0, y0, x1, y1;
  const int irad = f2i_sat(my_radius);
  tile_rect(o.mx, o.my, irad, cam.gx, cam.gy, x0, y0, x1, y1);
  if ((x1 - x0) * (y1 - y0) == 0) return;

  if (color_in) {
    o.rgb[0] = color_in[0]; o.rgb[1] = color_in[1]; o.rgb[2] = color_in[2];
  } else {
    sh_to_rgb(deg, sh, p_orig, cam.campos, o.rgb, o.clamped);
  }
  o.depth = p_view.z;
  o.vp[0] = p_view.x; o.vp[1] = p_view.y; o.vp[2] = p_view.z;
  o.op = opacity * coef;
  o.radius = irad;
  o.tiles = (y1 - y0) * (x1 - x0);
  o.rect = (unsigned)x0 | ((unsigned)y0 << 8) | ((unsigned)(x1 - x0) << 16) | ((unsigned)(y1 - y0) << 24);
}

}  // namespace rg
