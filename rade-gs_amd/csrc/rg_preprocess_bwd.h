// rg_preprocess_bwd.h -- per-Gaussian backward stage (host+device, fp32, -ffp-contract=off).
//
// One call does for one visible Gaussian what the reference spreads over two kernels:
//   computeCov2DCUDA        DGR/cuda_rasterizer/backward.cu:145-488
//   preprocessCUDA<3> bwd   backward.cu:560-628, with computeColorFromSH bwd :21-140 and
//                           computeCov3D bwd :492-555
// Input: the per-Gaussian sums the blend backward accumulated (SplatAcc); output: the gradients
// the operator returns.  The formulas are the reference's hand-derived ones, operation for
// operation; the oracle's copy of them is validated against float64 finite differences
// (tests/test_oracle_fd.py), and this header is validated against the oracle bit-for-bit on
// the host (tests/test_hostcheck.py) and on the device (tests/test_gpu_parity.py).
#pragma once
#include "rg_preprocess.h"

namespace rg {

struct SplatAcc {     // sums over all (pixel, this Gaussian) pairs
  float dcolor[3];
  float dts;
  float drp[2];       // d/d ray_plane   (already divided by focal, backward.cu:939-940)
  float dnrm[3];
  float dmean2D[3];   // x, y signed; z = sum of |.| (abs-grad used by densification)
  float dconic[3];    // conic x, y, w (the .z slot of the reference's float4 is never written)
  float dop;          // d/d (opacity*coef)
  float dvp[3];       // d/d view-space mean (coord map)
  float dcp[6];       // d/d camera_plane    (already divided by focal, backward.cu:917-922)
  // how the blend backward left the record (csrc/rg_streams.inc), both false for a record in the reference's own form:
  bool raw = false;      // dmean2D[0..1] and dconic[0..2] are RAW MOMENTS of h = opacity G dL/dalpha about the Gaussian's centre (sum h dx,
                         // sum h dy | sum h dx dx, sum h dx dy, sum h dy dy): the reference's sums are linear in them with coefficients this
                         // function derives anyway (the conic, the depth / camera planes), so they are formed here, once per Gaussian
  bool half_wh = false;  // the W/2, H/2 factors of dL_dmean2D (backward.cu:1002-1003) have not been applied yet
};

struct SplatBwd {
  float dmean2D[3];      // the returned dL_dmean2D row (x, y signed; z = abs-gradient)
  float sums_mean2D[3];  // the record's mean2D / conic sums in the reference's form (before W/2, H/2): what a raw record converts to
  float sums_conic[3];
  float dmean3D[3];
  float dopacity;
  float dcov3D[6];
  float dscale[3];
  float drot[4];
};

// dnormvdv(float3), auxiliary.h:124-134
RG_HD v3 dnormvdv(v3 v, v3 dv) {
  float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  v3 o;
  o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return o;
}

// SH backward: writes (deg+1)^2 rows of dsh, returns d/d mean through the view direction.
RG_HD v3 sh_bwd(int deg, const float* sh, v3 pos, const float campos[3], unsigned clamped, const float dcolor[3], float* dsh) {
  v3 dir_orig = sub(pos, mk3(campos[0], campos[1], campos[2]));
  v3 dir = div(dir_orig, len(dir_orig));
  v3 dRGB = mk3(dcolor[0], dcolor[1], dcolor[2]);
  dRGB.x *= (clamped & 1u) ? 0 : 1;
  dRGB.y *= (clamped & 2u) ? 0 : 1;
  dRGB.z *= (clamped & 4u) ? 0 : 1;
  v3 dx = mk3(0, 0, 0), dy = mk3(0, 0, 0), dz = mk3(0, 0, 0);
  const float x = dir.x, y = dir.y, z = dir.z;
#define SH(k) mk3(sh[3 * (k)], sh[3 * (k) + 1], sh[3 * (k) + 2])
#define PUT(k, wgt)                                  \
  if (dsh) {                                         \
    v3 tv = mul((wgt), dRGB);                        \
    dsh[3 * (k)] = tv.x; dsh[3 * (k) + 1] = tv.y; dsh[3 * (k) + 2] = tv.z; \
  }
  // Ordering note: inside every degree block all reads of sh[] come BEFORE the writes of dsh[] for the
  // same coefficients, so sh and dsh may alias (the kernel stages both through one LDS row).  The
  // statements are independent, so this order changes no value.
  PUT(0, RG_C0);
  if (deg > 0) {
    dx = mul(-RG_C1, SH(3));
    dy = mul(-RG_C1, SH(1));
    dz = mul(RG_C1, SH(2));
    PUT(1, -RG_C1 * y); PUT(2, RG_C1 * z); PUT(3, -RG_C1 * x);
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      dx = add(dx, add(add(add(mul(RG_C2_0 * y, SH(4)), mul(RG_C2_2 * 2.f * -x, SH(6))), mul(RG_C2_3 * z, SH(7))), mul(RG_C2_4 * 2.f * x, SH(8))));
      dy = add(dy, add(add(add(mul(RG_C2_0 * x, SH(4)), mul(RG_C2_1 * z, SH(5))), mul(RG_C2_2 * 2.f * -y, SH(6))), mul(RG_C2_4 * 2.f * -y, SH(8))));
      dz = add(dz, add(add(mul(RG_C2_1 * y, SH(5)), mul(RG_C2_2 * 2.f * 2.f * z, SH(6))), mul(RG_C2_3 * x, SH(7))));
      PUT(4, RG_C2_0 * xy); PUT(5, RG_C2_1 * yz); PUT(6, RG_C2_2 * (2.f * zz - xx - yy));
      PUT(7, RG_C2_3 * xz); PUT(8, RG_C2_4 * (xx - yy));
      if (deg > 2) {
        // scalar*vec3 first, further scalars then multiply the vec3 left to right (backward.cu:100-123)
        dx = add(dx, add(add(add(add(add(add(mul(mul(mul(mul(RG_C3_0, SH(9)), 3.f), 2.f), xy), mul(mul(RG_C3_1, SH(10)), yz)),
                                         mul(mul(mul(RG_C3_2, SH(11)), -2.f), xy)),
                                     mul(mul(mul(mul(RG_C3_3, SH(12)), -3.f), 2.f), xz)),
                                 mul(mul(RG_C3_4, SH(13)), (-3.f * xx + 4.f * zz - yy))),
                             mul(mul(mul(RG_C3_5, SH(14)), 2.f), xz)),
                         mul(mul(mul(RG_C3_6, SH(15)), 3.f), (xx - yy))));
        dy = add(dy, add(add(add(add(add(add(mul(mul(mul(RG_C3_0, SH(9)), 3.f), (xx - yy)), mul(mul(RG_C3_1, SH(10)), xz)),
                                         mul(mul(RG_C3_2, SH(11)), (-3.f * yy + 4.f * zz - xx))),
                                     mul(mul(mul(mul(RG_C3_3, SH(12)), -3.f), 2.f), yz)),
                                 mul(mul(mul(RG_C3_4, SH(13)), -2.f), xy)),
                             mul(mul(mul(RG_C3_5, SH(14)), -2.f), yz)),
                         mul(mul(mul(mul(RG_C3_6, SH(15)), -3.f), 2.f), xy)));
        dz = add(dz, add(add(add(add(mul(mul(RG_C3_1, SH(10)), xy), mul(mul(mul(mul(RG_C3_2, SH(11)), 4.f), 2.f), yz)),
                                 mul(mul(mul(RG_C3_3, SH(12)), 3.f), (2.f * zz - xx - yy))),
                             mul(mul(mul(mul(RG_C3_4, SH(13)), 4.f), 2.f), xz)),
                         mul(mul(RG_C3_5, SH(14)), (xx - yy))));
        PUT(9, RG_C3_0 * y * (3.f * xx - yy)); PUT(10, RG_C3_1 * xy * z); PUT(11, RG_C3_2 * y * (4.f * zz - xx - yy));
        PUT(12, RG_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy)); PUT(13, RG_C3_4 * x * (4.f * zz - xx - yy));
        PUT(14, RG_C3_5 * z * (xx - yy)); PUT(15, RG_C3_6 * x * (xx - 3.f * yy));
      }
    }
  }
#undef SH
#undef PUT
  v3 ddir = mk3(dot(dx, dRGB), dot(dy, dRGB), dot(dz, dRGB));
  return dnormvdv(dir_orig, ddir);
}

// The SH gradient is an outer product: dL/dsh[k] = w_k(dir) * dL/dRGB (clamp-masked).  w_k exactly as sh_bwd's PUT()
// weights; used to rebuild dL/dsh from per-view (dir, dRGB) pairs after a view-parallel exchange (view_parallel.py).
RG_HD void sh_basis(int deg, v3 pos, const float campos[3], float w[16]) {
  v3 dir_orig = sub(pos, mk3(campos[0], campos[1], campos[2]));
  v3 dir = div(dir_orig, len(dir_orig));
  const float x = dir.x, y = dir.y, z = dir.z;
#pragma unroll
  for (int k = 0; k < 16; k++) w[k] = 0.f;
  w[0] = RG_C0;
  if (deg > 0) {
    w[1] = -RG_C1 * y; w[2] = RG_C1 * z; w[3] = -RG_C1 * x;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      w[4] = RG_C2_0 * xy; w[5] = RG_C2_1 * yz; w[6] = RG_C2_2 * (2.f * zz - xx - yy); w[7] = RG_C2_3 * xz; w[8] = RG_C2_4 * (xx - yy);
      if (deg > 2) {
        w[9] = RG_C3_0 * y * (3.f * xx - yy); w[10] = RG_C3_1 * xy * z; w[11] = RG_C3_2 * y * (4.f * zz - xx - yy);
        w[12] = RG_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy); w[13] = RG_C3_4 * x * (4.f * zz - xx - yy);
        w[14] = RG_C3_5 * z * (xx - yy); w[15] = RG_C3_6 * x * (xx - 3.f * yy);
      }
    }
  }
}

// scale/quaternion backward (raw dL/dq: the caller's F.normalize owns that Jacobian, backward.cu:554)
RG_HD void cov3d_bwd(const float s3[3], float mod, const float q[4], const float d[6], float dscale[3], float drot[4]) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  m3 R = mk33(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
              2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
              2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
  m3 S = mk33(1, 0, 0, 0, 1, 0, 0, 0, 1);
  const float sx = mod * s3[0], sy = mod * s3[1], sz = mod * s3[2];
  S.c[0][0] = sx; S.c[1][1] = sy; S.c[2][2] = sz;
  m3 M = mul(S, R);
  m3 dSigma = mk33(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
  m3 dM = mul(scale_l(2.0f, M), dSigma);
  m3 Rt = transpose(R);
  m3 dMt = transpose(dM);
  dscale[0] = dot(col(Rt, 0), col(dMt, 0));
  dscale[1] = dot(col(Rt, 1), col(dMt, 1));
  dscale[2] = dot(col(Rt, 2), col(dMt, 2));
#pragma unroll
  for (int k = 0; k < 3; k++) {
    dMt.c[0][k] *= sx;
    dMt.c[1][k] *= sy;
    dMt.c[2][k] *= sz;
  }
#define T(c_, r_) dMt.c[c_][r_]
  drot[0] = 2 * z * (T(0, 1) - T(1, 0)) + 2 * y * (T(2, 0) - T(0, 2)) + 2 * x * (T(1, 2) - T(2, 1));
  drot[1] = 2 * y * (T(1, 0) + T(0, 1)) + 2 * z * (T(2, 0) + T(0, 2)) + 2 * r * (T(1, 2) - T(2, 1)) - 4 * x * (T(2, 2) + T(1, 1));
  drot[2] = 2 * x * (T(1, 0) + T(0, 1)) + 2 * r * (T(2, 0) - T(0, 2)) + 2 * z * (T(1, 2) + T(2, 1)) - 4 * y * (T(2, 2) + T(0, 0));
  drot[3] = 2 * r * (T(0, 1) - T(1, 0)) + 2 * x * (T(2, 0) + T(0, 2)) + 2 * y * (T(1, 2) + T(2, 1)) - 4 * z * (T(1, 1) + T(0, 0));
#undef T
}

// One VISIBLE Gaussian (radius > 0).  cov3D: the covariance the forward used (precomputed or
// re-derived from scale/quat by the caller with cov3d_from_scale_rot -- same bits as forward).
// op_combined: what the reference's kernel reads as `conic_opacity[idx].w` -- by default the conic gradient a.dconic[2] (upstream's
// argument slip, include/radegs.h::opacity_grad_intended), else opacity*coef as stored by the forward.  sh/dsh may be null (precomputed colours).
RG_HD void preprocess_bwd(v3 mean, const float* scale3, const float* quat4, const float cov3D[6], float op_combined, int deg,
                          const float* sh, unsigned clamped, const Camera& cam, const SplatAcc& a, float* dsh, SplatBwd& o) {
  Cov2D g;
  cov2d_common(mean, cam, cov3D, g);
  const v3 t = g.t;
  const float txtz = g.txtz, tytz = g.tytz;
  const float h_x = cam.focal_x, h_y = cam.focal_y, ks = cam.kernel_size;
  const float u2 = txtz * txtz, v2 = tytz * tytz, uv = txtz * tytz;
  const float det_0 = g.det0, det_1 = g.det1, coef = g.coef;
  const m3& T = g.T; const m3& Vrk = g.Vrk; const m3& W = g.W; const m3& cov2D = g.cov;
  const v3 dL_dnormal = mk3(a.dnrm[0], a.dnrm[1], a.dnrm[2]);
  const float dcp0x = a.dcp[0], dcp0y = a.dcp[1], dcp1x = a.dcp[2], dcp1y = a.dcp[3], dcp2x = a.dcp[4], dcp2y = a.dcp[5];
  const float drpx = a.drp[0], drpy = a.drp[1];

  m3 dL_dVrk = zero33(), dL_dnJ = zero33();
  v3 plane = mk3(0, 0, 0);
  float dL_du, dL_dv, dL_dl, l, nl;
  float cplx[3] = {0.f, 0.f, 0.f}, cply[3] = {0.f, 0.f, 0.f};   // camera planes without 1/focal (zero in the degenerate branch, as the forward's)
  if (g.uvh_mn.x != g.uvh_mn.x || g.D == 0) {  // backward.cu:262-272
    nl = 1; l = 1; dL_du = 0; dL_dv = 0; dL_dl = 0;
  } else {
    const v3 uvh = g.uvh, uvh_m = g.uvh_m, uvh_mn = g.uvh_mn;
    const float vb = dot(uvh_m, uvh), vbn = dot(uvh_mn, uvh);
    l = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
    m3 nJ = mk33(1 / t.z, 0.0f, -(t.x) / (t.z * t.z), 0.0f, 1 / t.z, -(t.y) / (t.z * t.z), t.x / l, t.y / l, t.z / l);
    m3 nJ_inv = mk33(v2 + 1, -uv, 0, -uv, u2 + 1, 0, -txtz, -tytz, 0);
    const float clamp_vb = fmaxf(vb, 0.0000001f), clamp_vbn = fmaxf(vbn, 0.0000001f);
    nl = u2 + v2 + 1;
    const float factor_normal = l / nl;
    v3 uvh_m_vb = div(uvh_mn, clamp_vbn);
    plane = mul(nJ_inv, uvh_m_vb);
    // planes WITHOUT the 1/focal factors: the blend backward pre-divided its sums (backward.cu:297-301)
    const float cpl0x = (-(v2 + 1) * t.z + plane.x * t.x) / nl, cpl0y = (uv * t.z + plane.y * t.x) / nl;
    const float cpl1x = (uv * t.z + plane.x * t.y) / nl, cpl1y = (-(u2 + 1) * t.z + plane.y * t.y) / nl;
    const float cpl2x = (t.x + plane.x * t.z) / nl, cpl2y = (t.y + plane.y * t.z) / nl;
    cplx[0] = cpl0x; cplx[1] = cpl1x; cplx[2] = cpl2x; cply[0] = cpl0y; cply[1] = cpl1y; cply[2] = cpl2y;
    const float rplx = plane.x * factor_normal, rply = plane.y * factor_normal;
    v3 ray_n = mk3(-plane.x * factor_normal, -plane.y * factor_normal, -1.0f);
    v3 cam_n = mul(nJ, ray_n);
    v3 nrm = normalize(cam_n);
    const float lv = len(cam_n);
    const v3 dn_lv = div(dL_dnormal, lv);
    v3 dL_dcam_n = sub(dn_lv, mul(nrm, dot(nrm, dn_lv)));
    v3 dL_dray_n = mul(transpose(nJ), dL_dcam_n);
    dL_dnJ = outer(dL_dcam_n, ray_n);
    dL_dl = (-plane.x * dL_dray_n.x - plane.y * dL_dray_n.y + plane.x * drpx + plane.y * drpy) / nl;
    const float dplx = (t.x * dcp0x + t.y * dcp1x + t.z * dcp2x - l * dL_dray_n.x + drpx * l) / nl;
    const float dply = (t.x * dcp0y + t.y * dcp1y + t.z * dcp2y - l * dL_dray_n.y + drpy * l) / nl;
    v3 dpl3 = mk3(dplx, dply, 0.0f);
    const float dL_dnl = (-dcp0x * cpl0x - dcp0y * cpl0y - dcp1x * cpl1x - dcp1y * cpl1y - dcp2x * cpl2x - dcp2y * cpl2y -
                          dL_dray_n.x * ray_n.x - dL_dray_n.y * ray_n.y - drpx * rplx - drpy * rply) / nl;
    const float tmp = dplx * plane.x + dply * plane.y;
    v3 W_uvh = mul(W, uvh);
    if (g.well) {
      dL_dVrk = neg(outer(mul(g.Vinv, W_uvh), mul(divs(g.Vinv, clamp_vb), add(mul(W_uvh, (-tmp)), mul(mul(W, transpose(nJ_inv)), dpl3)))));
    } else {
      const float dL_dvb = -tmp / clamp_vb;
      v3 nJi_dp = mul(transpose(nJ_inv), mk3(dplx / clamp_vb, dply / clamp_vb, 0.0f));
      m3 dVinv = outer(W_uvh, add(mul(W_uvh, dL_dvb), mul(W, nJi_dp)));
      v3 dvv = mul(add(dVinv, transpose(dVinv)), g.evmin);
      const float emin = g.min_id == 0 ? g.eig.d[0] : (g.min_id == 1 ? g.eig.d[1] : g.eig.d[2]);
#pragma unroll
      for (int j = 0; j < 3; j++) {
        if (j != g.min_id) {
          v3 evj = eig_vec(g.eig, j);
          const float sc = dot(evj, dvv) / fminf(emin - g.eig.d[j], -0.0000001f);
          dL_dVrk = add(dL_dVrk, outer(mul(evj, sc), g.evmin));
        }
      }
    }
    v3 dL_duvh = add(mul(2 * (-tmp), uvh_m_vb), mul(mul(divs(g.cam_inv, clamp_vb), transpose(nJ_inv)), dpl3));
    m3 dnJi = outer(dpl3, uvh_m_vb);
    dL_du = dL_dnl * 2 * txtz + dL_duvh.x + (dnJi.c[0][1] + dnJi.c[1][0]) * (-tytz) + 2 * dnJi.c[1][1] * txtz - dnJi.c[2][0] +
            (dcp0y * t.y + dcp1x * t.y + dcp1y * (-2 * t.x)) / nl;
    dL_dv = dL_dnl * 2 * tytz + dL_duvh.y + (dnJi.c[0][1] + dnJi.c[1][0]) * (-txtz) + 2 * dnJi.c[0][0] * tytz - dnJi.c[2][1] +
            (dcp0x * (-2 * t.y) + dcp0y * t.x + dcp1x * t.x) / nl;
  }

  // opacity-compensation chain, double-precision sub-expressions (backward.cu:367-375)
  const float opacity = (float)((double)op_combined / ((double)coef + 1e-6));
  const float dL_dcoef = a.dop * opacity;
  const float dL_dsqrtcoef = (float)((double)dL_dcoef * 0.5 * 1. / ((double)coef + 1e-6));
  const float dL_ddet0 = (float)((double)dL_dsqrtcoef / ((double)det_1 + 1e-6));
  const float dL_ddet1 = (float)((double)(dL_dsqrtcoef * det_0) * ((double)(-1.f) / ((double)(det_1 * det_1) + 1e-6)));
  const float c00 = cov2D.c[0][0], c01 = cov2D.c[0][1], c11 = cov2D.c[1][1];
  const float dcoef_da = dL_ddet0 * c11 + dL_ddet1 * (c11 + ks);
  const float dcoef_db = (float)((double)dL_ddet0 * (-2. * (double)c01) + (double)dL_ddet1 * (-2. * (double)c01));
  const float dcoef_dc = dL_ddet0 * c00 + dL_ddet1 * (c00 + ks);
  const float ca = c00 + ks, cb = c01, cc = c11 + ks;
  const float denom = ca * cc - cb * cb;
  float dL_da = 0, dL_db = 0, dL_dc = 0;
  const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  // the blend backward's mean2D / conic sums -- from a raw record: the moments times the coefficients of backward.cu:981-1012
  float dcx = a.dconic[0], dcy = a.dconic[1], dcz = a.dconic[2];
  float gx2 = a.dmean2D[0], gy2 = a.dmean2D[1];
  if (a.raw) {
    const float idet = 1.0f / denom;                              // the conic as the forward forms it (rg_preprocess.h: cvz / det, ...)
    const float kx = cc * idet, ky = -cb * idet, kz = ca * idet;
    const float shx = a.dmean2D[0], shy = a.dmean2D[1];
    const float rfac = l / nl;                                    // ray plane = plane * l / nl / focal (0 in the degenerate branch)
    gx2 = -(kx * shx + ky * shy) + (plane.x * rfac * a.dts + (cplx[0] * a.dvp[0] + cplx[1] * a.dvp[1] + cplx[2] * a.dvp[2])) / h_x;
    gy2 = -(ky * shx + kz * shy) + (plane.y * rfac * a.dts + (cply[0] * a.dvp[0] + cply[1] * a.dvp[1] + cply[2] * a.dvp[2])) / h_y;
    dcx = -0.5f * a.dconic[0]; dcy = -0.5f * a.dconic[1]; dcz = -0.5f * a.dconic[2];
  }
  o.sums_mean2D[0] = gx2; o.sums_mean2D[1] = gy2; o.sums_mean2D[2] = a.dmean2D[2];
  o.sums_conic[0] = dcx; o.sums_conic[1] = dcy; o.sums_conic[2] = dcz;
  if (a.half_wh) { gx2 *= 0.5f * (float)cam.W; gy2 *= 0.5f * (float)cam.H; }
  o.dmean2D[0] = gx2; o.dmean2D[1] = gy2; o.dmean2D[2] = a.dmean2D[2];
  float* dcov = o.dcov3D;
  o.dopacity = a.dop;
  if (denom2inv != 0) {
    dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
    dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
    dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
    if ((double)det_0 <= 1e-6 || (double)det_1 <= 1e-6) {
      o.dopacity = 0;
    } else {
      dL_da += dcoef_da; dL_dc += dcoef_dc; dL_db += dcoef_db;
      o.dopacity = a.dop * coef;
    }
#define TT(c_, r_) T.c[c_][r_]
    dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
    dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
    dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
    dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
    dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
    dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) dcov[i] = 0;
  }
  dcov[0] += dL_dVrk.c[0][0];
  dcov[3] += dL_dVrk.c[1][1];
  dcov[5] += dL_dVrk.c[2][2];
  dcov[1] += dL_dVrk.c[0][1] + dL_dVrk.c[1][0];
  dcov[2] += dL_dVrk.c[0][2] + dL_dVrk.c[2][0];
  dcov[4] += dL_dVrk.c[1][2] + dL_dVrk.c[2][1];

#define VV(c_, r_) Vrk.c[c_][r_]
  const float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da + (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
  const float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da + (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
  const float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da + (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
  const float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc + (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
  const float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc + (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
  const float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc + (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
#undef VV
#undef TT
  const float dL_dJ00 = W.c[0][0] * dL_dT00 + W.c[0][1] * dL_dT01 + W.c[0][2] * dL_dT02;
  const float dL_dJ02 = W.c[2][0] * dL_dT00 + W.c[2][1] * dL_dT01 + W.c[2][2] * dL_dT02;
  const float dL_dJ11 = W.c[1][0] * dL_dT10 + W.c[1][1] * dL_dT11 + W.c[1][2] * dL_dT12;
  const float dL_dJ12 = W.c[2][0] * dL_dT10 + W.c[2][1] * dL_dT11 + W.c[2][2] * dL_dT12;
  const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
  const float l3 = l * l * l;
#define NJ(c_, r_) dL_dnJ.c[c_][r_]
  const float dL_dtx = g.xmul * (-h_x * tz2 * dL_dJ02 + dL_du * tz - NJ(0, 2) * tz2 + NJ(2, 0) * (1 / l - t.x * t.x / l3) +
                                 NJ(2, 1) * (-t.x * t.y / l3) + NJ(2, 2) * (-t.x * t.z / l3) +
                                 (dcp0x * plane.x + dcp0y * plane.y + dcp2x) / nl + dL_dl * t.x / l);
  const float dL_dty = g.ymul * (-h_y * tz2 * dL_dJ12 + dL_dv * tz - NJ(1, 2) * tz2 + NJ(2, 0) * (-t.x * t.y / l3) +
                                 NJ(2, 1) * (1 / l - t.y * t.y / l3) + NJ(2, 2) * (-t.y * t.z / l3) +
                                 (dcp1x * plane.x + dcp1y * plane.y + dcp2y) / nl + dL_dl * t.y / l);
  const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12 -
                       (dL_du * t.x + dL_dv * t.y) * tz2 + (NJ(0, 0) + NJ(1, 1)) * (-tz2) + NJ(0, 2) * (2 * t.x * tz3) +
                       NJ(1, 2) * (2 * t.y * tz3) + (NJ(2, 0) * t.x + NJ(2, 1) * t.y) * (-t.z / l3) + NJ(2, 2) * (1 / l - t.z * t.z / l3) +
                       (dcp0x * (-(v2 + 1)) + dcp0y * uv + dcp1x * uv + dcp1y * (-(u2 + 1)) + dcp2x * plane.x + dcp2y * plane.y) / nl +
                       dL_dl * t.z / l;
#undef NJ
  v3 dmean = xform43_T(mk3(dL_dtx, dL_dty, dL_dtz), cam.view);

  // ---- second kernel of the reference: projection, ts, view_points, SH, scale/rot ----
  const float* pj = cam.proj;
  const float m_hw = pj[3] * mean.x + pj[7] * mean.y + pj[11] * mean.z + pj[15];
  const float m_w = 1.0f / (m_hw + 0.0000001f);
  const float mul1 = (pj[0] * mean.x + pj[4] * mean.y + pj[8] * mean.z + pj[12]) * m_w * m_w;
  const float mul2 = (pj[1] * mean.x + pj[5] * mean.y + pj[9] * mean.z + pj[13]) * m_w * m_w;
  const float d1x = (pj[0] * m_w - pj[3] * mul1) * gx2 + (pj[1] * m_w - pj[3] * mul2) * gy2;
  const float d1y = (pj[4] * m_w - pj[7] * mul1) * gx2 + (pj[5] * m_w - pj[7] * mul2) * gy2;
  const float d1z = (pj[8] * m_w - pj[11] * mul1) * gx2 + (pj[9] * m_w - pj[11] * mul2) * gy2;
  v3 mv = xform43(mean, cam.view);
  const float tt = sqrtf(mv.x * mv.x + mv.y * mv.y + mv.z * mv.z);
  const float dL_dt = a.dts;
  v3 d2 = xform43_T(mk3(a.dvp[0] + mv.x / tt * dL_dt, a.dvp[1] + mv.y / tt * dL_dt, a.dvp[2] + mv.z / tt * dL_dt), cam.view);
  dmean.x += d1x + d2.x;
  dmean.y += d1y + d2.y;
  dmean.z += d1z + d2.z;
  if (sh) {
    v3 ds = sh_bwd(deg, sh, mean, cam.campos, clamped, a.dcolor, dsh);
    dmean.x += ds.x; dmean.y += ds.y; dmean.z += ds.z;
  }
  o.dmean3D[0] = dmean.x; o.dmean3D[1] = dmean.y; o.dmean3D[2] = dmean.z;
  if (scale3) cov3d_bwd(scale3, cam.scale_modifier, quat4, dcov, o.dscale, o.drot);
}

}  // namespace rg
