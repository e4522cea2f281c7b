// rg_blend.h -- the per-(pixel, Gaussian) "decision chain" shared by the forward and the
// backward blend kernels, host+device.
//
// What must be reproducible bit-for-bit between the device and the CPU oracle is exactly the
// chain that feeds the thresholded decisions of the reference's blend loop
// (DGR/cuda_rasterizer/forward.cu:552-573, backward.cu:842-857):
//     d = mean2D - pix;  power = -0.5*(cx*dx*dx + cz*dy*dy) - cy*dx*dy;   power > 0  -> skip
//     alpha = min(0.99, op * exp(power));                                 alpha < 1/255 -> skip
//     test_T = T * (1 - alpha);                                           test_T < 1e-4 -> done
//     T > 0.5 (median bookkeeping)
// so these operations are written with one rounding each, in source order, and exp() is the
// fully specified exp_spec() below (CUDA's expf cannot be reproduced off-device).
// Everything downstream of the decisions (colour/depth/normal accumulation) is free to use fma.
#pragma once
#include "rg_math.h"

namespace rg {

// exp_spec(x) for x <= ~0:  k = rint(x*log2e);  r = x - k*ln2 (two-step Cody-Waite, fma);
// degree-5 polynomial (Cephes expf coefficients) in fma form;  result * 2^k through the
// exponent field.  x < -87 returns 0.  <= 1 ulp from expf on [-87, 0].
// k = rint(x*log2e) WITHOUT v_rndne / v_cvt (both half rate on gfx950): for |t| < 2^22, t + 1.5*2^23 lies in [2^23, 2^24), where
// one ulp is 1, so the addition itself rounds t to the nearest integer, ties to even (the constant is even) -- the same value rintf(t)
// gives; subtracting the constant is exact, and the integer k sits in the low mantissa bits of the sum.  Bit-identical to the rintf /
// (int32_t) form the oracle spells (tests/test_hostcheck.py compares the two over the whole argument range).
constexpr float kExpMagic = 12582912.0f;
RG_HD float exp_spec(float x) {
  if (x < -87.0f) return 0.0f;
  const float tm = x * 1.44269504088896341f + kExpMagic;   // == kExpMagic + rintf(x * log2e): see kExpMagic
  const float kf = tm - kExpMagic;
  float r = fmaf(kf, -0.693359375f, x);
  r = fmaf(kf, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  const float y = fmaf(p, r2, r) + 1.0f;
  union { float f; uint32_t i; } u, t;
  u.f = y;
  t.f = tm;
  u.i += t.i << 23;   // bits(tm) = bits(kExpMagic) + k and bits(kExpMagic) << 23 == 0 (mod 2^32): k lands in the exponent field
  return u.f;
}

// The same value as exp_spec(x) for x >= -87; below, exp_spec(-87) (~1.6e-38) instead of 0.  For callers that only use the result
// through `op * exp < 1/255` (any op <= 1 fails it either way): straight-line code, no branch between two pixels' evaluations.
RG_HD float exp_spec_floor(float x) {
  x = fmaxf(x, -87.0f);
  const float tm = x * 1.44269504088896341f + kExpMagic;
  const float kf = tm - kExpMagic;
  float r = fmaf(kf, -0.693359375f, x);
  r = fmaf(kf, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  const float y = fmaf(p, r2, r) + 1.0f;
  union { float f; uint32_t i; } u, t;
  u.f = y;
  t.f = tm;
  u.i += t.i << 23;   // bits(tm) = bits(kExpMagic) + k and bits(kExpMagic) << 23 == 0 (mod 2^32): k lands in the exponent field
  return u.f;
}

// Conservative skip threshold: any power below it gives alpha < 1/255 under the exact rule
// (margin 1e-3 in the exponent >> the 1e-6 relative error of exp_spec and of logf).  op <= 0
// gives +inf (always skip: alpha <= 0 < 1/255); NaN makes the prefilter a no-op and the exact
// rule decides.  This is only a shortcut -- it never changes a decision.
RG_HD float skip_threshold(float op) { return logf(1.0f / (255.0f * op)) - 1.0e-3f; }

// quadratic form of the conic; a_x = (cx*dx)*dx and b_xy = cy*dx are hoisted by callers that
// keep dx fixed across several pixels of one lane.
RG_HD float splat_power(float a_x, float b_xy, float cz, float dy) {
  const float s = a_x + (cz * dy) * dy;
  const float v = b_xy * dy;
  // -0.5f*s is exact, so this single fma rounds exactly like (-0.5f*s) - v.
  return fmaf(-0.5f, s, -v);
}

// ---- which 8x4-pixel blocks of a 16x16 tile can a splat reach?  (block lists of the sub-tile entry streams, rg_streams.inc) ----
// Bit b of the result (block column b & 1, block row b >> 1) is set unless NO pixel centre of the block can pass the blend loop's
// alpha >= 1/255 test.  The set {alpha can reach 1/255} is the ellipse  cx dx^2 + 2 cy dx dy + cz dy^2 <= M  around the mean,
// M = -2 thr + slack (thr: skip_threshold(), which already carries 1e-3 of margin in the exponent).  A block row is a slab
// dy in [va, va + 3]; the ellipse's x-extent over a slab is an interval whose ends are reached either at the ellipse's own
// x-extreme (when that point lies in the slab) or on the slab line nearest to it:
//     f+-(t) = (-cy t +- sqrt(cx M - det t^2)) / cx            (roots of the quadratic in dx at dy = t)
//     x_hi = f+(clamp(t*, a, b)),  x_lo = f-(clamp(-t*, a, b)),  t* = -cy hx / cz,  hx = sqrt(cz M / det),  [a, b] = slab ^ [-hy, hy]
// and the row's two blocks are kept when that interval reaches their columns: 4 slabs x 2 roots instead of 8 x 4 clamped edge
// minimisations.  Conservative by construction: slack covers the fp32 rounding of `power` at a pixel (1e-5 of the magnitude of its
// terms over the tile) plus 1e-2 for this function's own arithmetic (the root moves by < slack / (2 sqrt(disc)) for a relative
// error of 1e-4 in det, which `regular` guarantees), the discriminant is biased upwards, extents are widened by 0.01 px + 1e-4 hx.
// Irregular conics (NaN, not positive definite, determinant lost to cancellation) keep every block: the exact per-pixel rule
// decides, as for any entry of a tile-wide list.  thr > 0: not even the centre reaches 1/255.
#if defined(__HIP_DEVICE_COMPILE__)
#define RG_SQRT_APPROX(x) __builtin_amdgcn_sqrtf(x)
#define RG_RCP_APPROX(x) __builtin_amdgcn_rcpf(x)
#else
#define RG_SQRT_APPROX(x) sqrtf(x)
#define RG_RCP_APPROX(x) (1.0f / (x))
#endif
// The test in three pieces so that callers which visit several tiles of one splat share the work: the set-up once per splat
// (U, V: the largest |dx|, |dy| of any pixel centre that will be asked about), the x-extent once per slab, two compares per block.
struct EllipseSetup {
  int kind;   // 0: reaches nothing (thr > 0); 1: irregular, keep every block; 2: the fields below are valid
  float cy, det, cxM, icx, hx, hy, tstar, eps;
};
RG_HD EllipseSetup ellipse_setup(float mx, float my, float cx, float cy, float cz, float thr, float U, float V) {
  EllipseSetup e;
  e.cy = cy; e.det = 0.f; e.cxM = 0.f; e.icx = 0.f; e.hx = 0.f; e.hy = 0.f; e.tstar = 0.f; e.eps = 0.f;
  if (thr > 0.f) { e.kind = 0; return e; }
  const float cxcz = cx * cz;
  const float det = cxcz - cy * cy;
  const float chk = ((mx + my) + (cx + cy)) + ((cz + thr) + (U + V));   // NaN / inf - inf anywhere -> NaN
  const bool regular = (cx > 0.f) && (cz > 0.f) && (det > 1.0e-3f * cxcz) && (chk == chk) && (cxcz < 1.0e30f);
  if (!regular) { e.kind = 1; return e; }
  const float terms = fmaf(cx * U, U, fmaf(cz * V, V, 2.0f * fabsf(cy) * U * V));
  const float M = fmaf(2.0e-5f, terms, fmaf(-2.0f, thr, 1.0e-2f));
  const float rdet = RG_RCP_APPROX(det), icz = RG_RCP_APPROX(cz);
  e.kind = 2;
  e.det = det;
  e.icx = RG_RCP_APPROX(cx);
  e.cxM = cx * M * 1.000001f;
  e.hy = RG_SQRT_APPROX(e.cxM * rdet) * 1.0002f;
  e.hx = RG_SQRT_APPROX(cz * M * rdet) * 1.0002f;
  e.tstar = -cy * e.hx * icz;
  e.eps = fmaf(1.0e-4f, e.hx, 1.0e-2f);
  return e;
}
// x-extent [xlo, xhi] (relative to the mean, already widened by eps) of the ellipse over the slab dy in [va, va + 3]; false: misses it
RG_HD bool ellipse_slab(const EllipseSetup& e, float va, float& xlo, float& xhi) {
  const float vb = va + 3.f;
  const bool hit = (va <= e.hy) && (vb >= -e.hy);
  const float a = fmaxf(va, -e.hy), b = fminf(vb, e.hy);
  const float tp = fminf(fmaxf(e.tstar, a), b), tm = fminf(fmaxf(-e.tstar, a), b);
  const float dp = fmaxf(fmaf(-e.det * tp, tp, e.cxM), 0.f), dm = fmaxf(fmaf(-e.det * tm, tm, e.cxM), 0.f);
  const float hi = (RG_SQRT_APPROX(dp) - e.cy * tp) * e.icx, lo = (-RG_SQRT_APPROX(dm) - e.cy * tm) * e.icx;
  xhi = ((tp == e.tstar) ? e.hx : hi) + e.eps;
  xlo = ((tm == -e.tstar) ? -e.hx : lo) - e.eps;
  return hit;
}
// the two blocks of a block row: pixel columns [ua, ua + 7] and [ua + 8, ua + 15] relative to the mean
RG_HD uint32_t ellipse_cols(bool hit, float xlo, float xhi, float ua) {
  const bool c0 = hit && (xlo <= ua + 7.f) && (xhi >= ua), c1 = hit && (xlo <= ua + 15.f) && (xhi >= ua + 8.f);
  return (c0 ? 1u : 0u) | (c1 ? 2u : 0u);
}
RG_HD uint32_t ellipse_tile_mask(const EllipseSetup& e, float ua, float va0) {
  if (e.kind != 2) return e.kind == 1 ? 0xFFu : 0u;
  uint32_t mask = 0u;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float xlo, xhi;
    const bool hit = ellipse_slab(e, va0 + (float)(4 * r), xlo, xhi);
    mask |= ellipse_cols(hit, xlo, xhi, ua) << (2 * r);
  }
  return mask;
}
RG_HD uint32_t ellipse_block_mask(float mx, float my, float cx, float cy, float cz, float thr, float tile_x0, float tile_y0) {
  const float ua = tile_x0 - mx, va0 = tile_y0 - my;
  const float U = fmaxf(fabsf(ua), fabsf(ua + 15.f)), V = fmaxf(fabsf(va0), fabsf(va0 + 15.f));
  return ellipse_tile_mask(ellipse_setup(mx, my, cx, cy, cz, thr, U, V), ua, va0);
}

}  // namespace rg
