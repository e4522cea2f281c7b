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
RG_HD float exp_spec(float x) {
  if (x < -87.0f) return 0.0f;
  const float kf = rintf(x * 1.44269504088896341f);
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
  union { float f; int32_t i; } u;
  u.f = y;
  u.i += ((int32_t)kf) << 23;
  return u.f;
}

// The same value as exp_spec(x) for x >= -87; below, exp_spec(-87) (~1.6e-38) instead of 0.  For callers that only use the result
// through `op * exp < 1/255` (any op <= 1 fails it either way): straight-line code, no branch between two pixels' evaluations.
RG_HD float exp_spec_floor(float x) {
  x = fmaxf(x, -87.0f);
  const float kf = rintf(x * 1.44269504088896341f);
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
  union { float f; int32_t i; } u;
  u.f = y;
  u.i += ((int32_t)kf) << 23;
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

}  // namespace rg
