// TEST HARNESS (not product code): runs the product's host+device per-Gaussian math
// (rade-gs_amd/csrc/rg_*.h) on the CPU so it can be compared bit-for-bit with the oracle in a
// container that has no GPU.  The product never links this file.
#include <cmath>
#include <cstdint>
#include <cstring>
#include "rg_blend.h"
#include "rg_preprocess.h"
#include "rg_preprocess_bwd.h"

using namespace rg;

static Camera make_cam(const float* view, const float* proj, const float* campos, int W, int H, float tanfovx, float tanfovy,
                       float kernel_size, float scale_modifier) {
  Camera c;
  memcpy(c.view, view, 64);
  memcpy(c.proj, proj, 64);
  memcpy(c.campos, campos, 12);
  c.focal_y = H / (2.0f * tanfovy);
  c.focal_x = W / (2.0f * tanfovx);
  c.tan_fovx = tanfovx; c.tan_fovy = tanfovy; c.kernel_size = kernel_size; c.scale_modifier = scale_modifier;
  c.W = W; c.H = H; c.gx = (W + kTile - 1) / kTile; c.gy = (H + kTile - 1) / kTile;
  return c;
}

extern "C" {

// out_f: [P][27] = mx,my,cx,cy,cz,op,ts,rgb3,rp2,nrm3,cp6,vp3,depth ; out_i: [P][3] = radius,tiles,clamped
void hc_preprocess_fwd(int P, int deg, int M, const float* means, const float* scales, const float* rots, const float* cov3D,
                       const float* opac, const float* shs, const float* colors, const float* view, const float* proj,
                       const float* campos, int W, int H, float tanfovx, float tanfovy, float kernel_size, float scale_modifier,
                       float* out_f, int* out_i) {
  Camera cam = make_cam(view, proj, campos, W, H, tanfovx, tanfovy, kernel_size, scale_modifier);
  for (int i = 0; i < P; i++) {
    SplatFwd s;
    memset(&s, 0, sizeof(s));
    preprocess_fwd(mk3(means[3 * i], means[3 * i + 1], means[3 * i + 2]), scales ? scales + 3 * i : nullptr, rots ? rots + 4 * i : nullptr,
                   cov3D ? cov3D + 6 * i : nullptr, opac[i], deg, shs ? shs + (size_t)i * M * 3 : nullptr,
                   colors ? colors + 3 * i : nullptr, cam, s);
    float* f = out_f + (size_t)i * 27;
    f[0] = s.mx; f[1] = s.my; f[2] = s.cx; f[3] = s.cy; f[4] = s.cz; f[5] = s.op; f[6] = s.ts;
    for (int k = 0; k < 3; k++) f[7 + k] = s.rgb[k];
    for (int k = 0; k < 2; k++) f[10 + k] = s.rp[k];
    for (int k = 0; k < 3; k++) f[12 + k] = s.nrm[k];
    for (int k = 0; k < 6; k++) f[15 + k] = s.cp[k];
    for (int k = 0; k < 3; k++) f[21 + k] = s.vp[k];
    f[24] = s.depth; f[25] = 0; f[26] = 0;
    memcpy(&f[25], &s.rect, 4);   // packed tile rectangle, bit pattern carried in a float slot
    out_i[3 * i] = s.radius; out_i[3 * i + 1] = s.tiles; out_i[3 * i + 2] = (int)s.clamped;
  }
}

// INTE preprocess (integrate path): out_f [P][7] = icr0..icr5, well ; out_i [P] = radius
void hc_preprocess_inte(int P, int deg, int M, const float* means, const float* scales, const float* rots, const float* opac,
                        const float* shs, const float* view, const float* proj, const float* campos, int W, int H, float tanfovx,
                        float tanfovy, float kernel_size, float scale_modifier, float* out_f, int* out_i) {
  Camera cam = make_cam(view, proj, campos, W, H, tanfovx, tanfovy, kernel_size, scale_modifier);
  for (int i = 0; i < P; i++) {
    SplatFwd s;
    memset(&s, 0, sizeof(s));
    preprocess_fwd<true>(mk3(means[3 * i], means[3 * i + 1], means[3 * i + 2]), scales + 3 * i, rots + 4 * i, nullptr, opac[i], deg,
                         shs + (size_t)i * M * 3, nullptr, cam, s);
    for (int k = 0; k < 6; k++) out_f[(size_t)i * 7 + k] = s.icr[k];
    out_f[(size_t)i * 7 + 6] = s.well ? 1.0f : 0.0f;
    out_i[i] = s.radius;
  }
}

// acc: [P][25] in SplatAcc field order (dcolor3,dts,drp2,dnrm3,dmean2D3,dconic3,dop,dvp3,dcp6)
// out: [P][17] = dmean3D3,dopacity,dcov3D6,dscale3,drot4 ; dsh: [P][M][3]
void hc_preprocess_bwd(int P, int deg, int M, const float* means, const float* scales, const float* rots, const float* cov3D_pre,
                       const float* shs, const int* radii, const int* clamped, const float* op_combined, const float* view,
                       const float* proj, const float* campos, int W, int H, float tanfovx, float tanfovy, float kernel_size,
                       float scale_modifier, const float* acc, float* out, float* dsh) {
  Camera cam = make_cam(view, proj, campos, W, H, tanfovx, tanfovy, kernel_size, scale_modifier);
  for (int i = 0; i < P; i++) {
    float* o = out + (size_t)i * 17;
    for (int k = 0; k < 17; k++) o[k] = 0;
    if (!(radii[i] > 0)) continue;
    SplatAcc a;
    memcpy(&a, acc + (size_t)i * 25, sizeof(float) * 25);
    float cov[6];
    if (cov3D_pre) memcpy(cov, cov3D_pre + 6 * i, 24);
    else cov3d_from_scale_rot(scales + 3 * i, scale_modifier, rots + 4 * i, cov);
    SplatBwd b;
    memset(&b, 0, sizeof(b));
    preprocess_bwd(mk3(means[3 * i], means[3 * i + 1], means[3 * i + 2]), scales ? scales + 3 * i : nullptr, rots ? rots + 4 * i : nullptr,
                   cov, op_combined[i], deg, shs ? shs + (size_t)i * M * 3 : nullptr, (unsigned)clamped[i], cam, a,
                   dsh ? dsh + (size_t)i * M * 3 : nullptr, b);
    for (int k = 0; k < 3; k++) o[k] = b.dmean3D[k];
    o[3] = b.dopacity;
    for (int k = 0; k < 6; k++) o[4 + k] = b.dcov3D[k];
    for (int k = 0; k < 3; k++) o[10 + k] = b.dscale[k];
    for (int k = 0; k < 4; k++) o[13 + k] = b.drot[k];
  }
}

// Block masks of the sub-tile entry streams: ellipse_block_mask() against the brute-force truth "some pixel centre of the block
// passes the forward blend loop's own test" (power <= 0 and min(0.99, op exp_spec(power)) >= 1/255, evaluated exactly as the
// kernels do).  rec: [n][6] = mx, my, cx, cy, cz, op; the tile's first pixel is (tx0, ty0).  out: [n][2] = mask, truth.
void hc_block_masks(int n, const float* rec, float tx0, float ty0, unsigned* out) {
  for (int i = 0; i < n; i++) {
    const float* r = rec + 6 * (size_t)i;
    const float mx = r[0], my = r[1], cx = r[2], cy = r[3], cz = r[4], op = r[5];
    out[2 * i] = ellipse_block_mask(mx, my, cx, cy, cz, skip_threshold(op), tx0, ty0);
    unsigned truth = 0;
    for (int b = 0; b < 8; b++) {
      bool any = false;
      for (int py = 0; py < 4 && !any; py++)
        for (int px = 0; px < 8 && !any; px++) {
          const float x = tx0 + (b & 1) * 8 + px, y = ty0 + (b >> 1) * 4 + py;
          const float dx = mx - x, dy = my - y;
          const float power = splat_power((cx * dx) * dx, cy * dx, cz, dy);
          if (power > 0.0f) continue;
          const float alpha = fminf(0.99f, op * exp_spec(power));
          if (!(alpha < 1.0f / 255.0f)) any = true;
        }
      if (any) truth |= 1u << b;
    }
    out[2 * i + 1] = truth;
  }
}

// The same question asked the way emit_instances_kernel<true> asks it: one set-up per splat for its whole tile rectangle
// [tx0, tx1) x [ty0, ty1) (tile units), the slab extents of a tile row shared by its tiles.  out: [n][tiles][2] = mask, truth with
// tiles = (tx1-tx0)*(ty1-ty0), rows outer.
void hc_block_masks_rect(int n, const float* rec, int tx0, int ty0, int tx1, int ty1, unsigned* out) {
  const int tw = tx1 - tx0, th = ty1 - ty0;
  for (int i = 0; i < n; i++) {
    const float* r = rec + 6 * (size_t)i;
    const float mx = r[0], my = r[1], cx = r[2], cy = r[3], cz = r[4], op = r[5];
    const float U = fmaxf(fabsf((float)(tx0 * 16) - mx), fabsf((float)(tx1 * 16 - 1) - mx));
    const float V = fmaxf(fabsf((float)(ty0 * 16) - my), fabsf((float)(ty1 * 16 - 1) - my));
    const EllipseSetup e = ellipse_setup(mx, my, cx, cy, cz, skip_threshold(op), U, V);
    for (int y = ty0; y < ty1; y++) {
      float xl[4] = {0, 0, 0, 0}, xh[4] = {0, 0, 0, 0};
      bool hit[4] = {false, false, false, false};
      if (e.kind == 2)
        for (int q = 0; q < 4; q++) hit[q] = ellipse_slab(e, (float)(y * 16 + 4 * q) - my, xl[q], xh[q]);
      for (int x = tx0; x < tx1; x++) {
        unsigned mask = e.kind == 1 ? 0xFFu : 0u;
        if (e.kind == 2)
          for (int q = 0; q < 4; q++) mask |= ellipse_cols(hit[q], xl[q], xh[q], (float)(x * 16) - mx) << (2 * q);
        unsigned truth = 0;
        for (int b = 0; b < 8; b++) {
          bool any = false;
          for (int py = 0; py < 4 && !any; py++)
            for (int px = 0; px < 8 && !any; px++) {
              const float X = (float)(x * 16 + (b & 1) * 8 + px), Y = (float)(y * 16 + (b >> 1) * 4 + py);
              const float dx = mx - X, dy = my - Y;
              const float power = splat_power((cx * dx) * dx, cy * dx, cz, dy);
              if (power > 0.0f) continue;
              if (!(fminf(0.99f, op * exp_spec(power)) < 1.0f / 255.0f)) any = true;
            }
          if (any) truth |= 1u << b;
        }
        unsigned* o = out + 2 * (((size_t)i * th + (y - ty0)) * tw + (x - tx0));
        o[0] = mask; o[1] = truth;
      }
    }
  }
}

float hc_exp_spec(float x) { return exp_spec(x); }
// The specification as the oracle spells it (rintf + integer conversion); exp_spec / exp_spec_floor (rg_blend.h) reach k through a
// magic-number addition instead.  Counts the bit patterns in [lo_bits, hi_bits] (step `stride`) on which either differs from it.
static float exp_spec_rint_form(float x) {
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
long long hc_exp_spec_sweep(unsigned lo_bits, unsigned hi_bits, unsigned stride) {
  long long bad = 0;
  for (unsigned long long b = lo_bits; b <= hi_bits; b += stride) {
    float x; unsigned bb = (unsigned)b; memcpy(&x, &bb, 4);
    const float ref = exp_spec_rint_form(x), got = exp_spec(x), flo = exp_spec_floor(x);
    if (memcmp(&ref, &got, 4) != 0) bad++;
    if (!(x < -87.0f) && memcmp(&ref, &flo, 4) != 0) bad++;
  }
  return bad;
}
float hc_splat_power(float cx, float cy, float cz, float dx, float dy) { return splat_power((cx * dx) * dx, cy * dx, cz, dy); }
float hc_skip_threshold(float op) { return skip_threshold(op); }
int hc_sizeof_acc() { return (int)sizeof(SplatAcc); }
}
