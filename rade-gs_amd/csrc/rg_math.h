// rg_math.h -- fp32 vector/matrix helpers and the 3x3 symmetric eigen-solver used by the
// per-Gaussian stages (preprocess forward and backward).
//
// All functions are host+device so that tests/hostcheck can run the very same source on the
// CPU and compare it bit-for-bit with the oracle (no GPU in the build container).  Every
// translation unit that includes this header is compiled with -ffp-contract=off: one rounding
// per operation, fma only where spelled out.  Index-determining quantities (radii, tile
// rects, depth keys) therefore come out identical on gfx950 and on the host.
//
// Conventions (what the reference's glm types did; glm is not vendored there, see
// DGR/.gitmodules:1-3): m3 is column-major, m.c[col][row]; products accumulate left to right.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RG_HD __host__ __device__ __forceinline__
#else
#define RG_HD inline
#endif

namespace rg {

struct v3 {
  float x, y, z;
};
struct m3 {
  float c[3][3];  // c[col][row]
};

RG_HD v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
RG_HD v3 add(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RG_HD v3 sub(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RG_HD v3 mul(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
RG_HD v3 mul(float s, v3 a) { return v3{s * a.x, s * a.y, s * a.z}; }
RG_HD v3 div(v3 a, float s) { return v3{a.x / s, a.y / s, a.z / s}; }
RG_HD float dot(v3 a, v3 b) {
  float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
  return tx + ty + tz;
}
RG_HD float len(v3 a) { return sqrtf(dot(a, a)); }
RG_HD v3 normalize(v3 a) {
  float inv = 1.0f / sqrtf(dot(a, a));
  return mul(a, inv);
}
RG_HD float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// columns given in order, like glm::mat3(a,b,c, d,e,f, g,h,i)
RG_HD m3 mk33(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
  m3 m;
  m.c[0][0] = a; m.c[0][1] = b; m.c[0][2] = c;
  m.c[1][0] = d; m.c[1][1] = e; m.c[1][2] = f;
  m.c[2][0] = g; m.c[2][1] = h; m.c[2][2] = i;
  return m;
}
RG_HD m3 zero33() { return mk33(0, 0, 0, 0, 0, 0, 0, 0, 0); }
RG_HD v3 col(const m3& m, int k) { return v3{m.c[k][0], m.c[k][1], m.c[k][2]}; }
RG_HD m3 transpose(const m3& m) {
  return mk33(m.c[0][0], m.c[1][0], m.c[2][0], m.c[0][1], m.c[1][1], m.c[2][1], m.c[0][2], m.c[1][2], m.c[2][2]);
}
RG_HD m3 mul(const m3& A, const m3& B) {
  m3 o;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) o.c[c][r] = A.c[0][r] * B.c[c][0] + A.c[1][r] * B.c[c][1] + A.c[2][r] * B.c[c][2];
  return o;
}
RG_HD v3 mul(const m3& m, v3 v) {
  return v3{m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z, m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z,
            m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z};
}
RG_HD m3 scale(const m3& m, float s) {
  m3 o;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) o.c[c][r] = m.c[c][r] * s;
  return o;
}
RG_HD m3 scale_l(float s, const m3& m) {
  m3 o;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) o.c[c][r] = s * m.c[c][r];
  return o;
}
RG_HD m3 divs(const m3& m, float s) {
  m3 o;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) o.c[c][r] = m.c[c][r] / s;
  return o;
}
RG_HD m3 add(const m3& a, const m3& b) {
  m3 o;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) o.c[c][r] = a.c[c][r] + b.c[c][r];
  return o;
}
RG_HD m3 neg(const m3& a) {
  m3 o;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) o.c[c][r] = -a.c[c][r];
  return o;
}
// outer(colvec, rowvec): column i = colvec * rowvec[i]
RG_HD m3 outer(v3 cv, v3 rv) {
  m3 o;
  o.c[0][0] = cv.x * rv.x; o.c[0][1] = cv.y * rv.x; o.c[0][2] = cv.z * rv.x;
  o.c[1][0] = cv.x * rv.y; o.c[1][1] = cv.y * rv.y; o.c[1][2] = cv.z * rv.y;
  o.c[2][0] = cv.x * rv.z; o.c[2][1] = cv.y * rv.z; o.c[2][2] = cv.z * rv.z;
  return o;
}

// Transforms with the transposed 4x4 storage the callers hand over (auxiliary.h:74-113):
// flat m[0],m[4],m[8],m[12] is row 0 of the math matrix.
RG_HD v3 xform43(v3 p, const float* m) {
  return v3{m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
RG_HD v3 xform43_T(v3 p, const float* m) {
  return v3{m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z, m[8] * p.x + m[9] * p.y + m[10] * p.z};
}

RG_HD int imin(int a, int b) { return a < b ? a : b; }
RG_HD int imax(int a, int b) { return a > b ? a : b; }

// float -> int with v_cvt_i32_f32 semantics on every platform (NaN -> 0, saturating).
RG_HD int f2i_sat(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return -2147483647 - 1;
  return (int)v;
}

// ------------------------------------------------------------------------------------
// Symmetric 3x3 eigen-solver.  Same algorithm and operation order as the routine the
// reference carries in DGR/cuda_rasterizer/auxiliary.h:217-401 (Householder + QL implicit
// shifts, eps 1e-7, <= 30 sweeps, 0 on non-convergence), but specialised for N=3 with every
// array index a compile-time constant so d/e/a live in VGPRs instead of scratch memory.
// ------------------------------------------------------------------------------------
RG_HD bool tiny(float x) { return fabsf(x) <= 0.0000001f; }
RG_HD float sgn_like(float v, float s) { return s >= 0 ? fabsf(v) : -fabsf(v); }
RG_HD float hyp(float a, float b) {
  float aa = fabsf(a), ab = fabsf(b);
  if (aa > ab) {
    ab /= aa;
    ab *= ab;
    return aa * sqrtf(1.0f + ab);
  }
  if (tiny(ab)) return 0.0f;
  aa /= ab;
  aa *= aa;
  return ab * sqrtf(1.0f + aa);
}

struct Eig3 {
  float d[3];  // eigenvalues
  float e[3];
  float a[9];  // row-major work matrix; on exit column i is eigenvector i
};
#define RG_A(r, c) w.a[(r)*3 + (c)]

// One implicit-shift QL sweep for the block [L..M].  Returns false when the sweep hit the
// r ~ 0 early exit (the caller then just re-scans), true when it ran to completion.
template <int L, int M>
RG_HD void ql_sweep(Eig3& w) {
  float g = (w.d[L + 1] - w.d[L]) / (2 * w.e[L]);
  float r = hyp(g, 1.0f);
  g = w.d[M] - w.d[L] + w.e[L] / (g + sgn_like(r, g));
  float s = 1, c = 1, p = 0;
  bool early = false;
#pragma unroll
  for (int i = M - 1; i >= L; i--) {
    if (!early) {
      float f = s * w.e[i];
      const float b = c * w.e[i];
      w.e[i + 1] = r = hyp(f, g);
      if (tiny(r)) {
        w.d[i + 1] -= p;
        w.e[M] = 0;
        early = true;
      } else {
        s = f / r;
        c = g / r;
        g = w.d[i + 1] - p;
        r = (w.d[i] - g) * s + 2 * c * b;
        w.d[i + 1] = g + (p = s * r);
        g = c * r - b;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          f = RG_A(k, i + 1);
          RG_A(k, i + 1) = s * RG_A(k, i) + c * f;
          RG_A(k, i) = c * RG_A(k, i) - s * f;
        }
      }
    }
  }
  if (!early) {
    w.d[L] -= p;
    w.e[L] = g;
    w.e[M] = 0;
  }
}

// S given as its 6 unique entries s00,s01,s02,s11,s12,s22.  Returns 3, or 0 if QL failed.
RG_HD int sym_eigen3(float s00, float s01, float s02, float s11, float s12, float s22, Eig3& w) {
  RG_A(0, 0) = s00; RG_A(0, 1) = s01; RG_A(0, 2) = s02;
  RG_A(1, 0) = s01; RG_A(1, 1) = s11; RG_A(1, 2) = s12;
  RG_A(2, 0) = s02; RG_A(2, 1) = s12; RG_A(2, 2) = s22;

  // ---- Householder, row 2 (two sub-diagonal entries) ----
  {
    float h = 0, sc = 0;
    sc += fabsf(RG_A(2, 0));
    sc += fabsf(RG_A(2, 1));
    if (tiny(sc)) {
      w.e[2] = RG_A(2, 1);
    } else {
      RG_A(2, 0) /= sc; h += RG_A(2, 0) * RG_A(2, 0);
      RG_A(2, 1) /= sc; h += RG_A(2, 1) * RG_A(2, 1);
      float f = RG_A(2, 1);
      float g = (f >= 0) ? -sqrtf(h) : sqrtf(h);
      w.e[2] = sc * g;
      h -= f * g;
      RG_A(2, 1) = f - g;
      f = 0;
      // j = 0
      RG_A(0, 2) = RG_A(2, 0) / h;
      g = 0;
      g += RG_A(0, 0) * RG_A(2, 0);
      g += RG_A(1, 0) * RG_A(2, 1);
      w.e[0] = g / h;
      f += w.e[0] * RG_A(2, 0);
      // j = 1
      RG_A(1, 2) = RG_A(2, 1) / h;
      g = 0;
      g += RG_A(1, 0) * RG_A(2, 0);
      g += RG_A(1, 1) * RG_A(2, 1);
      w.e[1] = g / h;
      f += w.e[1] * RG_A(2, 1);
      const float hh = f / (h + h);
      // j = 0
      f = RG_A(2, 0);
      w.e[0] = g = w.e[0] - hh * f;
      RG_A(0, 0) -= (f * w.e[0] + g * RG_A(2, 0));
      // j = 1
      f = RG_A(2, 1);
      w.e[1] = g = w.e[1] - hh * f;
      RG_A(1, 0) -= (f * w.e[0] + g * RG_A(2, 0));
      RG_A(1, 1) -= (f * w.e[1] + g * RG_A(2, 1));
    }
    w.d[2] = h;
  }
  // ---- row 1 (single sub-diagonal entry: nothing to reduce) ----
  w.e[1] = RG_A(1, 0);
  w.d[1] = 0;
  w.d[0] = 0;
  w.e[0] = 0;
  // ---- accumulate transform: I = 0 ----
  w.d[0] = RG_A(0, 0);
  RG_A(0, 0) = 1;
  // I = 1 (L = 1)
  if (!tiny(w.d[1])) {
    float g = 0;
    g += RG_A(1, 0) * RG_A(0, 0);
    RG_A(0, 0) -= g * RG_A(0, 1);
  }
  w.d[1] = RG_A(1, 1);
  RG_A(1, 1) = 1;
  RG_A(0, 1) = RG_A(1, 0) = 0;
  // I = 2 (L = 2)
  if (!tiny(w.d[2])) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      float g = 0;
      g += RG_A(2, 0) * RG_A(0, j);
      g += RG_A(2, 1) * RG_A(1, j);
      RG_A(0, j) -= g * RG_A(0, 2);
      RG_A(1, j) -= g * RG_A(1, 2);
    }
  }
  w.d[2] = RG_A(2, 2);
  RG_A(2, 2) = 1;
  RG_A(0, 2) = RG_A(2, 0) = 0;
  RG_A(1, 2) = RG_A(2, 1) = 0;

  // ---- QL ----
  w.e[0] = w.e[1];
  w.e[1] = w.e[2];
  w.e[2] = 0;
  // l = 0
  for (int iter = 0;;) {
    int m;
    if (tiny(fabsf(w.e[0]))) m = 0;
    else if (tiny(fabsf(w.e[1]))) m = 1;
    else m = 2;
    if (m == 0) break;
    if (iter++ == 30) return 0;
    if (m == 1) ql_sweep<0, 1>(w);
    else ql_sweep<0, 2>(w);
  }
  // l = 1
  for (int iter = 0;;) {
    int m = tiny(fabsf(w.e[1])) ? 1 : 2;
    if (m == 1) break;
    if (iter++ == 30) return 0;
    ql_sweep<1, 2>(w);
  }
  // l = 2: nothing to do
  return 3;
}
#undef RG_A

RG_HD v3 eig_vec(const Eig3& w, int i) { return v3{w.a[0 * 3 + i], w.a[1 * 3 + i], w.a[2 * 3 + i]}; }

}  // namespace rg
