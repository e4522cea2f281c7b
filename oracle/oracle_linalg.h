// TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/README.md).
// Nothing under oracle/ may be imported, linked or executed by the product path.
//
// Tiny vec3 / mat3 value types with the semantics of the glm types the reference
// uses (glm is an un-vendored submodule of the reference: DGR/.gitmodules:1-3, so its
// semantics are restated here instead of included):
//   * mat3 is COLUMN-major: M3(a,b,c,d,e,f,g,h,i) has columns (a,b,c),(d,e,f),(g,h,i)
//     and m.c[col][row]                         (KAT: forward.cu:126-133 comment,
//     mat3(1..9)*(1,1,1) = (12,15,18) -- checked in tests/test_oracle_kat.py)
//   * products accumulate left to right, one rounding per operation
//     (build with -ffp-contract=off).
#pragma once
#include <cmath>

namespace orc {

template <class R> struct V2 { R x, y; };

template <class R> struct V3 {
  R x, y, z;
  R& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const R& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

template <class R> inline V3<R> operator+(const V3<R>& a, const V3<R>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> inline V3<R> operator-(const V3<R>& a, const V3<R>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> inline V3<R> operator*(const V3<R>& a, R s) { return {a.x * s, a.y * s, a.z * s}; }
template <class R> inline V3<R> operator*(R s, const V3<R>& a) { return {s * a.x, s * a.y, s * a.z}; }
template <class R> inline V3<R> operator/(const V3<R>& a, R s) { return {a.x / s, a.y / s, a.z / s}; }

// glm::dot for vec3: tmp = a*b; tmp.x + tmp.y + tmp.z
template <class R> inline R dot(const V3<R>& a, const V3<R>& b) {
  R tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
  return tx + ty + tz;
}
template <class R> inline R length(const V3<R>& a) { return std::sqrt(dot(a, a)); }
// glm::normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
template <class R> inline V3<R> normalize(const V3<R>& a) {
  R inv = R(1) / std::sqrt(dot(a, a));
  return a * inv;
}

template <class R> struct M3 {
  V3<R> c[3];  // columns
  M3() : c{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}} {}
  M3(R a, R b, R cc, R d, R e, R f, R g, R h, R i) : c{{a, b, cc}, {d, e, f}, {g, h, i}} {}
  V3<R>& operator[](int col) { return c[col]; }
  const V3<R>& operator[](int col) const { return c[col]; }
};

template <class R> inline M3<R> transpose(const M3<R>& m) {
  return M3<R>(m[0][0], m[1][0], m[2][0], m[0][1], m[1][1], m[2][1], m[0][2], m[1][2], m[2][2]);
}

// glm mat3*mat3: Result[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2]
template <class R> inline M3<R> operator*(const M3<R>& A, const M3<R>& B) {
  M3<R> out;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++)
      out[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2];
  return out;
}

// glm mat3*vec3: (m[0][r]*v.x + m[1][r]*v.y + m[2][r]*v.z)
template <class R> inline V3<R> operator*(const M3<R>& m, const V3<R>& v) {
  return {m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z,
          m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
          m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z};
}

template <class R> inline M3<R> operator*(const M3<R>& m, R s) {
  M3<R> o;
  for (int c = 0; c < 3; c++) o[c] = m[c] * s;
  return o;
}
template <class R> inline M3<R> operator*(R s, const M3<R>& m) {
  M3<R> o;
  for (int c = 0; c < 3; c++) o[c] = s * m[c];
  return o;
}
template <class R> inline M3<R> operator/(const M3<R>& m, R s) {
  M3<R> o;
  for (int c = 0; c < 3; c++) o[c] = m[c] / s;
  return o;
}
template <class R> inline M3<R> operator+(const M3<R>& a, const M3<R>& b) {
  M3<R> o;
  for (int c = 0; c < 3; c++) o[c] = a[c] + b[c];
  return o;
}
template <class R> inline M3<R> operator-(const M3<R>& a) {
  M3<R> o;
  for (int c = 0; c < 3; c++) o[c] = {-a[c].x, -a[c].y, -a[c].z};
  return o;
}
// glm::outerProduct(c, r): m[i] = c * r[i]
template <class R> inline M3<R> outer(const V3<R>& col, const V3<R>& row) {
  M3<R> o;
  for (int i = 0; i < 3; i++) o[i] = col * row[i];
  return o;
}

}  // namespace orc
