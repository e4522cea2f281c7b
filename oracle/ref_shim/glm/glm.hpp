// TEST INFRASTRUCTURE (oracle/_ref): the subset of GLM the reference's rasterizer uses.
//
// The reference depends on g-truc/glm as an UN-VENDORED git submodule (DGR/.gitmodules:1-3; third_party/glm/ is empty in the
// snapshot, commit unpinned) and no copy exists in this image.  This header restates the published semantics of the handful of
// GLM value types and functions the reference touches (call sites: forward.cu:23-304, backward.cu:21-555, auxiliary.h:182-401),
// operation for operation as GLM 0.9.9's scalar (non-SIMD) code path spells them, because the ORDER of the fp32 operations
// matters for bit-level comparisons:
//   * matrices are COLUMN-major: mat3(a,b,c, d,e,f, g,h,i) has columns (a,b,c),(d,e,f),(g,h,i); m[c][r]   (type_mat3x3.inl)
//   * mat3*vec3:   r.x = m[0][0]*v.x + m[1][0]*v.y + m[2][0]*v.z   (left to right)                          (type_mat3x3.inl)
//   * mat3*mat3:   R[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2]                            (type_mat3x3.inl)
//   * dot(a,b):    tmp = a*b;  tmp.x + tmp.y + tmp.z                                                         (detail/func_geometric.inl)
//   * length(v) = sqrt(dot(v,v));  normalize(v) = v * inversesqrt(dot(v,v)),  inversesqrt(x) = 1/sqrt(x)     (func_geometric.inl, func_exponential.inl)
//   * outerProduct(c, r): column i = c * r[i]                                                                (func_matrix.inl)
//   * vec / scalar divides every component (no reciprocal);  max(x, y) = (x < y) ? y : x                     (type_vec3.inl, func_common.inl)
// One difference from GLM: a default-constructed mat3 is ZERO here (GLM leaves it uninitialised).  The only place the reference reads
// such a matrix is the shadowed `inv_cov_ray` of the ill-conditioned INTE branch (forward.cu:196 vs :223), i.e. undefined upstream.
// The reference's own known-answer comment (forward.cu:126-133: mat3(1..9)*(1,1,1) = (12,15,18)) is checked against this header
// by tests/test_ref_parity.py::test_mark_visible_and_msb_and_matrix_convention.
#pragma once
#include <cmath>

namespace glm {
typedef int length_t;
enum qualifier { packed_highp, packed_mediump, packed_lowp, highp = packed_highp, mediump = packed_mediump, lowp = packed_lowp, defaultp = highp };

template <length_t L, typename T, qualifier Q = defaultp> struct vec;
template <length_t C, length_t R, typename T, qualifier Q = defaultp> struct mat;

template <typename T, qualifier Q> struct vec<2, T, Q> {
  T x, y;
  vec() = default;
  template <typename A, typename B> vec(A a, B b) : x(static_cast<T>(a)), y(static_cast<T>(b)) {}
  explicit vec(T s) : x(s), y(s) {}
  T& operator[](length_t i) { return (&x)[i]; }
  T const& operator[](length_t i) const { return (&x)[i]; }
};

template <typename T, qualifier Q> struct vec<3, T, Q> {
  T x, y, z;
  vec() = default;
  template <typename A, typename B, typename C> vec(A a, B b, C c) : x(static_cast<T>(a)), y(static_cast<T>(b)), z(static_cast<T>(c)) {}
  explicit vec(T s) : x(s), y(s), z(s) {}
  T& operator[](length_t i) { return (&x)[i]; }
  T const& operator[](length_t i) const { return (&x)[i]; }
  vec& operator+=(vec const& v) { x += v.x; y += v.y; z += v.z; return *this; }
  vec& operator-=(vec const& v) { x -= v.x; y -= v.y; z -= v.z; return *this; }
  vec& operator+=(T s) { x += s; y += s; z += s; return *this; }
  vec& operator*=(T s) { x *= s; y *= s; z *= s; return *this; }
  vec& operator/=(T s) { x /= s; y /= s; z /= s; return *this; }
};

template <typename T, qualifier Q> struct vec<4, T, Q> {
  T x, y, z, w;
  vec() = default;
  template <typename A, typename B, typename C, typename D>
  vec(A a, B b, C c, D d) : x(static_cast<T>(a)), y(static_cast<T>(b)), z(static_cast<T>(c)), w(static_cast<T>(d)) {}
  explicit vec(T s) : x(s), y(s), z(s), w(s) {}
  T& operator[](length_t i) { return (&x)[i]; }
  T const& operator[](length_t i) const { return (&x)[i]; }
};

typedef vec<2, float, defaultp> vec2;
typedef vec<3, float, defaultp> vec3;
typedef vec<4, float, defaultp> vec4;

// ---- vec3 arithmetic (component-wise, as type_vec3.inl)
template <typename T, qualifier Q> inline vec<3, T, Q> operator+(vec<3, T, Q> const& a, vec<3, T, Q> const& b) { return vec<3, T, Q>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T, qualifier Q> inline vec<3, T, Q> operator-(vec<3, T, Q> const& a, vec<3, T, Q> const& b) { return vec<3, T, Q>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T, qualifier Q> inline vec<3, T, Q> operator*(vec<3, T, Q> const& a, vec<3, T, Q> const& b) { return vec<3, T, Q>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename T, qualifier Q> inline vec<3, T, Q> operator-(vec<3, T, Q> const& a) { return vec<3, T, Q>(-a.x, -a.y, -a.z); }
template <typename T, qualifier Q> inline vec<3, T, Q> operator*(vec<3, T, Q> const& a, T s) { return vec<3, T, Q>(a.x * s, a.y * s, a.z * s); }
template <typename T, qualifier Q> inline vec<3, T, Q> operator*(T s, vec<3, T, Q> const& a) { return vec<3, T, Q>(s * a.x, s * a.y, s * a.z); }
template <typename T, qualifier Q> inline vec<3, T, Q> operator/(vec<3, T, Q> const& a, T s) { return vec<3, T, Q>(a.x / s, a.y / s, a.z / s); }
template <typename T, qualifier Q> inline vec<3, T, Q> operator+(vec<3, T, Q> const& a, T s) { return vec<3, T, Q>(a.x + s, a.y + s, a.z + s); }
// int scalars (the reference writes `2 * (-tmp) * v` with float operands, but also mat3 / vec3 built from int literals)
template <qualifier Q> inline vec<3, float, Q> operator*(int s, vec<3, float, Q> const& a) { return static_cast<float>(s) * a; }
template <qualifier Q> inline vec<3, float, Q> operator*(vec<3, float, Q> const& a, int s) { return a * static_cast<float>(s); }
template <qualifier Q> inline vec<3, float, Q> operator*(double s, vec<3, float, Q> const& a) { return static_cast<float>(s) * a; }

template <typename T> inline T abs(T x) { return x >= T(0) ? x : -x; }
template <typename T> inline T sqrt(T x) { return std::sqrt(x); }
template <typename T> inline T max(T x, T y) { return (x < y) ? y : x; }
template <typename T> inline T min(T x, T y) { return (y < x) ? y : x; }
template <typename T, qualifier Q> inline vec<3, T, Q> max(vec<3, T, Q> const& a, T s) { return vec<3, T, Q>(max(a.x, s), max(a.y, s), max(a.z, s)); }
template <typename T, qualifier Q> inline vec<3, T, Q> abs(vec<3, T, Q> const& a) { return vec<3, T, Q>(abs(a.x), abs(a.y), abs(a.z)); }

template <typename T, qualifier Q> inline T dot(vec<3, T, Q> const& a, vec<3, T, Q> const& b) { vec<3, T, Q> t(a * b); return t.x + t.y + t.z; }
template <typename T, qualifier Q> inline T dot(vec<2, T, Q> const& a, vec<2, T, Q> const& b) { return a.x * b.x + a.y * b.y; }
template <typename T, qualifier Q> inline T dot(vec<4, T, Q> const& a, vec<4, T, Q> const& b) { T t0 = a.x * b.x, t1 = a.y * b.y, t2 = a.z * b.z, t3 = a.w * b.w; return (t0 + t1) + (t2 + t3); }
template <typename T, qualifier Q> inline T length(vec<3, T, Q> const& v) { return std::sqrt(dot(v, v)); }
template <typename T, qualifier Q> inline T length(vec<4, T, Q> const& v) { return std::sqrt(dot(v, v)); }
template <typename T> inline T inversesqrt(T x) { return static_cast<T>(1) / std::sqrt(x); }
template <typename T, qualifier Q> inline vec<3, T, Q> normalize(vec<3, T, Q> const& v) { return v * inversesqrt(dot(v, v)); }

// ---- mat3 (column-major)
template <typename T, qualifier Q> struct mat<3, 3, T, Q> {
  typedef vec<3, T, Q> col_type;
  col_type value[3];
  mat() : value{col_type(0, 0, 0), col_type(0, 0, 0), col_type(0, 0, 0)} {}
  explicit mat(T s) : value{col_type(s, 0, 0), col_type(0, s, 0), col_type(0, 0, s)} {}
  template <typename X1, typename Y1, typename Z1, typename X2, typename Y2, typename Z2, typename X3, typename Y3, typename Z3>
  mat(X1 x1, Y1 y1, Z1 z1, X2 x2, Y2 y2, Z2 z2, X3 x3, Y3 y3, Z3 z3)
      : value{col_type(x1, y1, z1), col_type(x2, y2, z2), col_type(x3, y3, z3)} {}
  mat(col_type const& a, col_type const& b, col_type const& c) : value{a, b, c} {}
  col_type& operator[](length_t i) { return value[i]; }
  col_type const& operator[](length_t i) const { return value[i]; }
  mat& operator+=(mat const& m) { value[0] += m[0]; value[1] += m[1]; value[2] += m[2]; return *this; }
};
typedef mat<3, 3, float, defaultp> mat3;

template <typename T, qualifier Q> inline mat<3, 3, T, Q> transpose(mat<3, 3, T, Q> const& m) {
  mat<3, 3, T, Q> r;
  r[0][0] = m[0][0]; r[0][1] = m[1][0]; r[0][2] = m[2][0];
  r[1][0] = m[0][1]; r[1][1] = m[1][1]; r[1][2] = m[2][1];
  r[2][0] = m[0][2]; r[2][1] = m[1][2]; r[2][2] = m[2][2];
  return r;
}
template <typename T, qualifier Q> inline vec<3, T, Q> operator*(mat<3, 3, T, Q> const& m, vec<3, T, Q> const& v) {
  return vec<3, T, Q>(m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z,
                      m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
                      m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z);
}
template <typename T, qualifier Q> inline mat<3, 3, T, Q> operator*(mat<3, 3, T, Q> const& m1, mat<3, 3, T, Q> const& m2) {
  T const A00 = m1[0][0], A01 = m1[0][1], A02 = m1[0][2], A10 = m1[1][0], A11 = m1[1][1], A12 = m1[1][2], A20 = m1[2][0], A21 = m1[2][1], A22 = m1[2][2];
  T const B00 = m2[0][0], B01 = m2[0][1], B02 = m2[0][2], B10 = m2[1][0], B11 = m2[1][1], B12 = m2[1][2], B20 = m2[2][0], B21 = m2[2][1], B22 = m2[2][2];
  mat<3, 3, T, Q> r;
  r[0][0] = A00 * B00 + A10 * B01 + A20 * B02;
  r[0][1] = A01 * B00 + A11 * B01 + A21 * B02;
  r[0][2] = A02 * B00 + A12 * B01 + A22 * B02;
  r[1][0] = A00 * B10 + A10 * B11 + A20 * B12;
  r[1][1] = A01 * B10 + A11 * B11 + A21 * B12;
  r[1][2] = A02 * B10 + A12 * B11 + A22 * B12;
  r[2][0] = A00 * B20 + A10 * B21 + A20 * B22;
  r[2][1] = A01 * B20 + A11 * B21 + A21 * B22;
  r[2][2] = A02 * B20 + A12 * B21 + A22 * B22;
  return r;
}
template <typename T, qualifier Q> inline mat<3, 3, T, Q> operator*(mat<3, 3, T, Q> const& m, T s) { return mat<3, 3, T, Q>(m[0] * s, m[1] * s, m[2] * s); }
template <typename T, qualifier Q> inline mat<3, 3, T, Q> operator*(T s, mat<3, 3, T, Q> const& m) { return mat<3, 3, T, Q>(m[0] * s, m[1] * s, m[2] * s); }
template <typename T, qualifier Q> inline mat<3, 3, T, Q> operator/(mat<3, 3, T, Q> const& m, T s) { return mat<3, 3, T, Q>(m[0] / s, m[1] / s, m[2] / s); }
template <typename T, qualifier Q> inline mat<3, 3, T, Q> operator+(mat<3, 3, T, Q> const& a, mat<3, 3, T, Q> const& b) { return mat<3, 3, T, Q>(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
template <typename T, qualifier Q> inline mat<3, 3, T, Q> operator-(mat<3, 3, T, Q> const& a, mat<3, 3, T, Q> const& b) { return mat<3, 3, T, Q>(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
template <typename T, qualifier Q> inline mat<3, 3, T, Q> operator-(mat<3, 3, T, Q> const& a) { return mat<3, 3, T, Q>(-a[0], -a[1], -a[2]); }
template <typename T, qualifier Q> inline mat<3, 3, T, Q> outerProduct(vec<3, T, Q> const& c, vec<3, T, Q> const& r) {
  mat<3, 3, T, Q> m;
  for (length_t i = 0; i < 3; ++i) m[i] = c * r[i];
  return m;
}
}  // namespace glm
