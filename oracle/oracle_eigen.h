// TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/README.md).
//
// Symmetric 3x3 eigen-decomposition: Householder tridiagonalisation followed by
// QL with implicit shifts, restating the arithmetic of
//   DGR/cuda_rasterizer/auxiliary.h:182-401 (glm_modification::{equal,transferSign,
//   pythag,findEigenvaluesSymReal<3>}),
// i.e. the same operation sequence, eps = 1e-7, at most 30 QL sweeps per eigenvalue,
// return value 0 on non-convergence (callers zero the geometry outputs then:
// forward.cu:162-168, backward.cu:262-272).  Written 0-based; A(r,c) is the work
// matrix (row-major), dg/od the diagonal / off-diagonal of the tridiagonal form.
#pragma once
#include "oracle_linalg.h"

namespace orc {

template <class R> inline bool near0(R x) { return std::fabs(x - R(0)) <= R(0.0000001); }

template <class R> inline R sign_of(R v, R s) { return s >= 0 ? std::fabs(v) : -std::fabs(v); }

template <class R> inline R hypot_glm(R a, R b) {
  R aa = std::fabs(a), ab = std::fabs(b);
  if (aa > ab) {
    ab /= aa;
    ab *= ab;
    return aa * std::sqrt(R(1) + ab);
  }
  if (near0(ab)) return R(0);
  aa /= ab;
  aa *= aa;
  return ab * std::sqrt(R(1) + aa);
}

// Returns 3 on success, 0 if QL did not converge.  evec columns are eigenvectors.
template <class R> inline int sym_eigen3(const M3<R>& S, V3<R>& eval, M3<R>& evec) {
  R a[9], dg[3], od[3];
#define A(r, c) a[(r)*3 + (c)]
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) A(r, c) = S[c][r];

  // --- phase 1: Householder reduction (rows 2, then 1) ---
  for (int I = 2; I >= 1; --I) {
    const int L = I;  // number of sub-diagonal entries in row I
    R h = 0, scale = 0;
    if (L > 1) {
      for (int k = 0; k < L; k++) scale += std::fabs(A(I, k));
      if (near0(scale)) {
        od[I] = A(I, L - 1);
      } else {
        for (int k = 0; k < L; k++) {
          A(I, k) /= scale;
          h += A(I, k) * A(I, k);
        }
        R f = A(I, L - 1);
        R g = (f >= 0) ? -std::sqrt(h) : std::sqrt(h);
        od[I] = scale * g;
        h -= f * g;
        A(I, L - 1) = f - g;
        f = 0;
        for (int j = 0; j < L; j++) {
          A(j, I) = A(I, j) / h;
          g = 0;
          for (int k = 0; k <= j; k++) g += A(j, k) * A(I, k);
          for (int k = j + 1; k < L; k++) g += A(k, j) * A(I, k);
          od[j] = g / h;
          f += od[j] * A(I, j);
        }
        const R hh = f / (h + h);
        for (int j = 0; j < L; j++) {
          f = A(I, j);
          od[j] = g = od[j] - hh * f;
          for (int k = 0; k <= j; k++) A(j, k) -= (f * od[k] + g * A(I, k));
        }
      }
    } else {
      od[I] = A(I, L - 1);
    }
    dg[I] = h;
  }
  dg[0] = 0;
  od[0] = 0;
  // --- accumulate the orthogonal transform ---
  for (int I = 0; I < 3; I++) {
    const int L = I;
    if (!near0(dg[I])) {
      for (int j = 0; j < L; j++) {
        R g = 0;
        for (int k = 0; k < L; k++) g += A(I, k) * A(k, j);
        for (int k = 0; k < L; k++) A(k, j) -= g * A(k, I);
      }
    }
    dg[I] = A(I, I);
    A(I, I) = 1;
    for (int j = 0; j < L; j++) A(j, I) = A(I, j) = 0;
  }

  // --- phase 2: QL with implicit shifts ---
  od[0] = od[1];
  od[1] = od[2];
  od[2] = 0;
  for (int l = 0; l < 3; l++) {
    int iter = 0;
    int m;
    do {
      for (m = l; m <= 1; m++) {
        if (near0(std::fabs(od[m]))) break;
      }
      if (m != l) {
        if (iter++ == 30) return 0;
        R g = (dg[l + 1] - dg[l]) / (2 * od[l]);
        R r = hypot_glm<R>(g, R(1));
        g = dg[m] - dg[l] + od[l] / (g + sign_of(r, g));
        R s = 1, c = 1, p = 0;
        int i;
        for (i = m - 1; i >= l; i--) {
          R f = s * od[i];
          const R b = c * od[i];
          od[i + 1] = r = hypot_glm(f, g);
          if (near0(r)) {
            dg[i + 1] -= p;
            od[m] = 0;
            break;
          }
          s = f / r;
          c = g / r;
          g = dg[i + 1] - p;
          r = (dg[i] - g) * s + 2 * c * b;
          dg[i + 1] = g + (p = s * r);
          g = c * r - b;
          for (int k = 0; k < 3; k++) {
            f = A(k, i + 1);
            A(k, i + 1) = s * A(k, i) + c * f;
            A(k, i) = c * A(k, i) - s * f;
          }
        }
        if (near0(r) && i >= l) continue;
        dg[l] -= p;
        od[l] = g;
        od[m] = 0;
      }
    } while (m != l);
  }

  for (int i = 0; i < 3; i++) eval[i] = dg[i];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) evec[i][j] = A(j, i);
#undef A
  return 3;
}

}  // namespace orc
