// chol.h -- fully unrolled Cholesky solve of a DP x DP SPD system held in registers.
#pragma once
#include "lie_math.h"

namespace pplie {

// x = A^-1 (-g) by Cholesky A = L L^T (lower).  A non-positive pivot yields NaNs in x, which
// the host turns into the reference's "Cholesky decomposition failed" error (solver.py:214).
template <class T, int DP> struct Op_chol_solve {
  enum { IW0 = DP * DP, IW1 = DP, IW2 = 0, OW0 = DP, OW1 = 0 };
  static PP_HD void apply(const T* A, const T* g, const T*, T* x, T*) {
    // only 1 / L_jj is ever needed (the diagonal divides every entry below it and both triangular solves): one
    // reciprocal square root per column instead of a square root and 1 + 2 divisions -- a third of the instructions
    T L[DP * DP], inv[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      T d = A[j * DP + j];
#pragma unroll
      for (int k = 0; k < j; ++k) d -= L[j * DP + k] * L[j * DP + k];
      inv[j] = pp_rsqrt(d);        // d < 0 -> NaN, d = 0 -> inf: propagates to x
#pragma unroll
      for (int i = j + 1; i < DP; ++i) {
        T s = A[i * DP + j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[i * DP + k] * L[j * DP + k];
        L[i * DP + j] = s * inv[j];
      }
    }
    T y[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) {   // L y = -g
      T s = -g[i];
#pragma unroll
      for (int k = 0; k < i; ++k) s -= L[i * DP + k] * y[k];
      y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = DP - 1; i >= 0; --i) {   // L^T x = y
      T s = y[i];
#pragma unroll
      for (int k = i + 1; k < DP; ++k) s -= L[k * DP + i] * x[k];
      x[i] = s * inv[i];
    }
  }
};


// X = A^-1 for an SPD block A (Cholesky, then the columns of the identity one by one)
template <class T, int DP> PP_HD void Op_spd_inverse_apply(const T* A, T* X) {
#pragma unroll
  for (int c = 0; c < DP; ++c) {
    T g[DP], x[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) g[i] = (i == c) ? T(-1) : T(0);
    Op_chol_solve<T, DP>::apply(A, g, nullptr, x, nullptr);
#pragma unroll
    for (int i = 0; i < DP; ++i) X[i * DP + c] = x[i];
  }
}
template <class T> PP_HD void Op_chol6_solve(const T* A, const T* g, T* x) { Op_chol_solve<T, 6>::apply(A, g, nullptr, x, nullptr); }

}  // namespace pplie
