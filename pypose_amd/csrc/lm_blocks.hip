// lm_blocks.hip -- per-problem normal equations and damped Cholesky solves for the
// block-structured Levenberg-Marquardt / Gauss-Newton paths (include/pplie.h, "LM blocks").
//
// Reference: pypose/optim/optimizer.py:655-668 builds ONE dense A = J^T W J of size
// [sum N_param]^2 and factorises it with LAPACK (solver.py:213-216).  When the Jacobian is
// block diagonal (B independent problems: J_b is d_res x d_par with d_* <= 8) that is B tiny
// SPD systems; here each lane owns one problem, forms A_b and g_b in registers and solves it
// with a fully unrolled Cholesky.  Same row-map shell as the Lie ops: slabs of J / r / W are
// streamed HBM -> LDS -> registers with dwordx4 accesses.
#include "rowmap.h"

namespace pplie {

// A = J^T W J (DPxDP, row-major), g = J^T W r   (W optional, not assumed symmetric:
// the reference computes J_T = J.T @ weight, A = J_T @ J, b = -J_T @ R; optimizer.py:655-668)
template <class T, int DR, int DP, bool HAS_W> struct Op_normal_eq {
  enum { IW0 = DR * DP, IW1 = DR, IW2 = HAS_W ? DR * DR : 0, OW0 = DP * DP, OW1 = DP };
  static PP_HD void apply(const T* J, const T* r, const T* W, T* A, T* g) {
    T JtW[DP * DR];   // (J^T W)[p][k] = sum_i J[i][p] W[i][k]
#pragma unroll
    for (int p = 0; p < DP; ++p)
#pragma unroll
      for (int k = 0; k < DR; ++k) {
        if (HAS_W) {
          T acc = T(0);
#pragma unroll
          for (int i = 0; i < DR; ++i) acc += J[i * DP + p] * W[i * DR + k];
          JtW[p * DR + k] = acc;
        } else {
          JtW[p * DR + k] = J[k * DP + p];
        }
      }
#pragma unroll
    for (int p = 0; p < DP; ++p) {
#pragma unroll
      for (int q = 0; q < DP; ++q) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < DR; ++k) acc += JtW[p * DR + k] * J[k * DP + q];
        A[p * DP + q] = acc;
      }
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < DR; ++k) acc += JtW[p * DR + k] * r[k];
      g[p] = acc;
    }
  }
};

}  // namespace pplie
#include "chol.h"
namespace pplie {
// Ainv = A^-1 for SPD blocks (block-Jacobi preconditioner of the pose-graph PCG): Cholesky, then
// the columns of the identity are solved one by one.
template <class T, int DP> struct Op_spd_inverse {
  enum { IW0 = DP * DP, IW1 = 0, IW2 = 0, OW0 = DP * DP, OW1 = 0 };
  static PP_HD void apply(const T* A, const T*, const T*, T* X, T*) { Op_spd_inverse_apply<T, DP>(A, X); }
};

// 64-lane workgroups: the per-problem slabs are wide (up to 8x8 + 8x8 + 8 + 64 scalars)
template <class T, int DR, int DP> int normal_eq_launch(const void* J, const void* r, const void* W, void* A, void* g,
                                                        int64_t n, void* stream) {
  if (W) return launch_rowmap<T, Op_normal_eq<T, DR, DP, true>, 1, 64>(J, r, W, A, g, n, stream);
  return launch_rowmap<T, Op_normal_eq<T, DR, DP, false>, 1, 64>(J, r, nullptr, A, g, n, stream);
}

template <class T, int DR> int normal_eq_dp(int dp, const void* J, const void* r, const void* W, void* A, void* g,
                                            int64_t n, void* stream) {
  switch (dp) {
    case 3: return normal_eq_launch<T, DR, 3>(J, r, W, A, g, n, stream);
    case 4: return normal_eq_launch<T, DR, 4>(J, r, W, A, g, n, stream);
    case 5: return normal_eq_launch<T, DR, 5>(J, r, W, A, g, n, stream);
    case 6: return normal_eq_launch<T, DR, 6>(J, r, W, A, g, n, stream);
    case 7: return normal_eq_launch<T, DR, 7>(J, r, W, A, g, n, stream);
    case 8: return normal_eq_launch<T, DR, 8>(J, r, W, A, g, n, stream);
  }
  return PPLIE_EBADARG;
}

template <class T> int normal_eq_dispatch(int dr, int dp, const void* J, const void* r, const void* W, void* A, void* g,
                                          int64_t n, void* stream) {
  switch (dr) {
    case 3: return normal_eq_dp<T, 3>(dp, J, r, W, A, g, n, stream);
    case 4: return normal_eq_dp<T, 4>(dp, J, r, W, A, g, n, stream);
    case 6: return normal_eq_dp<T, 6>(dp, J, r, W, A, g, n, stream);
    case 7: return normal_eq_dp<T, 7>(dp, J, r, W, A, g, n, stream);
  }
  return PPLIE_EBADARG;
}

template <class T> int chol_dispatch(int dp, const void* A, const void* g, void* x, int64_t n, void* stream) {
  switch (dp) {
    case 3: return launch_rowmap<T, Op_chol_solve<T, 3>, 1, 64>(A, g, nullptr, x, nullptr, n, stream);
    case 4: return launch_rowmap<T, Op_chol_solve<T, 4>, 1, 64>(A, g, nullptr, x, nullptr, n, stream);
    case 5: return launch_rowmap<T, Op_chol_solve<T, 5>, 1, 64>(A, g, nullptr, x, nullptr, n, stream);
    case 6: return launch_rowmap<T, Op_chol_solve<T, 6>, 1, 64>(A, g, nullptr, x, nullptr, n, stream);
    case 7: return launch_rowmap<T, Op_chol_solve<T, 7>, 1, 64>(A, g, nullptr, x, nullptr, n, stream);
    case 8: return launch_rowmap<T, Op_chol_solve<T, 8>, 1, 64>(A, g, nullptr, x, nullptr, n, stream);
  }
  return PPLIE_EBADARG;
}

// x = (A with its diagonal clamped to [dmin, dmax] and scaled by s)^-1 (-g): the LM trial's damped solve on the RAW normal
// equations (optimizer.py:655-657, 666: clamp once, `A.diag += A.diag * damping` on every retry -> s = prod (1 + damping_i)).
// The diagonal is adjusted in registers; A is not written, so a retry is the same launch with a larger s -- three strided
// elementwise passes over the [n, dp, dp] blocks per trial gone.
template <class T> struct DampParam { T s, dmin, dmax; };
template <class T, int DP> struct Op_damped_chol_solve {
  enum { IW0 = DP * DP, IW1 = DP, IW2 = 0, OW0 = DP, OW1 = 0 };
  static PP_HD void apply(const T* A, const T* g, const T*, T* x, T*, DampParam<T> prm) {
    T Ad[DP * DP];
#pragma unroll
    for (int i = 0; i < DP * DP; ++i) Ad[i] = A[i];
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const T d0 = A[j * DP + j];
      // torch.clamp(min, max) semantics (NaN stays NaN)
      const T d = d0 < prm.dmin ? prm.dmin : (d0 > prm.dmax ? prm.dmax : d0);
      Ad[j * DP + j] = d * prm.s;
    }
    Op_chol_solve<T, DP>::apply(Ad, g, nullptr, x, nullptr);
  }
};
template <class T> int damped_chol_dispatch(int dp, const void* A, const void* g, void* x, int64_t n, double s, double dmin,
                                            double dmax, void* stream) {
  const DampParam<T> prm{(T)s, (T)dmin, (T)dmax};
#define PPLIE_DCS(DPN)                                                                                                 \
  case DPN:                                                                                                            \
    return launch_rowmap<T, Op_damped_chol_solve<T, DPN>, 1, 64, false, DampParam<T>>(A, g, nullptr, x, nullptr, n, stream, \
                                                                                      kGridCap, prm);
  switch (dp) {
    PPLIE_DCS(3) PPLIE_DCS(4) PPLIE_DCS(5) PPLIE_DCS(6) PPLIE_DCS(7) PPLIE_DCS(8)
  }
#undef PPLIE_DCS
  return PPLIE_EBADARG;
}

template <class T> int spd_inverse_dispatch(int dp, const void* A, void* X, int64_t n, void* stream) {
  switch (dp) {
    case 3: return launch_rowmap<T, Op_spd_inverse<T, 3>, 1, 64>(A, nullptr, nullptr, X, nullptr, n, stream);
    case 6: return launch_rowmap<T, Op_spd_inverse<T, 6>, 1, 64>(A, nullptr, nullptr, X, nullptr, n, stream);
    case 7: return launch_rowmap<T, Op_spd_inverse<T, 7>, 1, 64>(A, nullptr, nullptr, X, nullptr, n, stream);
  }
  return PPLIE_EBADARG;
}
}  // namespace pplie

extern "C" int pplie_block_spd_inverse_f32(const void* A, void* X, int64_t n, int dp, void* stream) {
  return pplie::spd_inverse_dispatch<float>(dp, A, X, n, stream);
}
extern "C" int pplie_block_spd_inverse_f64(const void* A, void* X, int64_t n, int dp, void* stream) {
  return pplie::spd_inverse_dispatch<double>(dp, A, X, n, stream);
}
extern "C" int pplie_block_normal_eq_f32(const void* J, const void* r, const void* W, void* A, void* g, int64_t n, int dr,
                                         int dp, void* stream) {
  return pplie::normal_eq_dispatch<float>(dr, dp, J, r, W, A, g, n, stream);
}
extern "C" int pplie_block_normal_eq_f64(const void* J, const void* r, const void* W, void* A, void* g, int64_t n, int dr,
                                         int dp, void* stream) {
  return pplie::normal_eq_dispatch<double>(dr, dp, J, r, W, A, g, n, stream);
}
extern "C" int pplie_block_chol_solve_f32(const void* A, const void* g, void* x, int64_t n, int dp, void* stream) {
  return pplie::chol_dispatch<float>(dp, A, g, x, n, stream);
}
extern "C" int pplie_block_chol_solve_f64(const void* A, const void* g, void* x, int64_t n, int dp, void* stream) {
  return pplie::chol_dispatch<double>(dp, A, g, x, n, stream);
}
extern "C" int pplie_block_damped_chol_solve_f32(const void* A, const void* g, void* x, int64_t n, int dp, double s, double dmin,
                                                 double dmax, void* stream) {
  return pplie::damped_chol_dispatch<float>(dp, A, g, x, n, s, dmin, dmax, stream);
}
extern "C" int pplie_block_damped_chol_solve_f64(const void* A, const void* g, void* x, int64_t n, int dp, double s, double dmin,
                                                 double dmax, void* stream) {
  return pplie::damped_chol_dispatch<double>(dp, A, g, x, n, s, dmin, dmax, stream);
}
