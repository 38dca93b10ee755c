// robust.h -- the reference's robust kernels rho(x) on x = |r|^2 (pypose/optim/kernel.py:37-297) and their first derivatives in
// closed form, as a launch-wide (kind, p0, p1) parameter of the linearisation kernels.
//
// The reference's correctors (pypose/optim/corrector.py:69-96 FastTriggs, :132-167 Triggs) obtain rho' by autograd
// (torch.autograd.functional.jacobian of sum(rho(x)), one graph per corrected step) and then scale the residual rows and their
// Jacobian rows by sqrt(rho') with element-wise tensor ops.  Every built-in kernel is concave (rho'' <= 0 everywhere), so the
// row mask of Triggs' second-order term (corrector.py:160: rows with x != 0 and rho'' > 0) is empty for all of them and Triggs
// reduces to the same sqrt(rho') scaling; user-defined kernels keep the autograd route (optim/corrector.py).
#pragma once
#include "lie_math.h"

namespace pplie {

enum RobustKind { RK_NONE = 0, RK_HUBER = 1, RK_PSEUDOHUBER = 2, RK_CAUCHY = 3, RK_SOFTLONE = 4, RK_ARCTAN = 5, RK_SCALE = 6,
                  RK_TOLERANT = 7 };

// p0 = delta (Tolerant: a), p1 = unused (Tolerant: b)
template <class T> struct RobustParam { int kind; T p0, p1; };

// rho(x)  (kernel.py:48-53, 93-94, 133-134, 174-175, 213-214, 255-258, 297)
// (no FMA contraction in these two: several kernels are differences of nearly equal numbers -- 2 (d sqrt(1/d^2 + x) - 1) at small
//  x -- and fma(d, root, -1) keeps product digits the reference's separate multiply and subtract round away: a converged loss
//  the reference reports as exactly 0 came out as 5e-16)
template <class T> PP_HD T robust_rho(const RobustParam<T>& k, T x) {
#pragma clang fp contract(off)
  const T d = k.p0, d2 = d * d;
  switch (k.kind) {
    case RK_HUBER: { const T root = pp_sqrt(x); return root < d ? x : T(2) * d * root - d2; }
    case RK_PSEUDOHUBER: return T(2) * d2 * (pp_sqrt(x / d2 + T(1)) - T(1));
    case RK_CAUCHY: return d2 * pp_log(x / d2 + T(1));
    case RK_SOFTLONE: return T(2) * (d * pp_sqrt(T(1) / d2 + x) - T(1));
    case RK_ARCTAN: return d2 * pp_atan(x / d2);
    case RK_SCALE: return d * x;
    case RK_TOLERANT: return k.p1 * pp_log(T(1) + pp_exp((x - k.p0) / k.p1)) - k.p1 * pp_log(T(1) + pp_exp(-k.p0 / k.p1));
    default: return x;
  }
}
// rho'(x): what autograd returns for the expressions above
template <class T> PP_HD T robust_rho1(const RobustParam<T>& k, T x) {
#pragma clang fp contract(off)
  const T d = k.p0, d2 = d * d;
  switch (k.kind) {
    case RK_HUBER: { const T root = pp_sqrt(x); return root < d ? T(1) : d / root; }
    case RK_PSEUDOHUBER: return T(1) / pp_sqrt(x / d2 + T(1));
    case RK_CAUCHY: return T(1) / (x / d2 + T(1));
    case RK_SOFTLONE: return d / pp_sqrt(T(1) / d2 + x);
    case RK_ARCTAN: { const T u = x / d2; return T(1) / (T(1) + u * u); }
    case RK_SCALE: return d;
    case RK_TOLERANT: { const T e = pp_exp((x - k.p0) / k.p1); return e / (T(1) + e); }
    default: return T(1);
  }
}
// FastTriggs' row scale sqrt(rho'(|r|^2)) (corrector.py:91-93)
template <class T, int DR> PP_HD T robust_row_scale(const RobustParam<T>& k, const T* r) {
  T x = T(0);
#pragma unroll
  for (int i = 0; i < DR; ++i) x += r[i] * r[i];
  return pp_sqrt(robust_rho1<T>(k, x));
}

}  // namespace pplie
