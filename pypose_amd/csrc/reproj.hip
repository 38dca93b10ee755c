// reproj.hip -- pinhole reprojection residual of one (camera pose, point) pair with its closed-form Jacobian blocks:
// the linearisation of pypose.reprojerr / point2pixel (function/geometry.py:60-113, 171-226 -> homo2cart :37-57) for
// bundle adjustment.  Row ops of the uniform C ABI (include/pplie.h):
//
//   pplie_se3_reproj_fwd   (pose [n,7], point [n,3], cam [n,11]) -> r [n,2]
//   pplie_se3_reproj_lin   (pose, point, cam)                    -> r [n,2], J [n,18]
//   pplie_reproj_vjp       (J [n,18], g [n,2])                   -> gpose [n,7], gpoint [n,3]
//
// cam = the 3x3 intrinsic matrix row-major (9) followed by the observed pixel (2); r = homo2cart(K (X . p)) - pixel.
// J = [d r / d pose (2x6, the left-tangent convention of SE3_Act.backward, operation.py:561-568: [I | -skew(q)]),
//      d r / d point (2x3)], row-major 2 x 9.  Algorithmic bytes per observation: 84 read + 8 (fwd) / 80 (lin) written.
#include "rowmap.h"

namespace pplie {

template <class S> PP_HD void se3_reproj_core(const S* X, const S* p, const S* cam, S* r, S* J) {
  typedef typename Num<S>::base T;
  S qa[3];
  se3_act<S>(X, p, qa);
  const V3<S> q = v3(qa);
  const V3<S> k0 = v3(cam), k1 = v3(cam + 3), k2 = v3(cam + 6);
  const S hx = dot(k0, q), hy = dot(k1, q), hz = dot(k2, q);
  // homo2cart: divide by pm(hz) * max(|hz|, tiny), pm(0) = +1 (geometry.py:55-57)
  const S az = pp_abs(hz);
  const bool clamped = pp_val(az) < Num<T>::min_normal();
  const S den = (pp_val(hz) < T(0) ? S(T(-1)) : S(T(1))) * (clamped ? S(Num<T>::min_normal()) : az);
  const S inv = S(T(1)) / den;
  const S px = hx * inv, py = hy * inv;
  r[0] = px - cam[9];
  r[1] = py - cam[10];
  if (J) {
    // d pix / d h = [[inv, 0, -px inv], [0, inv, -py inv]]  (the clamp passes no gradient: third column 0 when clamped)
    const S dz = clamped ? S(T(0)) : inv;
    const V3<S> m0 = inv * k0 - (px * dz) * k2, m1 = inv * k1 - (py * dz) * k2;       // rows of D = (d pix / d h) K
    const V3<S> qv = v3(X + 3);
    const S qw = X[6];
    put(m0, J);
    put(cross(q, m0), J + 3);                       // m (-skew(q)) = q x m
    put(adj_rotate_T(qv, qw, m0), J + 6);           // m R = (R^T m)^T
    put(m1, J + 9);
    put(cross(q, m1), J + 12);
    put(adj_rotate_T(qv, qw, m1), J + 15);
  }
}
template <class S> PP_HD void se3_reproj(const S* X, const S* p, const S* cam, S* r) { se3_reproj_core<S>(X, p, cam, r, nullptr); }
template <class S> PP_HD void se3_reproj_lin(const S* X, const S* p, const S* cam, S* r, S* J) { se3_reproj_core<S>(X, p, cam, r, J); }
template <class S> PP_HD void reproj_vjp(const S* J, const S* g, S* gX, S* gp) {
  typedef typename Num<S>::base T;
#pragma unroll
  for (int c = 0; c < 6; ++c) gX[c] = g[0] * J[c] + g[1] * J[9 + c];
  gX[6] = S(T(0));
#pragma unroll
  for (int c = 0; c < 3; ++c) gp[c] = g[0] * J[6 + c] + g[1] * J[15 + c];
}

template <class T> struct Op_se3_reproj_fwd {
  enum { IW0 = 7, IW1 = 3, IW2 = 11, OW0 = 2, OW1 = 0 };
  static PP_HD void apply(const T* a, const T* b, const T* c, T* o, T*) { se3_reproj<T>(a, b, c, o); }
};
PPLIE_OP_3_2(Op_se3_reproj_lin, se3_reproj_lin, 7, 3, 11, 2, 18)
PPLIE_OP_2_2(Op_reproj_vjp, reproj_vjp, 18, 2, 7, 3)
// 41 scalars per row: a 256-row tile is 42 KB of LDS (3 workgroups per CU); 128 x 1 measured 141.8 -> 128.6 us at 4 M observations
// (round 4; 64 x 1: 130.8; two rows per lane, rolled or not: 190-245)
PPLIE_TILE_EX(Op_se3_reproj_lin, 1, 128, false)

}  // namespace pplie

PPLIE_EXPORT(pplie_se3_reproj_fwd, pplie::Op_se3_reproj_fwd)
PPLIE_EXPORT(pplie_se3_reproj_lin, pplie::Op_se3_reproj_lin)
PPLIE_EXPORT(pplie_reproj_vjp, pplie::Op_reproj_vjp)
