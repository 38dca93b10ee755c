// convert.hip -- data-format conversions either side of the Lie hot path:
//   rotation matrix -> quaternion   (pypose/lietensor/convert.py:8-146  mat2SO3; mat2SE3/Sim3/RxSO3 build on it)
//   Euler angles    -> quaternion   (convert.py:607-663 euler2SO3)
//   quaternion      -> Euler angles (lietensor.py:1147-1173 LieTensor.euler)
// Row kernels on the same LDS-slab shell as the group ops.  Each forward is templated on its scalar type;
// the backward kernels push Dual<T> sweeps through the very same code, i.e. they differentiate the branch the
// value takes -- what the reference's autograd does through its masked sums / torch.where.
#include "rowmap.h"

namespace pplie {

template <class T> struct ScalarParam { T v; };

// mat2SO3 (convert.py:97-146).  m = row-major 3x3 R; the reference works on rt = R^T.
// Four candidates q_k / (2 sqrt(t_k)), selected by  rt22 < atol,  rt00 > rt11,  rt00 < -rt11.
template <class S, class B> PP_HD void mat2so3(const S* m, S* q, B atol) {
  typedef typename Num<S>::base T;
  const S rt00 = m[0], rt01 = m[3], rt02 = m[6], rt10 = m[1], rt11 = m[4], rt12 = m[7], rt20 = m[2], rt21 = m[5], rt22 = m[8];
  const bool d2 = pp_val(rt22) < (T)atol;
  const S one = S(T(1));
  S w, x, y, z, t;
  if (d2) {
    if (pp_val(rt00) > pp_val(rt11)) {
      t = one + rt00 - rt11 - rt22;
      w = rt12 - rt21; x = t; y = rt01 + rt10; z = rt20 + rt02;
    } else {
      t = one - rt00 + rt11 - rt22;
      w = rt20 - rt02; x = rt01 + rt10; y = t; z = rt12 + rt21;
    }
  } else {
    if (pp_val(rt00) < -pp_val(rt11)) {
      t = one - rt00 - rt11 + rt22;
      w = rt01 - rt10; x = rt20 + rt02; y = rt12 + rt21; z = t;
    } else {
      t = one + rt00 + rt11 + rt22;
      w = t; x = rt12 - rt21; y = rt20 - rt02; z = rt01 - rt10;
    }
  }
  const S inv = one / (S(T(2)) * pp_sqrt(t));
  q[0] = x * inv; q[1] = y * inv; q[2] = z * inv; q[3] = w * inv;      // wxyz -> xyzw (:143-144)
}

// euler2SO3 (convert.py:650-663): [roll, pitch, yaw] -> [x, y, z, w]
template <class S> PP_HD void euler2so3(const S* e, S* q) {
  typedef typename Num<S>::base T;
  S sr, cr, sp, cp, sy, cy;
  pp_sincos(e[0] * S(T(0.5)), sr, cr);
  pp_sincos(e[1] * S(T(0.5)), sp, cp);
  pp_sincos(e[2] * S(T(0.5)), sy, cy);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}

// LieTensor.euler (lietensor.py:1151-1173): [x, y, z, w] -> [roll, pitch, yaw]; near the pitch = +-pi/2
// singularity (|t2| >= 1 - eps) roll = 0 and yaw = -2 pm(t2) atan2(x, w).
// The branch / clamp decisions hinge on the last bit of t2 when the pitch is exactly +-pi/2, so they are
// taken once, in plain T with every product and sum rounded separately (as the reference's tensor ops are),
// and shared by the forward and the Dual sweeps of the backward.
struct EulerBranch { bool regular; int clamp; float sgn; };
template <class T> PP_HD EulerBranch so3_euler_branch(const T* q, T eps) {
#pragma clang fp contract(off)
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T xx = x * x, yy = y * y, zz = z * z, ww = w * w;
  const T wy = w * y, zx = z * x;
  const T num = T(2) * (wy - zx);
  const T den = ((xx + yy) + zz) + ww;
  const T t2 = num / den;
  EulerBranch b;
  b.regular = (t2 < T(0) ? -t2 : t2) < T(1) - eps;
  b.clamp = t2 > T(1) ? 1 : (t2 < T(-1) ? -1 : 0);
  b.sgn = t2 < T(0) ? -1.f : 1.f;                                  // pm(): +1 at 0 (basics/ops.py:24)
  return b;
}
template <class S> PP_HD void so3_euler(const S* q, S* e, EulerBranch br) {
  typedef typename Num<S>::base T;
  const S x = q[0], y = q[1], z = q[2], w = q[3];
  const S xx = x * x, yy = y * y, zz = z * z, ww = w * w;
  const S two = S(T(2));
  if (br.regular) {
    e[0] = pp_atan2(two * (w * x + y * z), (ww + zz) - (xx + yy));
    e[2] = pp_atan2(two * (w * z + x * y), (ww + xx) - (yy + zz));
  } else {
    e[0] = S(T(0));
    e[2] = S(T(-2) * T(br.sgn)) * pp_atan2(x, w);
  }
  // asin(clamp(t2, -1, 1)): the clamp passes no gradient outside [-1, 1]
  if (br.clamp > 0) e[1] = S(pp_asin(T(1)));
  else if (br.clamp < 0) e[1] = S(pp_asin(T(-1)));
  else {
    S t2 = two * (w * y - z * x) / (xx + yy + zz + ww);
    const T v = pp_val(t2);                                        // (a contracted product may land one ulp outside)
    if (v > T(1) || v < T(-1)) t2 = t2 + S((v > T(1) ? T(1) : T(-1)) - v);
    e[1] = pp_asin(t2);
  }
}

// backward of an IW -> OW row function F by IW forward-mode sweeps: gin_k = sum_j g_j dF_j/din_k
#define PPLIE_DUAL_BWD(IW, OW, CALL)                                   \
  for (int k = 0; k < IW; ++k) {                                       \
    Dual<T> in[IW], out[OW];                                           \
    for (int i = 0; i < IW; ++i) in[i] = Dual<T>(a[i], i == k ? T(1) : T(0)); \
    CALL;                                                              \
    T acc = T(0);                                                      \
    for (int j = 0; j < OW; ++j) acc += g[j] * out[j].d;               \
    gin[k] = acc;                                                      \
  }

template <class T> struct Op_mat2so3_fwd {
  enum { IW0 = 9, IW1 = 0, IW2 = 0, OW0 = 4, OW1 = 0 };
  static PP_HD void apply(const T* a, const T*, const T*, T* o, T*, ScalarParam<T> prm) { mat2so3<T, T>(a, o, prm.v); }
};
template <class T> struct Op_mat2so3_bwd {
  enum { IW0 = 9, IW1 = 4, IW2 = 0, OW0 = 9, OW1 = 0 };
  static PP_HD void apply(const T* a, const T* g, const T*, T* gin, T*, ScalarParam<T> prm) {
    PPLIE_DUAL_BWD(9, 4, (mat2so3<Dual<T>, T>(in, out, prm.v)))
  }
};
template <class T> struct Op_so3_euler_fwd {
  enum { IW0 = 4, IW1 = 0, IW2 = 0, OW0 = 3, OW1 = 0 };
  static PP_HD void apply(const T* a, const T*, const T*, T* o, T*, ScalarParam<T> prm) { so3_euler<T>(a, o, so3_euler_branch<T>(a, prm.v)); }
};
template <class T> struct Op_so3_euler_bwd {
  enum { IW0 = 4, IW1 = 3, IW2 = 0, OW0 = 4, OW1 = 0 };
  static PP_HD void apply(const T* a, const T* g, const T*, T* gin, T*, ScalarParam<T> prm) {
    const EulerBranch br = so3_euler_branch<T>(a, prm.v);
    PPLIE_DUAL_BWD(4, 3, (so3_euler<Dual<T>>(in, out, br)))
  }
};
template <class T> PP_HD void euler2so3_bwd(const T* a, const T* g, T* gin) { PPLIE_DUAL_BWD(3, 4, (euler2so3<Dual<T>>(in, out))) }
PPLIE_OP_1_1(Op_euler2so3_fwd, euler2so3, 3, 4)
PPLIE_OP_2_1(Op_euler2so3_bwd, euler2so3_bwd, 3, 4, 3)

template <class T, template <class> class OP>
int launch_param(const void* i0, const void* i1, void* o0, double prm, int64_t n, void* stream) {
  ScalarParam<T> p{(T)prm};
  return launch_rowmap<T, OP<T>, 1, 256, false, ScalarParam<T>>(i0, i1, nullptr, o0, nullptr, n, stream, kGridCap, p);
}
}  // namespace pplie

PPLIE_EXPORT(pplie_euler2so3_fwd, pplie::Op_euler2so3_fwd)
PPLIE_EXPORT(pplie_euler2so3_bwd, pplie::Op_euler2so3_bwd)

#define PPLIE_EXPORT_PARAM_FWD(SYM, OP)                                                                        \
  extern "C" int SYM##_f32(const void* in, void* out, double prm, int64_t n, void* stream) {                   \
    return pplie::launch_param<float, OP>(in, nullptr, out, prm, n, stream);                                   \
  }                                                                                                            \
  extern "C" int SYM##_f64(const void* in, void* out, double prm, int64_t n, void* stream) {                   \
    return pplie::launch_param<double, OP>(in, nullptr, out, prm, n, stream);                                  \
  }
#define PPLIE_EXPORT_PARAM_BWD(SYM, OP)                                                                        \
  extern "C" int SYM##_f32(const void* in, const void* g, void* gin, double prm, int64_t n, void* stream) {    \
    return pplie::launch_param<float, OP>(in, g, gin, prm, n, stream);                                         \
  }                                                                                                            \
  extern "C" int SYM##_f64(const void* in, const void* g, void* gin, double prm, int64_t n, void* stream) {    \
    return pplie::launch_param<double, OP>(in, g, gin, prm, n, stream);                                        \
  }
PPLIE_EXPORT_PARAM_FWD(pplie_mat2so3_fwd, pplie::Op_mat2so3_fwd)
PPLIE_EXPORT_PARAM_BWD(pplie_mat2so3_bwd, pplie::Op_mat2so3_bwd)
PPLIE_EXPORT_PARAM_FWD(pplie_so3_euler_fwd, pplie::Op_so3_euler_fwd)
PPLIE_EXPORT_PARAM_BWD(pplie_so3_euler_bwd, pplie::Op_so3_euler_bwd)
