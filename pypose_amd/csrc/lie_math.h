// lie_math.h -- per-row Lie-group arithmetic shared by every HIP kernel in this library.
//
// Everything here is a register-resident, matrix-free restatement of what the reference
// builds out of [B,3,3]/[B,6,6]/[B,7,7] temporaries (pypose/lietensor/operation.py).  One
// call = one row (one group element); the kernels in rowmap.h are thin load/transpose/store
// shells around these functions.  The functions are templated on the scalar type S so that
// the same code serves fp32 kernels, fp64 kernels and the forward-mode Dual<T> used for the
// Jinvp backward.  They compile with hipcc (device) and with g++ (tests/hostmath builds this
// header for the CPU-only arithmetic check -- that build is test infrastructure, never a
// product fallback).
//
// Conventions (reference: pypose/lietensor/lietensor.py:196-198,303-305,354-356,447-449,
// 494-496,590-592,638-640,731-733): quaternion [x,y,z,w]; SE3 [t(3),q(4)]; Sim3 [t,q,s];
// RxSO3 [q,s]; tangents se3 [tau,phi], sim3 [tau,phi,sigma], rxso3 [phi,sigma].
#pragma once
#include <cmath>
#include <cstdint>
#include <cfloat>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PP_HD __host__ __device__ __forceinline__
#else
#define PP_HD inline
#endif

namespace pplie {

// ---------------------------------------------------------------------------------------
// numeric traits
// ---------------------------------------------------------------------------------------
template <class T> struct Num;
template <> struct Num<float> {
  typedef float base;
  static PP_HD float eps() { return 1.1920928955078125e-07f; }   // torch.finfo(float32).eps
  static PP_HD float max() { return FLT_MAX; }
  static PP_HD float min_normal() { return FLT_MIN; }
  // below theta^2 < series2 the coefficient functions B, C, D, E use their power series (the closed forms cancel
  // catastrophically in fp32: SURVEY.md section 7 "hard parts").  The 8-term series are good to 2-3 ulp up to
  // theta = pi (tests/hostmath: max rel. error 2.7e-7 on [0, 9.9]), so every rotation angle a Log can return takes the
  // series and the closed-form branch is dynamically rare -- with the switch at 1.5 rad nearly every wave executed both
  // sides.  F = (1 - (t/2)cot(t/2))/t^2 has a pole at 2 pi: its series keeps the old switch.
  static PP_HD float series2() { return 9.9f; }
  static PP_HD float seriesF2() { return 2.25f; }
};
template <> struct Num<double> {
  typedef double base;
  static PP_HD double eps() { return 2.220446049250313e-16; }     // torch.finfo(float64).eps
  static PP_HD double max() { return DBL_MAX; }
  static PP_HD double min_normal() { return DBL_MIN; }
  static PP_HD double series2() { return 0.0625; }
  static PP_HD double seriesF2() { return 0.0625; }
};

PP_HD float pp_sin(float x) { return ::sinf(x); }
PP_HD double pp_sin(double x) { return ::sin(x); }
PP_HD float pp_cos(float x) { return ::cosf(x); }
PP_HD double pp_cos(double x) { return ::cos(x); }
PP_HD float pp_sqrt(float x) { return ::sqrtf(x); }
PP_HD double pp_sqrt(double x) { return ::sqrt(x); }
// 1 / sqrt(x): v_rsq_f32 (1 ulp) on the device; NaN for x < 0, inf at 0
#if defined(__HIP_DEVICE_COMPILE__)
PP_HD float pp_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else
PP_HD float pp_rsqrt(float x) { return 1.0f / ::sqrtf(x); }
#endif
PP_HD double pp_rsqrt(double x) { return 1.0 / ::sqrt(x); }
PP_HD float pp_atan(float x) { return ::atanf(x); }
PP_HD double pp_atan(double x) { return ::atan(x); }
PP_HD float pp_exp(float x) { return ::expf(x); }
PP_HD double pp_exp(double x) { return ::exp(x); }
PP_HD float pp_expm1(float x) { return ::expm1f(x); }
PP_HD double pp_expm1(double x) { return ::expm1(x); }
PP_HD float pp_log(float x) { return ::logf(x); }
PP_HD double pp_log(double x) { return ::log(x); }
PP_HD float pp_atan2(float y, float x) { return ::atan2f(y, x); }
PP_HD double pp_atan2(double y, double x) { return ::atan2(y, x); }
PP_HD float pp_asin(float x) { return ::asinf(x); }
PP_HD double pp_asin(double x) { return ::asin(x); }
PP_HD float pp_abs(float x) { return ::fabsf(x); }
PP_HD double pp_abs(double x) { return ::fabs(x); }
PP_HD float pp_val(float x) { return x; }
PP_HD double pp_val(double x) { return x; }
PP_HD bool pp_isnan(float x) { return x != x; }
PP_HD bool pp_isnan(double x) { return x != x; }

// sin and cos of one fp32 angle in ~30 instructions: 3-constant Cody-Waite reduction by pi/2
// (exact enough for |x| < 2^15; every angle on the hot path is a rotation angle or half of one)
// + the classic degree-9 / degree-8 minimax polynomials on [-pi/4, pi/4] (<= 1 ulp).  Larger
// arguments are reduced in double.  The library's sinf+cosf pair costs ~4x as much because each
// call carries its own Payne-Hanek reduction.
PP_HD void pp_sincos(float x, float& s, float& c) {
  float kf, r;
  int k;
  if (::fabsf(x) < 32768.0f) {
    kf = ::rintf(x * 0.63661977236758134f);
    k = (int)kf;
    r = ::fmaf(kf, -1.57079625129699707031e+00f, x);
    r = ::fmaf(kf, -7.54978941586159635335e-08f, r);
    r = ::fmaf(kf, -5.39030252995776476554e-15f, r);
  } else {
    // rare (an angle beyond 5000 turns): reduce in double instead of dragging libm's Payne-Hanek
    // tables and its register footprint into every kernel; inf / nan fall through as nan
    double xd = (double)x;
    double kd = ::rint(xd * 0.63661977236758134308);
    double rd = ::fma(kd, -1.57079632679489655800e+00, xd);
    rd = ::fma(kd, -6.12323399573676603587e-17, rd);
    k = (int)(((long long)kd) & 3);
    r = (float)rd;
  }
  float z = r * r;
  float ps = ::fmaf(::fmaf(::fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
  float pc = ::fmaf(::fmaf(::fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                    ::fmaf(-0.5f, z, 1.0f));
  float ss = (k & 1) ? pc : ps;
  float cc = (k & 1) ? ps : pc;
  s = (k & 2) ? -ss : ss;
  c = ((k + 1) & 2) ? -cc : cc;
}
PP_HD void pp_sincos(double x, double& s, double& c) {
  s = ::sin(x);
  c = ::cos(x);
}

// ---------------------------------------------------------------------------------------
// Forward-mode dual numbers: value + one directional derivative.  Every row function below is
// templated on its scalar type, so instantiating it with Dual<T> differentiates exactly the
// branch the value takes -- this is how the Jinvp / Jr backwards are obtained (the reference
// lets autograd differentiate through so3_Jl_inv / calcQ: lietensor.py:257-264, 343-351).
// ---------------------------------------------------------------------------------------
template <class T> struct Dual {
  T v, d;
  PP_HD Dual() : v(T(0)), d(T(0)) {}
  PP_HD explicit Dual(T a) : v(a), d(T(0)) {}
  PP_HD Dual(T a, T b) : v(a), d(b) {}
};
template <class T> struct Num<Dual<T>> { typedef T base; };
template <class T> PP_HD Dual<T> operator+(Dual<T> a, Dual<T> b) { return Dual<T>(a.v + b.v, a.d + b.d); }
template <class T> PP_HD Dual<T> operator-(Dual<T> a, Dual<T> b) { return Dual<T>(a.v - b.v, a.d - b.d); }
template <class T> PP_HD Dual<T> operator-(Dual<T> a) { return Dual<T>(-a.v, -a.d); }
template <class T> PP_HD Dual<T> operator*(Dual<T> a, Dual<T> b) { return Dual<T>(a.v * b.v, a.d * b.v + a.v * b.d); }
template <class T> PP_HD Dual<T> operator/(Dual<T> a, Dual<T> b) {
  T q = a.v / b.v;
  return Dual<T>(q, (a.d - q * b.d) / b.v);
}
template <class T> PP_HD T pp_val(Dual<T> a) { return a.v; }
template <class T> PP_HD Dual<T> pp_sqrt(Dual<T> a) {
  T r = pp_sqrt(a.v);
  return Dual<T>(r, a.d == T(0) ? T(0) : a.d / (T(2) * r));
}
template <class T> PP_HD Dual<T> pp_sin(Dual<T> a) { return Dual<T>(pp_sin(a.v), pp_cos(a.v) * a.d); }
template <class T> PP_HD Dual<T> pp_cos(Dual<T> a) { return Dual<T>(pp_cos(a.v), -pp_sin(a.v) * a.d); }
template <class T> PP_HD void pp_sincos(Dual<T> a, Dual<T>& s, Dual<T>& c) {
  T sv, cv;
  pp_sincos(a.v, sv, cv);
  s = Dual<T>(sv, cv * a.d);
  c = Dual<T>(cv, -sv * a.d);
}
template <class T> PP_HD Dual<T> pp_atan(Dual<T> a) { return Dual<T>(pp_atan(a.v), a.d / (T(1) + a.v * a.v)); }
template <class T> PP_HD Dual<T> pp_exp(Dual<T> a) { T e = pp_exp(a.v); return Dual<T>(e, e * a.d); }
template <class T> PP_HD Dual<T> pp_expm1(Dual<T> a) { T e = pp_expm1(a.v); return Dual<T>(e, (e + T(1)) * a.d); }
template <class T> PP_HD Dual<T> pp_log(Dual<T> a) { return Dual<T>(pp_log(a.v), a.d / a.v); }
template <class T> PP_HD Dual<T> pp_abs(Dual<T> a) { return a.v < T(0) ? -a : a; }
template <class T> PP_HD Dual<T> pp_atan2(Dual<T> y, Dual<T> x) {
  return Dual<T>(pp_atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / (x.v * x.v + y.v * y.v));
}
template <class T> PP_HD Dual<T> pp_asin(Dual<T> a) { return Dual<T>(pp_asin(a.v), a.d / pp_sqrt(T(1) - a.v * a.v)); }

// torch.nan_to_num default semantics: nan -> 0, +inf -> max, -inf -> lowest
template <class S> PP_HD S pp_nan_to_num(S x) {
  typedef typename Num<S>::base B;
  B v = pp_val(x);
  if (v != v) return S(B(0));
  if (v > Num<B>::max()) return S(Num<B>::max());
  if (v < -Num<B>::max()) return S(-Num<B>::max());
  return x;   // (for Dual: a replaced value drops its derivative, as autograd through nan_to_num does)
}

// pypose.basics.pm: sign(sign(x)*2+1) -> +1 at 0 (reference basics/ops.py:24)
template <class S> PP_HD S pp_pm(S x) {
  typedef typename Num<S>::base B;
  B v = pp_val(x);
  if (v != v) return x;
  return v < B(0) ? S(B(-1)) : S(B(1));
}

// ---------------------------------------------------------------------------------------
// tiny 3-vector
// ---------------------------------------------------------------------------------------
template <class S> struct V3 {
  S x, y, z;
};
template <class S> PP_HD V3<S> v3(S x, S y, S z) {
  V3<S> r;
  r.x = x; r.y = y; r.z = z;
  return r;
}
template <class S> PP_HD V3<S> v3(const S* p) { return v3(p[0], p[1], p[2]); }
template <class S> PP_HD void put(const V3<S>& a, S* p) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
template <class S> PP_HD V3<S> operator+(const V3<S>& a, const V3<S>& b) { return v3<S>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class S> PP_HD V3<S> operator-(const V3<S>& a, const V3<S>& b) { return v3<S>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class S> PP_HD V3<S> operator-(const V3<S>& a) { return v3<S>(-a.x, -a.y, -a.z); }
template <class S> PP_HD V3<S> operator*(S s, const V3<S>& a) { return v3<S>(s * a.x, s * a.y, s * a.z); }
template <class S> PP_HD V3<S> cross(const V3<S>& a, const V3<S>& b) {
  return v3<S>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <class S> PP_HD S dot(const V3<S>& a, const V3<S>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class S> PP_HD S norm2(const V3<S>& a) { return dot(a, a); }

// ---------------------------------------------------------------------------------------
// rotation coefficient functions of theta = |phi|
//   B = (1-cos t)/t^2, C = (t - sin t)/t^3                      so3_Jl   (operation.py:7-20)
//   D = (t^2 + 2cos t - 2)/(2 t^4), E = (2t - 3 sin t + t cos t)/(2 t^5)   calcQ (:37-58)
//   F = (1 - t cos(t/2) / (2 sin(t/2)))/t^2                      so3_Jl_inv (:23-32)
// The reference evaluates the closed forms for every theta > eps and a 2-term Taylor below;
// here the power series is used for theta^2 < Num::series2 (it contains the reference's
// Taylor branch as its leading terms) and cancellation-free closed forms above.
// ---------------------------------------------------------------------------------------
template <class S> struct RotCoef {
  S B, C, D, E;
};

template <class S> PP_HD S poly8(S t, S c0, S c1, S c2, S c3, S c4, S c5, S c6, S c7) {
  return c0 + t * (c1 + t * (c2 + t * (c3 + t * (c4 + t * (c5 + t * (c6 + t * c7))))));
}

// B and C only (forward paths)
template <class S> PP_HD void rot_coef_BC(S th2, S& B, S& C) {
  typedef typename Num<S>::base T;
  if (pp_val(th2) < Num<T>::series2()) {
    B = poly8<S>(th2, S(T(1.0 / 2)), S(T(-1.0 / 24)), S(T(1.0 / 720)), S(T(-1.0 / 40320)), S(T(1.0 / 3628800)),
                 S(T(-1.0 / 479001600)), S(T(1.0 / 87178291200.0)), S(T(-1.0 / 20922789888000.0)));
    C = poly8<S>(th2, S(T(1.0 / 6)), S(T(-1.0 / 120)), S(T(1.0 / 5040)), S(T(-1.0 / 362880)), S(T(1.0 / 39916800)),
                 S(T(-1.0 / 6227020800.0)), S(T(1.0 / 1307674368000.0)), S(T(-1.0 / 355687428096000.0)));
  } else {
    S th = pp_sqrt(th2);
    S sh, ch;
    pp_sincos(S(T(0.5)) * th, sh, ch);
    B = S(T(2)) * sh * sh / th2;                        // 2 sin^2(t/2) / t^2: no cancellation
    C = (th - S(T(2)) * sh * ch) / (th2 * th);          // sin t = 2 sin(t/2) cos(t/2)
  }
}

// power series of B, C, D, E in t = theta^2 (used below series2(); no trigonometric call)
template <class S> PP_HD RotCoef<S> rot_coef_series(S th2) {
  typedef typename Num<S>::base T;
  RotCoef<S> k;
  k.B = poly8<S>(th2, S(T(1.0 / 2)), S(T(-1.0 / 24)), S(T(1.0 / 720)), S(T(-1.0 / 40320)), S(T(1.0 / 3628800)),
                 S(T(-1.0 / 479001600)), S(T(1.0 / 87178291200.0)), S(T(-1.0 / 20922789888000.0)));
  k.C = poly8<S>(th2, S(T(1.0 / 6)), S(T(-1.0 / 120)), S(T(1.0 / 5040)), S(T(-1.0 / 362880)), S(T(1.0 / 39916800)),
                 S(T(-1.0 / 6227020800.0)), S(T(1.0 / 1307674368000.0)), S(T(-1.0 / 355687428096000.0)));
  // D = sum (-1)^j t^j / (2j+4)!
  k.D = poly8<S>(th2, S(T(1.0 / 24)), S(T(-1.0 / 720)), S(T(1.0 / 40320)), S(T(-1.0 / 3628800)), S(T(1.0 / 479001600)),
                 S(T(-1.0 / 87178291200.0)), S(T(1.0 / 20922789888000.0)), S(T(-1.0 / 6402373705728000.0)));
  // E = sum (-1)^j (j+1) t^j / (2j+5)!
  k.E = poly8<S>(th2, S(T(1.0 / 120)), S(T(-2.0 / 5040)), S(T(3.0 / 362880)), S(T(-4.0 / 39916800)),
                 S(T(5.0 / 6227020800.0)), S(T(-6.0 / 1307674368000.0)), S(T(7.0 / 355687428096000.0)),
                 S(T(-8.0 / 121645100408832000.0)));
  return k;
}

template <class S> PP_HD RotCoef<S> rot_coef(S th2) {
  typedef typename Num<S>::base T;
  if (pp_val(th2) < Num<T>::series2()) return rot_coef_series(th2);
  RotCoef<S> k;
  S th = pp_sqrt(th2);
  S sh, ch;
  pp_sincos(S(T(0.5)) * th, sh, ch);
  k.B = S(T(2)) * sh * sh / th2;
  k.C = (th - S(T(2)) * sh * ch) / (th2 * th);
  k.D = (S(T(0.5)) - k.B) / th2;                       // == (t^2 + 2cos t - 2)/(2 t^4)
  k.E = (S(T(3)) * k.C - k.B) / (S(T(2)) * th2);       // == (2t - 3 sin t + t cos t)/(2 t^5)
  return k;
}

// F of so3_Jl_inv
template <class S> PP_HD S rot_coef_F_series(S th2) {
  typedef typename Num<S>::base T;
  // (t/2)cot(t/2) = 1 - t^2/12 - t^4/720 - t^6/30240 - ... (Bernoulli numbers)
  return poly8<S>(th2, S(T(1.0 / 12)), S(T(1.0 / 720)), S(T(1.0 / 30240)), S(T(1.0 / 1209600)), S(T(1.0 / 47900160)),
                  S(T(691.0 / 1307674368000.0)), S(T(1.0 / 74724249600.0)), S(T(3617.0 / 10670622842880000.0)));
}
template <class S> PP_HD S rot_coef_F(S th2) {
  typedef typename Num<S>::base T;
  if (pp_val(th2) < Num<T>::seriesF2()) return rot_coef_F_series(th2);
  S th = pp_sqrt(th2);
  S sh, ch;
  pp_sincos(S(T(0.5)) * th, sh, ch);
  return pp_nan_to_num((S(T(1)) - th * ch / (S(T(2)) * sh)) / th2);
}

// Jl(phi) v = v + B phi x v + C phi x (phi x v)            (operation.py:7-20)
template <class S> PP_HD V3<S> jl_apply(S B, S C, const V3<S>& phi, const V3<S>& v) {
  V3<S> a = cross(phi, v);
  return v + B * a + C * cross(phi, a);
}
// Jl_inv(phi) v = v - 1/2 phi x v + F phi x (phi x v)      (operation.py:23-32)
template <class S> PP_HD V3<S> jlinv_apply(S F, const V3<S>& phi, const V3<S>& v) {
  typedef typename Num<S>::base T;
  V3<S> a = cross(phi, v);
  return v - S(T(0.5)) * a + F * cross(phi, a);
}

// Q(tau,phi) v with the four calcQ coefficient groups (operation.py:56-57); every
// Phi..Tau..Phi factor is a nested cross product.
template <class S> PP_HD V3<S> q_apply(const RotCoef<S>& k, const V3<S>& tau, const V3<S>& phi, const V3<S>& v) {
  typedef typename Num<S>::base T;
  V3<S> pv = cross(phi, v);       // Phi v
  V3<S> tv = cross(tau, v);       // Tau v
  V3<S> ppv = cross(phi, pv);     // Phi^2 v
  V3<S> tpv = cross(tau, pv);     // Tau Phi v
  V3<S> ptv = cross(phi, tv);     // Phi Tau v
  V3<S> ptpv = cross(phi, tpv);   // Phi Tau Phi v
  V3<S> pptv = cross(phi, ptv);   // Phi^2 Tau v
  V3<S> tppv = cross(tau, ppv);   // Tau Phi^2 v
  V3<S> ptppv = cross(phi, tppv); // Phi Tau Phi^2 v
  V3<S> pptpv = cross(phi, ptpv); // Phi^2 Tau Phi v
  return S(T(0.5)) * tv + k.C * (ptv + tpv + ptpv) + k.D * (pptv + tppv - S(T(3)) * ptpv) + k.E * (ptppv + pptpv);
}

// ---------------------------------------------------------------------------------------
// quaternion helpers  (SO3_Act operation.py:519-525, SO3_Mul :832-837)
// ---------------------------------------------------------------------------------------
template <class S> PP_HD V3<S> quat_rotate(const V3<S>& qv, S qw, const V3<S>& p) {
  V3<S> uv = cross(qv, p);
  uv = uv + uv;
  return p + qw * uv + cross(qv, uv);
}
// R^T p = rotation by the conjugate
template <class S> PP_HD V3<S> quat_rotate_inv(const V3<S>& qv, S qw, const V3<S>& p) { return quat_rotate(-qv, qw, p); }

// R v with R = SO3_Adj(q) = 2w(wI + K(v)) - I + 2 v v^T (operation.py:175-179; SO3_Matrix is the
// same matrix, :182-183).  It equals the SO3_Act rotation only for unit quaternions; the
// reference multiplies by this matrix in every Adj / Matrix / backward product and by the
// cross-product form in every forward Act, so non-normalised inputs must take the same routes.
template <class S> PP_HD V3<S> adj_rotate(const V3<S>& qv, S qw, const V3<S>& p) {
  typedef typename Num<S>::base T;
  return (S(T(2)) * qw * qw - S(T(1))) * p + (S(T(2)) * qw) * cross(qv, p) + (S(T(2)) * dot(qv, p)) * qv;
}
template <class S> PP_HD V3<S> adj_rotate_T(const V3<S>& qv, S qw, const V3<S>& p) { return adj_rotate(-qv, qw, p); }

template <class S> PP_HD void quat_mul(const S* X, const S* Y, S* Z) {
  V3<S> xv = v3(X), yv = v3(Y);
  S xw = X[3], yw = Y[3];
  V3<S> zv = xw * yv + yw * xv + cross(xv, yv);   // Xw*Yv + Xv*Yw + Xv x Yv
  put(zv, Z);
  Z[3] = xw * yw - dot(xv, yv);
}


// Jl_inv(x) p for each algebra (defined with the Jinvp backward at the end of this header)
template <class S> PP_HD void so3_jlinv_p(const S* x, const S* p, S* out);
template <class S> PP_HD void se3_jlinv_p(const S* x, const S* p, S* out);
template <class S> PP_HD void rxso3_jlinv_p(const S* x, const S* p, S* out);
template <class S> PP_HD void sim3_jlinv_p(const S* x, const S* p, S* out);

// ---------------------------------------------------------------------------------------
// SO3
// ---------------------------------------------------------------------------------------
// so3_Exp.forward (operation.py:343-357)
template <class S> PP_HD void so3_exp(const S* x, S* q) {
  typedef typename Num<S>::base T;
  V3<S> phi = v3(x);
  S th2 = norm2(phi);
  S th = pp_sqrt(th2);
  S imag, real;
  if (pp_val(th) > Num<T>::eps()) {
    S sh;
    pp_sincos(S(T(0.5)) * th, sh, real);
    imag = sh / th;
  } else {
    S th4 = th2 * th2;
    imag = S(T(0.5)) - S(T(1.0 / 48.0)) * th2 + S(T(1.0 / 3840.0)) * th4;
    real = S(T(1.0)) - S(T(1.0 / 8.0)) * th2 + S(T(1.0 / 384.0)) * th4;
  }
  put(imag * phi, q);
  q[3] = real;
}

// SO3_Log.forward (operation.py:307-324): three-way branch, atan (not atan2), no
// canonicalisation of the quaternion sign.
template <class S> PP_HD void so3_log(const S* q, S* x) {
  typedef typename Num<S>::base T;
  V3<S> v = v3(q);
  S w = q[3];
  S vn = pp_sqrt(norm2(v));
  bool vbig = pp_val(vn) > Num<T>::eps();
  bool wbig = pp_val(pp_abs(w)) > Num<T>::eps();
  S f;
  if (vbig && wbig)
    f = pp_nan_to_num(S(T(2)) * pp_atan(vn / w) / vn);
  else if (vbig)
    f = pp_nan_to_num(pp_pm(w) * S(T(3.14159265358979323846)) / vn);
  else
    f = pp_nan_to_num(S(T(2)) * (S(T(1)) / w - vn * vn / (S(T(3)) * w * w * w)));
  put(f * v, x);
}

// so3_Exp.backward: g[:3] @ so3_Jl(x) == Jl(-x) g   (operation.py:366-370)
template <class S> PP_HD void so3_exp_bwd(const S* x, const S* g, S* gx) {
  V3<S> phi = v3(x);
  S B, C;
  rot_coef_BC(norm2(phi), B, C);
  put(jl_apply(B, C, -phi, v3(g)), gx);
}
// SO3_Log.backward: [g @ so3_Jl_inv(y), 0]             (operation.py:332-337)
template <class S> PP_HD void so3_log_bwd(const S* y, const S* g, S* gX) {
  typedef typename Num<S>::base T;
  V3<S> phi = v3(y);
  S F = rot_coef_F(norm2(phi));
  put(jlinv_apply(F, -phi, v3(g)), gX);
  gX[3] = S(T(0));
}

// SO3_Act (operation.py:519-525) and its backward (:535-542):
//   X_grad = [g @ skew(-out), 0] = [out x g, 0]; p_grad = g @ R = R^T g
template <class S> PP_HD void so3_act(const S* X, const S* p, S* out) { put(quat_rotate(v3(X), X[3], v3(p)), out); }
template <class S> PP_HD void so3_act_bwd(const S* X, const S* out, const S* g, S* gX, S* gp) {
  typedef typename Num<S>::base T;
  put(cross(v3(out), v3(g)), gX);
  gX[3] = S(T(0));
  put(adj_rotate_T(v3(X), X[3], v3(g)), gp);
}
// SO3_Act4 (:626-645): rotate xyz, carry w; X_grad from the 3 leading comps of out/g
template <class S> PP_HD void so3_act4(const S* X, const S* p, S* out) {
  put(quat_rotate(v3(X), X[3], v3(p)), out);
  out[3] = p[3];
}
template <class S> PP_HD void so3_act4_bwd(const S* X, const S* out, const S* g, S* gX, S* gp) {
  typedef typename Num<S>::base T;
  put(cross(v3(out), v3(g)), gX);
  gX[3] = S(T(0));
  put(adj_rotate_T(v3(X), X[3], v3(g)), gp);
  gp[3] = g[3];
}

// SO3_Mul (:832-852)
template <class S> PP_HD void so3_mul(const S* X, const S* Y, S* Z) { quat_mul(X, Y, Z); }
template <class S> PP_HD void so3_mul_bwd(const S* X, const S* g, S* gX, S* gY) {
  typedef typename Num<S>::base T;
  gX[0] = g[0]; gX[1] = g[1]; gX[2] = g[2]; gX[3] = S(T(0));
  put(adj_rotate_T(v3(X), X[3], v3(g)), gY);   // g @ Adj(X) = R^T g
  gY[3] = S(T(0));
}
// SO3_Inv (:933-949)
template <class S> PP_HD void so3_inv(const S* X, S* Y) {
  Y[0] = -X[0]; Y[1] = -X[1]; Y[2] = -X[2]; Y[3] = X[3];
}
template <class S> PP_HD void so3_inv_bwd(const S* Y, const S* g, S* gX) {
  typedef typename Num<S>::base T;
  put(-adj_rotate_T(v3(Y), Y[3], v3(g)), gX);
  gX[3] = S(T(0));
}
// SO3_AdjXa (:728-748): out = R a; X_grad = [-g @ skew(out), 0] = [out x g ... sign below]
//   (-g^T K(out))^T = -K(out)^T g = K(out) g = out x g ; a_grad = R^T g
template <class S> PP_HD void so3_adj(const S* X, const S* a, S* out) { put(adj_rotate(v3(X), X[3], v3(a)), out); }
template <class S> PP_HD void so3_adj_bwd(const S* X, const S* out, const S* g, S* gX, S* ga) {
  typedef typename Num<S>::base T;
  put(cross(v3(out), v3(g)), gX);
  gX[3] = S(T(0));
  put(adj_rotate_T(v3(X), X[3], v3(g)), ga);
}
// SO3_AdjTXa (:1027-1044): out = R^T a; a_grad = R g; X_grad = [-a @ skew(a_grad),0] = [a_grad x a, 0]
template <class S> PP_HD void so3_adjt(const S* X, const S* a, S* out) { put(adj_rotate_T(v3(X), X[3], v3(a)), out); }
template <class S> PP_HD void so3_adjt_bwd(const S* X, const S* a, const S* g, S* gX, S* ga) {
  typedef typename Num<S>::base T;
  V3<S> ag = adj_rotate(v3(X), X[3], v3(g));
  put(ag, ga);
  put(cross(ag, v3(a)), gX);
  gX[3] = S(T(0));
}
// SO3 Jinvp (lietensor.py:257-264): so3_Jl_inv(Log X) p
template <class S> PP_HD void so3_jinvp(const S* X, const S* p, S* out) {
  S x[3];
  so3_log(X, x);
  so3_jlinv_p(x, p, out);
}
// so3.Jr (lietensor.py:343-351): I - c1 K + c2 K^2 where theta > eps else I; row-major 3x3
template <class S> PP_HD void so3_jr(const S* x, S* J) {
  typedef typename Num<S>::base T;
  V3<S> phi = v3(x);
  S th2 = norm2(phi);
  S B = S(T(0)), C = S(T(0));
  if (pp_val(pp_sqrt(th2)) > Num<T>::eps()) rot_coef_BC(th2, B, C);
  // Jr = I - B K + C K^2, K = skew(phi), K^2 = phi phi^T - th2 I
  S px = phi.x, py = phi.y, pz = phi.z;
  J[0] = S(T(1)) + C * (px * px - th2);
  J[1] = B * pz + C * px * py;
  J[2] = -B * py + C * px * pz;
  J[3] = -B * pz + C * px * py;
  J[4] = S(T(1)) + C * (py * py - th2);
  J[5] = B * px + C * py * pz;
  J[6] = B * py + C * px * pz;
  J[7] = -B * px + C * py * pz;
  J[8] = S(T(1)) + C * (pz * pz - th2);
}

// ---------------------------------------------------------------------------------------
// SE3
// ---------------------------------------------------------------------------------------
// se3_Exp.forward (operation.py:401-405): t = Jl(phi) tau, q = Exp(phi).  One sincos of the
// half angle feeds the quaternion and both Jl coefficients (sin t = 2 sin(t/2) cos(t/2)).
template <class S> PP_HD void se3_exp(const S* x, S* X) {
  typedef typename Num<S>::base T;
  V3<S> tau = v3(x), phi = v3(x + 3);
  S th2 = norm2(phi);
  S th = pp_sqrt(th2);
  S sh = S(T(0)), ch = S(T(1)), imag, B, C;
  bool big = pp_val(th) > Num<T>::eps();
  if (big) {
    pp_sincos(S(T(0.5)) * th, sh, ch);
    imag = sh / th;
  } else {   // so3_Exp Taylor branch (operation.py:354-355)
    S th4 = th2 * th2;
    imag = S(T(0.5)) - S(T(1.0 / 48.0)) * th2 + S(T(1.0 / 3840.0)) * th4;
    ch = S(T(1.0)) - S(T(1.0 / 8.0)) * th2 + S(T(1.0 / 384.0)) * th4;
  }
  if (pp_val(th2) < Num<T>::series2()) {
    rot_coef_BC(th2, B, C);
  } else {
    B = S(T(2)) * sh * sh / th2;
    C = (th - S(T(2)) * sh * ch) / (th2 * th);
  }
  put(jl_apply(B, C, phi, tau), X);
  put(imag * phi, X + 3);
  X[6] = ch;
}
// SE3_Log.forward (:376-382): phi = Log(q), tau = Jl_inv(phi) t.  In the regular branch of
// SO3_Log (|v| > eps, |w| > eps) the half angle is atan(|v|/w), so the cot(theta/2) that
// so3_Jl_inv needs (operation.py:29-30) is exactly |w|/|v| -- no trigonometric call at all.
template <class S> PP_HD void se3_log(const S* X, S* x) {
  typedef typename Num<S>::base T;
  V3<S> v = v3(X + 3);
  S w = X[6];
  S vn = pp_sqrt(norm2(v));
  if (pp_val(vn) > Num<T>::eps() && pp_val(pp_abs(w)) > Num<T>::eps()) {
    S half = pp_atan(vn / w);
    V3<S> phi = pp_nan_to_num(S(T(2)) * half / vn) * v;
    put(phi, x + 3);
    S th2 = norm2(phi);
    S F;
    if (pp_val(th2) < Num<T>::seriesF2())
      F = rot_coef_F(th2);
    else
      F = pp_nan_to_num((S(T(1)) - pp_abs(half) * pp_abs(w) / vn) / th2);
    put(jlinv_apply(F, phi, v3(X)), x);
  } else {
    so3_log(X + 3, x + 3);
    V3<S> phi = v3(x + 3);
    put(jlinv_apply(rot_coef_F(norm2(phi)), phi, v3(X)), x);
  }
}
// se3_Exp.backward (:413-418): g[:6] @ se3_Jl(x);  se3_Jl = [[J,Q],[0,J]]  (:61-65)
//   out_tau = J^T g_tau ; out_phi = Q^T g_tau + J^T g_phi ; J^T = J(-phi), Q^T = Q(-tau,-phi)
template <class S> PP_HD void se3_exp_bwd(const S* x, const S* g, S* gx) {
  V3<S> tau = v3(x), phi = v3(x + 3);
  V3<S> gt = v3(g), gp = v3(g + 3);
  RotCoef<S> k = rot_coef(norm2(phi));
  V3<S> nphi = -phi;
  put(jl_apply(k.B, k.C, nphi, gt), gx);
  put(q_apply(k, -tau, nphi, gt) + jl_apply(k.B, k.C, nphi, gp), gx + 3);
}
// SE3_Log.backward (:389-395): [g @ se3_Jl_inv(y), 0];
//   se3_Jl_inv = [[Ji, -Ji Q Ji],[0, Ji]] (:68-75)
//   out_tau = Ji^T g_tau ; out_phi = -Ji^T Q^T Ji^T g_tau + Ji^T g_phi
template <class S> PP_HD void se3_log_bwd(const S* y, const S* g, S* gX) {
  typedef typename Num<S>::base T;
  V3<S> tau = v3(y), phi = v3(y + 3);
  V3<S> gt = v3(g), gp = v3(g + 3);
  S th2 = norm2(phi);
  RotCoef<S> k = rot_coef(th2);
  S F = rot_coef_F(th2);
  V3<S> nphi = -phi;
  V3<S> a = jlinv_apply(F, nphi, gt);
  put(a, gX);
  V3<S> b = q_apply(k, -tau, nphi, a);
  put(jlinv_apply(F, nphi, gp - b), gX + 3);
  gX[6] = S(T(0));
}
// SE3_Act (:548-568)
template <class S> PP_HD void se3_act(const S* X, const S* p, S* out) {
  put(v3(X) + quat_rotate(v3(X + 3), X[6], v3(p)), out);
}
//   X_grad = [g @ [I | skew(-out)], 0] = [g, out x g, 0]; p_grad = R^T g
template <class S> PP_HD void se3_act_bwd(const S* X, const S* out, const S* g, S* gX, S* gp) {
  typedef typename Num<S>::base T;
  gX[0] = g[0]; gX[1] = g[1]; gX[2] = g[2];
  put(cross(v3(out), v3(g)), gX + 3);
  gX[6] = S(T(0));
  put(adj_rotate_T(v3(X + 3), X[6], v3(g)), gp);
}
// SE3_Act4 (:651-671): t = R p3 + t*p4, out4 = p4
//   X_grad = g @ SE3_Act4_Jacobian(out) (:229-234): J[:3,:3] = I*out4, J[:3,3:] = skew(-out3); row 4 zero
//   p_grad = g @ SE3_Matrix4x4(X) = [R^T g3, t.g3 + g4]
template <class S> PP_HD void se3_act4(const S* X, const S* p, S* out) {
  put(quat_rotate(v3(X + 3), X[6], v3(p)) + p[3] * v3(X), out);
  out[3] = p[3];
}
template <class S> PP_HD void se3_act4_bwd(const S* X, const S* out, const S* g, S* gX, S* gp) {
  typedef typename Num<S>::base T;
  V3<S> g3 = v3(g);
  put(out[3] * g3, gX);
  put(cross(v3(out), g3), gX + 3);
  gX[6] = S(T(0));
  put(adj_rotate_T(v3(X + 3), X[6], g3), gp);
  gp[3] = dot(v3(X), g3) + g[3];
}
// SE3_Mul (:858-877)
template <class S> PP_HD void se3_mul(const S* X, const S* Y, S* Z) {
  put(v3(X) + quat_rotate(v3(X + 3), X[6], v3(Y)), Z);
  quat_mul(X + 3, Y + 3, Z + 3);
}
// SE3 Adj(X) = [[R, tx R],[0,R]] (:202-210).  Adj [u;w] = [R u + t x (R w), R w]
//   row-vector product g @ Adj = Adj^T g = [R^T g_t, R^T(g_t x t) ... ] derived:
//   Adj^T = [[R^T, 0],[(tx R)^T, R^T]] ; (tx R)^T g_t = R^T tx^T g_t = -R^T (t x g_t) = R^T (g_t x t)
template <class S> PP_HD void se3_adjT_apply(const S* X, const V3<S>& gt, const V3<S>& gp, V3<S>& ot, V3<S>& op) {
  V3<S> qv = v3(X + 3);
  S qw = X[6];
  ot = adj_rotate_T(qv, qw, gt);
  op = adj_rotate_T(qv, qw, cross(gt, v3(X)) + gp);
}
template <class S> PP_HD void se3_adj_apply(const S* X, const V3<S>& u, const V3<S>& w, V3<S>& ot, V3<S>& op) {
  V3<S> qv = v3(X + 3);
  S qw = X[6];
  op = adj_rotate(qv, qw, w);
  ot = adj_rotate(qv, qw, u) + cross(v3(X), op);
}
template <class S> PP_HD void se3_mul_bwd(const S* X, const S* g, S* gX, S* gY) {
  typedef typename Num<S>::base T;
  for (int i = 0; i < 6; ++i) gX[i] = g[i];
  gX[6] = S(T(0));
  V3<S> ot, op;
  se3_adjT_apply(X, v3(g), v3(g + 3), ot, op);
  put(ot, gY);
  put(op, gY + 3);
  gY[6] = S(T(0));
}
// SE3_Inv (:955-973)
template <class S> PP_HD void se3_inv(const S* X, S* Y) {
  so3_inv(X + 3, Y + 3);
  put(-quat_rotate(v3(Y + 3), Y[6], v3(X)), Y);
}
template <class S> PP_HD void se3_inv_bwd(const S* Y, const S* g, S* gX) {
  typedef typename Num<S>::base T;
  V3<S> ot, op;
  se3_adjT_apply(Y, v3(g), v3(g + 3), ot, op);
  put(-ot, gX);
  put(-op, gX + 3);
  gX[6] = S(T(0));
}
// se3_adj(x) = [[Phi, Tau],[0, Phi]] (:77-83); row-vector g @ adj = [g_t x phi, g_t x tau + g_p x phi]
//   since (g^T K(a))^T = K(a)^T g = -a x g = g x a
template <class S> PP_HD void se3_rowvec_adj(const V3<S>& gt, const V3<S>& gp, const V3<S>& tau, const V3<S>& phi, V3<S>& ot, V3<S>& op) {
  ot = cross(gt, phi);
  op = cross(gt, tau) + cross(gp, phi);
}
// SE3_AdjXa (:754-774)
template <class S> PP_HD void se3_adj(const S* X, const S* a, S* out) {
  V3<S> ot, op;
  se3_adj_apply(X, v3(a), v3(a + 3), ot, op);
  put(ot, out);
  put(op, out + 3);
}
template <class S> PP_HD void se3_adj_bwd(const S* X, const S* out, const S* g, S* gX, S* ga) {
  typedef typename Num<S>::base T;
  V3<S> ot, op;
  se3_rowvec_adj(v3(g), v3(g + 3), v3(out), v3(out + 3), ot, op);
  put(-ot, gX);
  put(-op, gX + 3);
  gX[6] = S(T(0));
  se3_adjT_apply(X, v3(g), v3(g + 3), ot, op);
  put(ot, ga);
  put(op, ga + 3);
}
// SE3_AdjTXa (:1050-1067): out = Adj(X^-1) a; a_grad = Adj(X) g; X_grad = [-a @ adj(a_grad), 0]
template <class S> PP_HD void se3_adjt(const S* X, const S* a, S* out) {
  S Y[7];
  se3_inv(X, Y);
  se3_adj(Y, a, out);
}
template <class S> PP_HD void se3_adjt_bwd(const S* X, const S* a, const S* g, S* gX, S* ga) {
  typedef typename Num<S>::base T;
  V3<S> at, ap, ot, op;
  se3_adj_apply(X, v3(g), v3(g + 3), at, ap);
  put(at, ga);
  put(ap, ga + 3);
  se3_rowvec_adj(v3(a), v3(a + 3), at, ap, ot, op);
  put(-ot, gX);
  put(-op, gX + 3);
  gX[6] = S(T(0));
}
// SE3 Jinvp (lietensor.py:422-429): se3_Jl_inv(Log X) p = [Ji p_t - Ji Q Ji p_p, Ji p_p]
// Jinvp = Jl_inv(Log X) p (lietensor.py:422-429) without a single sincos: with phi = f v (f = SO3_Log's factor, its three
// branches, operation.py:315-322) the half angle is atan(|v|/w) (or +-pi/2 when |w| <= eps), so its sine and cosine are
// |v|/|q| and |w|/|q| -- every coefficient of Jl_inv and Q comes out of the quaternion with one atan and two square
// roots; small angles take the power series.  (se3_log + se3_jlinv_p, the composition, evaluates sincos twice.)
template <class S> PP_HD void se3_jinvp(const S* X, const S* p, S* out) {
  typedef typename Num<S>::base T;
  const V3<S> v = v3(X + 3);
  const S w = X[6];
  const S vn2 = norm2(v);
  const S vn = pp_sqrt(vn2);
  const bool vbig = pp_val(vn) > Num<T>::eps();
  const bool wbig = pp_val(pp_abs(w)) > Num<T>::eps();
  S f;
  if (vbig && wbig)
    f = pp_nan_to_num(S(T(2)) * pp_atan(vn / w) / vn);
  else if (vbig)
    f = pp_nan_to_num(pp_pm(w) * S(T(3.14159265358979323846)) / vn);
  else
    f = pp_nan_to_num(S(T(2)) * (S(T(1)) / w - vn2 / (S(T(3)) * w * w * w)));
  const V3<S> phi = f * v;
  const S th2 = norm2(phi);
  RotCoef<S> k;
  S F;
  const bool kser = pp_val(th2) < Num<T>::series2(), fser = pp_val(th2) < Num<T>::seriesF2();
  if (kser) k = rot_coef_series(th2);
  if (fser) F = rot_coef_F_series(th2);
  if (!kser || !fser) {
    const S th = pp_abs(f) * vn;
    if (!kser) {
      const S rn = S(T(1)) / pp_sqrt(vn2 + w * w);
      const S sh = vn * rn, ch = pp_abs(w) * rn;               // sin, cos of theta/2
      k.B = S(T(2)) * sh * sh / th2;
      k.C = (th - S(T(2)) * sh * ch) / (th2 * th);
      k.D = (S(T(0.5)) - k.B) / th2;
      k.E = (S(T(3)) * k.C - k.B) / (S(T(2)) * th2);
    }
    if (!fser) F = pp_nan_to_num((S(T(1)) - S(T(0.5)) * th * pp_abs(w) / vn) / th2);   // (theta/2) cot(theta/2) = (theta/2) |w|/|v|
  }
  const V3<S> tau = jlinv_apply(F, phi, v3(X));
  const V3<S> b = jlinv_apply(F, phi, v3(p + 3));
  put(b, out + 3);
  put(jlinv_apply(F, phi, v3(p) - q_apply(k, tau, phi, b)), out);
}

// ---------------------------------------------------------------------------------------
// RxSO3  [q(4), s]  /  rxso3 [phi(3), sigma]
// ---------------------------------------------------------------------------------------
template <class S> PP_HD void rxso3_exp(const S* x, S* X) {   // :447-451
  so3_exp(x, X);
  X[4] = pp_exp(x[3]);
}
template <class S> PP_HD void rxso3_log(const S* X, S* x) {   // :424-428
  so3_log(X, x);
  x[3] = pp_log(X[4]);
}
// rxso3_Jl = blockdiag(so3_Jl, 1) (:132-135)
template <class S> PP_HD void rxso3_exp_bwd(const S* x, const S* g, S* gx) {
  so3_exp_bwd(x, g, gx);
  gx[3] = g[3];
}
template <class S> PP_HD void rxso3_log_bwd(const S* y, const S* g, S* gX) {   // :436-441
  typedef typename Num<S>::base T;
  V3<S> phi = v3(y);
  put(jlinv_apply(rot_coef_F(norm2(phi)), -phi, v3(g)), gX);
  gX[3] = g[3];
  gX[4] = S(T(0));
}
template <class S> PP_HD void rxso3_act(const S* X, const S* p, S* out) {     // :574-577
  put(X[4] * quat_rotate(v3(X), X[3], v3(p)), out);
}
// RxSO3_Act.backward (:587-594): X_grad = g @ [skew(-out) | out] ; p_grad = g @ (s R) = s R^T g
template <class S> PP_HD void rxso3_act_bwd(const S* X, const S* out, const S* g, S* gX, S* gp) {
  typedef typename Num<S>::base T;
  put(cross(v3(out), v3(g)), gX);
  gX[3] = dot(v3(g), v3(out));
  gX[4] = S(T(0));
  put(X[4] * adj_rotate_T(v3(X), X[3], v3(g)), gp);
}
template <class S> PP_HD void rxso3_act4(const S* X, const S* p, S* out) {    // :677-680
  rxso3_act(X, p, out);
  out[3] = p[3];
}
// RxSO3_Act4.backward (:690-696): J (:261-265) rows 0..2 = [skew(-out3) | out3], row 3 zero;
//   p_grad = g @ Matrix4x4 (:255-258) = [s R^T g3, g4]
template <class S> PP_HD void rxso3_act4_bwd(const S* X, const S* out, const S* g, S* gX, S* gp) {
  rxso3_act_bwd(X, out, g, gX, gp);
  gp[3] = g[3];
}
template <class S> PP_HD void rxso3_mul(const S* X, const S* Y, S* Z) {       // :883-887
  quat_mul(X, Y, Z);
  Z[4] = X[4] * Y[4];
}
// RxSO3_Adj = blockdiag(R, 1) (:237-240)
template <class S> PP_HD void rxso3_mul_bwd(const S* X, const S* g, S* gX, S* gY) {
  typedef typename Num<S>::base T;
  for (int i = 0; i < 4; ++i) gX[i] = g[i];
  gX[4] = S(T(0));
  put(adj_rotate_T(v3(X), X[3], v3(g)), gY);
  gY[3] = g[3];
  gY[4] = S(T(0));
}
template <class S> PP_HD void rxso3_inv(const S* X, S* Y) {                   // :979-984
  typedef typename Num<S>::base T;
  so3_inv(X, Y);
  Y[4] = S(T(1)) / X[4];
}
template <class S> PP_HD void rxso3_inv_bwd(const S* Y, const S* g, S* gX) {  // :992-997
  typedef typename Num<S>::base T;
  put(-adj_rotate_T(v3(Y), Y[3], v3(g)), gX);
  gX[3] = -g[3];
  gX[4] = S(T(0));
}
template <class S> PP_HD void rxso3_adj(const S* X, const S* a, S* out) {     // :780-783
  put(adj_rotate(v3(X), X[3], v3(a)), out);
  out[3] = a[3];
}
// RxSO3_AdjXa.backward (:795-800): X_grad = -g @ rxso3_adj(out) (4x4 with skew(out3) top-left, :142-145)
template <class S> PP_HD void rxso3_adj_bwd(const S* X, const S* out, const S* g, S* gX, S* ga) {
  typedef typename Num<S>::base T;
  put(cross(v3(out), v3(g)), gX);
  gX[3] = S(T(0));
  gX[4] = S(T(0));
  put(adj_rotate_T(v3(X), X[3], v3(g)), ga);
  ga[3] = g[3];
}
template <class S> PP_HD void rxso3_adjt(const S* X, const S* a, S* out) {    // :1073-1076
  put(adj_rotate_T(v3(X), X[3], v3(a)), out);
  out[3] = a[3];
}
template <class S> PP_HD void rxso3_adjt_bwd(const S* X, const S* a, const S* g, S* gX, S* ga) {  // :1084-1090
  typedef typename Num<S>::base T;
  V3<S> ag = adj_rotate(v3(X), X[3], v3(g));
  put(ag, ga);
  ga[3] = g[3];
  put(cross(ag, v3(a)), gX);
  gX[3] = S(T(0));
  gX[4] = S(T(0));
}
template <class S> PP_HD void rxso3_jinvp(const S* X, const S* p, S* out) {   // lietensor.py:700-707
  S x[4];
  rxso3_log(X, x);
  rxso3_jlinv_p(x, p, out);
}

// ---------------------------------------------------------------------------------------
// Sim3  [t(3), q(4), s]  /  sim3 [tau(3), phi(3), sigma]
// ---------------------------------------------------------------------------------------
// rxso3_Ws coefficients (operation.py:85-129): W = A K + B K^2 + C I with the reference's
// four-way (sigma, theta) branch -- including its condition-3 B formula exactly as written.
template <class S> PP_HD void ws_coef(const V3<S>& phi, S sigma, S& A, S& B, S& C) {
  typedef typename Num<S>::base T;
  S th2 = norm2(phi);
  S th = pp_sqrt(th2);
  bool sl = pp_val(pp_abs(sigma)) > Num<T>::eps();
  bool tl = pp_val(th) > Num<T>::eps();
  S scale = pp_exp(sigma);
  S s2 = sigma * sigma;
  if (!sl && !tl) {
    C = S(T(1)); A = S(T(0.5)); B = S(T(1.0 / 6));
  } else if (!sl && tl) {
    C = S(T(1));
    rot_coef_BC(th2, A, B);   // (1-cos t)/t^2, (t - sin t)/t^3, cancellation-free
  } else if (sl && !tl) {
    C = pp_expm1(sigma) / sigma;                       // (e^s - 1)/s without the fp32 cancellation
    A = (S(T(1)) + (sigma - S(T(1))) * scale) / s2;
    B = (S(T(0.5)) * s2 * scale + scale - S(T(1)) - s2 * scale) / (s2 * sigma);   // sic: reference :115
  } else {
    C = pp_expm1(sigma) / sigma;
    S c = th2 + s2;
    if (sizeof(T) == 4 && pp_val(c) < T(1.0 / 64)) {
      // fp32, sigma AND theta small: the closed forms below cancel -- a sigma and (1 - b) theta agree to a fraction c / 2 of
      // themselves and 1 - b carries an absolute rounding error of one ulp of 1, so A is wrong by ~2e-7 / c (1 % at
      // sigma = theta = 3e-3; 3 rows in 300 k random ones left the 1e-5 band of the fp64 oracle, profiles/r06).  There, the
      // coefficients' own series  A = int_0^1 e^{u sigma} sin(u theta) / theta du = sum sigma^n (-theta^2)^m / (n! (2m+1)! (n+2m+2)),
      // B = int_0^1 e^{u sigma} (1 - cos(u theta)) / theta^2 du = sum sigma^n (-theta^2)^m / (n! (2m+2)! (n+2m+3)),
      // to total degree 6 (next term < 1e-10 at c = 1/64).  fp64 keeps the closed forms (their error there is 2e-16 / c).
      const S g = sigma, t = th2;
      A = S(T(1.0 / 2)) + g * (S(T(1.0 / 3)) + g * (S(T(1.0 / 8)) + g * (S(T(1.0 / 30)) + g * (S(T(1.0 / 144)) + g * (S(T(1.0 / 840)) + g * S(T(1.0 / 5760)))))))
          - t * (S(T(1.0 / 24)) + g * (S(T(1.0 / 30)) + g * (S(T(1.0 / 72)) + g * (S(T(1.0 / 252)) + g * S(T(1.0 / 1152)))))
                 - t * (S(T(1.0 / 720)) + g * (S(T(1.0 / 840)) + g * S(T(1.0 / 1920))) - t * S(T(1.0 / 40320))));
      B = S(T(1.0 / 6)) + g * (S(T(1.0 / 8)) + g * (S(T(1.0 / 20)) + g * (S(T(1.0 / 72)) + g * (S(T(1.0 / 336)) + g * S(T(1.0 / 1920))))))
          - t * (S(T(1.0 / 120)) + g * (S(T(1.0 / 144)) + g * (S(T(1.0 / 336)) + g * S(T(1.0 / 1152))))
                 - t * (S(T(1.0 / 5040)) + g * S(T(1.0 / 5760))));
    } else {
      S a = scale * pp_sin(th), b = scale * pp_cos(th);
      A = (a * sigma + (S(T(1)) - b) * th) / (th * c);
      B = (C - ((b - S(T(1))) * sigma + a * th) / c) * (S(T(1)) / th2);
    }
  }
}
// W v = A phi x v + B phi x (phi x v) + C v
template <class S> PP_HD V3<S> ws_apply(S A, S B, S C, const V3<S>& phi, const V3<S>& v) {
  V3<S> a = cross(phi, v);
  return A * a + B * cross(phi, a) + C * v;
}
// W^-1 v in closed form (the reference calls torch .inverse() on the 3x3, operation.py:473).
// W = C I + A K + B K^2 acts as C on the phi axis and as (alpha I + A K) on the plane normal
// to phi, alpha = C - B theta^2; hence W^-1 = a I + b K + c K^2 with
//   a = 1/C, b = -A/(alpha^2 + A^2 theta^2), c = (1/C - alpha/(alpha^2 + A^2 theta^2))/theta^2.
template <class S> PP_HD V3<S> ws_inv_apply(S A, S B, S C, const V3<S>& phi, const V3<S>& v) {
  typedef typename Num<S>::base T;
  S th2 = norm2(phi);
  S alpha = C - B * th2;
  S den = alpha * alpha + A * A * th2;
  S a = S(T(1)) / C;
  S b = -A / den;
  V3<S> kv = cross(phi, v);
  V3<S> r = a * v + b * kv;
  if (pp_val(th2) >= Num<T>::min_normal()) {      // (a subnormal divisor has no fast reciprocal; the term is O(theta^2) there)
    S c = (a - alpha / den) / th2;
    r = r + c * cross(phi, kv);
  }
  return r;
}
template <class S> PP_HD void sim3_exp(const S* x, S* X) {    // :495-500
  V3<S> tau = v3(x), phi = v3(x + 3);
  S A, B, C;
  ws_coef(phi, x[6], A, B, C);
  put(ws_apply(A, B, C, phi, tau), X);
  rxso3_exp(x + 3, X + 3);
}
template <class S> PP_HD void sim3_log(const S* X, S* x) {    // :470-476
  rxso3_log(X + 3, x + 3);
  V3<S> phi = v3(x + 3);
  S A, B, C;
  ws_coef(phi, x[6], A, B, C);
  put(ws_inv_apply(A, B, C, phi, v3(X)), x);
}
// sim3_adj (:147-156): Xi = [[Phi + sigma I, Tau, -tau],[0, Phi, 0],[0,0,0]];
// row-vector product g @ Xi (7 comps): [g_t x phi + sigma g_t, g_t x tau + g_p x phi, -g_t . tau]
template <class S> PP_HD void sim3_rowvec_adj(const S* g, const V3<S>& tau, const V3<S>& phi, S sigma, S* o) {
  V3<S> gt = v3(g), gp = v3(g + 3);
  put(cross(gt, phi) + sigma * gt, o);
  put(cross(gt, tau) + cross(gp, phi), o + 3);
  o[6] = -dot(gt, tau);
}
// sim3_Exp.backward (:509-513): g[:7] @ sim3_Jl, sim3_Jl = sum_{k<=5} Xi^k/(k+1)! (:159-164)
template <class S> PP_HD void sim3_exp_bwd(const S* x, const S* g, S* gx) {
  typedef typename Num<S>::base T;
  V3<S> tau = v3(x), phi = v3(x + 3);
  S sigma = x[6];
  const T cf[5] = {T(1.0 / 2), T(1.0 / 6), T(1.0 / 24), T(1.0 / 120), T(1.0 / 720)};
  S p[7], q[7];
  for (int i = 0; i < 7; ++i) { p[i] = g[i]; gx[i] = g[i]; }
  for (int k = 0; k < 5; ++k) {
    sim3_rowvec_adj(p, tau, phi, sigma, q);
    for (int i = 0; i < 7; ++i) { p[i] = q[i]; gx[i] = gx[i] + S(cf[k]) * q[i]; }
  }
}
// Sim3_Log.backward (:484-489): [g @ sim3_Jl_inv(y), 0], Jl_inv = I - Xi/2 + Xi^2/12 - Xi^4/720 (:167-172)
template <class S> PP_HD void sim3_log_bwd(const S* y, const S* g, S* gX) {
  typedef typename Num<S>::base T;
  V3<S> tau = v3(y), phi = v3(y + 3);
  S sigma = y[6];
  S p1[7], p2[7], p3[7], p4[7];
  sim3_rowvec_adj(g, tau, phi, sigma, p1);
  sim3_rowvec_adj(p1, tau, phi, sigma, p2);
  sim3_rowvec_adj(p2, tau, phi, sigma, p3);
  sim3_rowvec_adj(p3, tau, phi, sigma, p4);
  for (int i = 0; i < 7; ++i)
    gX[i] = g[i] - S(T(0.5)) * p1[i] + S(T(1.0 / 12)) * p2[i] - S(T(1.0 / 720)) * p4[i];
  gX[7] = S(T(0));
}
template <class S> PP_HD void sim3_act(const S* X, const S* p, S* out) {      // :600-603
  put(v3(X) + X[7] * quat_rotate(v3(X + 3), X[6], v3(p)), out);
}
// Sim3_Act.backward (:613-620): J = [I | skew(-out) | out]; p_grad = s R^T g
template <class S> PP_HD void sim3_act_bwd(const S* X, const S* out, const S* g, S* gX, S* gp) {
  typedef typename Num<S>::base T;
  gX[0] = g[0]; gX[1] = g[1]; gX[2] = g[2];
  put(cross(v3(out), v3(g)), gX + 3);
  gX[6] = dot(v3(g), v3(out));
  gX[7] = S(T(0));
  put(X[7] * adj_rotate_T(v3(X + 3), X[6], v3(g)), gp);
}
template <class S> PP_HD void sim3_act4(const S* X, const S* p, S* out) {     // :702-706
  put(X[7] * quat_rotate(v3(X + 3), X[6], v3(p)) + p[3] * v3(X), out);
  out[3] = p[3];
}
// Sim3_Act4.backward (:716-722): J (:297-301) = [SE3_Act4_Jacobian | out3]; p_grad = g @ Sim3_Matrix4x4
template <class S> PP_HD void sim3_act4_bwd(const S* X, const S* out, const S* g, S* gX, S* gp) {
  typedef typename Num<S>::base T;
  V3<S> g3 = v3(g);
  put(out[3] * g3, gX);
  put(cross(v3(out), g3), gX + 3);
  gX[6] = dot(g3, v3(out));
  gX[7] = S(T(0));
  put(X[7] * adj_rotate_T(v3(X + 3), X[6], g3), gp);
  gp[3] = dot(v3(X), g3) + g[3];
}
template <class S> PP_HD void sim3_mul(const S* X, const S* Y, S* Z) {        // :908-912
  put(v3(X) + X[7] * quat_rotate(v3(X + 3), X[6], v3(Y)), Z);
  quat_mul(X + 3, Y + 3, Z + 3);
  Z[7] = X[7] * Y[7];
}
// Sim3_Adj (:268-276) = [[sR, tx R, -t],[0, R, 0],[0,0,1]]
//   Adj [u;w;c] = [s R u + t x (R w) - c t, R w, c]
//   g @ Adj = [s R^T g_t, R^T (g_t x t) + R^T g_p, -g_t.t + g_s]
template <class S> PP_HD void sim3_adj_apply(const S* X, const S* a, S* o) {
  V3<S> qv = v3(X + 3);
  S qw = X[6];
  V3<S> t = v3(X);
  V3<S> rw = adj_rotate(qv, qw, v3(a + 3));
  put(X[7] * adj_rotate(qv, qw, v3(a)) + cross(t, rw) - a[6] * t, o);
  put(rw, o + 3);
  o[6] = a[6];
}
template <class S> PP_HD void sim3_adjT_apply(const S* X, const S* g, S* o) {
  V3<S> qv = v3(X + 3);
  S qw = X[6];
  V3<S> t = v3(X), gt = v3(g);
  put(X[7] * adj_rotate_T(qv, qw, gt), o);
  put(adj_rotate_T(qv, qw, cross(gt, t) + v3(g + 3)), o + 3);
  o[6] = g[6] - dot(gt, t);
}
template <class S> PP_HD void sim3_mul_bwd(const S* X, const S* g, S* gX, S* gY) {   // :921-927
  typedef typename Num<S>::base T;
  for (int i = 0; i < 7; ++i) gX[i] = g[i];
  gX[7] = S(T(0));
  sim3_adjT_apply(X, g, gY);
  gY[7] = S(T(0));
}
template <class S> PP_HD void sim3_inv(const S* X, S* Y) {                     // :1003-1008
  rxso3_inv(X + 3, Y + 3);
  put(-(Y[7] * quat_rotate(v3(Y + 3), Y[6], v3(X))), Y);
}
template <class S> PP_HD void sim3_inv_bwd(const S* Y, const S* g, S* gX) {    // :1016-1021
  typedef typename Num<S>::base T;
  S o[7];
  sim3_adjT_apply(Y, g, o);
  for (int i = 0; i < 7; ++i) gX[i] = -o[i];
  gX[7] = S(T(0));
}
template <class S> PP_HD void sim3_adj(const S* X, const S* a, S* out) { sim3_adj_apply(X, a, out); }   // :806-810
template <class S> PP_HD void sim3_adj_bwd(const S* X, const S* out, const S* g, S* gX, S* ga) {      // :820-826
  typedef typename Num<S>::base T;
  S o[7];
  sim3_rowvec_adj(g, v3(out), v3(out + 3), out[6], o);
  for (int i = 0; i < 7; ++i) gX[i] = -o[i];
  gX[7] = S(T(0));
  sim3_adjT_apply(X, g, ga);
}
template <class S> PP_HD void sim3_adjt(const S* X, const S* a, S* out) {      // :1096-1099
  S Y[8];
  sim3_inv(X, Y);
  sim3_adj_apply(Y, a, out);
}
template <class S> PP_HD void sim3_adjt_bwd(const S* X, const S* a, const S* g, S* gX, S* ga) {   // :1107-1113
  typedef typename Num<S>::base T;
  sim3_adj_apply(X, g, ga);
  S o[7];
  sim3_rowvec_adj(a, v3(ga), v3(ga + 3), ga[6], o);
  for (int i = 0; i < 7; ++i) gX[i] = -o[i];
  gX[7] = S(T(0));
}
// Xi v (column product) for Jinvp: [phi x u + sigma u + tau x w - c tau, phi x w, 0]
template <class S> PP_HD void sim3_adj_colvec(const S* v, const V3<S>& tau, const V3<S>& phi, S sigma, S* o) {
  typedef typename Num<S>::base T;
  V3<S> u = v3(v), w = v3(v + 3);
  put(cross(phi, u) + sigma * u + cross(tau, w) - v[6] * tau, o);
  put(cross(phi, w), o + 3);
  o[6] = S(T(0));
}
template <class S> PP_HD void sim3_jinvp(const S* X, const S* p, S* out) {     // lietensor.py:556-563
  S x[7];
  sim3_log(X, x);
  sim3_jlinv_p(x, p, out);
}

// ---------------------------------------------------------------------------------------
// Jinvp backward (the reference differentiates Jl_inv(Log X) p with plain autograd through
// so3_Jl_inv / calcQ and through <Group>_Log's custom backward; lietensor.py:257-264, 422-429,
// 556-563, 700-707):
//   x = Log X ;  f(x, p) = Jl_inv(x) p
//   gp = Jl_inv(x)^T g                      (= the tangent part of <Group>_Log.backward(x, g))
//   h  = d(g . f)/dx   by dx forward-mode sweeps (Dual<T>)
//   gX = <Group>_Log.backward(x, h) = [h @ Jl_inv(x), 0]
// ---------------------------------------------------------------------------------------
template <class S> PP_HD void so3_jlinv_p(const S* x, const S* p, S* out) {
  V3<S> phi = v3(x);
  put(jlinv_apply(rot_coef_F(norm2(phi)), phi, v3(p)), out);
}
template <class S> PP_HD void se3_jlinv_p(const S* x, const S* p, S* out) {
  V3<S> tau = v3(x), phi = v3(x + 3);
  S th2 = norm2(phi);
  RotCoef<S> k = rot_coef(th2);
  S F = rot_coef_F(th2);
  V3<S> b = jlinv_apply(F, phi, v3(p + 3));
  put(b, out + 3);
  put(jlinv_apply(F, phi, v3(p) - q_apply(k, tau, phi, b)), out);
}
template <class S> PP_HD void rxso3_jlinv_p(const S* x, const S* p, S* out) {
  so3_jlinv_p(x, p, out);
  out[3] = p[3];
}
template <class S> PP_HD void sim3_jlinv_p(const S* x, const S* p, S* out) {
  typedef typename Num<S>::base T;
  V3<S> tau = v3(x), phi = v3(x + 3);
  S sigma = x[6];
  S p1[7], p2[7], p3[7], p4[7];
  sim3_adj_colvec(p, tau, phi, sigma, p1);
  sim3_adj_colvec(p1, tau, phi, sigma, p2);
  sim3_adj_colvec(p2, tau, phi, sigma, p3);
  sim3_adj_colvec(p3, tau, phi, sigma, p4);
  for (int i = 0; i < 7; ++i)
    out[i] = p[i] - S(T(0.5)) * p1[i] + S(T(1.0 / 12)) * p2[i] - S(T(1.0 / 720)) * p4[i];
}

#define PPLIE_JINVP_BWD(g, DA, DG)                                                                  \
  template <class T> PP_HD void g##_jinvp_bwd(const T* X, const T* p, const T* gr, T* gX, T* gp) { \
    T x[DA], h[DA];                                                                                 \
    g##_log<T>(X, x);                                                                               \
    for (int k = 0; k < DA; ++k) {                                                                  \
      Dual<T> xd[DA], pd[DA], od[DA];                                                               \
      for (int i = 0; i < DA; ++i) {                                                                \
        xd[i] = Dual<T>(x[i], i == k ? T(1) : T(0));                                                \
        pd[i] = Dual<T>(p[i]);                                                                      \
      }                                                                                             \
      g##_jlinv_p<Dual<T>>(xd, pd, od);                                                             \
      T acc = T(0);                                                                                 \
      for (int i = 0; i < DA; ++i) acc += gr[i] * od[i].d;                                          \
      h[k] = acc;                                                                                   \
    }                                                                                               \
    g##_log_bwd<T>(x, h, gX);                                                                       \
    T tmp[DG];                                                                                      \
    g##_log_bwd<T>(x, gr, tmp);                                                                     \
    for (int i = 0; i < DA; ++i) gp[i] = tmp[i];                                                    \
  }
PPLIE_JINVP_BWD(sim3, 7, 8)

// so3 / rxso3 / se3: the generic macro re-evaluates the coefficient functions (series or sincos closed forms)
// with Dual arithmetic in every one of its DA sweeps.  They depend on theta^2 only, so their values and
// theta^2-derivatives are taken ONCE (one Dual evaluation seeded with d theta^2 = 1) and every sweep only chains
// d theta^2 / d phi_k = 2 phi_k through the cross products (se3: 1.45 ms -> see DESIGN 3.2 at 10^7 rows).
template <class T> PP_HD Dual<T> chain(Dual<T> f, T dth2) { return Dual<T>(f.v, f.d * dth2); }

template <class T> PP_HD void so3_jinvp_bwd(const T* X, const T* p, const T* gr, T* gX, T* gp) {
  T x[3], h[3];
  so3_log<T>(X, x);
  const Dual<T> F1 = rot_coef_F(Dual<T>(x[0] * x[0] + x[1] * x[1] + x[2] * x[2], T(1)));
  for (int k = 0; k < 3; ++k) {
    Dual<T> xd[3], pd[3];
    for (int i = 0; i < 3; ++i) { xd[i] = Dual<T>(x[i], i == k ? T(1) : T(0)); pd[i] = Dual<T>(p[i]); }
    V3<Dual<T>> o = jlinv_apply(chain(F1, T(2) * x[k]), v3(xd), v3(pd));
    h[k] = gr[0] * o.x.d + gr[1] * o.y.d + gr[2] * o.z.d;
  }
  so3_log_bwd<T>(x, h, gX);
  T tmp[4];
  so3_log_bwd<T>(x, gr, tmp);
  for (int i = 0; i < 3; ++i) gp[i] = tmp[i];
}
template <class T> PP_HD void rxso3_jinvp_bwd(const T* X, const T* p, const T* gr, T* gX, T* gp) {
  // rxso3_Jl_inv = blockdiag(so3_Jl_inv, 1): the scale component passes straight through
  T x[4], h[4];
  rxso3_log<T>(X, x);
  const Dual<T> F1 = rot_coef_F(Dual<T>(x[0] * x[0] + x[1] * x[1] + x[2] * x[2], T(1)));
  for (int k = 0; k < 3; ++k) {
    Dual<T> xd[3], pd[3];
    for (int i = 0; i < 3; ++i) { xd[i] = Dual<T>(x[i], i == k ? T(1) : T(0)); pd[i] = Dual<T>(p[i]); }
    V3<Dual<T>> o = jlinv_apply(chain(F1, T(2) * x[k]), v3(xd), v3(pd));
    h[k] = gr[0] * o.x.d + gr[1] * o.y.d + gr[2] * o.z.d;
  }
  h[3] = T(0);
  rxso3_log_bwd<T>(x, h, gX);
  T tmp[5];
  rxso3_log_bwd<T>(x, gr, tmp);
  for (int i = 0; i < 4; ++i) gp[i] = tmp[i];
}
// reverse-mode pieces (for c = a x b: abar += b x cbar, bbar += cbar x a)
//   y = Jl_inv(phi) w = w - 1/2 a1 + F a2, a1 = phi x w, a2 = phi x a1
template <class T> PP_HD void jlinv_apply_rev(T F, const V3<T>& phi, const V3<T>& w, const V3<T>& ybar, V3<T>& phibar, V3<T>& wbar, T& Fbar) {
  const V3<T> a1 = cross(phi, w), a2 = cross(phi, a1);
  Fbar += dot(ybar, a2);
  const V3<T> a2bar = F * ybar;
  const V3<T> a1bar = T(-0.5) * ybar + cross(a2bar, phi);
  phibar = phibar + cross(a1, a2bar) + cross(w, a1bar);
  wbar = wbar + ybar + cross(a1bar, phi);
}
//   q = Q(tau, phi) v (q_apply above), adjoints of tau, phi, v and of the coefficients C, D, E
template <class T> PP_HD void q_apply_rev(const RotCoef<T>& k, const V3<T>& tau, const V3<T>& phi, const V3<T>& v, const V3<T>& qbar,
                                          V3<T>& taubar, V3<T>& phibar, V3<T>& vbar, T& Cbar, T& Dbar, T& Ebar) {
  const V3<T> pv = cross(phi, v), tv = cross(tau, v), ppv = cross(phi, pv), tpv = cross(tau, pv), ptv = cross(phi, tv);
  const V3<T> ptpv = cross(phi, tpv), pptv = cross(phi, ptv), tppv = cross(tau, ppv), ptppv = cross(phi, tppv), pptpv = cross(phi, ptpv);
  Cbar += dot(qbar, ptv + tpv + ptpv);
  Dbar += dot(qbar, pptv + tppv - T(3) * ptpv);
  Ebar += dot(qbar, ptppv + pptpv);
  V3<T> tvb = T(0.5) * qbar, ptvb = k.C * qbar, tpvb = k.C * qbar, ptpvb = (k.C - T(3) * k.D) * qbar;
  const V3<T> pptvb = k.D * qbar, ptppvb = k.E * qbar, pptpvb = k.E * qbar;
  V3<T> tppvb = k.D * qbar;
  phibar = phibar + cross(ptpv, pptpvb);  ptpvb = ptpvb + cross(pptpvb, phi);      // pptpv = phi x ptpv
  phibar = phibar + cross(tppv, ptppvb);  tppvb = tppvb + cross(ptppvb, phi);      // ptppv = phi x tppv
  taubar = taubar + cross(ppv, tppvb);    V3<T> ppvb = cross(tppvb, tau);          // tppv  = tau x ppv
  phibar = phibar + cross(ptv, pptvb);    ptvb = ptvb + cross(pptvb, phi);         // pptv  = phi x ptv
  phibar = phibar + cross(tpv, ptpvb);    tpvb = tpvb + cross(ptpvb, phi);         // ptpv  = phi x tpv
  phibar = phibar + cross(tv, ptvb);      tvb = tvb + cross(ptvb, phi);            // ptv   = phi x tv
  taubar = taubar + cross(pv, tpvb);      V3<T> pvb = cross(tpvb, tau);            // tpv   = tau x pv
  phibar = phibar + cross(pv, ppvb);      pvb = pvb + cross(ppvb, phi);            // ppv   = phi x pv
  taubar = taubar + cross(v, tvb);        vbar = vbar + cross(tvb, tau);           // tv    = tau x v
  phibar = phibar + cross(v, pvb);        vbar = vbar + cross(pvb, phi);           // pv    = phi x v
}

// se3: ONE reverse sweep through  b = Jinv(phi) p_phi,  t = Jinv(phi) (p_tau - Q(tau, phi) b)  instead of six
// forward (Dual) sweeps that each redo the value computation; the coefficient functions and their theta^2
// derivatives come from one Dual evaluation.
template <class T> PP_HD void se3_jinvp_bwd(const T* X, const T* p, const T* gr, T* gX, T* gp) {
  T x[6], h[6];
  se3_log<T>(X, x);
  const V3<T> tau = v3(x), phi = v3(x + 3);
  const Dual<T> th2(norm2(phi), T(1));
  const RotCoef<Dual<T>> k1 = rot_coef(th2);
  const Dual<T> F1 = rot_coef_F(th2);
  RotCoef<T> k;
  k.B = k1.B.v; k.C = k1.C.v; k.D = k1.D.v; k.E = k1.E.v;
  const T F = F1.v;
  // forward values
  const V3<T> b = jlinv_apply(F, phi, v3(p + 3));
  const V3<T> w = v3(p) - q_apply(k, tau, phi, b);
  // reverse
  const V3<T> z = v3<T>(T(0), T(0), T(0));
  V3<T> phibar = z, taubar = z, wbar = z, bbar = v3(gr + 3), ppbar = z;
  T Fbar = T(0), Cbar = T(0), Dbar = T(0), Ebar = T(0);
  jlinv_apply_rev(F, phi, w, v3(gr), phibar, wbar, Fbar);                     // t = Jinv(phi) w
  q_apply_rev(k, tau, phi, b, -wbar, taubar, phibar, bbar, Cbar, Dbar, Ebar);  // w = p_tau - Q b
  jlinv_apply_rev(F, phi, v3(p + 3), bbar, phibar, ppbar, Fbar);              // b = Jinv(phi) p_phi
  const T th2bar = Fbar * F1.d + Cbar * k1.C.d + Dbar * k1.D.d + Ebar * k1.E.d;
  phibar = phibar + (T(2) * th2bar) * phi;
  put(taubar, h);
  put(phibar, h + 3);
  se3_log_bwd<T>(x, h, gX);
  T tmp[7];
  se3_log_bwd<T>(x, gr, tmp);
  for (int i = 0; i < 6; ++i) gp[i] = tmp[i];
}

// so3.Jr backward: gx_k = sum_ij G_ij dJr_ij/dx_k (the reference: autograd through lietensor.py:343-351)
template <class T> PP_HD void so3_jr_bwd(const T* x, const T* G, T* gx) {
  for (int k = 0; k < 3; ++k) {
    Dual<T> xd[3], J[9];
    for (int i = 0; i < 3; ++i) xd[i] = Dual<T>(x[i], i == k ? T(1) : T(0));
    so3_jr<Dual<T>>(xd, J);
    T acc = T(0);
    for (int i = 0; i < 9; ++i) acc += G[i] * J[i].d;
    gx[k] = acc;
  }
}

}  // namespace pplie
