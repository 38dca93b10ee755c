// lm_generic.hip -- a whole Levenberg-Marquardt STEP on the device for ANY residual of the form
//
//     r = Log(L * P^s * R) [- b]          (kind 0)          or          r = (L * P^s * R) . a [- b]       (kind 1, points)
//
// with P the optimised parameter (one group element per problem: SO3 / SE3 / Sim3 / RxSO3), s = +1 or -1, and L, R constant
// group elements per problem (either may be absent).  Every chain of Mul / Inv over one occurrence of P and any number of
// constants reduces to this normal form (the host folds the constants to the left and to the right of P into L and R once,
// pypose_amd/optim/fused.py match_lpr), so one kernel family covers what used to fall off the 30x cliff between the
// hand-derived InvNet program (csrc/lm_step.hip: r = Log(P X)) and the generic block path (d_res batched autograd sweeps
// through HBM, then separate normal-equation / Cholesky / retraction / loss launches):  Log(P^-1 X),  P.Act(a) - b,  Log(A P B),
// their SO3 / Sim3 / RxSO3 variants ...
//
// Per problem, in registers (one problem per lane): the forward chain, the d_res rows of the Jacobian by the reference's own
// backward rules applied to unit cotangents --
//     e_i @ Jl_inv(y)        <g>_log_bwd   (pypose/lietensor/operation.py:385-395)        or  e_i @ J_act(q)   <g>_act_bwd (:535-543)
//     . @ Adj(L)             <g>_mul_bwd   (:846-852: Y_grad = g @ Adj(X))
//     -(.) @ Adj(P^-1)       <g>_inv_bwd   (:945-949), when s = -1
// -- i.e. the batched backward sweeps of the block path, in registers (the unit vectors fold at compile time); then
// A = J^T J with the clamped, damped diagonal (optimizer.py:655-657, :666), Cholesky (solver.py:213-216), the retraction
// P' = Exp(d) P (lietensor.py:60-65), the residual again at P' (:673) and the two dot products of the gain ratio
// (strategy.py:144, :261).  Loop state, decision and retries exactly as in lm_step.hip (lm_common.h).
#include "rowmap.h"
#include "chol.h"
#include "gridsync.h"
#include "lm_common.h"

namespace pplie {

template <class T, int GID> struct Grp;
#define PPLIE_GRP(ID, g, DA_, DG_)                                                                            \
  template <class T> struct Grp<T, ID> {                                                                      \
    enum { DA = DA_, DG = DG_ };                                                                              \
    static PP_HD void exp(const T* x, T* X) { g##_exp<T>(x, X); }                                             \
    static PP_HD void log(const T* X, T* x) { g##_log<T>(X, x); }                                             \
    static PP_HD void mul(const T* X, const T* Y, T* Z) { g##_mul<T>(X, Y, Z); }                              \
    static PP_HD void inv(const T* X, T* Y) { g##_inv<T>(X, Y); }                                             \
    static PP_HD void log_bwd(const T* y, const T* c, T* gX) { g##_log_bwd<T>(y, c, gX); }                    \
    static PP_HD void mul_bwd(const T* X, const T* c, T* gX, T* gY) { g##_mul_bwd<T>(X, c, gX, gY); }         \
    static PP_HD void inv_bwd(const T* Y, const T* c, T* gX) { g##_inv_bwd<T>(Y, c, gX); }                    \
    static PP_HD void act(const T* X, const T* p, T* o) { g##_act<T>(X, p, o); }                              \
    static PP_HD void act_bwd(const T* X, const T* o, const T* c, T* gX, T* gp) { g##_act_bwd<T>(X, o, c, gX, gp); } \
  };
PPLIE_GRP(0, so3, 3, 4)
PPLIE_GRP(1, se3, 6, 7)
PPLIE_GRP(2, sim3, 7, 8)
PPLIE_GRP(3, rxso3, 4, 5)
#undef PPLIE_GRP

// the operands of one program (device pointers; L, R, b may be null; a only for kind 1)
struct LprArgs {
  const void* L; const void* R; const void* a; const void* b;
  int sign;         // +1: P,  -1: P^-1
};

template <int W, class T> __device__ __forceinline__ void ld_row(const T* p, int64_t row, T* r) {
#pragma unroll
  for (int k = 0; k < W; ++k) r[k] = p[row * W + k];
}
template <int W, class T> __device__ __forceinline__ void st_row(T* p, int64_t row, const T* r) {
#pragma unroll
  for (int k = 0; k < W; ++k) p[row * W + k] = r[k];
}

// forward chain at the pose `P`: Y = P^s, Z = L Y R, and the residual (y = Log Z for kind 0, q = Z . a for kind 1)
// (hasL / hasR / hasB instead of null pointers: a local array whose ADDRESS is selected at run time cannot stay in registers)
template <class T, int GID, int KIND>
PP_HD void lpr_forward(const T* P, const T* Lr, bool hasL, const T* Rr, bool hasR, int sign, const T* av, const T* bv, bool hasB, T* Y,
                       T* Z, T* y, T* r) {
  typedef Grp<T, GID> G;
  constexpr int DG = G::DG, DR = KIND == 0 ? (int)G::DA : 3;
  if (sign > 0) {
#pragma unroll
    for (int k = 0; k < DG; ++k) Y[k] = P[k];
  } else {
    G::inv(P, Y);
  }
  T W[DG];
  if (hasL) G::mul(Lr, Y, W);
  else {
#pragma unroll
    for (int k = 0; k < DG; ++k) W[k] = Y[k];
  }
  if (hasR) G::mul(W, Rr, Z);
  else {
#pragma unroll
    for (int k = 0; k < DG; ++k) Z[k] = W[k];
  }
  if (KIND == 0) G::log(Z, y);
  else G::act(Z, av, y);
#pragma unroll
  for (int k = 0; k < DR; ++k) r[k] = hasB ? y[k] - bv[k] : y[k];
}

// one LM trial of one problem: candidate pose `pn` and the four summands
template <class T, int GID, int KIND>
PP_HD void lm_lpr_row(const T* plin, const T* Lr, bool hasL, const T* Rr, bool hasR, int sign, const T* av, const T* bv, bool hasB, T s,
                      T dmin, T dmax, T* pn, T& a_new, T& a_old, T& a_jj, T& a_jr) {
  typedef Grp<T, GID> G;
  constexpr int DA = G::DA, DG = G::DG, DR = KIND == 0 ? (int)G::DA : 3;
  T Y[DG], Z[DG], y[DR > 3 ? DR : 3], r[DR];
  lpr_forward<T, GID, KIND>(plin, Lr, hasL, Rr, hasR, sign, av, bv, hasB, Y, Z, y, r);
  // Jacobian rows: the reference's backward rules on unit cotangents
  T J[DR][DA];
#pragma unroll
  for (int i = 0; i < DR; ++i) {
    T g1[DG];
    if (KIND == 0) {
      T e[DA];
#pragma unroll
      for (int k = 0; k < DA; ++k) e[k] = k == i ? T(1) : T(0);
      G::log_bwd(y, e, g1);                              // [e_i @ Jl_inv(y), 0]
    } else {
      T e[3], gp[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) e[k] = k == i ? T(1) : T(0);
      G::act_bwd(Z, y, e, g1, gp);                       // [e_i @ J_act(q), 0]
    }
    if (hasL) {
      T gx[DG], gy[DG];
      G::mul_bwd(Lr, g1, gx, gy);                        // through Z = L (Y R): . @ Adj(L)
#pragma unroll
      for (int k = 0; k < DG; ++k) g1[k] = gy[k];
    }
    if (sign < 0) {
      T gx[DG];
      G::inv_bwd(Y, g1, gx);                             // through Y = P^-1: -(.) @ Adj(Y)
#pragma unroll
      for (int k = 0; k < DG; ++k) g1[k] = gx[k];
    }
#pragma unroll
    for (int k = 0; k < DA; ++k) J[i][k] = g1[k];
  }
  // normal equations, clamp + damping on the diagonal
  T A[DA * DA], g[DA], lam[DA];
#pragma unroll
  for (int p = 0; p < DA; ++p) {
    T gs = T(0);
#pragma unroll
    for (int i = 0; i < DR; ++i) gs += J[i][p] * r[i];
    g[p] = gs;
#pragma unroll
    for (int q = 0; q <= p; ++q) {
      T as = T(0);
#pragma unroll
      for (int i = 0; i < DR; ++i) as += J[i][p] * J[i][q];
      A[p * DA + q] = as;
      A[q * DA + p] = as;
    }
  }
#pragma unroll
  for (int p = 0; p < DA; ++p) {
    const T d0 = A[p * DA + p];
    T d = d0 < dmin ? dmin : (d0 > dmax ? dmax : d0);
    d *= s;
    A[p * DA + p] = d;
    lam[p] = d - d0;
  }
  T d[DG];
  Op_chol_solve<T, DA>::apply(A, g, nullptr, d, nullptr);
  T E[DG];
  G::exp(d, E);
  G::mul(E, plin, pn);
  // the new loss at the candidate as stored
  T Y2[DG], Z2[DG], y2[DR > 3 ? DR : 3], r2[DR];
  lpr_forward<T, GID, KIND>(pn, Lr, hasL, Rr, hasR, sign, av, bv, hasB, Y2, Z2, y2, r2);
  T nn = T(0), oo = T(0), jr = T(0), ld = T(0);
#pragma unroll
  for (int i = 0; i < DR; ++i) { nn += r2[i] * r2[i]; oo += r[i] * r[i]; }
#pragma unroll
  for (int p = 0; p < DA; ++p) { jr += d[p] * g[p]; ld += lam[p] * d[p] * d[p]; }
  // (J d).r = d.g ;  |J d|^2 = d^T (J^T J) d = -d.g - sum_p Lambda_p d_p^2   (since (J^T J + Lambda) d = -g)
  a_new += nn; a_old += oo; a_jj += -jr - ld; a_jr += jr;
}

// trial of the rows  tid, tid + stride, ...: reads the linearisation point from Plin, writes the candidate to Pout (may alias
// Plin: a lane reads its row before it writes it) and, if given, Plin's rows to save; one row of partial sums per workgroup
template <class T, int GID, int KIND, int BLOCK>
__device__ __forceinline__ void lm_lpr_rows(const T* Plin, T* Pout, T* save, const LprArgs& ar, T s, T dmin, T dmax, int64_t n,
                                            T* partial_row) {
  typedef Grp<T, GID> G;
  constexpr int DG = G::DG, DW = KIND == 0 ? (int)G::DA : 3;
  T a_new = T(0), a_old = T(0), a_jj = T(0), a_jr = T(0);
  const T* L = static_cast<const T*>(ar.L);
  const T* R = static_cast<const T*>(ar.R);
  const T* av = static_cast<const T*>(ar.a);
  const T* bv = static_cast<const T*>(ar.b);
  for (int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x; row < n; row += (int64_t)gridDim.x * BLOCK) {
    T p[DG], l[DG], rr[DG], a3[3], b[DW], pn[DG];
    ld_row<DG>(Plin, row, p);
#pragma unroll
    for (int k = 0; k < DG; ++k) { l[k] = T(0); rr[k] = T(0); }
#pragma unroll
    for (int k = 0; k < DW; ++k) b[k] = T(0);
    a3[0] = a3[1] = a3[2] = T(0);
    if (L) ld_row<DG>(L, row, l);
    if (R) ld_row<DG>(R, row, rr);
    if (KIND == 1) ld_row<3>(av, row, a3);
    if (bv) ld_row<DW>(bv, row, b);
    if (save) st_row<DG>(save, row, p);
    lm_lpr_row<T, GID, KIND>(p, l, L != nullptr, rr, R != nullptr, ar.sign, a3, b, bv != nullptr, s, dmin, dmax, pn, a_new, a_old, a_jj, a_jr);
    st_row<DG>(Pout, row, pn);
  }
  T v0 = wg_sum<T, BLOCK>(a_new), v1 = wg_sum<T, BLOCK>(a_old), v2 = wg_sum<T, BLOCK>(a_jj), v3_ = wg_sum<T, BLOCK>(a_jr);
  if (threadIdx.x == 0) { partial_row[0] = v0; partial_row[1] = v1; partial_row[2] = v2; partial_row[3] = v3_; }
}

constexpr int kLprBlock = 256;

template <class T, int GID, int KIND>
__global__ void __launch_bounds__(kLprBlock)
lm_lpr_trial_kernel(T* P, T* save, LprArgs ar, T* partials, const double* st, LmCfg cfg, int64_t n) {
  const double sc = 1.0 + ((cfg.flags & LMF_HOST_STATE) ? cfg.host_damping : st[ST_DAMPING]);
  lm_lpr_rows<T, GID, KIND, kLprBlock>(P, P, save, ar, (T)sc, (T)cfg.dmin, (T)cfg.dmax, n, partials + (size_t)blockIdx.x * 4);
}

// decision + retries: the structure of lm_se3inv_finish_kernel (lm_step.hip)
template <class T, int GID, int KIND>
__global__ void __launch_bounds__(kLprBlock)
lm_lpr_finish_kernel(T* P, T* save, LprArgs ar, T* partials, int first_rows, const double* st_in, double* st_out, unsigned* bar,
                     LmCfg cfg, int64_t n, T* loss_out, T* last_out) {
  typedef Grp<T, GID> G;
  __shared__ double verdict[2];
  {
    double v[4];
    lm_reduce_partials<T, kLprBlock, true>(partials, first_rows, v);
    if (threadIdx.x == 0) {
      double o[ST_SIZE];
      lm_decide(st_in, o, cfg, true, v[0], v[1], v[2], v[3]);
      if (blockIdx.x == 0) lm_store_state<T>(o, st_out, loss_out, last_out);
      verdict[0] = o[ST_DONE];
      verdict[1] = o[ST_FAILED];
    }
    __syncthreads();
  }
  if (verdict[1] != 0.0) {                 // the reference's solver raised before the update (solver.py:214): P unchanged
    for (int64_t i = (int64_t)blockIdx.x * kLprBlock + threadIdx.x; i < n * G::DG; i += (int64_t)gridDim.x * kLprBlock) P[i] = save[i];
    return;
  }
  if (verdict[0] != 0.0) return;
  __threadfence();
  grid_barrier(bar, bar + 1);
  for (int it = 0; it <= cfg.reject; ++it) {
    const T s = (T)(ld_state(st_out, ST_SCALE) * (1.0 + ld_state(st_out, ST_DAMPING)));
    lm_lpr_rows<T, GID, KIND, kLprBlock>(save, P, nullptr, ar, s, (T)cfg.dmin, (T)cfg.dmax, n, partials + (size_t)blockIdx.x * 4);
    __threadfence();
    grid_barrier(bar, bar + 1);
    if (blockIdx.x == 0) {
      double v[4];
      lm_reduce_partials<T, kLprBlock, false>(partials, gridDim.x, v);
      if (threadIdx.x == 0) {
        double in[ST_SIZE], o[ST_SIZE];
#pragma unroll
        for (int i = 0; i <= ST_PL_STOP; ++i) in[i] = ld_state(st_out, i);
        lm_decide(in, o, cfg, false, v[0], v[1], v[2], v[3]);
        lm_store_state<T>(o, st_out, loss_out, last_out);
        __threadfence();
      }
    }
    grid_barrier(bar, bar + 1);
    if (ld_state(st_out, ST_FAILED) != 0.0) {
      for (int64_t i = (int64_t)blockIdx.x * kLprBlock + threadIdx.x; i < n * G::DG; i += (int64_t)gridDim.x * kLprBlock) P[i] = save[i];
      return;
    }
    if (ld_state(st_out, ST_DONE) != 0.0) return;
  }
}

template <class T, int GID, int KIND>
int lm_lpr_step_launch(void* P, const LprArgs& ar, void* save, void* partials, const void* st_in, void* st_out, void* sync,
                       const LmCfg* cfg, int64_t n, void* loss_out, void* last_out, hipStream_t s) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const int64_t nt = (n + kLprBlock - 1) / kLprBlock;
  int cap = cfg->grid_cap > 0 ? cfg->grid_cap : 8 * cus;
  if (cap > kStepPartials) cap = kStepPartials;
  const int grid = (int)(nt < cap ? nt : cap);
  hipLaunchKernelGGL((lm_lpr_trial_kernel<T, GID, KIND>), dim3(grid), dim3(kLprBlock), 0, s, (T*)P, (T*)save, ar, (T*)partials,
                     (const double*)st_in, *cfg, n);
  if (hipGetLastError() != hipSuccess) return PPLIE_ELAUNCH;
  const int fgrid = (int)(nt < kFinishGrid ? nt : kFinishGrid);
  hipLaunchKernelGGL((lm_lpr_finish_kernel<T, GID, KIND>), dim3(fgrid), dim3(kLprBlock), 0, s, (T*)P, (T*)save, ar, (T*)partials, grid,
                     (const double*)st_in, (double*)st_out, (unsigned*)sync, *cfg, n, (T*)loss_out, (T*)last_out);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

template <class T>
int lm_lpr_step(int group, int kind, int sign, void* P, const void* L, const void* R, const void* a, const void* b, void* save,
                void* partials, const void* st_in, void* st_out, void* sync, const LmCfg* cfg, int64_t n, void* loss_out, void* last_out,
                void* stream) {
  if (n < 0 || !cfg || !P || !save || !partials || !st_in || !st_out || !sync || st_in == st_out) return PPLIE_EBADARG;
  if ((sign != 1 && sign != -1) || (kind != 0 && kind != 1) || (kind == 1 && !a) || group < 0 || group > 3) return PPLIE_EBADARG;
  if (n == 0) return PPLIE_OK;
  const LprArgs ar = {L, R, a, b, sign};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define CASE(G, K) \
  if (group == G && kind == K) return lm_lpr_step_launch<T, G, K>(P, ar, save, partials, st_in, st_out, sync, cfg, n, loss_out, last_out, s);
  CASE(0, 0) CASE(0, 1) CASE(1, 0) CASE(1, 1) CASE(2, 0) CASE(2, 1) CASE(3, 0) CASE(3, 1)
#undef CASE
  return PPLIE_EBADARG;
}
}  // namespace pplie

extern "C" int pplie_lm_lpr_step_f32(int group, int kind, int sign, void* P, const void* L, const void* R, const void* a, const void* b,
                                     void* save, void* partials, const void* st_in, void* st_out, void* sync, const void* cfg, int64_t n,
                                     void* loss_out, void* last_out, void* stream) {
  return pplie::lm_lpr_step<float>(group, kind, sign, P, L, R, a, b, save, partials, st_in, st_out, sync,
                                   static_cast<const pplie::LmCfg*>(cfg), n, loss_out, last_out, stream);
}
extern "C" int pplie_lm_lpr_step_f64(int group, int kind, int sign, void* P, const void* L, const void* R, const void* a, const void* b,
                                     void* save, void* partials, const void* st_in, void* st_out, void* sync, const void* cfg, int64_t n,
                                     void* loss_out, void* last_out, void* stream) {
  return pplie::lm_lpr_step<double>(group, kind, sign, P, L, R, a, b, save, partials, st_in, st_out, sync,
                                    static_cast<const pplie::LmCfg*>(cfg), n, loss_out, last_out, stream);
}
