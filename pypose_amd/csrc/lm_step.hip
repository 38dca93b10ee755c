// lm_step.hip -- a whole Levenberg-Marquardt STEP of B independent pose-inversion problems on the device
// (BASELINE configs[2]: the reference's README InvNet, README.md:120-129,  r_b = Log(P_b * X_b)).
//
// pypose/optim/optimizer.py:644-679 is a host loop: linearise, then { damp, solve, update, loss, strategy.update,
// accept / reject } until a trial is accepted, every `<` on a device scalar a synchronisation.  Here the loop's
// state lives in device memory and the decisions are taken there:
//
//   lmstate (double[PPLIE_LM_STATE]):  damping, radius, down (the three entries of the param group the strategies
//            rewrite, strategy.py:143-151, 260-274), the compounded damping factor of the current step
//            (optimizer.py:666: A.diag += A.diag * damping on every retry), last, loss, reject count, flags
//   trial kernel   one problem per lane, everything between the 56 B it reads (P, X) and the 28 + 28 B it writes
//            (P' in place, the linearisation point P into `save` for a possible retry) stays in registers:
//            r = Log(P X), J = se3_Jl_inv(r) = [[Ji, -Ji Q Ji], [0, Ji]]  (operation.py:68-75), A = J^T J with the
//            clamped diagonal times the damping factor (:655-657, :666), 6x6 Cholesky (solver.py:213-216),
//            P' = Exp(d) P (lietensor.py:442-444), |Log(P' X)|^2 (:673) and the two dot products of the gain ratio
//            (strategy.py:144, :261).  Ends in one row of partial sums per workgroup.
//   finish kernel  launched right behind the trial kernel, 64 workgroups.  Every workgroup adds up the partial
//            sums in the same fixed order and runs `decide` (strategy update, accept / reject, reject count, loss) on
//            them -- identical bits everywhere, so all workgroups know the outcome without communicating; workgroup 0
//            stores the new state.  Accepted (the common case): done.  Otherwise it repeats { trial from the saved
//            linearisation point with the compounded damping, grid barrier, decide, grid barrier } until a trial is
//            accepted (the reference's loop always ends in an accepted trial: the reject count is bounded), or
//            restores P when the factorisation failed.  The state is double-buffered (st_in of a step is st_out of
//            the previous one) so that no workgroup reads a word another one is writing.
//
// A step is therefore two launches and no host synchronisation; the host mirrors of loss / damping are read back
// lazily (pypose_amd/optim/fused.py).  Sharded runs (LM(group=...)) use the same trial kernel without the in-kernel
// decision, all-reduce the four sums and launch `decide` as a one-thread kernel.
//
// Arithmetic: J is never formed.  With K = [phi]x, theta^2 = |phi|^2, s = phi.tau:
//   Ji       = (1 - F theta^2) I + F phi phi^T - K/2                       (so3_Jl_inv, operation.py:23-32)
//   Ji^T Ji  = I + c K^2,  c = 2F - 1/4 - F^2 theta^2                       (K^4 = -theta^2 K^2)
//   Q        = [(1/2 - D theta^2) tau + s (2D - C) phi]x + C (tau phi^T + phi tau^T) - 2 s E phi phi^T
//              + 2 s (E theta^2 - C) I                                      (calcQ, operation.py:37-58, every
//              Phi..Tau..Phi product reduced with  [a]x[b]x = b a^T - (a.b) I  and  [a]x[b]x[a]x = -(a.b)[a]x)
//   N = Q Ji;  A11 = Ji^T Ji;  A12 = -A11 N;  A22 = A11 - N^T A12;  g = [Ji^T tau, phi - N^T Ji^T tau]
//   J d = [Ji (d_tau - N d_phi), Ji d_phi]
#include "rowmap.h"
#include "chol.h"
#include "gridsync.h"
#include "lm_common.h"

namespace pplie {

#if defined(__HIP_DEVICE_COMPILE__)
template <int W, class T> PP_HD void row_get(const T* p, T* r) { row_ld<W>(p, r); }
#else
template <int W, class T> PP_HD void row_get(const T* p, T* r) { for (int i = 0; i < W; ++i) r[i] = p[i]; }
#endif

// SE3 Log that also hands back theta^2 and the so3_Jl_inv coefficient F it computed (se3_log in lie_math.h)
template <class T> PP_HD void se3_log_ext(const T* X, T* x, T& th2, T& F, T& sh, T& ch, bool& regular) {
  V3<T> v = v3(X + 3);
  T w = X[6];
  T vn2 = norm2(v);
  T vn = pp_sqrt(vn2);
  regular = vn > Num<T>::eps() && pp_abs(w) > Num<T>::eps();
  V3<T> phi;
  if (regular) {
    T half = pp_atan(vn / w);
    phi = pp_nan_to_num(T(2) * half / vn) * v;
    th2 = norm2(phi);
    T iq = T(1) / pp_sqrt(vn2 + w * w);           // sin / cos of |theta| / 2 straight from the quaternion
    sh = vn * iq;
    ch = pp_abs(w) * iq;
    if (th2 < Num<T>::seriesF2())
      F = rot_coef_F_series(th2);
    else
      F = pp_nan_to_num((T(1) - pp_abs(half) * pp_abs(w) / vn) / th2);
  } else {
    T p3[3];
    so3_log(X + 3, p3);
    phi = v3(p3);
    th2 = norm2(phi);
    F = rot_coef_F(th2);
    sh = T(0); ch = T(1);
  }
  put(phi, x + 3);
  put(jlinv_apply(F, phi, v3(X)), x);
}

// |Log(X)|^2 of an SE3 element without forming the tangent: with phi = f v, tau = Jl_inv(phi) t = a t - (phi x t)/2 + F (phi.t) phi
// (a = 1 - F theta^2), the three pieces are mutually orthogonal except t and phi, so
//   |tau|^2 = a^2 |t|^2 + (theta^2 |t|^2 - (phi.t)^2)/4 + F^2 (phi.t)^2 theta^2 + 2 a F (phi.t)^2.
// Same branch structure as se3_log (lie_math.h); the irregular rows take se3_log itself.
template <class T> PP_HD T se3_log_norm2(const T* X) {
  const V3<T> t = v3(X), v = v3(X + 3);
  const T w = X[6];
  const T vn2 = norm2(v);
  const T vn = pp_sqrt(vn2);
  if (vn > Num<T>::eps() && pp_abs(w) > Num<T>::eps()) {
    const T half = pp_atan(vn / w);
    const T f = pp_nan_to_num(T(2) * half / vn);
    const T th2 = f * f * vn2;
    T F;
    if (th2 < Num<T>::seriesF2())
      F = rot_coef_F_series(th2);
    else
      F = pp_nan_to_num((T(1) - pp_abs(half) * pp_abs(w) / vn) / th2);
    const T pt = f * dot(v, t), tt = norm2(t), a = T(1) - F * th2, pt2 = pt * pt;
    return th2 + a * a * tt + T(0.25) * (th2 * tt - pt2) + F * pt2 * (F * th2 + T(2) * a);
  }
  T x[6];
  se3_log<T>(X, x);
  return norm2(v3(x)) + norm2(v3(x + 3));
}

// calcQ coefficients C, D, E at theta^2 (rot_coef of lie_math.h with the half-angle sine / cosine already known)
template <class T> PP_HD void calcq_coef(T th2, T sh, T ch, bool regular, T& C, T& D, T& E) {
  if (th2 < Num<T>::series2() || !regular) {
    RotCoef<T> k = rot_coef(th2);
    C = k.C; D = k.D; E = k.E;
  } else {
    T th = pp_sqrt(th2);
    T i2 = T(1) / th2;
    T B = T(2) * sh * sh * i2;
    C = (th - T(2) * sh * ch) * i2 / th;
    D = (T(0.5) - B) * i2;
    E = (T(3) * C - B) * (T(0.5) * i2);
  }
}

// one LM trial of one problem; returns P' and the four summands.  prow / xrow: the problem's rows (in LDS on the device:
// P is read again after the solve instead of being held in registers across it)
template <class T>
PP_HD void lm_se3inv_row(const T* prow, const T* xrow, T s, T dmin, T dmax, T* pn, T& a_new, T& a_old, T& a_jj, T& a_jr) {
  T z0[7], r[6], th2, F, sh, ch;
  bool regular;
  {
    T pc[7], x[7];
    row_get<7>(prow, pc);
    row_get<7>(xrow, x);
    se3_mul<T>(pc, x, z0);
  }
  se3_log_ext<T>(z0, r, th2, F, sh, ch, regular);
  T C, D, E;
  calcq_coef<T>(th2, sh, ch, regular, C, D, E);
  const T tx = r[0], ty = r[1], tz = r[2], px = r[3], py = r[4], pz = r[5];
  const T sd = px * tx + py * ty + pz * tz;
  // Ji (row major)
  const T a = T(1) - F * th2;
  const T fxy = F * px * py, fxz = F * px * pz, fyz = F * py * pz;
  const T hx = T(0.5) * px, hy = T(0.5) * py, hz = T(0.5) * pz;
  const T Ji[9] = {a + F * px * px, fxy + hz, fxz - hy,
                   fxy - hz, a + F * py * py, fyz + hx,
                   fxz + hy, fyz - hx, a + F * pz * pz};
  // Q
  const T al = T(0.5) - D * th2, be = sd * (T(2) * D - C), gam = T(2) * sd * (E * th2 - C), e2 = T(-2) * sd * E;
  const T qx = al * tx + be * px, qy = al * ty + be * py, qz = al * tz + be * pz;
  const T ex = e2 * px, ey = e2 * py, ez = e2 * pz;          // e2 phi
  const T cx = C * tx, cy = C * ty, cz = C * tz;             // C tau
  const T s00 = T(2) * cx * px + ex * px + gam, s11 = T(2) * cy * py + ey * py + gam, s22 = T(2) * cz * pz + ez * pz + gam;
  const T s01 = cx * py + cy * px + ex * py, s02 = cx * pz + cz * px + ex * pz, s12 = cy * pz + cz * py + ey * pz;
  const T Q[9] = {s00, s01 - qz, s02 + qy,
                  s01 + qz, s11, s12 - qx,
                  s02 - qy, s12 + qx, s22};
  T N[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) N[i * 3 + j] = Q[i * 3] * Ji[j] + Q[i * 3 + 1] * Ji[3 + j] + Q[i * 3 + 2] * Ji[6 + j];
  // gradient g = J^T r
  T g[6];
#pragma unroll
  for (int j = 0; j < 3; ++j) g[j] = Ji[j] * tx + Ji[3 + j] * ty + Ji[6 + j] * tz;
  g[3] = px - (N[0] * g[0] + N[3] * g[1] + N[6] * g[2]);
  g[4] = py - (N[1] * g[0] + N[4] * g[1] + N[7] * g[2]);
  g[5] = pz - (N[2] * g[0] + N[5] * g[1] + N[8] * g[2]);
  // A = J^T J, lower triangle of a 6x6
  const T c = T(2) * F - T(0.25) - F * F * th2;
  const T k1 = T(1) - c * th2;
  const T cpx = c * px, cpy = c * py, cpz = c * pz;
  const T A11[6] = {k1 + cpx * px, cpx * py, k1 + cpy * py, cpx * pz, cpy * pz, k1 + cpz * pz};   // 00 10 11 20 21 22
  T pN[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) pN[j] = px * N[j] + py * N[3 + j] + pz * N[6 + j];
  T A12[9];
  const T cp[3] = {cpx, cpy, cpz};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) A12[i * 3 + j] = -(k1 * N[i * 3 + j] + cp[i] * pN[j]);
  T A[36];
  A[0] = A11[0];
  A[6] = A11[1]; A[7] = A11[2];
  A[12] = A11[3]; A[13] = A11[4]; A[14] = A11[5];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) A[(3 + i) * 6 + j] = A12[j * 3 + i];
  const int lo[6][2] = {{0, 0}, {1, 0}, {1, 1}, {2, 0}, {2, 1}, {2, 2}};
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    const int i = lo[e][0], j = lo[e][1];
    A[(3 + i) * 6 + 3 + j] = A11[e] - (N[i] * A12[j] + N[3 + i] * A12[3 + j] + N[6 + i] * A12[6 + j]);
  }
  T lam[6];                                     // what clamping + damping added to the diagonal
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const T d0 = A[p * 6 + p];
    T d = d0 < dmin ? dmin : (d0 > dmax ? dmax : d0);
    d *= s;
    A[p * 6 + p] = d;
    lam[p] = d - d0;
  }
  T d6[6];
  Op_chol6_solve(A, g, d6);
  T Ex[7], zn[7];
  se3_exp<T>(d6, Ex);
  {
    T pc[7], x[7];
    row_get<7>(prow, pc);
    row_get<7>(xrow, x);
    se3_mul<T>(Ex, pc, pn);
    // the new loss is evaluated at the parameter value that is stored, P' as rounded -- not at Exp(d) (P X), whose
    // rounding errors cancel against those of r = Log(P X) and report a loss far below the stored parameter's
    se3_mul<T>(pn, x, zn);
  }
  const T nn = se3_log_norm2<T>(zn);
  // gain-ratio terms without forming J d:  (J d).r = d.(J^T r) = d.g,  and since (J^T J + Lambda) d = -g with Lambda the
  // diagonal that clamping + damping added,  |J d|^2 = d^T (J^T J) d = -d.g - sum_p Lambda_p d_p^2
  T jr = T(0), ld = T(0), oo = T(0);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    jr += d6[i] * g[i];
    ld += lam[i] * d6[i] * d6[i];
    oo += r[i] * r[i];
  }
  const T jj = -jr - ld;
  a_new += nn; a_old += oo; a_jj += jj; a_jr += jr;
}

// asynchronous HBM -> LDS copy of one full [BLOCK, 7] slab (global_load_lds_dwordx4: 1 KiB per wave instruction, no
// staging registers; the destination of a wave's instruction is its wave-uniform base + lane * 16 B, i.e. the slab
// arrives in memory order).  Completion: the issuing wave's vmcnt, then a barrier (MI355X_MICROARCH.md LDS-DMA rules).
template <class T, int BLOCK> __device__ __forceinline__ void glds_slab(const T* g, T* s) {
  constexpr int NV = BLOCK * 7 * (int)sizeof(T) / 16;          // 16-byte chunks, a multiple of 64
  static_assert(NV % 64 == 0, "a slab is a whole number of wave instructions");
  const int lane = threadIdx.x & 63, w64 = threadIdx.x & ~63;
#pragma unroll
  for (int k = 0; k * BLOCK < NV; ++k) {
    const int base = k * BLOCK + w64;                           // wave-uniform
    if (base < NV)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)g + (size_t)(base + lane) * 16),
                                       (__attribute__((address_space(3))) void*)((char*)s + (size_t)base * 16), 16, 0, 0);
  }
}

// trial over the tiles  blockIdx.x, blockIdx.x + gridDim.x, ...: reads the linearisation point from Plin, writes the
// candidate to Pout (may alias Plin: every tile is loaded before any row of it is written) and, if given, Plin's rows to
// save.  Software pipeline, one barrier per tile: while a tile is being computed (~1000 VALU instructions per wave) the
// next tile's P and X slabs are in flight straight into the other LDS buffer and the previous tile's candidates are on
// their way out -- the kernel sits at 3 waves / SIMD, too few to hide a serial load -> compute -> store chain.
// The LDS of a trial workgroup: two buffers of (P slab | X slab).  A tile's candidates P' are written over its X slab (a
// lane's x row is dead once its first product is formed), so no output buffers are needed: 28 KB per workgroup in fp32,
// four workgroups per CU.  The two buffers are SEPARATE __shared__ objects on purpose: hipcc orders every LDS access
// behind each LDS-DMA it cannot prove disjoint (vmcnt(0) before the access), which would serialise the prefetch; distinct
// variables carry distinct alias scopes.
template <class T, int BLOCK> struct LmLds {
  T* b0; T* b1;
};
#define PPLIE_LM_LDS(T, BLOCK, name)                                         \
  __shared__ __attribute__((aligned(16))) T name##_0[BLOCK * 14];            \
  __shared__ __attribute__((aligned(16))) T name##_1[BLOCK * 14];            \
  LmLds<T, BLOCK> name = {name##_0, name##_1}

template <class T, int BLOCK>
__device__ __forceinline__ void lm_trial_tiles(const T* Plin, const T* X, T* Pout, T* save, T s, T dmin, T dmax, int64_t n,
                                               const LmLds<T, BLOCK>& L, T* partial_row) {
  constexpr int SL = BLOCK * 7;
  T a_new = T(0), a_old = T(0), a_jj = T(0), a_jr = T(0);
  const int64_t ntiles = (n + BLOCK - 1) / BLOCK;
  const int64_t nfull = n / BLOCK;    // tiles below this index are complete
  int64_t tile = blockIdx.x, prev = -1;
  if (tile < nfull) {
    glds_slab<T, BLOCK>(Plin + tile * SL, L.b0);
    glds_slab<T, BLOCK>(X + tile * SL, L.b0 + SL);
  }
  // One tile.  `cur` holds (or is about to hold) its inputs; `nxt` still holds the previous tile's candidates in its X
  // slab and then receives the next tile's inputs.  A wave stores exactly the 16-byte chunks of nxt's X slab that its own
  // LDS-DMA instructions overwrite afterwards (slab_s2g and glds_slab share the chunk -> lane map), so program order
  // within the wave is all the ordering that hand-over needs.
  auto stage = [&](T* cur, T* nxt) {
    const bool full = tile < nfull;
    const int rows = full ? BLOCK : (int)(n - tile * BLOCK);
    if (!full) {                      // the ragged last tile: plain loads
      slab_g2s<T, BLOCK, SL, false>(Plin + tile * SL, cur, rows * 7, false);
      slab_g2s<T, BLOCK, SL, false>(X + tile * SL, cur + SL, rows * 7, false);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (prev >= 0) slab_s2g<T, BLOCK, SL, true>(nxt + SL, Pout + prev * SL, SL, true);   // (a previous tile is always full)
    if (save) slab_s2g<T, BLOCK, SL, true>(cur, save + tile * SL, rows * 7, full);
    const int64_t next = tile + gridDim.x;
    if (next < nfull) {
      glds_slab<T, BLOCK>(Plin + next * SL, nxt);
      glds_slab<T, BLOCK>(X + next * SL, nxt + SL);
    }
    const int t = threadIdx.x;
    if (t < rows) {
      T pn[7];
      lm_se3inv_row<T>(cur + t * 7, cur + SL + t * 7, s, dmin, dmax, pn, a_new, a_old, a_jj, a_jr);
      row_st<7>(cur + SL + t * 7, pn);
    }
    prev = tile;
    tile = next;
  };
  int last = 1;
  while (tile < ntiles) {
    stage(L.b0, L.b1);
    last = 0;
    if (tile >= ntiles) break;
    stage(L.b1, L.b0);
    last = 1;
  }
  __syncthreads();
  if (prev >= 0) {
    const bool full = prev < nfull;
    const int valid = full ? SL : (int)(n - prev * BLOCK) * 7;
    if (last == 0) slab_s2g<T, BLOCK, SL, true>(L.b0 + SL, Pout + prev * SL, valid, full);
    else slab_s2g<T, BLOCK, SL, true>(L.b1 + SL, Pout + prev * SL, valid, full);
  }
  T v0 = wg_sum<T, BLOCK>(a_new), v1 = wg_sum<T, BLOCK>(a_old), v2 = wg_sum<T, BLOCK>(a_jj), v3_ = wg_sum<T, BLOCK>(a_jr);
  if (threadIdx.x == 0) {
    partial_row[0] = v0; partial_row[1] = v1; partial_row[2] = v2; partial_row[3] = v3_;
  }
}

// One trial of every problem, per-workgroup partial sums only.  FIRST: linearise at P, keep P in `save`; else a retry
// from the saved linearisation point with the compounded damping.  No same-address atomics anywhere: 4k returning
// atomics on one word serialise at ~25 ns each on this part (measured: 164 us instead of 60 for this kernel).
template <class T, int BLOCK, bool FIRST, int WAVES>
__global__ void __launch_bounds__(BLOCK, WAVES)
lm_se3inv_trial2_kernel(T* P, const T* __restrict__ X, T* save, T* partials, const double* st, LmCfg cfg, int64_t n) {
  if (FIRST && cfg.plateau_max_steps > 0 && st[ST_PL_STOP] != 0.0) return;      // the device-side StopOnPlateau has stopped the run
  PPLIE_LM_LDS(T, BLOCK, lds);
  double sc;
  if (FIRST) sc = 1.0 + ((cfg.flags & LMF_HOST_STATE) ? cfg.host_damping : st[ST_DAMPING]);
  else sc = st[ST_SCALE] * (1.0 + st[ST_DAMPING]);
  lm_trial_tiles<T, BLOCK>(FIRST ? P : save, X, P, FIRST ? save : nullptr, (T)sc, (T)cfg.dmin, (T)cfg.dmax, n, lds,
                           partials + (size_t)blockIdx.x * 4);
}

template <class T, int BLOCK>
__global__ void __launch_bounds__(BLOCK) lm_reduce_kernel(const T* partials, int rows, T* sums) {
  double v[4];
  lm_reduce_partials<T, BLOCK, true>(partials, rows, v);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sums[k] = (T)v[k];
  }
}

template <class T>
__global__ void lm_decide_kernel(const double* st_in, double* st_out, LmCfg cfg, int first, const T* sums, T* loss_out, T* last_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double o[ST_SIZE];
    lm_decide(st_in, o, cfg, first != 0, (double)sums[0], (double)sums[1], (double)sums[2], (double)sums[3]);
    lm_store_state<T>(o, st_out, loss_out, last_out);
  }
}

// Everything after the first trial of a step.  EVERY workgroup adds up the trial kernel's partial sums (same rows, same
// order, same bits) and evaluates the decision from the previous step's state `st_in`, so all of them know the outcome
// without exchanging a word; workgroup 0 stores the new state to `st_out` (the other buffer: nobody reads what it
// writes during this phase).  Accepted (the common case): return.  Failed factorisation: put the linearisation point
// back.  Rejected: { retry trial from the saved point with the compounded damping, grid barrier, workgroup 0 decides,
// grid barrier } until a trial is accepted -- the reference's loop always ends in an accepted trial.
template <class T, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
lm_se3inv_finish_kernel(T* P, const T* __restrict__ X, T* save, T* partials, int first_rows, const double* st_in, double* st_out,
                        unsigned* bar, LmCfg cfg, int64_t n, T* loss_out, T* last_out) {
  PPLIE_LM_LDS(T, BLOCK, lds);
  __shared__ double verdict[2];
  if (cfg.plateau_max_steps > 0 && st_in[ST_PL_STOP] != 0.0) {
    // stopped: the trial kernel did nothing; the state (and the loss scalars of this step's slot) pass through unchanged
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      double o[ST_SIZE];
#pragma unroll
      for (int i = 0; i <= ST_PL_STOP; ++i) o[i] = st_in[i];
      lm_store_state<T>(o, st_out, loss_out, last_out);
    }
    return;
  }
  {
    double v[4];
    lm_reduce_partials<T, BLOCK, true>(partials, first_rows, v);
    if (threadIdx.x == 0) {
      double o[ST_SIZE];
      lm_decide(st_in, o, cfg, true, v[0], v[1], v[2], v[3]);
      if (blockIdx.x == 0) lm_store_state<T>(o, st_out, loss_out, last_out, cfg.plateau_flag);
      verdict[0] = o[ST_DONE];
      verdict[1] = o[ST_FAILED];
    }
    __syncthreads();
  }
  if (verdict[1] != 0.0) {                 // the reference's solver raised before the update (solver.py:214): P unchanged
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n * 7; i += (int64_t)gridDim.x * BLOCK) P[i] = save[i];
    return;
  }
  if (verdict[0] != 0.0) return;
  __threadfence();
  grid_barrier(bar, bar + 1);              // workgroup 0's state is visible
  for (int it = 0; it <= cfg.reject; ++it) {
    const T s = (T)(ld_state(st_out, ST_SCALE) * (1.0 + ld_state(st_out, ST_DAMPING)));
    lm_trial_tiles<T, BLOCK>(save, X, P, nullptr, s, (T)cfg.dmin, (T)cfg.dmax, n, lds, partials + (size_t)blockIdx.x * 4);
    __threadfence();
    grid_barrier(bar, bar + 1);
    if (blockIdx.x == 0) {
      double v[4];
      lm_reduce_partials<T, BLOCK, false>(partials, gridDim.x, v);
      if (threadIdx.x == 0) {
        double in[ST_SIZE], o[ST_SIZE];
#pragma unroll
        for (int i = 0; i <= ST_PL_STOP; ++i) in[i] = ld_state(st_out, i);
        lm_decide(in, o, cfg, false, v[0], v[1], v[2], v[3]);
        lm_store_state<T>(o, st_out, loss_out, last_out, cfg.plateau_flag);
        __threadfence();
      }
    }
    grid_barrier(bar, bar + 1);
    if (ld_state(st_out, ST_FAILED) != 0.0) {
      for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n * 7; i += (int64_t)gridDim.x * BLOCK) P[i] = save[i];
      return;
    }
    if (ld_state(st_out, ST_DONE) != 0.0) return;
  }
}

// workgroup size: 256 problems per tile in fp32, 128 in fp64 (four [BLOCK, 7] slabs of LDS per workgroup)
template <class T> struct LmBlock { static constexpr int v = sizeof(T) == 4 ? 256 : 128; };

template <class T, bool FIRST>
int lm_launch_trial(void* P, const void* X, void* save, void* partials, const void* st, const LmCfg* cfg, int64_t n, hipStream_t s,
                    int* rows) {
  constexpr int BLOCK = LmBlock<T>::v;
  const int64_t nt = (n + BLOCK - 1) / BLOCK;
  // three workgroups per CU, each walking ~5 tiles at configs[2]: measured fastest on MI355X (768: 33.6 us / step, 1024: 34.4,
  // one workgroup per tile: 37.5) -- the software pipeline wants several tiles per workgroup
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  int cap = cfg->grid_cap > 0 ? cfg->grid_cap : 3 * cus;
  if (cap > kStepPartials) cap = kStepPartials;
  const int grid = (int)(nt < cap ? nt : cap);
  *rows = grid;
  // (fp64: the 168-VGPR budget of the three-workgroup variant spills 200+ registers to scratch -- the reference's test precision takes
  //  the 256-VGPR build, which keeps everything in registers / AGPRs: profiles/r06/kernel_resources.txt)
  if constexpr (sizeof(T) == 8) {
    hipLaunchKernelGGL((lm_se3inv_trial2_kernel<T, BLOCK, FIRST, 4>), dim3(grid), dim3(BLOCK), 0, s, (T*)P, (const T*)X, (T*)save,
                       (T*)partials, (const double*)st, *cfg, n);
  } else {
    if (cfg->flags & LMF_OCC4)
      hipLaunchKernelGGL((lm_se3inv_trial2_kernel<T, BLOCK, FIRST, 4>), dim3(grid), dim3(BLOCK), 0, s, (T*)P, (const T*)X, (T*)save,
                         (T*)partials, (const double*)st, *cfg, n);
    else
      hipLaunchKernelGGL((lm_se3inv_trial2_kernel<T, BLOCK, FIRST, 3>), dim3(grid), dim3(BLOCK), 0, s, (T*)P, (const T*)X, (T*)save,
                         (T*)partials, (const double*)st, *cfg, n);
  }
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

static bool lm_args_ok(const void* P, const void* X, const void* save, const void* partials, const void* st, const LmCfg* cfg, int64_t n) {
  return n >= 0 && cfg && P && X && save && partials && st && aligned16(P) && aligned16(X) && aligned16(save) && aligned16(partials);
}

// a whole step: first trial + finish (decision, retries)
template <class T>
int lm_se3inv_step(void* P, const void* X, void* save, void* partials, const void* st_in, void* st_out, void* sync, const LmCfg* cfg,
                   int64_t n, void* loss_out, void* last_out, void* stream) {
  if (!lm_args_ok(P, X, save, partials, st_in, cfg, n) || !st_out || !sync || st_in == st_out) return PPLIE_EBADARG;
  if (n == 0) return PPLIE_OK;
  constexpr int BLOCK = LmBlock<T>::v;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rows = 0;
  int code = lm_launch_trial<T, true>(P, X, save, partials, st_in, cfg, n, s, &rows);
  if (code != PPLIE_OK) return code;
  const int64_t nt = (n + BLOCK - 1) / BLOCK;
  const int fgrid = (int)(nt < kFinishGrid ? nt : kFinishGrid);
  hipLaunchKernelGGL((lm_se3inv_finish_kernel<T, BLOCK>), dim3(fgrid), dim3(BLOCK), 0, s, (T*)P, (const T*)X, (T*)save, (T*)partials, rows,
                     (const double*)st_in, (double*)st_out, (unsigned*)sync, *cfg, n, (T*)loss_out, (T*)last_out);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

// sharded runs: one trial + the reduction of its partial sums to sums[4] (to be all-reduced, then pplie_lm_decide)
template <class T>
int lm_se3inv_trial_sums(void* P, const void* X, void* save, void* partials, const void* st, const LmCfg* cfg, int first, int64_t n,
                         void* sums, void* stream) {
  if (!lm_args_ok(P, X, save, partials, st, cfg, n) || !sums) return PPLIE_EBADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rows = 0;
  if (n > 0) {
    int code = first ? lm_launch_trial<T, true>(P, X, save, partials, st, cfg, n, s, &rows)
                     : lm_launch_trial<T, false>(P, X, save, partials, st, cfg, n, s, &rows);
    if (code != PPLIE_OK) return code;
  }
  hipLaunchKernelGGL((lm_reduce_kernel<T, 256>), dim3(1), dim3(256), 0, s, (const T*)partials, rows, (T*)sums);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

template <class T>
int lm_decide_launch(const void* st_in, void* st_out, const LmCfg* cfg, int first, const void* sums, void* loss_out, void* last_out,
                     void* stream) {
  if (!st_in || !st_out || !cfg || !sums) return PPLIE_EBADARG;
  hipLaunchKernelGGL((lm_decide_kernel<T>), dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), (const double*)st_in,
                     (double*)st_out, *cfg, first, (const T*)sums, (T*)loss_out, (T*)last_out);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace pplie

#define PPLIE_LM_EXPORT(SFX, T)                                                                                                    \
  extern "C" int pplie_lm_se3inv_step_##SFX(void* P, const void* X, void* save, void* partials, const void* st_in, void* st_out,   \
                                            void* sync, const void* cfg, int64_t n, void* loss_out, void* last_out, void* stream) { \
    return pplie::lm_se3inv_step<T>(P, X, save, partials, st_in, st_out, sync, static_cast<const pplie::LmCfg*>(cfg), n, loss_out, \
                                    last_out, stream);                                                                             \
  }                                                                                                                                \
  extern "C" int pplie_lm_se3inv_trial_sums_##SFX(void* P, const void* X, void* save, void* partials, const void* st,              \
                                                  const void* cfg, int first, int64_t n, void* sums, void* stream) {               \
    return pplie::lm_se3inv_trial_sums<T>(P, X, save, partials, st, static_cast<const pplie::LmCfg*>(cfg), first, n, sums, stream); \
  }                                                                                                                                \
  extern "C" int pplie_lm_decide_##SFX(const void* st_in, void* st_out, const void* cfg, int first, const void* sums,              \
                                       void* loss_out, void* last_out, void* stream) {                                             \
    return pplie::lm_decide_launch<T>(st_in, st_out, static_cast<const pplie::LmCfg*>(cfg), first, sums, loss_out, last_out, stream); \
  }
PPLIE_LM_EXPORT(f32, float)
PPLIE_LM_EXPORT(f64, double)
