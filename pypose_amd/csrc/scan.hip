// scan.hip -- single-pass scans: cumulative group products and IMU pre-integration.
//
// Reference: pypose/basics/ops.py:27-36 scans with Hillis-Steele over the WHOLE sequence:
// ceil(log2 L) rounds of index_select x2 + op + index_copy_, i.e. O(L log L) work and ~3 log2 L
// launches (11 rounds x 4 kernels at L = 1025), and pypose/module/imu_preintegrator.py:314-465
// chains two such scans (SO3 and 9x9 matrices) with a dozen eager ops and [B,F+1,9,9] temporaries.
// Here one wavefront owns one sequence and walks it in 64-element chunks: a wave-level
// shuffle scan inside the chunk (6 steps), a carried prefix between chunks -- O(L) work, ONE
// launch, every element read and written once.
//
//   pplie_scan_<group>     in-place inclusive product scan of [outer, L, inner, W] group elements
//   pplie_imu_integrate    dt/gyro/acc -> rot/vel/pos (+ the per-step terms the covariance needs)
//   pplie_imu_cov          9x9 covariance by the backward recurrence S_k = Bc_k + A_k S_{k+1} A_k^T
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lie_math.h"

namespace pplie {

enum { SC_OK = 0, SC_EBADARG = -1, SC_ELAUNCH = -2 };

template <class T, int W> __device__ __forceinline__ void shfl_up_vec(const T* v, T* u, int off) {
#pragma unroll
  for (int k = 0; k < W; ++k) u[k] = __shfl_up(v[k], off, 64);
}
template <class T, int W> __device__ __forceinline__ void bcast_vec(const T* v, T* u, int lane) {
#pragma unroll
  for (int k = 0; k < W; ++k) u[k] = __shfl(v[k], lane, 64);
}

// group products: c = a * b
template <class T> struct MulSO3 { enum { W = 4 }; static __device__ __forceinline__ void mul(const T* a, const T* b, T* c) { so3_mul<T>(a, b, c); } };
template <class T> struct MulSE3 { enum { W = 7 }; static __device__ __forceinline__ void mul(const T* a, const T* b, T* c) { se3_mul<T>(a, b, c); } };
template <class T> struct MulSim3 { enum { W = 8 }; static __device__ __forceinline__ void mul(const T* a, const T* b, T* c) { sim3_mul<T>(a, b, c); } };
template <class T> struct MulRxSO3 { enum { W = 5 }; static __device__ __forceinline__ void mul(const T* a, const T* b, T* c) { rxso3_mul<T>(a, b, c); } };

// acc (+) x  with acc the earlier prefix: left ? x * acc : acc * x   (basics/ops.py:49-56)
template <class T, class G> __device__ __forceinline__ void combine(const T* acc, const T* x, T* out, bool left) {
  T tmp[G::W];
  if (left) G::mul(x, acc, tmp); else G::mul(acc, x, tmp);
#pragma unroll
  for (int k = 0; k < G::W; ++k) out[k] = tmp[k];
}

// inclusive wave scan of v (one element per lane) with the prefix `carry` (valid iff has_carry)
template <class T, class G>
__device__ __forceinline__ void wave_scan(T* v, const T* carry, bool has_carry, bool left, int lane) {
  constexpr int W = G::W;
  if (has_carry && lane == 0) combine<T, G>(carry, v, v, left);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    T u[W];
    shfl_up_vec<T, W>(v, u, off);
    if (lane >= off) combine<T, G>(u, v, v, left);
  }
}

template <class T, class G, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
scan_kernel(T* __restrict__ data, int64_t nseq, int64_t L, int64_t inner, int left) {
  constexpr int W = G::W;
  const int lane = threadIdx.x & 63;
  const int64_t seq = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (seq >= nseq) return;
  const int64_t o = seq / inner, in = seq % inner;
  T carry[W];
#pragma unroll
  for (int k = 0; k < W; ++k) carry[k] = T(0);
  for (int64_t c0 = 0; c0 < L; c0 += 64) {
    const int64_t i = c0 + lane;
    const bool valid = i < L;
    T* p = data + ((o * L + (valid ? i : 0)) * inner + in) * W;
    T v[W];
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = valid ? p[k] : T(0);
    wave_scan<T, G>(v, carry, c0 > 0, left != 0, lane);
    if (valid) {
#pragma unroll
      for (int k = 0; k < W; ++k) p[k] = v[k];
    }
    bcast_vec<T, W>(v, carry, 63);
  }
}

template <class T, class G> int scan_launch(void* data, int64_t nseq, int64_t L, int64_t inner, int left, void* stream) {
  if (nseq < 0 || L < 0 || inner <= 0) return SC_EBADARG;
  if (nseq == 0 || L == 0) return SC_OK;
  if (!data) return SC_EBADARG;
  constexpr int WAVES = 4;
  int64_t blocks = (nseq + WAVES - 1) / WAVES;
  hipLaunchKernelGGL((scan_kernel<T, G, WAVES>), dim3((unsigned)blocks), dim3(WAVES * 64), 0,
                     reinterpret_cast<hipStream_t>(stream), static_cast<T*>(data), nseq, L, inner, left);
  return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
}

// ---------------------------------------------------------------------------------------------
// IMU pre-integration (imu_preintegrator.py:314-426): one wavefront per sequence.
//   dr_f   = Exp(gyro_f dt_f)                                  (:360)
//   P_f    = dr_0 ... dr_f  (incre_r[f+1]);  Pex_f = incre_r[f] (:361-362)
//   a_f    = acc_f - (init_rot * P_f)^-1 g     or  acc_f - rot_f^-1 g if rot is given   (:364-370)
//   u_f    = Pex_f a_f ;  Dv = cumsum(u dt) ; Dp = cumsum(Dv_excl dt + u dt^2/2) ; Dt = cumsum(dt)   (:372-381)
//   rot = init_rot * Dr ; vel = init_vel + init_rot Dv ; pos = init_pos + init_rot Dp + init_vel Dt (:422-426)
// aux (for the covariance): Rk = dr, Rij = [Rij0 *] Dr, a.
// ---------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ T wave_scan_add(T v, T carry, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    T u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  return v + carry;
}

template <class T, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
imu_integrate_kernel(const T* __restrict__ dt, const T* __restrict__ gyro, const T* __restrict__ acc,
                     const T* __restrict__ rot_known,                       // [B,F,4] or null
                     const T* __restrict__ init_rot, const T* __restrict__ init_vel, const T* __restrict__ init_pos,  // [B,4],[B,3],[B,3]
                     const T* __restrict__ rij0,                            // [B,4] or null
                     T gx, T gy, T gz,
                     T* __restrict__ out_rot, T* __restrict__ out_vel, T* __restrict__ out_pos,   // [B,F,4],[B,F,3],[B,F,3]
                     T* __restrict__ aux_rk, T* __restrict__ aux_rij, T* __restrict__ aux_a,      // [B,F,4],[B,F,4],[B,F,3] or null
                     int64_t B, int64_t F) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (b >= B) return;
  T R0[4], v0[3], p0[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) R0[k] = init_rot[b * 4 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { v0[k] = init_vel[b * 3 + k]; p0[k] = init_pos[b * 3 + k]; }
  const V3<T> g = v3<T>(gx, gy, gz);
  T cR[4] = {T(0), T(0), T(0), T(1)};          // carried incre_r (identity before the first step)
  T cV[3] = {T(0), T(0), T(0)}, cP[3] = {T(0), T(0), T(0)}, cT = T(0);
  for (int64_t c0 = 0; c0 < F; c0 += 64) {
    const int64_t f = c0 + lane;
    const bool valid = f < F;
    const int64_t row = b * F + (valid ? f : 0);
    const T h = valid ? dt[row] : T(0);
    T w[3], am[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { w[k] = valid ? gyro[row * 3 + k] * h : T(0); am[k] = valid ? acc[row * 3 + k] : T(0); }
    T dr[4];
    so3_exp<T>(w, dr);
    T P[4] = {dr[0], dr[1], dr[2], dr[3]};
    wave_scan<T, MulSO3<T>>(P, cR, true, false, lane);            // inclusive: incre_r[f+1]
    T Pex[4];
    shfl_up_vec<T, 4>(P, Pex, 1);
    if (lane == 0) { Pex[0] = cR[0]; Pex[1] = cR[1]; Pex[2] = cR[2]; Pex[3] = cR[3]; }   // incre_r[f]
    // acceleration in the body frame minus gravity
    T Rw[4];
    if (rot_known) {
#pragma unroll
      for (int k = 0; k < 4; ++k) Rw[k] = valid ? rot_known[row * 4 + k] : (k == 3 ? T(1) : T(0));
    } else {
      so3_mul<T>(R0, P, Rw);
    }
    V3<T> gi = quat_rotate(-v3(Rw), Rw[3], g);                   // Rw^-1 g  (SO3_Inv then SO3_Act)
    V3<T> a = v3(am) - gi;
    V3<T> u = quat_rotate(v3(Pex), Pex[3], a);
    T dv[3] = {u.x * h, u.y * h, u.z * h};
    T Dv[3], Dp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) Dv[k] = wave_scan_add(dv[k], cV[k], lane);
    T hh = T(0.5) * h * h;
    T uu[3] = {u.x, u.y, u.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      T dp = (Dv[k] - dv[k]) * h + uu[k] * hh;                   // incre_v[f] * dt + R a dt^2 / 2
      Dp[k] = wave_scan_add(dp, cP[k], lane);
    }
    T Dt = wave_scan_add(h, cT, lane);
    if (valid) {
      T Rf[4];
      so3_mul<T>(R0, P, Rf);
      V3<T> vel = v3(v0) + quat_rotate(v3(R0), R0[3], v3(Dv));
      V3<T> pos = v3(p0) + quat_rotate(v3(R0), R0[3], v3(Dp)) + Dt * v3(v0);
#pragma unroll
      for (int k = 0; k < 4; ++k) out_rot[row * 4 + k] = Rf[k];
      put(vel, out_vel + row * 3);
      put(pos, out_pos + row * 3);
      if (aux_rk) {
        T Rij[4];
        if (rij0) {
          T q0[4] = {rij0[b * 4], rij0[b * 4 + 1], rij0[b * 4 + 2], rij0[b * 4 + 3]};
          so3_mul<T>(q0, P, Rij);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) Rij[k] = P[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { aux_rk[row * 4 + k] = dr[k]; aux_rij[row * 4 + k] = Rij[k]; }
        put(a, aux_a + row * 3);
      }
    }
    bcast_vec<T, 4>(P, cR, 63);
#pragma unroll
    for (int k = 0; k < 3; ++k) { cV[k] = __shfl(Dv[k], 63, 64); cP[k] = __shfl(Dp[k], 63, 64); }
    cT = __shfl(Dt, 63, 64);
  }
}

// ---------------------------------------------------------------------------------------------
// Covariance (imu_preintegrator.py:428-465).  The reference's code evaluates
//   cov = sum_{k=0..F} P_k Bc_k P_k^T,   P_k = A_k A_{k+1} ... A_{F-1} (P_F = I),   Bc_0 = init_cov
// through a cumprod of flipped 9x9 matrices (:462-464).  Note this is NOT the textbook recursion of
// its docstring (the products share their RIGHT factors).  One backward pass per sequence:
//   P <- A_k P  (k = F-1 .. 0),  cov += P Bc_k P^T
// A_k = I9 with [0:3,0:3] = Rk^T, [3:6,0:3] = -Rij Ha dt, [6:9,0:3] = -Rij Ha dt^2/2, [6:9,3:6] = dt I (:442-448)
// Bc_{k+1} = (Bg Cg Bg^T + Ba Ca Ba^T)/dt, Bg[0:3] = Jr(Rk) dt, Ba[3:6] = Rij dt, Ba[6:9] = Rij dt^2/2 (:451-460)
//          = V (dt Cg) V^T + U (dt Ca) U^T  with  V = [I;0;0] Jr,  U = [0;I;dt/2 I] Rij,
// so  P Bc P^T = (P0 Jr)(dt Cg)(P0 Jr)^T + ((P1 + dt/2 P2) Rij)(dt Ca)(...)^T  (P = [P0 P1 P2] column blocks):
// two rank-3 updates instead of two 9x9x9 products.  One lane per sequence, P and cov in registers.
// ---------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ void quat_to_matrix(const T* q, T* M) {
  // columns = images of the basis vectors under SO3_Act (matrix(): lietensor.py:281-285)
  V3<T> qv = v3(q);
  V3<T> c0 = quat_rotate(qv, q[3], v3<T>(T(1), T(0), T(0)));
  V3<T> c1 = quat_rotate(qv, q[3], v3<T>(T(0), T(1), T(0)));
  V3<T> c2 = quat_rotate(qv, q[3], v3<T>(T(0), T(0), T(1)));
  M[0] = c0.x; M[1] = c1.x; M[2] = c2.x;
  M[3] = c0.y; M[4] = c1.y; M[5] = c2.y;
  M[6] = c0.z; M[7] = c1.z; M[8] = c2.z;
}

// Nine lanes cooperate on one sequence (seven sequences per wavefront): lane r owns row r of P and
// row r of cov.  P <- A_k P needs rows 0..2 of P (27 broadcast shuffles inside the group) and, for
// rows 6..8, the lane three below; the rank-3 updates need every lane's V / U row (54 shuffles).
// The recurrence over the F steps is inherently sequential, so the parallelism is 9 x B lanes.
template <class T>
__global__ void __launch_bounds__(256)
imu_cov_kernel(const T* __restrict__ dt, const T* __restrict__ rk, const T* __restrict__ rij, const T* __restrict__ a,
               const T* __restrict__ init_cov,                 // [B,9,9]
               const T* __restrict__ gyro_cov, int64_t gc_sb, int64_t gc_sf,   // [.,.,3] with strides (elements)
               const T* __restrict__ acc_cov, int64_t ac_sb, int64_t ac_sf,
               T* __restrict__ cov, int64_t B, int64_t F) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / 9, r = lane - sub * 9;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  int64_t b = wave * 7 + sub;
  const bool live = (sub < 7) && (b < B);
  if (!live) b = 0;                       // idle lanes shadow sequence 0 (they must execute the shuffles)
  const int g0 = (sub < 7 ? sub : 0) * 9; // first lane of this group
  const int blk = r / 3, ri = r - blk * 3;  // row block (0: rotation, 1: velocity, 2: position) and row inside it

  T C[9], P[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { C[c] = T(0); P[c] = (c == r) ? T(1) : T(0); }

  // Per-step quantities.  With ~0.5 wavefront per SIMD nothing else hides memory latency, so the raw
  // inputs of step k-2 are fetched while steps k and k-1 are being consumed (software pipeline), and
  // what is derived from a step (its matrices) is computed once and used twice: in Bc_{k} and in A_{k-1}.
  struct Raw { T h, q[4], qij[4], a[3], cg[3], ca[3]; };
  struct Step { T h, coef[3], V_J[9], U_R[9], cg[3], ca[3]; };   // coef: this lane's weights on old rows 0..2
  auto fetch = [&](int64_t k, Raw& w) {
    const int64_t row = b * F + k;
    w.h = dt[row];
#pragma unroll
    for (int i = 0; i < 4; ++i) { w.q[i] = rk[row * 4 + i]; w.qij[i] = rij[row * 4 + i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      w.a[i] = a[row * 3 + i];
      w.cg[i] = gyro_cov[b * gc_sb + k * gc_sf + i];
      w.ca[i] = acc_cov[b * ac_sb + k * ac_sf + i];
    }
  };
  auto derive = [&](const Raw& w, Step& st) {
    T Rk[9], Rj[9], phi[3];
    quat_to_matrix<T>(w.q, Rk);
    quat_to_matrix<T>(w.qij, Rj);
    so3_log<T>(w.q, phi);
    so3_jr<T>(phi, st.V_J);                 // V = P[:,0:3] Jr
#pragma unroll
    for (int i = 0; i < 9; ++i) st.U_R[i] = Rj[i];   // U = (P[:,3:6] + h/2 P[:,6:9]) Rij
    st.h = w.h;
    const T Ha[9] = {T(0), -w.a[2], w.a[1], w.a[2], T(0), -w.a[0], -w.a[1], w.a[0], T(0)};
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      T m1 = T(0);
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) m1 += Rj[ri * 3 + jj] * Ha[jj * 3 + l];
      m1 = -m1 * w.h;
      st.coef[l] = blk == 0 ? Rk[l * 3 + ri] : (blk == 1 ? m1 : T(0.5) * w.h * m1);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { st.cg[i] = w.h * w.cg[i]; st.ca[i] = w.h * w.ca[i]; }
  };
  const T mine = blk == 0 ? T(0) : T(1);    // rows 3..8 keep their own old row
  auto advance = [&](const Step& st) {      // P <- A_k P
    const T below = blk == 2 ? st.h : T(0); // rows 6..8 add dt * (old row three above)
    T Pn[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const T p0 = __shfl(P[c], g0 + 0, 64), p1 = __shfl(P[c], g0 + 1, 64), p2 = __shfl(P[c], g0 + 2, 64);
      const T up = __shfl(P[c], g0 + (r >= 3 ? r - 3 : r), 64);
      Pn[c] = mine * P[c] + below * up + st.coef[0] * p0 + st.coef[1] * p1 + st.coef[2] * p2;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) P[c] = Pn[c];
  };
  auto accumulate = [&](const Step& st) {   // cov += P Bc P^T with Bc built from this step
    T V[3], U[3], Vr[3], Ur[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      T v = T(0), u = T(0);
#pragma unroll
      for (int l = 0; l < 3; ++l) {
        v += P[l] * st.V_J[l * 3 + jj];
        u += (P[3 + l] + T(0.5) * st.h * P[6 + l]) * st.U_R[l * 3 + jj];
      }
      Vr[jj] = v; Ur[jj] = u;
      V[jj] = v * st.cg[jj];
      U[jj] = u * st.ca[jj];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      T sacc = T(0);
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        sacc += V[jj] * __shfl(Vr[jj], g0 + c, 64);
        sacc += U[jj] * __shfl(Ur[jj], g0 + c, 64);
      }
      C[c] += sacc;
    }
  };

  if (F > 0) {
    Raw nxt;
    Step prev, cur;                          // prev: step k-1 (for Bc_k), cur: step k (for A_k)
    fetch(F - 1, nxt);
    derive(nxt, prev);
    if (F > 1) fetch(F - 2, nxt);
    for (int64_t k = F; k >= 1; --k) {
      if (k < F) advance(cur);
      accumulate(prev);
      cur = prev;
      if (k >= 2) {
        derive(nxt, prev);                   // step k-2, fetched one iteration ago
        if (k >= 3) fetch(k - 3, nxt);
      }
    }
    advance(cur);                            // P_0 = A_0 P_1
  }
  {                                          // Bc_0 = init_cov
    T t[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      T sacc = T(0);
#pragma unroll
      for (int l = 0; l < 9; ++l) sacc += P[l] * init_cov[b * 81 + l * 9 + c];
      t[c] = sacc;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      T sacc = T(0);
#pragma unroll
      for (int l = 0; l < 9; ++l) sacc += t[l] * __shfl(P[l], g0 + c, 64);
      C[c] += sacc;
    }
  }
  if (live) {
#pragma unroll
    for (int c = 0; c < 9; ++c) cov[b * 81 + r * 9 + c] = C[c];
  }
}

template <class T>
int imu_integrate_launch(const void* dt, const void* gyro, const void* acc, const void* rot, const void* r0, const void* v0,
                         const void* p0, const void* rij0, const double* g, void* orot, void* ovel, void* opos, void* ark,
                         void* arij, void* aa, int64_t B, int64_t F, void* stream) {
  if (B < 0 || F < 0) return SC_EBADARG;
  if (B == 0 || F == 0) return SC_OK;
  if (!dt || !gyro || !acc || !r0 || !v0 || !p0 || !g || !orot || !ovel || !opos) return SC_EBADARG;
  if ((ark != nullptr) != (arij != nullptr) || (ark != nullptr) != (aa != nullptr)) return SC_EBADARG;
  constexpr int WAVES = 4;
  int64_t blocks = (B + WAVES - 1) / WAVES;
  hipLaunchKernelGGL((imu_integrate_kernel<T, WAVES>), dim3((unsigned)blocks), dim3(WAVES * 64), 0,
                     reinterpret_cast<hipStream_t>(stream), (const T*)dt, (const T*)gyro, (const T*)acc, (const T*)rot,
                     (const T*)r0, (const T*)v0, (const T*)p0, (const T*)rij0, (T)g[0], (T)g[1], (T)g[2], (T*)orot, (T*)ovel,
                     (T*)opos, (T*)ark, (T*)arij, (T*)aa, B, F);
  return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
}

template <class T>
int imu_cov_launch(const void* dt, const void* rk, const void* rij, const void* a, const void* init_cov, const void* gc,
                   int64_t gc_sb, int64_t gc_sf, const void* ac, int64_t ac_sb, int64_t ac_sf, void* cov, int64_t B, int64_t F,
                   void* stream) {
  if (B < 0 || F < 0) return SC_EBADARG;
  if (B == 0) return SC_OK;
  if (!dt || !rk || !rij || !a || !init_cov || !gc || !ac || !cov) return SC_EBADARG;
  int64_t waves = (B + 6) / 7;              // seven sequences per wavefront, nine lanes each
  int64_t blocks = (waves + 3) / 4;
  hipLaunchKernelGGL((imu_cov_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)dt, (const T*)rk, (const T*)rij, (const T*)a, (const T*)init_cov, (const T*)gc, gc_sb, gc_sf,
                     (const T*)ac, ac_sb, ac_sf, (T*)cov, B, F);
  return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
}
}  // namespace pplie

#define PPLIE_SCAN_EXPORT(g, G)                                                                                     \
  extern "C" int pplie_scan_##g##_f32(void* data, int64_t nseq, int64_t L, int64_t inner, int left, void* stream) { \
    return pplie::scan_launch<float, pplie::G<float>>(data, nseq, L, inner, left, stream);                          \
  }                                                                                                                 \
  extern "C" int pplie_scan_##g##_f64(void* data, int64_t nseq, int64_t L, int64_t inner, int left, void* stream) { \
    return pplie::scan_launch<double, pplie::G<double>>(data, nseq, L, inner, left, stream);                        \
  }
PPLIE_SCAN_EXPORT(so3, MulSO3)
PPLIE_SCAN_EXPORT(se3, MulSE3)
PPLIE_SCAN_EXPORT(sim3, MulSim3)
PPLIE_SCAN_EXPORT(rxso3, MulRxSO3)

extern "C" int pplie_imu_integrate_f32(const void* dt, const void* gyro, const void* acc, const void* rot, const void* init_rot,
                                       const void* init_vel, const void* init_pos, const void* rij0, const double* gravity,
                                       void* out_rot, void* out_vel, void* out_pos, void* aux_rk, void* aux_rij, void* aux_a,
                                       int64_t B, int64_t F, void* stream) {
  return pplie::imu_integrate_launch<float>(dt, gyro, acc, rot, init_rot, init_vel, init_pos, rij0, gravity, out_rot, out_vel,
                                            out_pos, aux_rk, aux_rij, aux_a, B, F, stream);
}
extern "C" int pplie_imu_integrate_f64(const void* dt, const void* gyro, const void* acc, const void* rot, const void* init_rot,
                                       const void* init_vel, const void* init_pos, const void* rij0, const double* gravity,
                                       void* out_rot, void* out_vel, void* out_pos, void* aux_rk, void* aux_rij, void* aux_a,
                                       int64_t B, int64_t F, void* stream) {
  return pplie::imu_integrate_launch<double>(dt, gyro, acc, rot, init_rot, init_vel, init_pos, rij0, gravity, out_rot, out_vel,
                                             out_pos, aux_rk, aux_rij, aux_a, B, F, stream);
}
extern "C" int pplie_imu_cov_f32(const void* dt, const void* rk, const void* rij, const void* a, const void* init_cov,
                                 const void* gyro_cov, int64_t gc_sb, int64_t gc_sf, const void* acc_cov, int64_t ac_sb,
                                 int64_t ac_sf, void* cov, int64_t B, int64_t F, void* stream) {
  return pplie::imu_cov_launch<float>(dt, rk, rij, a, init_cov, gyro_cov, gc_sb, gc_sf, acc_cov, ac_sb, ac_sf, cov, B, F, stream);
}
extern "C" int pplie_imu_cov_f64(const void* dt, const void* rk, const void* rij, const void* a, const void* init_cov,
                                 const void* gyro_cov, int64_t gc_sb, int64_t gc_sf, const void* acc_cov, int64_t ac_sb,
                                 int64_t ac_sf, void* cov, int64_t B, int64_t F, void* stream) {
  return pplie::imu_cov_launch<double>(dt, rk, rij, a, init_cov, gyro_cov, gc_sb, gc_sf, acc_cov, ac_sb, ac_sf, cov, B, F, stream);
}
