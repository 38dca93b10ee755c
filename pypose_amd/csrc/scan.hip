// scan.hip -- single-pass scans: cumulative group products and IMU pre-integration.
//
// Reference: pypose/basics/ops.py:27-36 scans with Hillis-Steele over the WHOLE sequence:
// ceil(log2 L) rounds of index_select x2 + op + index_copy_, i.e. O(L log L) work and ~3 log2 L
// launches (11 rounds x 4 kernels at L = 1025), and pypose/module/imu_preintegrator.py:314-465
// chains two such scans (SO3 and 9x9 matrices) with a dozen eager ops and [B,F+1,9,9] temporaries.
// Here one wavefront owns one sequence and walks it in 64-element chunks: a wave-level scan on DPP
// cross-lane moves inside the chunk (7 steps), a carried prefix between chunks -- O(L) work, ONE
// launch, every element read and written once.
//
//   pplie_scan_<group>     in-place inclusive product scan of [outer, L, inner, W] group elements
//   pplie_imu_integrate    dt/gyro/acc -> rot/vel/pos (+ the per-step terms the covariance needs)
//   pplie_imu_cov          9x9 covariance  sum_k P_k Bc_k P_k^T  (P_k = A_k ... A_{F-1}) from suffix scans + one reduction
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lie_math.h"
#include "dpp.h"

namespace pplie {

enum { SC_OK = 0, SC_EBADARG = -1, SC_ELAUNCH = -2 };

template <class T, int W> __device__ __forceinline__ void shfl_up_vec(const T* v, T* u, int off) {
#pragma unroll
  for (int k = 0; k < W; ++k) u[k] = __shfl_up(v[k], off, 64);
}
template <class T, int W> __device__ __forceinline__ void bcast_vec(const T* v, T* u, int lane) {   // lane is always 63
#pragma unroll
  for (int k = 0; k < W; ++k) u[k] = lane_bcast63(v[k]);
}

// group products: c = a * b
// (ident(k): component k of the group identity)
// (inv(a, o): o = a^-1;  adjT(X, g, o): o = [Adj(X)^T g[:W-1], 0] -- the Y-gradient of Mul, operation.py:846-852)
template <class T> struct MulSO3 { enum { W = 4 }; static __device__ __forceinline__ void mul(const T* a, const T* b, T* c) { so3_mul<T>(a, b, c); }
  static __device__ __forceinline__ T ident(int k) { return k == 3 ? T(1) : T(0); }
  static __device__ __forceinline__ void inv(const T* a, T* o) { so3_inv<T>(a, o); }
  static __device__ __forceinline__ void adjT(const T* X, const T* g, T* o) { T gx[W]; so3_mul_bwd<T>(X, g, gx, o); } };
template <class T> struct MulSE3 { enum { W = 7 }; static __device__ __forceinline__ void mul(const T* a, const T* b, T* c) { se3_mul<T>(a, b, c); }
  static __device__ __forceinline__ T ident(int k) { return k == 6 ? T(1) : T(0); }
  static __device__ __forceinline__ void inv(const T* a, T* o) { se3_inv<T>(a, o); }
  static __device__ __forceinline__ void adjT(const T* X, const T* g, T* o) { T gx[W]; se3_mul_bwd<T>(X, g, gx, o); } };
template <class T> struct MulSim3 { enum { W = 8 }; static __device__ __forceinline__ void mul(const T* a, const T* b, T* c) { sim3_mul<T>(a, b, c); }
  static __device__ __forceinline__ T ident(int k) { return (k == 6 || k == 7) ? T(1) : T(0); }
  static __device__ __forceinline__ void inv(const T* a, T* o) { sim3_inv<T>(a, o); }
  static __device__ __forceinline__ void adjT(const T* X, const T* g, T* o) { T gx[W]; sim3_mul_bwd<T>(X, g, gx, o); } };
template <class T> struct MulRxSO3 { enum { W = 5 }; static __device__ __forceinline__ void mul(const T* a, const T* b, T* c) { rxso3_mul<T>(a, b, c); }
  static __device__ __forceinline__ T ident(int k) { return (k == 3 || k == 4) ? T(1) : T(0); }
  static __device__ __forceinline__ void inv(const T* a, T* o) { rxso3_inv<T>(a, o); }
  static __device__ __forceinline__ void adjT(const T* X, const T* g, T* o) { T gx[W]; rxso3_mul_bwd<T>(X, g, gx, o); } };

// acc (+) x  with acc the earlier prefix: left ? x * acc : acc * x   (basics/ops.py:49-56)
template <class T, class G> __device__ __forceinline__ void combine(const T* acc, const T* x, T* out, bool left) {
  T tmp[G::W];
  if (left) G::mul(x, acc, tmp); else G::mul(acc, x, tmp);
#pragma unroll
  for (int k = 0; k < G::W; ++k) out[k] = tmp[k];
}

// inclusive wave scan of v (one element per lane) with the prefix `carry` (valid iff has_carry): 7 DPP steps,
// lanes a step does not reach combine with the identity
template <class T, class G, int CTRL, int RM, int BM>
__device__ __forceinline__ void scan_step(const T* src, T* v, bool left) {
  constexpr int W = G::W;
  T u[W];
#pragma unroll
  for (int k = 0; k < W; ++k) u[k] = dpp_mov<CTRL, RM, BM>(G::ident(k), src[k]);
  combine<T, G>(u, v, v, left);
}
template <class T, class G>
__device__ __forceinline__ void wave_scan(T* v, const T* carry, bool has_carry, bool left, int lane) {
  constexpr int W = G::W;
  T v0[W];
#pragma unroll
  for (int k = 0; k < W; ++k) v0[k] = v[k];
  scan_step<T, G, DPP_ROW_SHR1, 0xf, 0xf>(v0, v, left);
  scan_step<T, G, DPP_ROW_SHR2, 0xf, 0xf>(v0, v, left);
  scan_step<T, G, DPP_ROW_SHR3, 0xf, 0xf>(v0, v, left);
  scan_step<T, G, DPP_ROW_SHR4, 0xf, 0xe>(v, v, left);
  scan_step<T, G, DPP_ROW_SHR8, 0xf, 0xc>(v, v, left);
  scan_step<T, G, DPP_ROW_BCAST15, 0xa, 0xf>(v, v, left);
  scan_step<T, G, DPP_ROW_BCAST31, 0xc, 0xf>(v, v, left);
  if (has_carry) combine<T, G>(carry, v, v, left);               // uniform branch: every lane takes the carry
}

// K consecutive elements per lane: their running products are formed inside the lane, one wave scan combines the 64
// lane totals, and the exclusive prefix is folded back into the K values -- the cross-lane scan (the dominant cost: 7
// DPP steps of W-wide group products) is paid once per 64 K elements.
template <class T, class G, int WAVES, int K, bool LEFT>
__global__ void __launch_bounds__(WAVES * 64)
scan_kernel(T* __restrict__ data, int64_t nseq, int64_t L, int64_t inner) {
  constexpr int W = G::W;
  const int lane = threadIdx.x & 63;
  const int64_t seq = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (seq >= nseq) return;
  const int64_t o = seq / inner, in = seq % inner;
  constexpr bool lf = LEFT;            // compile-time operand order: as a runtime flag it costs 8 v_cndmask per scan step
  T carry[W];
#pragma unroll
  for (int k = 0; k < W; ++k) carry[k] = G::ident(k);
  // software pipeline: the next chunk's elements are requested before this chunk's scan runs (one wave owns the whole
  // sequence, so every chunk would otherwise pay the full HBM latency on its critical path).  The fetch is nothing but loads from
  // clamped addresses: a predicate on the load (`ok ? p[k] : d`) or a select right behind it puts an s_waitcnt INSIDE the fetch
  // and the "prefetch" completes before the scan starts (how these kernels ran until round 4; the same holds for all fetches below).
  T nxt[K][W];
  auto fetch = [&](int64_t c0) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int64_t i = c0 + (int64_t)lane * K + j;
      const bool valid = i < L;
      const T* p = data + ((o * L + (valid ? i : 0)) * inner + in) * W;
#pragma unroll
      for (int k = 0; k < W; ++k) nxt[j][k] = p[k];          // raw (the address is clamped); the padding is selected at the consumer
    }
  };
  fetch(0);
  for (int64_t c0 = 0; c0 < L; c0 += 64 * K) {
    T v[K][W];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool valid = c0 + (int64_t)lane * K + j < L;
#pragma unroll
      for (int k = 0; k < W; ++k) v[j][k] = valid ? nxt[j][k] : G::ident(k);     // padding: the identity changes nothing
    }
    if (c0 + 64 * K < L) fetch(c0 + 64 * K);      // (reads rows this iteration does not write: the scan is in place)
#pragma unroll
    for (int j = 1; j < K; ++j) combine<T, G>(v[j - 1], v[j], v[j], lf);     // running products inside the lane
    T tot[W];
#pragma unroll
    for (int k = 0; k < W; ++k) tot[k] = v[K - 1][k];
    wave_scan<T, G>(tot, carry, c0 > 0, lf, lane);                          // inclusive over lane totals (+ carry)
    if (K > 1) {
      T ex[W];                                                              // what precedes this lane's first element
#pragma unroll
      for (int k = 0; k < W; ++k) ex[k] = lane_shift_up1(tot[k], carry[k]);
      if (c0 > 0 || lane > 0) {
#pragma unroll
        for (int j = 0; j < K - 1; ++j) combine<T, G>(ex, v[j], v[j], lf);
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) v[K - 1][k] = tot[k];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int64_t i = c0 + (int64_t)lane * K + j;
      if (i < L) {
        T* p = data + ((o * L + i) * inner + in) * W;
#pragma unroll
        for (int k = 0; k < W; ++k) p[k] = v[j][k];
      }
    }
    bcast_vec<T, W>(tot, carry, 63);
  }
}

template <class T, class G> int scan_launch(void* data, int64_t nseq, int64_t L, int64_t inner, int left, void* stream) {
  if (nseq < 0 || L < 0 || inner <= 0) return SC_EBADARG;
  if (nseq == 0 || L == 0) return SC_OK;
  if (!data) return SC_EBADARG;
  constexpr int WAVES = 4;
  int64_t blocks = (nseq + WAVES - 1) / WAVES;
#define PPLIE_SCAN(KK, LF)                                                                                   \
  hipLaunchKernelGGL((scan_kernel<T, G, WAVES, KK, LF>), dim3((unsigned)blocks), dim3(WAVES * 64), 0,          \
                     reinterpret_cast<hipStream_t>(stream), static_cast<T*>(data), nseq, L, inner)
  if (L >= 256) {               // (K = 4 measured no better than K = 1 at [4096, 1025] SO3: 48 us; K = 2: 43 us)
    if (left) PPLIE_SCAN(2, true); else PPLIE_SCAN(2, false);
  } else {
    if (left) PPLIE_SCAN(1, true); else PPLIE_SCAN(1, false);
  }
#undef PPLIE_SCAN
  return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
}

// ---------------------------------------------------------------------------------------------
// Product scan of small square matrices [nseq, L, D, D] along L, in place (pypose/basics/ops.py:48-56 cumprod_ on plain tensors:
// ops = b @ a / a @ b; the reference's use is the [B, F + 1, 9, 9] propagation matrices of the IMU covariance,
// module/imu_preintegrator.py:462 -- SURVEY 8(b) `scan_mat9`).  The reference runs ceil(log2 L) rounds of index_select / bmm /
// index_copy_ over the whole tensor (10 rounds x 3 passes at L = 1024); here a workgroup walks one sequence once: thread (i, j) keeps
// entry (i, j) of the running product, the product of the previous step sits in LDS (double-buffered), the next matrix is in
// flight while the current one is multiplied.  2 D^2 x sizeof(T) bytes per element moved -- the algorithmic minimum; the walk's
// latency (one barrier per step) is hidden by the other sequences' workgroups on the CU.
// ---------------------------------------------------------------------------------------------
template <class T, int D, bool LEFT>
__global__ void __launch_bounds__(128)
scan_mat_kernel(T* __restrict__ data, int64_t nseq, int64_t L) {
  __shared__ T P[2][D * D], A[2][D * D];
  const int t = threadIdx.x, i = t / D, j = t % D;
  const bool act = t < D * D;
  for (int64_t seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
    T* base = data + seq * L * (D * D);
    T nxt = act ? base[t] : T(0);                        // element t of matrix 0
    if (act) P[0][t] = nxt;                              // out_0 = x_0
    if (act && L > 1) nxt = base[(D * D) + t];
    __syncthreads();
    for (int64_t k = 1; k < L; ++k) {
      const int cur = (int)(k & 1), prev = cur ^ 1;
      if (act) A[cur][t] = nxt;
      if (act && k + 1 < L) nxt = base[(k + 1) * (D * D) + t];          // (in flight during this step's product)
      __syncthreads();
      if (act) {
        T acc = T(0);
        // LEFT: out_k = x_k @ out_{k-1};  else out_k = out_{k-1} @ x_k
#pragma unroll
        for (int m = 0; m < D; ++m) acc += LEFT ? A[cur][i * D + m] * P[prev][m * D + j] : P[prev][i * D + m] * A[cur][m * D + j];
        P[cur][t] = acc;
        base[k * (D * D) + t] = acc;
      }
      // (one barrier per step: P[cur] / A[cur] are read by the next step, which writes P[prev] / A[prev] -- last read before
      //  this step's barrier)
    }
    __syncthreads();
  }
}
template <class T> int scan_mat_launch(void* data, int64_t nseq, int64_t L, int d, int left, void* stream) {
  if (nseq < 0 || L < 0 || d < 1) return SC_EBADARG;
  if (nseq == 0 || L == 0) return SC_OK;
  if (!data) return SC_EBADARG;
  const int64_t cap = 256 * 8;                            // eight workgroups per CU cover one another's barriers
  const unsigned grid = (unsigned)(nseq < cap ? nseq : cap);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define PPLIE_SCAN_MAT(DD)                                                                                              \
  {                                                                                                                     \
    if (left) hipLaunchKernelGGL((scan_mat_kernel<T, DD, true>), dim3(grid), dim3(128), 0, st, static_cast<T*>(data), nseq, L);  \
    else hipLaunchKernelGGL((scan_mat_kernel<T, DD, false>), dim3(grid), dim3(128), 0, st, static_cast<T*>(data), nseq, L);      \
  }
  switch (d) {
    case 2: PPLIE_SCAN_MAT(2) break;
    case 3: PPLIE_SCAN_MAT(3) break;
    case 4: PPLIE_SCAN_MAT(4) break;
    case 6: PPLIE_SCAN_MAT(6) break;
    case 7: PPLIE_SCAN_MAT(7) break;
    case 9: PPLIE_SCAN_MAT(9) break;
    default: return SC_EBADARG;
  }
#undef PPLIE_SCAN_MAT
  return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
}

// ---------------------------------------------------------------------------------------------
// Backward of the product scan (the cotangent of  y = cumprod(x)  in the reference's gradient convention: gradients of
// group elements are left-tangent vectors zero-padded to the embedding width, operation.py:846-852).  The reference gets
// it by differentiating the log2(L) Hillis-Steele rounds (basics/ops.py:27-36: index_select / Mul / index_copy_ per round
// plus their backward nodes); composing the Mul rules gives one reverse pass:
//   right products  y_i = x_1 ... x_i :  gx_k = Adj(y_{k-1})^T  sum_{i>=k} g_i                                 (y_0 = 1)
//       a reverse PLAIN sum of the cotangents (they all live in the same left frame), then one transport.
//       Reads the scan's OUTPUT y (shifted by one) and g.
//   left products   y_i = x_i ... x_1 :  gx_k = g_k + Adj(x_{k+1})^T gx_{k+1}
//       a reverse scan of the affine maps  s -> c + Adj(M)^T s  (pairs (M, c); composition (M_A, c_A) o (M_B, c_B) =
//       (M_B M_A, c_A + Adj(M_A)^T c_B) for a range A followed by the later range B): inside a lane sequentially, across the
//       lanes by the 7-step DPP tree, across chunks by a carried cotangent.  Every transport is by a product of the factors
//       BETWEEN the two elements -- what the reference's tree does, at O(L) work; transporting through absolute poses
//       (y_i y_a^-1) or through a chunk-wide anchor instead was measured 10x (SE3 random walk) to 70x (Sim3 with compounding
//       scales) less accurate than the reference's formulation in fp32.  Reads the scan's INPUT x (shifted by one) and g.
// One wavefront per sequence, chunks walked from the END of the sequence, lanes in reverse order (a prefix over the lanes
// is a suffix over positions), K consecutive elements per lane; 3 W scalars of traffic per element.
// Element 0 is never the OUTPUT of a product in the reference's rounds (ops.py:34-35 overwrites indices >= step only): y_0 IS
// x_0 and its cotangent passes through whole, last embedding component included.
// ---------------------------------------------------------------------------------------------
template <class T, class G, int WAVES, int K>
__global__ void __launch_bounds__(WAVES * 64)
scan_bwd_right_kernel(const T* __restrict__ y, const T* __restrict__ g, T* __restrict__ gx, int64_t nseq, int64_t L, int64_t inner) {
  constexpr int W = G::W, D = W - 1;
  const int lane = threadIdx.x & 63, rl = 63 - lane;
  const int64_t seq = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (seq >= nseq) return;
  const int64_t o = seq / inner, in = seq % inner;
  auto row = [&](int64_t i) { return ((o * L + i) * inner + in) * W; };
  const T g0D = g[row(0) + D];     // (element 0's padding component passes through: read once, not under a branch in the loop)
  T C[W];                 // the sum over everything behind this chunk
#pragma unroll
  for (int k = 0; k < W; ++k) C[k] = T(0);
  const int64_t CH = 64 * K, nch = (L + CH - 1) / CH;
  // software pipeline: the next (earlier) chunk's rows are requested before this chunk's sums run -- one wave owns the whole
  // sequence, so every chunk would otherwise pay the full HBM latency on its critical path
  T ny[K][W], nu[K][W];
  auto fetch = [&](int64_t c0) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int64_t i = c0 + (int64_t)rl * K + j;
      const bool valid = i < L;
      const T* pg = g + row(valid ? i : 0);
#pragma unroll
      for (int k = 0; k < D; ++k) nu[j][k] = pg[k];               // (raw: selected at the consumer)
      nu[j][D] = T(0);
      const bool has = valid && i > 0;                            // y_{i-1}
      const T* py = y + row(has ? i - 1 : 0);
#pragma unroll
      for (int k = 0; k < W; ++k) ny[j][k] = py[k];
    }
  };
  fetch((nch - 1) * CH);
  for (int64_t c = nch - 1; c >= 0; --c) {
    const int64_t c0 = c * CH;
    T yv[K][W], u[K][W];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int64_t i = c0 + (int64_t)rl * K + j;
      const bool valid = i < L, has = valid && i > 0;
#pragma unroll
      for (int k = 0; k < W; ++k) { yv[j][k] = has ? ny[j][k] : G::ident(k); u[j][k] = (valid && k < D) ? nu[j][k] : T(0); }
    }
    if (c > 0) fetch(c0 - CH);
    // suffix sums over positions: inside the lane, then over the (reversed) lanes
#pragma unroll
    for (int j = K - 2; j >= 0; --j)
#pragma unroll
      for (int k = 0; k < D; ++k) u[j][k] += u[j + 1][k];
    T ex[W];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const T tot = wave_prefix_add<T>(u[0][k]);
      ex[k] = lane_shift_up1(tot, T(0)) + C[k];                   // everything behind this lane's K elements
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
#pragma unroll
      for (int k = 0; k < D; ++k) u[j][k] += ex[k];
      u[j][D] = T(0);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) C[k] = lane_bcast63(u[0][k]);     // the sum from position c0 on
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int64_t i = c0 + (int64_t)rl * K + j;
      if (i < L) {
        T out[W];
        G::adjT(yv[j], u[j], out);
        T* p = gx + row(i);
#pragma unroll
        for (int k = 0; k < D; ++k) p[k] = out[k];
        p[D] = i == 0 ? g0D : T(0);
      }
    }
  }
}

// (M, c) <- (M, c) o (Ma, ca):  the range held by this lane followed by the LATER range (Ma, ca) of the lower lanes
template <class T, class G> __device__ __forceinline__ void affine_after(T* M, T* c, const T* Ma, const T* ca) {
  constexpr int W = G::W;
  T t[W], Mn[W];
  G::adjT(M, ca, t);
  G::mul(Ma, M, Mn);
#pragma unroll
  for (int k = 0; k < W; ++k) { M[k] = Mn[k]; c[k] += t[k]; }
}
template <class T, class G, int CTRL, int RM, int BM>
__device__ __forceinline__ void affine_scan_step(const T* srcM, const T* srcc, T* M, T* c) {
  constexpr int W = G::W;
  T Ma[W], ca[W];
#pragma unroll
  for (int k = 0; k < W; ++k) { Ma[k] = dpp_mov<CTRL, RM, BM>(G::ident(k), srcM[k]); ca[k] = dpp_mov<CTRL, RM, BM>(T(0), srcc[k]); }
  affine_after<T, G>(M, c, Ma, ca);
}

template <class T, class G, int WAVES, int K>
__global__ void __launch_bounds__(WAVES * 64)
scan_bwd_left_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ gx, int64_t nseq, int64_t L, int64_t inner) {
  constexpr int W = G::W, D = W - 1;
  const int lane = threadIdx.x & 63, rl = 63 - lane;
  const int64_t seq = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (seq >= nseq) return;
  const int64_t o = seq / inner, in = seq % inner;
  auto row = [&](int64_t i) { return ((o * L + i) * inner + in) * W; };
  const T g0D = g[row(0) + D];     // (element 0's padding component passes through: read once, not under a branch in the loop)
  T C[W];                 // gx of the later chunk's first element (zero behind the end of the sequence)
#pragma unroll
  for (int k = 0; k < W; ++k) C[k] = T(0);
  const int64_t CH = 64 * K, nch = (L + CH - 1) / CH;
  T nM[K][W], nc[K][W];   // software pipeline as in the right-product kernel
  auto fetch = [&](int64_t c0) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int64_t i = c0 + (int64_t)rl * K + j;
      const bool valid = i < L, nxt = i + 1 < L;
      const T* pg = g + row(valid ? i : 0);
      const T* px = x + row(nxt ? i + 1 : 0);
#pragma unroll
      for (int k = 0; k < D; ++k) nc[j][k] = pg[k];               // (raw: selected at the consumer)
      nc[j][D] = T(0);
#pragma unroll
      for (int k = 0; k < W; ++k) nM[j][k] = px[k];
    }
  };
  fetch((nch - 1) * CH);
  for (int64_t ch = nch - 1; ch >= 0; --ch) {
    const int64_t c0 = ch * CH;
    // element i carries the map  s -> g_i + Adj(x_{i+1})^T s ; M[j], c[j] become the lane-local composites of [j .. K-1]
    T M[K][W], c[K][W];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int64_t i = c0 + (int64_t)rl * K + j;
      const bool valid = i < L, nxt = i + 1 < L;
#pragma unroll
      for (int k = 0; k < W; ++k) { M[j][k] = nxt ? nM[j][k] : G::ident(k); c[j][k] = (valid && k < D) ? nc[j][k] : T(0); }
    }
    if (ch > 0) fetch(c0 - CH);
#pragma unroll
    for (int j = K - 2; j >= 0; --j) affine_after<T, G>(M[j], c[j], M[j + 1], c[j + 1]);
    // inclusive scan of the lane totals over the lanes (lower lanes = later positions)
    T Mt[W], ct[W], M0[W], c0v[W];
#pragma unroll
    for (int k = 0; k < W; ++k) { Mt[k] = M0[k] = M[0][k]; ct[k] = c0v[k] = c[0][k]; }
    affine_scan_step<T, G, DPP_ROW_SHR1, 0xf, 0xf>(M0, c0v, Mt, ct);
    affine_scan_step<T, G, DPP_ROW_SHR2, 0xf, 0xf>(M0, c0v, Mt, ct);
    affine_scan_step<T, G, DPP_ROW_SHR3, 0xf, 0xf>(M0, c0v, Mt, ct);
    affine_scan_step<T, G, DPP_ROW_SHR4, 0xf, 0xe>(Mt, ct, Mt, ct);
    affine_scan_step<T, G, DPP_ROW_SHR8, 0xf, 0xc>(Mt, ct, Mt, ct);
    affine_scan_step<T, G, DPP_ROW_BCAST15, 0xa, 0xf>(Mt, ct, Mt, ct);
    affine_scan_step<T, G, DPP_ROW_BCAST31, 0xc, 0xf>(Mt, ct, Mt, ct);
    // what follows this lane's K elements: the exclusive composite applied to the carried cotangent
    T Me[W], ce[W], S[W], t[W];
#pragma unroll
    for (int k = 0; k < W; ++k) { Me[k] = lane_shift_up1(Mt[k], G::ident(k)); ce[k] = lane_shift_up1(ct[k], T(0)); }
    G::adjT(Me, C, t);
#pragma unroll
    for (int k = 0; k < W; ++k) S[k] = ce[k] + t[k];
    T first[W];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int64_t i = c0 + (int64_t)rl * K + j;
      T out[W];
      G::adjT(M[j], S, out);
#pragma unroll
      for (int k = 0; k < D; ++k) out[k] += c[j][k];
      if (j == 0) {
#pragma unroll
        for (int k = 0; k < W; ++k) first[k] = out[k];
      }
      if (i < L) {
        T* p = gx + row(i);
#pragma unroll
        for (int k = 0; k < D; ++k) p[k] = out[k];
        p[D] = i == 0 ? g0D : T(0);
      }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) C[k] = lane_bcast63(first[k]);
  }
}

template <class T, class G> int scan_bwd_launch(const void* x, const void* y, const void* g, void* gx, int64_t nseq, int64_t L,
                                                int64_t inner, int left, void* stream) {
  if (nseq < 0 || L < 0 || inner <= 0) return SC_EBADARG;
  if (nseq == 0 || L == 0) return SC_OK;
  if (!g || !gx || (left ? !x : !y)) return SC_EBADARG;
  const void* xy = left ? x : y;
  constexpr int WAVES = 4;
  int64_t blocks = (nseq + WAVES - 1) / WAVES;
#define PPLIE_SCANB(KERNEL, KK)                                                                                       \
  hipLaunchKernelGGL((KERNEL<T, G, WAVES, KK>), dim3((unsigned)blocks), dim3(WAVES * 64), 0,                           \
                     reinterpret_cast<hipStream_t>(stream), static_cast<const T*>(xy), static_cast<const T*>(g),         \
                     static_cast<T*>(gx), nseq, L, inner)
  if (L >= 256) {
    if (left) PPLIE_SCANB(scan_bwd_left_kernel, 2); else PPLIE_SCANB(scan_bwd_right_kernel, 2);
  } else {
    if (left) PPLIE_SCANB(scan_bwd_left_kernel, 1); else PPLIE_SCANB(scan_bwd_right_kernel, 1);
  }
#undef PPLIE_SCANB
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
template <class T> __device__ __forceinline__ T wave_scan_add(T v, T carry, int lane) { return wave_prefix_add(v) + carry; }

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
  // software pipeline: the inputs of chunk c+1 are requested before the scans of chunk c run (one wave owns a
  // whole sequence, so without this every chunk would pay the full HBM latency on its critical path)
  T nh = T(0), ng[3] = {T(0), T(0), T(0)}, na[3] = {T(0), T(0), T(0)}, nr[4] = {T(0), T(0), T(0), T(1)};
  auto fetch = [&](int64_t c0) {
    const int64_t f = c0 + lane;
    const bool ok = f < F;
    const int64_t row = b * F + (ok ? f : 0);
    nh = dt[row];                                    // (raw loads from clamped rows; selected at the consumer)
#pragma unroll
    for (int k = 0; k < 3; ++k) { ng[k] = gyro[row * 3 + k]; na[k] = acc[row * 3 + k]; }
    const T* rkp = rot_known ? rot_known : init_rot;  // (a branch around these loads would end in an s_waitcnt at its join)
    const int64_t rrow = rot_known ? row : b;
#pragma unroll
    for (int k = 0; k < 4; ++k) nr[k] = rkp[rrow * 4 + k];
  };
  fetch(0);
  for (int64_t c0 = 0; c0 < F; c0 += 64) {
    const int64_t f = c0 + lane;
    const bool valid = f < F;
    const int64_t row = b * F + (valid ? f : 0);
    const T h = valid ? nh : T(0);
    T w[3], am[3], rkn[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) { w[k] = (valid ? ng[k] : T(0)) * h; am[k] = valid ? na[k] : T(0); }
#pragma unroll
    for (int k = 0; k < 4; ++k) rkn[k] = valid ? nr[k] : (k == 3 ? T(1) : T(0));
    if (c0 + 64 < F) fetch(c0 + 64);
    T dr[4];
    so3_exp<T>(w, dr);
    T P[4] = {dr[0], dr[1], dr[2], dr[3]};
    wave_scan<T, MulSO3<T>>(P, cR, true, false, lane);            // inclusive: incre_r[f+1]
    T Pex[4];                                                     // incre_r[f]: lane-1's product, the carry for lane 0
#pragma unroll
    for (int k = 0; k < 4; ++k) Pex[k] = lane_shift_up1(P[k], cR[k]);
    // acceleration in the body frame minus gravity
    T Rw[4];
    if (rot_known) {
#pragma unroll
      for (int k = 0; k < 4; ++k) Rw[k] = rkn[k];
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
    for (int k = 0; k < 3; ++k) { cV[k] = lane_bcast63(Dv[k]); cP[k] = lane_bcast63(Dp[k]); }
    cT = lane_bcast63(Dt);
  }
}

// ---------------------------------------------------------------------------------------------
// Backward of imu_integrate (training THROUGH the pre-integrator: examples/module/imu/imu_corrector.py:97 back-propagates a
// position / rotation loss to corrected gyro / acc).  The reference differentiates the composed graph of :359-384 and
// :422-426 -- two cumsum nodes, Act / Inv / Mul / Exp nodes and the log2(F) rounds of the product scan.  In world-frame
// quantities (Q_f = rot_f = r0 dr_0 ... dr_f, Q_{-1} = r0, U_f = R(Q_{f-1}) a_f, a_f = acc_f - R(Qw_f)^T g) the whole
// backward is three reverse plain sums per sequence and element-wise algebra:
//   Sp_j = sum_{f>=j} Gp_f                                   (cotangent reaching the position increments)
//   Sv_j = sum_{m>=j} (Gv_m + h_{m+1} Sp_{m+1})              = suffix(Gv + h Sp)_j - h_j Sp_j
//   gU_j = h_j Sv_j + h_j^2/2 Sp_j ;  g_acc_j = R(Q_{j-1})^T gU_j
//   gQ_j = Gr_j + U_{j+1} x gU_{j+1} + [rot not given] g x (R(Q_j) g_acc_j)      (left-tangent cotangent of Q_j)
//   SQ_k = sum_{i>=k} gQ_i                                   = suffix(A + c)_k - c_k,  c_j = U_j x gU_j
//   g_dr_k = R(Q_{k-1})^T SQ_k ;  g_phi_k = g_dr_k @ Jl(phi_k) ;  g_gyro_k = h_k g_phi_k
//   g_dt_k = Sp_k . vel_k + Sv_k . U_k + gyro_k . g_phi_k
// Reads per step: dt, gyro, acc, rot_f, rot_{f-1} (saved outputs), the three cotangents [, vel_f for g_dt]; writes g_gyro,
// g_acc [, g_dt].  One wavefront per sequence, chunks of 64 steps from the END, lanes reversed (prefix over lanes = suffix
// over steps).  Gradients of the initial state are sums of these outputs (host side, module/imu_preintegrator.py).
// ---------------------------------------------------------------------------------------------
template <class T, int WAVES, bool KNOWN>
__global__ void __launch_bounds__(WAVES * 64)
imu_integrate_bwd_kernel(const T* __restrict__ dt, const T* __restrict__ gyro, const T* __restrict__ acc,
                         const T* __restrict__ rot_known, const T* __restrict__ rot_out, const T* __restrict__ vel_out,
                         const T* __restrict__ init_rot, T gx, T gy, T gz,
                         const T* __restrict__ g_rot, const T* __restrict__ g_vel, const T* __restrict__ g_pos,
                         T* __restrict__ o_gyro, T* __restrict__ o_acc, T* __restrict__ o_dt, int64_t B, int64_t F) {
  const int lane = threadIdx.x & 63, rl = 63 - lane;
  const int64_t b = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (b >= B) return;
  const V3<T> g = v3<T>(gx, gy, gz);
  T cSp[3] = {T(0), T(0), T(0)}, cT2[3] = {T(0), T(0), T(0)}, cS3[3] = {T(0), T(0), T(0)};
  const int64_t nch = (F + 63) / 64;
  // software pipeline: the next (earlier) chunk's 24 scalars per step are in flight while this chunk's sums run
  T n_h, n_gy[3], n_am[3], n_Q[4], n_Qm[4], n_Rw[4], n_Gp[3], n_Gv[3], n_Gr[3], n_vel[3];
  // An absent cotangent (NULL) reads a stream of the same shape that IS there and the consumer selects zero: a launch-uniform
  // `if (g_pos)` around the load is a scalar branch whose join waits for vmcnt(0) -- twelve serialized round trips per chunk.
  const bool has_gp = g_pos != nullptr, has_gv = g_vel != nullptr, has_gr = g_rot != nullptr, has_dt = o_dt != nullptr;
  const T* s_gp = has_gp ? g_pos : gyro;
  const T* s_gv = has_gv ? g_vel : gyro;
  const T* s_gr = has_gr ? g_rot : rot_out;
  const T* s_vel = has_dt ? vel_out : gyro;
  auto fetch = [&](int64_t c) {
    const int64_t f = c * 64 + rl;
    const bool valid = f < F;
    const int64_t row = b * F + (valid ? f : 0);
    n_h = dt[row];                                    // (raw loads from clamped rows; selected at the consumer)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      n_gy[k] = gyro[row * 3 + k];
      n_am[k] = acc[row * 3 + k];
      n_Gp[k] = s_gp[row * 3 + k];
      n_Gv[k] = s_gv[row * 3 + k];
      n_Gr[k] = s_gr[row * 4 + k];
      n_vel[k] = s_vel[row * 3 + k];
    }
    const T* pm = (valid && f > 0) ? rot_out + (row - 1) * 4 : init_rot + b * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      n_Q[k] = rot_out[row * 4 + k];
      n_Qm[k] = pm[k];
      if (KNOWN) n_Rw[k] = rot_known[row * 4 + k];
    }
  };
  fetch(nch - 1);
  for (int64_t c = nch - 1; c >= 0; --c) {
    const int64_t f = c * 64 + rl;
    const bool valid = f < F;
    const int64_t row = b * F + (valid ? f : 0);
    const T h = valid ? n_h : T(0);
    T gyv[3], w[3], am[3], Q[4], Qm[4], Rw[4], Gp[3], Gv[3], Gr[3], velv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      gyv[k] = valid ? n_gy[k] : T(0); w[k] = gyv[k] * h; am[k] = valid ? n_am[k] : T(0);
      Gp[k] = (valid && has_gp) ? n_Gp[k] : T(0); Gv[k] = (valid && has_gv) ? n_Gv[k] : T(0);
      Gr[k] = (valid && has_gr) ? n_Gr[k] : T(0); velv[k] = (valid && has_dt) ? n_vel[k] : T(0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T idk = k == 3 ? T(1) : T(0);
      Q[k] = valid ? n_Q[k] : idk; Qm[k] = valid ? n_Qm[k] : idk; Rw[k] = KNOWN ? (valid ? n_Rw[k] : idk) : Q[k];
    }
    if (c > 0) fetch(c - 1);
    T Sp[3], T2[3], Sv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) Sp[k] = wave_prefix_add<T>(Gp[k]) + cSp[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      T2[k] = wave_prefix_add<T>(Gv[k] + h * Sp[k]) + cT2[k];
      Sv[k] = T2[k] - h * Sp[k];
    }
    const T hh = T(0.5) * h * h;
    const V3<T> gU = v3<T>(h * Sv[0] + hh * Sp[0], h * Sv[1] + hh * Sp[1], h * Sv[2] + hh * Sp[2]);
    const V3<T> a = v3(am) - quat_rotate(-v3(Rw), Rw[3], g);
    const V3<T> U = quat_rotate(v3(Qm), Qm[3], a);
    const V3<T> ga = adj_rotate_T(v3(Qm), Qm[3], gU);
    const V3<T> cc = cross(U, gU);
    V3<T> A = v3(Gr);
    if (!KNOWN) A = A + cross(g, adj_rotate(v3(Q), Q[3], ga));
    T s3[3] = {A.x + cc.x, A.y + cc.y, A.z + cc.z}, S3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) S3[k] = wave_prefix_add<T>(s3[k]) + cS3[k];
    const V3<T> SQ = v3<T>(S3[0] - cc.x, S3[1] - cc.y, S3[2] - cc.z);
    const V3<T> gdr = adj_rotate_T(v3(Qm), Qm[3], SQ);
    T gdr3[3] = {gdr.x, gdr.y, gdr.z}, gphi[3];
    so3_exp_bwd<T>(w, gdr3, gphi);
    if (valid) {
#pragma unroll
      for (int k = 0; k < 3; ++k) o_gyro[row * 3 + k] = h * gphi[k];
      put(ga, o_acc + row * 3);
      if (o_dt) {
        T acc_dt = Sv[0] * U.x + Sv[1] * U.y + Sv[2] * U.z;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc_dt += Sp[k] * velv[k] + gyv[k] * gphi[k];
        o_dt[row] = acc_dt;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { cSp[k] = lane_bcast63(Sp[k]); cT2[k] = lane_bcast63(T2[k]); cS3[k] = lane_bcast63(S3[k]); }
  }
}

template <class T>
int imu_integrate_bwd_launch(const void* dt, const void* gyro, const void* acc, const void* rot_known, const void* rot_out,
                             const void* vel_out, const void* init_rot, const double* gravity, const void* g_rot,
                             const void* g_vel, const void* g_pos, void* o_gyro, void* o_acc, void* o_dt, int64_t B, int64_t F,
                             void* stream) {
  if (B < 0 || F < 0) return SC_EBADARG;
  if (B == 0 || F == 0) return SC_OK;
  if (!dt || !gyro || !acc || !rot_out || !init_rot || !gravity || !o_gyro || !o_acc || (o_dt && !vel_out)) return SC_EBADARG;
  constexpr int WAVES = 4;
  int64_t blocks = (B + WAVES - 1) / WAVES;
#define PPLIE_IMUB(KN)                                                                                                   \
  hipLaunchKernelGGL((imu_integrate_bwd_kernel<T, WAVES, KN>), dim3((unsigned)blocks), dim3(WAVES * 64), 0,                \
                     reinterpret_cast<hipStream_t>(stream), (const T*)dt, (const T*)gyro, (const T*)acc, (const T*)rot_known, \
                     (const T*)rot_out, (const T*)vel_out, (const T*)init_rot, (T)gravity[0], (T)gravity[1], (T)gravity[2],  \
                     (const T*)g_rot, (const T*)g_vel, (const T*)g_pos, (T*)o_gyro, (T*)o_acc, (T*)o_dt, B, F)
  if (rot_known) PPLIE_IMUB(true); else PPLIE_IMUB(false);
#undef PPLIE_IMUB
  return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
}

// K consecutive steps per lane (chunks of 64 K steps): the local products / sums of a lane's K steps are formed
// sequentially, ONE wave scan runs over the 64 lane totals, and the K per-step values are rebuilt from the exclusive
// prefix.  The kernel above is bound by the instruction issue of its cross-lane scans (~800 VALU instructions per
// 64-step chunk, ~300 of them the SO3 product scan); here a scan is paid once per K steps.  States only: the variant
// that also writes the covariance's aux streams is store-bound and gains nothing (launcher below).
template <class T, int WAVES, int K, bool KNOWN>
__global__ void __launch_bounds__(WAVES * 64)
imu_integrate_multi_kernel(const T* __restrict__ dt, const T* __restrict__ gyro, const T* __restrict__ acc,
                           const T* __restrict__ rot_known, const T* __restrict__ init_rot, const T* __restrict__ init_vel,
                           const T* __restrict__ init_pos, T gx, T gy, T gz,
                           T* __restrict__ out_rot, T* __restrict__ out_vel, T* __restrict__ out_pos, int64_t B, int F) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (b >= B) return;
  // this sequence's streams: 64-bit base once, 32-bit offsets inside the sequence (4 F < 2^31, checked by the launcher)
  dt += b * F; gyro += b * F * 3; acc += b * F * 3;
  out_rot += b * F * 4; out_vel += b * F * 3; out_pos += b * F * 3;
  if (KNOWN) rot_known += b * F * 4;
  T R0[4], v0[3], p0[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) R0[k] = init_rot[b * 4 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { v0[k] = init_vel[b * 3 + k]; p0[k] = init_pos[b * 3 + k]; }
  const V3<T> g = v3<T>(gx, gy, gz);
  T cR[4] = {T(0), T(0), T(0), T(1)};
  T cV[3] = {T(0), T(0), T(0)}, cP[3] = {T(0), T(0), T(0)}, cT = T(0);
  T nh[K], ng[K][3], na[K][3], nr[KNOWN ? K : 1][4];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int f = c0 + lane * K + j;
      const bool ok = f < F;
      const int row = ok ? f : 0;
      nh[j] = dt[row];                                // (raw loads from clamped rows; selected at the consumer)
#pragma unroll
      for (int k = 0; k < 3; ++k) { ng[j][k] = gyro[row * 3 + k]; na[j][k] = acc[row * 3 + k]; }
      if (KNOWN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) nr[j][k] = rot_known[row * 4 + k];
      }
    }
  };
  fetch(0);
  constexpr int CH = 64 * K;
  for (int c0 = 0; c0 < F; c0 += CH) {
    T h[K], am[K][3], rkn[KNOWN ? K : 1][4], dr[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool ok = c0 + lane * K + j < F;
      h[j] = ok ? nh[j] : T(0);
      T w[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { w[k] = (ok ? ng[j][k] : T(0)) * h[j]; am[j][k] = ok ? na[j][k] : T(0); }
      if (KNOWN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rkn[j][k] = ok ? nr[j][k] : (k == 3 ? T(1) : T(0));
      }
      so3_exp<T>(w, dr[j]);
    }
    if (c0 + CH < F) fetch(c0 + CH);
    // products: L[j] = dr[0] .. dr[j] inside the lane, P = scan of the lane totals, E = the product before this lane
    T L[K][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) L[0][k] = dr[0][k];
#pragma unroll
    for (int j = 1; j < K; ++j) so3_mul<T>(L[j - 1], dr[j], L[j]);
    T P[4] = {L[K - 1][0], L[K - 1][1], L[K - 1][2], L[K - 1][3]};
    wave_scan<T, MulSO3<T>>(P, cR, true, false, lane);
    T E[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) E[k] = lane_shift_up1(P[k], cR[k]);
    T Pj[K][4];                                                  // incre_r[f+1] of each of the lane's steps
#pragma unroll
    for (int j = 0; j < K - 1; ++j) so3_mul<T>(E, L[j], Pj[j]);
#pragma unroll
    for (int k = 0; k < 4; ++k) Pj[K - 1][k] = P[k];
    T u[K][3], dv[K][3];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      T Rw[4];
      if (KNOWN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) Rw[k] = rkn[j][k];
      } else {
        so3_mul<T>(R0, Pj[j], Rw);
      }
      const V3<T> gi = quat_rotate(-v3(Rw), Rw[3], g);
      const V3<T> aj = v3(am[j]) - gi;
      const T* Pex = j == 0 ? E : Pj[j - 1];                     // incre_r[f]
      const V3<T> uj = quat_rotate(v3(Pex), Pex[3], aj);
      put(uj, u[j]);
#pragma unroll
      for (int k = 0; k < 3; ++k) dv[j][k] = u[j][k] * h[j];
    }
    // velocity / position / time: exclusive lane prefix + running sum inside the lane
    T Dv[K][3], Dp[K][3], Dt[K], endV[3], endP[3], endT;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      T tot = dv[0][k];
#pragma unroll
      for (int j = 1; j < K; ++j) tot += dv[j][k];
      endV[k] = wave_scan_add(tot, cV[k], lane);
      T run = lane_shift_up1(endV[k], cV[k]);
#pragma unroll
      for (int j = 0; j < K; ++j) { run += dv[j][k]; Dv[j][k] = run; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      T dp[K];
      T tot = T(0);
#pragma unroll
      for (int j = 0; j < K; ++j) {
        dp[j] = (Dv[j][k] - dv[j][k]) * h[j] + u[j][k] * (T(0.5) * h[j] * h[j]);   // incre_v[f] dt + R a dt^2 / 2
        tot += dp[j];
      }
      endP[k] = wave_scan_add(tot, cP[k], lane);
      T run = lane_shift_up1(endP[k], cP[k]);
#pragma unroll
      for (int j = 0; j < K; ++j) { run += dp[j]; Dp[j][k] = run; }
    }
    {
      T tot = h[0];
#pragma unroll
      for (int j = 1; j < K; ++j) tot += h[j];
      endT = wave_scan_add(tot, cT, lane);
      T run = lane_shift_up1(endT, cT);
#pragma unroll
      for (int j = 0; j < K; ++j) { run += h[j]; Dt[j] = run; }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int f = c0 + lane * K + j;
      if (f < F) {
        T Rf[4];
        so3_mul<T>(R0, Pj[j], Rf);
        const V3<T> vel = v3(v0) + quat_rotate(v3(R0), R0[3], v3(Dv[j]));
        const V3<T> pos = v3(p0) + quat_rotate(v3(R0), R0[3], v3(Dp[j])) + Dt[j] * v3(v0);
#pragma unroll
        for (int k = 0; k < 4; ++k) out_rot[f * 4 + k] = Rf[k];
        put(vel, out_vel + f * 3);
        put(pos, out_pos + f * 3);
      }
    }
    bcast_vec<T, 4>(P, cR, 63);
#pragma unroll
    for (int k = 0; k < 3; ++k) { cV[k] = lane_bcast63(endV[k]); cP[k] = lane_bcast63(endP[k]); }
    cT = lane_bcast63(endT);
  }
}

// ---------------------------------------------------------------------------------------------
// Covariance (imu_preintegrator.py:428-465).  The reference's code evaluates
//   cov = sum_{k=0..F} P_k Bc_k P_k^T,   P_k = A_k A_{k+1} ... A_{F-1} (P_F = I),   Bc_0 = init_cov
// through a cumprod of flipped 9x9 matrices (:462-464).  Note this is NOT the textbook recursion of
// its docstring (the products share their RIGHT factors).
// A_k = I9 with [0:3,0:3] = Rk^T, [3:6,0:3] = -Rij Ha dt, [6:9,0:3] = -Rij Ha dt^2/2, [6:9,3:6] = dt I (:442-448)
// Bc_{k+1} = (Bg Cg Bg^T + Ba Ca Ba^T)/dt, Bg[0:3] = Jr(Rk) dt, Ba[3:6] = Rij dt, Ba[6:9] = Rij dt^2/2 (:451-460)
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

// ---------------------------------------------------------------------------------------------
// Covariance as scans + a reduction.
// The transition matrices are block lower-triangular with identity blocks,
//   A_k = [[Rk^T, 0, 0], [M1_k, I, 0], [h_k/2 M1_k, h_k I, I]],   M1_k = -h_k Rij_k skew(a_k),
// and so is every product of them:  P_k = A_k ... A_{F-1} = [[S_k, 0, 0], [X_k, I, 0], [Y_k, t_k I, I]] with
//   S_k = Rk_k^T S_{k+1}                       (a suffix product of the inverse gyro increments: one SO3 scan)
//   X_k = M1_k S_{k+1} + X_{k+1}               (suffix sum of G_k := M1_k S_{k+1})
//   Y_k = h_k/2 G_k + h_k X_{k+1} + Y_{k+1}    (suffix sum)
//   t_k = h_k + t_{k+1}                        (suffix sum)
// so all P_k come from wave-level scans (lanes hold the chunk's steps in reverse order), and  cov = P_0 C_0 P_0^T + sum_{j=0}^{F-1} P_{j+1} Bc(j) P_{j+1}^T,
//   P Bc P^T = V (h Cg) V^T + U (h Ca) U^T,  V = [S; X; Y] Jr_j,  U = [0; Rij_j; (t + h_j/2) Rij_j],
// is a sum over steps that each lane accumulates privately (45 symmetric entries) and the wave reduces
// once.  One wavefront per sequence walks it backwards in 64-step chunks; no step depends on another
// except through the scans' carried values.
// ---------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ void mat3_mul(const T* A, const T* B, T* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

template <class T, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
imu_cov_scan_kernel(const T* __restrict__ dt, const T* __restrict__ rk, const T* __restrict__ rij, const T* __restrict__ a,
                    const T* __restrict__ init_cov, const T* __restrict__ gyro_cov, int64_t gc_sb, int64_t gc_sf,
                    const T* __restrict__ acc_cov, int64_t ac_sb, int64_t ac_sf, T* __restrict__ cov, int64_t B, int64_t F) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (b >= B) return;
  // carried suffix values at the first step after the current chunk (P_F = I)
  T cS[4] = {T(0), T(0), T(0), T(1)}, cX[9], cY[9], ct = T(0);
#pragma unroll
  for (int i = 0; i < 9; ++i) { cX[i] = T(0); cY[i] = T(0); }
  T acc[45];                                   // upper triangle of this lane's share of cov
#pragma unroll
  for (int i = 0; i < 45; ++i) acc[i] = T(0);

  const int64_t nchunks = (F + 63) / 64;
  // software pipeline (as in imu_integrate_kernel): the next (earlier) chunk's inputs are in flight during the scans
  T nh = T(0), nq[4] = {T(0), T(0), T(0), T(1)}, nqij[4] = {T(0), T(0), T(0), T(1)}, nav[3] = {T(0), T(0), T(0)};
  auto fetch = [&](int64_t ch) {
    const int64_t jj = ch * 64 + (63 - lane);
    const bool ok = jj < F;
    const int64_t row = b * F + (ok ? jj : 0);
    nh = dt[row];                                    // (raw loads from clamped rows; selected at the consumer)
#pragma unroll
    for (int i = 0; i < 4; ++i) { nq[i] = rk[row * 4 + i]; nqij[i] = rij[row * 4 + i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) nav[i] = a[row * 3 + i];
  };
  fetch(nchunks - 1);
  for (int64_t ch = nchunks - 1; ch >= 0; --ch) {
    const int64_t j = ch * 64 + (63 - lane);   // lanes walk the chunk backwards: a suffix over steps is a prefix over lanes
    const bool valid = j < F;
    const T h = valid ? nh : T(0);
    T q[4], qij[4], av[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const T idk = i == 3 ? T(1) : T(0); q[i] = valid ? nq[i] : idk; qij[i] = valid ? nqij[i] : idk; }
#pragma unroll
    for (int i = 0; i < 3; ++i) av[i] = valid ? nav[i] : T(0);
    if (ch > 0) fetch(ch - 1);
    // ---- S: suffix product of inverse increments, s_j = dr_j^-1 * s_{j+1}
    T sv[4] = {-q[0], -q[1], -q[2], q[3]};
    wave_scan<T, MulSO3<T>>(sv, cS, true, true, lane);             // inclusive S_j = dr_j^-1 ... dr_last^-1 * carry
    T sx[4];                                   // exclusive S_{j+1}
#pragma unroll
    for (int i = 0; i < 4; ++i) sx[i] = lane_shift_up1(sv[i], cS[i]);
    // ---- G_j = M1_j S_{j+1},  M1_j = -h Rij skew(a)
    T Rj[9], Sx[9], M1[9], G[9];
    quat_to_matrix<T>(qij, Rj);
    quat_to_matrix<T>(sx, Sx);
    {
      const T Ha[9] = {T(0), -av[2], av[1], av[2], T(0), -av[0], -av[1], av[0], T(0)};
      mat3_mul<T>(Rj, Ha, M1);
#pragma unroll
      for (int i = 0; i < 9; ++i) M1[i] = -h * M1[i];
    }
    mat3_mul<T>(M1, Sx, G);
    // ---- X (inclusive) and its exclusive shift; Z_j = h/2 G_j + h X_{j+1}; Y ; t
    T X[9], Xx[9], Y[9], Yx[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      X[i] = wave_prefix_add(G[i]) + cX[i];
      Xx[i] = lane_shift_up1(X[i], cX[i]);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      Y[i] = wave_prefix_add(T(0.5) * h * G[i] + h * Xx[i]) + cY[i];
      Yx[i] = lane_shift_up1(Y[i], cY[i]);
    }
    const T tin = wave_prefix_add(h) + ct;
    const T tx = lane_shift_up1(tin, ct);
    // ---- this step's term  P_{j+1} Bc(j) P_{j+1}^T
    if (valid) {
      T phi[3], Jr[9], V[27];
      so3_log<T>(q, phi);
      so3_jr<T>(phi, Jr);
      mat3_mul<T>(Sx, Jr, V);                  // rows 0..2
      mat3_mul<T>(Xx, Jr, V + 9);              // rows 3..5
      mat3_mul<T>(Yx, Jr, V + 18);             // rows 6..8
      T dg[3], da[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { dg[i] = h * gyro_cov[b * gc_sb + j * gc_sf + i]; da[i] = h * acc_cov[b * ac_sb + j * ac_sf + i]; }
      const T tu = tx + T(0.5) * h;            // U = [0; Rj; tu Rj]
      // V (h Cg) V^T with the scaled copy Vd = V diag(h Cg) formed once: 3 FMAs per entry of the upper triangle
      T Vd[27];
#pragma unroll
      for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) Vd[r * 3 + k] = V[r * 3 + k] * dg[k];
      // Rj (h Ca) Rj^T once (symmetric 3x3); U (h Ca) U^T is that block times 1, tu or tu^2
      T Wa[9];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = rr; cc < 3; ++cc) {
          const T w = Rj[rr * 3] * da[0] * Rj[cc * 3] + Rj[rr * 3 + 1] * da[1] * Rj[cc * 3 + 1] + Rj[rr * 3 + 2] * da[2] * Rj[cc * 3 + 2];
          Wa[rr * 3 + cc] = w;
          Wa[cc * 3 + rr] = w;
        }
#pragma unroll
      for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int c = r; c < 9; ++c) {
          const int e = r * 9 - (r * (r - 1)) / 2 + (c - r);      // index in the packed upper triangle
          T sacc = Vd[r * 3] * V[c * 3] + Vd[r * 3 + 1] * V[c * 3 + 1] + Vd[r * 3 + 2] * V[c * 3 + 2];
          if (r >= 3) {                        // U rows 0..2 are zero
            const T f = (r < 6 ? T(1) : tu) * (c < 6 ? T(1) : tu);
            sacc += f * Wa[(r % 3) * 3 + (c % 3)];
          }
          acc[e] += sacc;
        }
    }
    // ---- carries for the next (earlier) chunk: the inclusive values of the chunk's first step (lane 63)
#pragma unroll
    for (int i = 0; i < 4; ++i) cS[i] = lane_bcast63(sv[i]);
#pragma unroll
    for (int i = 0; i < 9; ++i) { cX[i] = lane_bcast63(X[i]); cY[i] = lane_bcast63(Y[i]); }
    ct = lane_bcast63(tin);
  }
  // wave reduction of the per-lane partial sums
#pragma unroll
  for (int e = 0; e < 45; ++e) {
    T v = acc[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    acc[e] = v;
  }
  if (lane == 0) {
    // P_0 = [[S_0,0,0],[X_0,I,0],[Y_0,t_0 I,I]] (the carries) ; cov = P_0 init_cov P_0^T + sum
    T P[81], S0[9];
#pragma unroll
    for (int i = 0; i < 81; ++i) P[i] = T(0);
    quat_to_matrix<T>(cS, S0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        P[i * 9 + jj] = S0[i * 3 + jj];
        P[(3 + i) * 9 + jj] = cX[i * 3 + jj];
        P[(6 + i) * 9 + jj] = cY[i * 3 + jj];
      }
#pragma unroll
    for (int i = 0; i < 3; ++i) { P[(3 + i) * 9 + 3 + i] = T(1); P[(6 + i) * 9 + 3 + i] = ct; P[(6 + i) * 9 + 6 + i] = T(1); }
    // the accumulated step terms are symmetric by construction; P_0 init_cov P_0^T is formed in full
    // (the reference does not symmetrise a caller-supplied init_cov)
    // (fully unrolled: a run-time index into acc[] would move the 45 accumulators of EVERY lane to scratch
    //  memory for the whole kernel -- measured as 706 MB of HBM writes per launch before this pragma)
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int c = r; c < 9; ++c) {
        const int e = r * 9 - (r * (r - 1)) / 2 + (c - r);
        cov[b * 81 + r * 9 + c] = acc[e];
        cov[b * 81 + c * 9 + r] = acc[e];
      }
    for (int r = 0; r < 9; ++r) {
      T tl[9];
      for (int l = 0; l < 9; ++l) {
        T s0 = T(0);
        for (int m = 0; m < 9; ++m) s0 += P[r * 9 + m] * init_cov[b * 81 + m * 9 + l];
        tl[l] = s0;
      }
      for (int c = 0; c < 9; ++c) {
        T s1 = T(0);
        for (int l = 0; l < 9; ++l) s1 += tl[l] * P[c * 9 + l];
        cov[b * 81 + r * 9 + c] += s1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Covariance, version 2: a LANE owns LS consecutive steps (a segment) and walks them sequentially; the cross-lane scan
// is paid once per 64 LS steps instead of once per 64, and the kernel reads the RAW inputs (dt, gyro, acc) plus the
// integrated rotations instead of three auxiliary streams the integration kernel had to write for it.
//
//   per chunk of 64 LS steps (processed from the end of the sequence backwards; lane l holds segment 63 - l):
//     walk 1   the segment's own product  L = A_first ... A_last  in the closed structure [[S,0,0],[X,I,0],[Y,tI,I]]
//              (recurrences above, started from the identity)
//     scan     inclusive products over the lanes (later segments first) and the carry of the later chunks:
//              P_after(segment) = product of everything behind it            -- 7 DPP steps on 23 values
//     walk 2   the same steps again from P_after, adding each step's  V (h Cg) V^T + U (h Ca) U^T  to 45 accumulators
//   a step costs ~240 (walk 1) + ~590 (walk 2) VALU instructions + 1/LS of the scan (~1300), against ~1750 for the
//   one-step-per-lane kernel above with its 19 scalar scans and SO3 scan per 64 steps.
// Per step it needs: h, dr = Exp(gyro h) (recomputed), a = acc - Rw^-1 g (Rw: the integrated or the known rotation),
// Rij = C * Rout (C = Rij0 * r0^-1 per sequence), and Jr(Log dr) = Jr(gyro h) for |gyro h| < pi.
// ---------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ void quat_matrix(const T* q, T* M) {
  // the matrix of SO3_Act (p + 2 w v x p + 2 v x (v x p)) column by column, in closed form: (1 - 2|v|^2) I + 2 v v^T + 2 w [v]x
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  M[0] = T(1) - T(2) * (yy + zz); M[1] = T(2) * (xy - wz);        M[2] = T(2) * (xz + wy);
  M[3] = T(2) * (xy + wz);        M[4] = T(1) - T(2) * (xx + zz); M[5] = T(2) * (yz - wx);
  M[6] = T(2) * (xz - wy);        M[7] = T(2) * (yz + wx);        M[8] = T(1) - T(2) * (xx + yy);
}

// the structured 9x9 products: element = { S quaternion [0:4], X [4:13], Y [13:22], t [22] };  c = a b
template <class T> struct MulImuP {
  enum { W = 23 };
  static __device__ __forceinline__ void mul(const T* a, const T* b, T* c) {
    T Sb[9], XS[9], YS[9], q[4];
    quat_matrix<T>(b, Sb);
    mat3_mul<T>(a + 4, Sb, XS);
    mat3_mul<T>(a + 13, Sb, YS);
    so3_mul<T>(a, b, q);
    const T ta = a[22];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const T bx = b[4 + i];
      c[13 + i] = YS[i] + ta * bx + b[13 + i];
      c[4 + i] = XS[i] + bx;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = q[i];
    c[22] = ta + b[22];
  }
  static __device__ __forceinline__ T ident(int k) { return k == 3 ? T(1) : T(0); }
};

// one step of the backward recurrence  P <- A_j P  on the structured state (S, X, Y, t); optionally (ACC) first adds the
// step's noise term  P Bc_j P^T  to the accumulators
// ISO (with ACC): the accelerometer noise is isotropic and the same for every step (the module's default: one scalar), so
// Rij Ca Rij^T = ca I and the U-term of a step is ca h [0; I; tu I][..]^T: three SCALAR sums (h, h tu, h tu^2 in uacc) instead of
// a 3x3 congruence and 39 accumulator updates per step
template <class T, bool ACC, bool ISO = false>
__device__ __forceinline__ void imu_cov_step(T* P, T h, const T* gy, const T* av, const T* qij, const T* dg, const T* da, T* acc,
                                             T* uacc = nullptr) {
  T S[9], Rj[9], M1[9], G[9];
  quat_matrix<T>(P, S);
  quat_matrix<T>(qij, Rj);
  T w[3] = {gy[0] * h, gy[1] * h, gy[2] * h};
  T dr[4];
  so3_exp<T>(w, dr);
  if (ACC) {
    T phi[3] = {w[0], w[1], w[2]};
    if (!(w[0] * w[0] + w[1] * w[1] + w[2] * w[2] < T(9.8))) so3_log<T>(dr, phi);   // Log(Exp(w)) = w below pi
    T Jr[9], V[27];
    so3_jr<T>(phi, Jr);
    mat3_mul<T>(S, Jr, V);
    mat3_mul<T>(P + 4, Jr, V + 9);
    mat3_mul<T>(P + 13, Jr, V + 18);
    const T tu = P[22] + T(0.5) * h;            // U = [0; Rj; tu Rj]
    T Vd[27];
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) Vd[r * 3 + k] = V[r * 3 + k] * (h * dg[k]);
    T Wa[9];
    if (ISO) {
      uacc[0] += h;
      uacc[1] += h * tu;
      uacc[2] += h * tu * tu;
    } else {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = rr; cc < 3; ++cc) {
          const T ww = h * (Rj[rr * 3] * da[0] * Rj[cc * 3] + Rj[rr * 3 + 1] * da[1] * Rj[cc * 3 + 1] + Rj[rr * 3 + 2] * da[2] * Rj[cc * 3 + 2]);
          Wa[rr * 3 + cc] = ww;
          Wa[cc * 3 + rr] = ww;
        }
    }
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int c = r; c < 9; ++c) {
        const int e = r * 9 - (r * (r - 1)) / 2 + (c - r);
        T sacc = Vd[r * 3] * V[c * 3] + Vd[r * 3 + 1] * V[c * 3 + 1] + Vd[r * 3 + 2] * V[c * 3 + 2];
        if (!ISO && r >= 3) {
          const T f = (r < 6 ? T(1) : tu) * (c < 6 ? T(1) : tu);
          sacc += f * Wa[(r % 3) * 3 + (c % 3)];
        }
        acc[e] += sacc;
      }
  }
  // M1 = -h Rj skew(a): column i of Rj skew(a) is Rj (a x e_i)
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    M1[r * 3 + 0] = -h * (av[2] * Rj[r * 3 + 1] - av[1] * Rj[r * 3 + 2]);
    M1[r * 3 + 1] = -h * (av[0] * Rj[r * 3 + 2] - av[2] * Rj[r * 3 + 0]);
    M1[r * 3 + 2] = -h * (av[1] * Rj[r * 3 + 0] - av[0] * Rj[r * 3 + 1]);
  }
  mat3_mul<T>(M1, S, G);
  const T hh = T(0.5) * h;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const T x0 = P[4 + i];
    P[13 + i] += hh * G[i] + h * x0;
    P[4 + i] = x0 + G[i];
  }
  const T si[4] = {-dr[0], -dr[1], -dr[2], dr[3]};
  T q[4];
  so3_mul<T>(si, P, q);
#pragma unroll
  for (int i = 0; i < 4; ++i) P[i] = q[i];
  P[22] += h;
}

// (fp64 gets the whole register file: under the three-workgroup budget of 168 VGPRs the LS = 2 build spilled 548 registers, 1.2 KB of
//  scratch per lane -- the reference's tests run the module in fp64)
template <class T, int WAVES, int LS, bool ISO>
__global__ void __launch_bounds__(WAVES * 64, sizeof(T) == 8 ? 1 : 3)
imu_cov_seg_kernel(const T* __restrict__ dt, const T* __restrict__ gyro, const T* __restrict__ accel, const T* __restrict__ rout,
                   const T* __restrict__ rw, const T* __restrict__ C, const T* __restrict__ init_cov, const T* __restrict__ gyro_cov,
                   int64_t gc_sb, int64_t gc_sf, const T* __restrict__ acc_cov, int64_t ac_sb, int64_t ac_sf, T g0, T g1, T g2,
                   T* __restrict__ cov, int64_t B, int64_t F, int64_t in_mask = -1) {
  // in_mask (tools/micro/imu_cov_chain.hip; the library passes -1): sequence b reads the INPUT streams of sequence b & in_mask --
  // a wave-uniform AND on the row address, the same instructions and registers otherwise.  With a small mask the inputs of all B
  // sequences are a few MB that stay cache-resident: what this kernel costs when its bytes are free.
  __shared__ T sd[WAVES * LS * 11 * 64];          // [wave][step of the segment][field][lane]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * WAVES + wv;
  if (b >= B) return;
  T carry[23];
#pragma unroll
  for (int i = 0; i < 23; ++i) carry[i] = MulImuP<T>::ident(i);
  T acc[45];
#pragma unroll
  for (int i = 0; i < 45; ++i) acc[i] = T(0);
  T cq[4] = {T(0), T(0), T(0), T(1)};
  if (C) {
#pragma unroll
    for (int i = 0; i < 4; ++i) cq[i] = C[b * 4 + i];
  }
  const bool per_step_cov = gc_sf != 0 || ac_sf != 0;
  T dg0[3], da0[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { dg0[i] = gyro_cov[b * gc_sb + i]; da0[i] = acc_cov[b * ac_sb + i]; }
  // ISO: the caller vouches for ONE isotropic accelerometer variance for every sequence and step (launcher below)
  constexpr bool iso = ISO;
  T uacc[3] = {T(0), T(0), T(0)};
  const int64_t CH = 64 * LS;
  const int64_t nchunks = (F + CH - 1) / CH;
  const int seg = 63 - lane;                      // lanes walk a chunk's segments backwards: a suffix over steps is a prefix over lanes
  for (int64_t ch = nchunks - 1; ch >= 0; --ch) {
    const int64_t j0 = ch * CH + (int64_t)seg * LS;
    // ---- this segment's steps: raw inputs -> (h, gyro, a, Rij), parked in lane-private LDS slots for the two walks
    // (11 values per step; held in registers the unrolled walks need > 256 VGPRs).  Unrolled: the raw loads of all LS steps are
    // independent and go out together -- rolled up, every step of the segment paid its own HBM round trip, LS of them per chunk
    // on the critical path of a wave that has only two others on its SIMD to hide behind.
#pragma unroll
    for (int s = 0; s < LS; ++s) {
      const int64_t j = j0 + s;
      const bool ok = j < F;
      const int64_t row = (b & in_mask) * F + (ok ? j : 0);
      T ro[4], rwq[4], ac[3], gyv[3], qq[4];
      // (unconditional loads from the clamped row, THEN the selects: `ok ? p[i] : d` is a branch around every load with a wait at
      //  its join -- the LS steps' loads went out one after the other)
      T hv = dt[row];
#pragma unroll
      for (int i = 0; i < 3; ++i) { gyv[i] = gyro[row * 3 + i]; ac[i] = accel[row * 3 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { ro[i] = rout[row * 4 + i]; rwq[i] = rw[row * 4 + i]; }
      hv = ok ? hv : T(0);
#pragma unroll
      for (int i = 0; i < 3; ++i) { gyv[i] = ok ? gyv[i] : T(0); ac[i] = ok ? ac[i] : T(0); }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const T idk = i == 3 ? T(1) : T(0); ro[i] = ok ? ro[i] : idk; rwq[i] = ok ? rwq[i] : idk; }
      const V3<T> gb = quat_rotate_inv(v3(rwq), rwq[3], v3<T>(g0, g1, g2));       // Rw^-1 g
      if (C) so3_mul<T>(cq, ro, qq);
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) qq[i] = ro[i];
      }
      T* slot = sd + ((size_t)(wv * LS + s) * 11) * 64 + lane;
      slot[0] = hv;
#pragma unroll
      for (int i = 0; i < 3; ++i) slot[(1 + i) * 64] = gyv[i];
      slot[4 * 64] = ok ? ac[0] - gb.x : T(0); slot[5 * 64] = ok ? ac[1] - gb.y : T(0); slot[6 * 64] = ok ? ac[2] - gb.z : T(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) slot[(7 + i) * 64] = qq[i];
    }
    // ---- walk 1: the segment's own product
    T P[23];
#pragma unroll
    for (int i = 0; i < 23; ++i) P[i] = MulImuP<T>::ident(i);
#pragma unroll 1
    for (int s = LS - 1; s >= 0; --s) {
      const T* slot = sd + ((size_t)(wv * LS + s) * 11) * 64 + lane;
      const T gy[3] = {slot[64], slot[128], slot[192]}, av[3] = {slot[256], slot[320], slot[384]};
      const T qij[4] = {slot[448], slot[512], slot[576], slot[640]};
      imu_cov_step<T, false>(P, slot[0], gy, av, qij, nullptr, nullptr, nullptr);
    }
    // ---- scan over the lanes (later segments sit in lower lanes) and the later chunks' carry.
    // Incl(l) = L_l Incl(l-1), Incl(-1) = carry, in the structured form (S, X, Y, t) with  c = a b:  S_c = S_a S_b,
    // X_c = X_a R(S_b) + X_b,  Y_c = Y_a R(S_b) + t_a X_b + Y_b,  t_c = t_a + t_b.  Unrolled over the lanes, only the rotation
    // part is a PRODUCT scan; with E(l) = Incl(l-1) the rest are plain prefix sums of transported terms:
    //   X_incl = carry.X + prefix( X_l R(S_E(l)) ),   Y_incl = carry.Y + prefix( Y_l R(S_E(l)) + t_l X_E(l) ),   t_incl = carry.t + prefix(t_l)
    // -- 7 DPP steps of a quaternion product and 19 scalar prefix sums instead of 7 steps of the 23-value structured product
    // (two 3x3 matrix products each): ~400 instead of ~1390 VALU instructions per chunk.
    T Q[23];                                     // exclusive: everything behind this segment
    {
      T Sq[4] = {P[0], P[1], P[2], P[3]}, cS[4] = {carry[0], carry[1], carry[2], carry[3]};
      wave_scan<T, MulSO3<T>>(Sq, cS, true, true, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) Q[i] = lane_shift_up1(Sq[i], cS[i]);
      T Re[9], Xh[9], Yh[9];
      quat_matrix<T>(Q, Re);
      mat3_mul<T>(P + 4, Re, Xh);
      mat3_mul<T>(P + 13, Re, Yh);
      const T tl = P[22];
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const T xi = wave_prefix_add<T>(Xh[i]) + carry[4 + i];
        Q[4 + i] = lane_shift_up1(xi, carry[4 + i]);
        const T yi = wave_prefix_add<T>(Yh[i] + tl * Q[4 + i]) + carry[13 + i];
        Q[13 + i] = lane_shift_up1(yi, carry[13 + i]);
        carry[4 + i] = lane_bcast63(xi);
        carry[13 + i] = lane_bcast63(yi);
      }
      const T ti = wave_prefix_add<T>(tl) + carry[22];
      Q[22] = lane_shift_up1(ti, carry[22]);
      carry[22] = lane_bcast63(ti);
#pragma unroll
      for (int i = 0; i < 4; ++i) carry[i] = lane_bcast63(Sq[i]);
    }
    // ---- walk 2: the true suffix products and the noise terms
#pragma unroll 1
    for (int s = LS - 1; s >= 0; --s) {
      T dg[3], da[3];
      if (per_step_cov) {
        const int64_t j = j0 + s < F ? j0 + s : 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) { dg[i] = gyro_cov[b * gc_sb + j * gc_sf + i]; da[i] = acc_cov[b * ac_sb + j * ac_sf + i]; }
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) { dg[i] = dg0[i]; da[i] = da0[i]; }
      }
      const T* slot = sd + ((size_t)(wv * LS + s) * 11) * 64 + lane;
      const T gy[3] = {slot[64], slot[128], slot[192]}, av[3] = {slot[256], slot[320], slot[384]};
      const T qij[4] = {slot[448], slot[512], slot[576], slot[640]};
      imu_cov_step<T, true, ISO>(Q, slot[0], gy, av, qij, dg, da, acc, uacc);
    }
  }
  if (iso) {                                      // ca (sum h) I, ca (sum h tu) I, ca (sum h tu^2) I on the diagonals of the U blocks
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      acc[(3 + i) * 9 - ((3 + i) * (2 + i)) / 2] += da0[0] * uacc[0];                    // (3+i, 3+i)
      acc[(3 + i) * 9 - ((3 + i) * (2 + i)) / 2 + 3] += da0[0] * uacc[1];                // (3+i, 6+i)
      acc[(6 + i) * 9 - ((6 + i) * (5 + i)) / 2] += da0[0] * uacc[2];                    // (6+i, 6+i)
    }
  }
#pragma unroll
  for (int e = 0; e < 45; ++e) {
    T v = acc[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    acc[e] = v;
  }
  if (lane == 0) {
    // P_0 = the final carry; cov = P_0 init_cov P_0^T + sum (init_cov is not symmetrised, as in the reference)
    T Pm[81], S0[9];
#pragma unroll
    for (int i = 0; i < 81; ++i) Pm[i] = T(0);
    quat_matrix<T>(carry, S0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        Pm[i * 9 + jj] = S0[i * 3 + jj];
        Pm[(3 + i) * 9 + jj] = carry[4 + i * 3 + jj];
        Pm[(6 + i) * 9 + jj] = carry[13 + i * 3 + jj];
      }
#pragma unroll
    for (int i = 0; i < 3; ++i) { Pm[(3 + i) * 9 + 3 + i] = T(1); Pm[(6 + i) * 9 + 3 + i] = carry[22]; Pm[(6 + i) * 9 + 6 + i] = T(1); }
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int c = r; c < 9; ++c) {
        const int e = r * 9 - (r * (r - 1)) / 2 + (c - r);
        cov[b * 81 + r * 9 + c] = acc[e];
        cov[b * 81 + c * 9 + r] = acc[e];
      }
    for (int r = 0; r < 9; ++r) {
      T tl[9];
      for (int l = 0; l < 9; ++l) {
        T s0 = T(0);
        for (int mm = 0; mm < 9; ++mm) s0 += Pm[r * 9 + mm] * init_cov[b * 81 + mm * 9 + l];
        tl[l] = s0;
      }
      for (int c = 0; c < 9; ++c) {
        T s1 = T(0);
        for (int l = 0; l < 9; ++l) s1 += tl[l] * Pm[c * 9 + l];
        cov[b * 81 + r * 9 + c] += s1;
      }
    }
  }
}

template <class T>
int imu_cov2_launch(const void* dt, const void* gyro, const void* acc, const void* rout, const void* rw, const void* C,
                    const void* init_cov, const void* gc, int64_t gc_sb, int64_t gc_sf, const void* ac, int64_t ac_sb, int64_t ac_sf,
                    const double* g, void* cov, int64_t B, int64_t F, void* stream) {
  if (B < 0 || F < 0) return SC_EBADARG;
  if (B == 0) return SC_OK;
  if (!dt || !gyro || !acc || !rout || !rw || !init_cov || !gc || !ac || !cov || !g) return SC_EBADARG;
  if (F == 0) {
    hipMemcpyAsync(cov, init_cov, (size_t)B * 81 * sizeof(T), hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
  }
  constexpr int WAVES = 2;
  const int64_t blocks = (B + WAVES - 1) / WAVES;
  // ac_sb == 0 && ac_sf == -1: the caller states that acc_cov is ONE isotropic variance (acc_cov[0] == [1] == [2]) shared by every
  // sequence and step -- the module's default noise model -- which turns the accelerometer part of every step's noise term into
  // three scalar sums (imu_cov_step ISO)
  const bool iso = ac_sb == 0 && ac_sf == -1;
  if (iso) ac_sf = 0;
#define PPLIE_COV2K(LSN, ISOV)                                                                                                  \
  hipLaunchKernelGGL((imu_cov_seg_kernel<T, WAVES, LSN, ISOV>), dim3((unsigned)blocks), dim3(WAVES * 64), 0,                    \
                     reinterpret_cast<hipStream_t>(stream), (const T*)dt, (const T*)gyro, (const T*)acc, (const T*)rout,        \
                     (const T*)rw, (const T*)C, (const T*)init_cov, (const T*)gc, gc_sb, gc_sf, (const T*)ac, ac_sb, ac_sf, (T)g[0], \
                     (T)g[1], (T)g[2], (T*)cov, B, F)
#define PPLIE_COV2(LSN) { if (iso) PPLIE_COV2K(LSN, true); else PPLIE_COV2K(LSN, false); }
  const char* env = getenv("PPLIE_IMU_COV_LS");                   // tuning switch: steps per lane
  const int ls = env ? atoi(env) : 4;
  if (F <= 128 || ls == 2) PPLIE_COV2(2)
  else if (ls == 8) PPLIE_COV2(8)
  else PPLIE_COV2(4)
#undef PPLIE_COV2
#undef PPLIE_COV2K
  return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
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
  const char* env = getenv("PPLIE_IMU_STEPS_PER_LANE");          // tuning switch: 1 = one step per lane
  const int kk = env ? atoi(env) : 2;
  // two steps per lane pay off when the kernel only writes the states (77 vs 94 us at 4096 x 1024); with the covariance's
  // aux streams the stores dominate and both variants take ~100 us: those keep one step per lane (K = 4: 98 us, slower)
  if (kk >= 2 && F >= 256 && F < (int64_t(1) << 29) && ark == nullptr) {
#define PPLIE_IMU_MULTI(KN)                                                                                                  \
  hipLaunchKernelGGL((imu_integrate_multi_kernel<T, WAVES, 2, KN>), dim3((unsigned)blocks), dim3(WAVES * 64), 0,             \
                     reinterpret_cast<hipStream_t>(stream), (const T*)dt, (const T*)gyro, (const T*)acc, (const T*)rot,       \
                     (const T*)r0, (const T*)v0, (const T*)p0, (T)g[0], (T)g[1], (T)g[2], (T*)orot, (T*)ovel, (T*)opos, B, (int)F)
    if (rot != nullptr) PPLIE_IMU_MULTI(true);
    else PPLIE_IMU_MULTI(false);
#undef PPLIE_IMU_MULTI
  }
  else
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
  if (F == 0) {                               // no steps: cov = init_cov
    hipMemcpyAsync(cov, init_cov, (size_t)B * 81 * sizeof(T), hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
  }
  constexpr int WAVES = 2;
  int64_t blocks = (B + WAVES - 1) / WAVES;
  hipLaunchKernelGGL((imu_cov_scan_kernel<T, WAVES>), dim3((unsigned)blocks), dim3(WAVES * 64), 0,
                     reinterpret_cast<hipStream_t>(stream), (const T*)dt, (const T*)rk, (const T*)rij, (const T*)a,
                     (const T*)init_cov, (const T*)gc, gc_sb, gc_sf, (const T*)ac, ac_sb, ac_sf, (T*)cov, B, F);
  return hipGetLastError() == hipSuccess ? SC_OK : SC_ELAUNCH;
}
}  // namespace pplie

// matrices [nseq, L, d, d] (d in {2, 3, 4, 6, 7, 9}); pplie_scan_mat9 = the reference's one use
extern "C" int pplie_scan_mat_f32(void* data, int64_t nseq, int64_t L, int d, int left, void* stream) {
  return pplie::scan_mat_launch<float>(data, nseq, L, d, left, stream);
}
extern "C" int pplie_scan_mat_f64(void* data, int64_t nseq, int64_t L, int d, int left, void* stream) {
  return pplie::scan_mat_launch<double>(data, nseq, L, d, left, stream);
}
extern "C" int pplie_scan_mat9_f32(void* data, int64_t nseq, int64_t L, int left, void* stream) {
  return pplie::scan_mat_launch<float>(data, nseq, L, 9, left, stream);
}
extern "C" int pplie_scan_mat9_f64(void* data, int64_t nseq, int64_t L, int left, void* stream) {
  return pplie::scan_mat_launch<double>(data, nseq, L, 9, left, stream);
}
#define PPLIE_SCAN_EXPORT(g, G)                                                                                     \
  extern "C" int pplie_scan_##g##_f32(void* data, int64_t nseq, int64_t L, int64_t inner, int left, void* stream) { \
    return pplie::scan_launch<float, pplie::G<float>>(data, nseq, L, inner, left, stream);                          \
  }                                                                                                                 \
  extern "C" int pplie_scan_##g##_f64(void* data, int64_t nseq, int64_t L, int64_t inner, int left, void* stream) { \
    return pplie::scan_launch<double, pplie::G<double>>(data, nseq, L, inner, left, stream);                        \
  }
#define PPLIE_SCAN_BWD_EXPORT(g, G)                                                                                 \
  extern "C" int pplie_scan_##g##_bwd_f32(const void* x, const void* y, const void* gy, void* gx, int64_t nseq, int64_t L,  \
                                          int64_t inner, int left, void* stream) {                                  \
    return pplie::scan_bwd_launch<float, pplie::G<float>>(x, y, gy, gx, nseq, L, inner, left, stream);              \
  }                                                                                                                 \
  extern "C" int pplie_scan_##g##_bwd_f64(const void* x, const void* y, const void* gy, void* gx, int64_t nseq, int64_t L,  \
                                          int64_t inner, int left, void* stream) {                                  \
    return pplie::scan_bwd_launch<double, pplie::G<double>>(x, y, gy, gx, nseq, L, inner, left, stream);            \
  }
PPLIE_SCAN_BWD_EXPORT(so3, MulSO3)
PPLIE_SCAN_BWD_EXPORT(se3, MulSE3)
PPLIE_SCAN_BWD_EXPORT(sim3, MulSim3)
PPLIE_SCAN_BWD_EXPORT(rxso3, MulRxSO3)
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
#define PPLIE_IMU_BWD_EXPORT(sfx, T)                                                                                    \
  extern "C" int pplie_imu_integrate_bwd_##sfx(const void* dt, const void* gyro, const void* acc, const void* rot,            \
                                               const void* rot_out, const void* vel_out, const void* init_rot,                \
                                               const double* gravity, const void* g_rot, const void* g_vel, const void* g_pos,  \
                                               void* g_gyro, void* g_acc, void* g_dt, int64_t B, int64_t F, void* stream) {   \
    return pplie::imu_integrate_bwd_launch<T>(dt, gyro, acc, rot, rot_out, vel_out, init_rot, gravity, g_rot, g_vel, g_pos,   \
                                              g_gyro, g_acc, g_dt, B, F, stream);                                           \
  }
PPLIE_IMU_BWD_EXPORT(f32, float)
PPLIE_IMU_BWD_EXPORT(f64, double)
extern "C" int pplie_imu_cov2_f32(const void* dt, const void* gyro, const void* acc, const void* rot_out, const void* rot_world,
                                  const void* C, const void* init_cov, const void* gyro_cov, int64_t gc_sb, int64_t gc_sf,
                                  const void* acc_cov, int64_t ac_sb, int64_t ac_sf, const double* gravity, void* cov, int64_t B,
                                  int64_t F, void* stream) {
  return pplie::imu_cov2_launch<float>(dt, gyro, acc, rot_out, rot_world, C, init_cov, gyro_cov, gc_sb, gc_sf, acc_cov, ac_sb, ac_sf,
                                       gravity, cov, B, F, stream);
}
extern "C" int pplie_imu_cov2_f64(const void* dt, const void* gyro, const void* acc, const void* rot_out, const void* rot_world,
                                  const void* C, const void* init_cov, const void* gyro_cov, int64_t gc_sb, int64_t gc_sf,
                                  const void* acc_cov, int64_t ac_sb, int64_t ac_sf, const double* gravity, void* cov, int64_t B,
                                  int64_t F, void* stream) {
  return pplie::imu_cov2_launch<double>(dt, gyro, acc, rot_out, rot_world, C, init_cov, gyro_cov, gc_sb, gc_sf, acc_cov, ac_sb, ac_sf,
                                        gravity, cov, B, F, stream);
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
