// csr_pcg.hip -- Jacobi-preconditioned conjugate gradient on a scalar CSR matrix, two launches per iteration.
//
// What it is for: pypose/optim/optimizer.py:630-643, 663-668 -- the reference's `LM(sparse=True)` branch builds A = J^T J as a
// torch.sparse_csr matrix and hands it to `bae.utils.pysolvers.PCG` (solver.py:358-364), a third-party solver this tree replaces
// (pypose_amd/compat/bae/utils/pysolvers.py).  That stand-in ran the textbook loop in torch: a hipSPARSE SpMV and ~10 small
// element-wise / reduction launches per iteration.  Here an iteration is
//   K1 (pplie_csr_pcg_spmv):  q = A p;  pq += p.q;  qz += q.z;  qmq += q.(Minv q)          (16 lanes per row: CSR-vector)
//   K2 (pplie_csr_pcg_step):  alpha = rho / pq;  x += alpha p;  r -= alpha q;  z = Minv r;  rho' += r.z;  rr += r.r;
//                             beta = (rho - 2 alpha qz + alpha^2 qmq) / rho;  p = z + beta p;  ++it
// (the same recurrence trick as csrc/graph.hip's pcg2: with a diagonal preconditioner rho' is known from node-local products, so
// the direction update does not wait for a third reduction; the exact r.z replaces it in the next iteration's alpha).
// The stop test |r|^2 <= tol^2 |b|^2 runs on the device: the K1 that finds it raises it[2] and every later launch returns at once,
// so the host queues chunks of iterations and reads four ints per chunk; it[0] = iterations done when it stopped -- the count of
// a loop that tests after every update, as the stand-in's (and the reference CG's, solver.py:319) does.
// Scalars: two alternating sets of slot-spread accumulators (one float atomic per workgroup per quantity, 32 addresses a line
// apart), as csrc/graph.hip.   scal: T[2][8][32][32]: rho, pq, rr, bn2, qz, qmq.   Index type: int64 (torch's) or int32.
#include "rowmap.h"

namespace pplie {
namespace csr {
constexpr int kSlots = 32, kStride = 32;
enum { Q_RHO = 0, Q_PQ = 1, Q_RR = 2, Q_BN2 = 3, Q_QZ = 4, Q_QMQ = 5, Q_COUNT = 8 };
template <class T> __device__ __forceinline__ T* sq(T* scal, int set, int q) { return scal + (size_t)((set * Q_COUNT + q) * kSlots) * kStride; }
template <class T> __device__ __forceinline__ void slot_add(T* base, T v) { atomicAdd(base + (blockIdx.x & (kSlots - 1)) * kStride, v); }
template <class T> __device__ __forceinline__ T slot_total(const T* base) {
  T s = T(0);
#pragma unroll
  for (int k = 0; k < kSlots; ++k) s += base[k * kStride];
  return s;
}
// totals of NQ quantities, fetched once per workgroup (wave 0) and handed to everybody through LDS
template <class T, int NQ> __device__ __forceinline__ void totals_wg(const T* const (&base)[NQ], T (&out)[NQ]) {
  __shared__ T tot[NQ];
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      T v = lane < kSlots ? base[k][lane * kStride] : T(0);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) tot[k] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NQ; ++k) out[k] = tot[k];
}

// sum over the 16 lanes of a row group (result in every lane of the group)
template <class T> __device__ __forceinline__ T group16_sum(T v) {
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// x = 0, Minv = 1 / diag(A) (1 where the diagonal is absent or zero), r = b, z = Minv r, p = z, rho += r.z, bn2 += b.b
template <class T, class I>
__global__ void __launch_bounds__(256)
prepare_kernel(const I* __restrict__ crow, const I* __restrict__ col, const T* __restrict__ val, const T* __restrict__ b,
               T* __restrict__ minv, T* __restrict__ x, T* __restrict__ r, T* __restrict__ z, T* __restrict__ p, T* scal, int64_t n) {
  const int lane16 = threadIdx.x & 15;
  T a_rho = T(0), a_bn = T(0);
  for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; row < n; row += ((int64_t)gridDim.x * 256) >> 4) {
    const I beg = crow[row], end = crow[row + 1];
    T d = T(0);
    for (I c = beg + lane16; c < end; c += 16)
      if (col[c] == (I)row) d += val[c];
    d = group16_sum(d);
    if (lane16 == 0) {
      const T mi = d != T(0) ? T(1) / d : T(1);
      const T bi = b[row], zi = mi * bi;
      minv[row] = mi; x[row] = T(0); r[row] = bi; z[row] = zi; p[row] = zi;
      a_rho += bi * zi;
      a_bn += bi * bi;
    }
  }
  const T s1 = block_sum(a_rho), s2 = block_sum(a_bn);
  if (threadIdx.x == 0) { slot_add(sq(scal, 0, Q_RHO), s1); slot_add(sq(scal, 0, Q_BN2), s2); }
}

template <class T, class I>
__global__ void __launch_bounds__(256)
spmv_kernel(const I* __restrict__ crow, const I* __restrict__ col, const T* __restrict__ val, const T* __restrict__ p,
            const T* __restrict__ z, const T* __restrict__ minv, T* __restrict__ q, T* scal, int* it, int64_t n, T tol2) {
  if (it[2] != 0) return;
  const int done = it[0], a = done & 1;
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      if (done > 0) {                                   // the residual of the iteration just finished: the stop test
        const T rr = slot_total(sq(scal, a ^ 1, Q_RR)), bn2 = slot_total(sq(scal, 0, Q_BN2));
        if (!(rr == rr)) it[2] = 2;
        else if (rr <= tol2 * bn2) it[2] = 1;           // (this launch's q is not applied: the step kernel sees the flag)
      }
      it[1] = done;
    }
    __syncthreads();
    if (threadIdx.x < 5 * kSlots) {                     // clear rho, pq, rr, qz, qmq of the idle set (bn2 stays)
      const int qi = threadIdx.x / kSlots, quant = qi < 3 ? qi : qi + 1;
      sq(scal, a ^ 1, quant)[(threadIdx.x % kSlots) * kStride] = T(0);
    }
  }
  const int lane16 = threadIdx.x & 15;
  T a_pq = T(0), a_qz = T(0), a_qmq = T(0);
  for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; row < n; row += ((int64_t)gridDim.x * 256) >> 4) {
    const I beg = crow[row], end = crow[row + 1];
    T acc = T(0);
    for (I c = beg + lane16; c < end; c += 16) acc += val[c] * p[col[c]];
    acc = group16_sum(acc);
    if (lane16 == 0) {
      q[row] = acc;
      a_pq += acc * p[row];
      a_qz += acc * z[row];
      a_qmq += acc * acc * minv[row];
    }
  }
  const T s1 = block_sum(a_pq), s2 = block_sum(a_qz), s3 = block_sum(a_qmq);
  if (threadIdx.x == 0) { slot_add(sq(scal, a, Q_PQ), s1); slot_add(sq(scal, a, Q_QZ), s2); slot_add(sq(scal, a, Q_QMQ), s3); }
}

template <class T>
__global__ void __launch_bounds__(256)
step_kernel(T* __restrict__ x, T* __restrict__ r, T* __restrict__ p, const T* __restrict__ q, T* __restrict__ z,
            const T* __restrict__ minv, T* scal, int* it, int64_t n) {
  if (it[2] != 0) return;
  const int done = it[1], a = done & 1;
  const T* const bases[4] = {sq(scal, a, Q_RHO), sq(scal, a, Q_PQ), sq(scal, a, Q_QZ), sq(scal, a, Q_QMQ)};
  T tv[4];
  totals_wg<T, 4>(bases, tv);
  constexpr T tiny = sizeof(T) == 4 ? T(1e-30) : T(1e-290);
  const T rho = tv[0], pq = tv[1], qz = tv[2], qmq = tv[3];
  const T alpha = pq > tiny ? rho / pq : T(0);
  T rho_rec = rho - T(2) * alpha * qz + alpha * alpha * qmq;
  if (rho_rec < T(0)) rho_rec = T(0);
  const T beta = rho > tiny ? rho_rec / rho : T(0);
  T a1 = T(0), a2 = T(0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const T pe = p[i];
    const T re = r[i] - alpha * q[i];
    const T ze = minv[i] * re;
    x[i] += alpha * pe;
    r[i] = re;
    z[i] = ze;
    p[i] = ze + beta * pe;
    a1 += re * ze;
    a2 += re * re;
  }
  const T s1 = block_sum(a1), s2 = block_sum(a2);
  if (threadIdx.x == 0) {
    slot_add(sq(scal, a ^ 1, Q_RHO), s1);
    slot_add(sq(scal, a, Q_RR), s2);
    if (blockIdx.x == 0) it[0] = done + 1;
  }
}

// the diagonal of a CSR matrix in place: v <- clamp(v, lo, hi) * scale   (diagonal_op_ of the reference's sparse branch:
// optimizer.py:643 clamps to [min, max], :664 multiplies by 1 + damping)
template <class T, class I>
__global__ void __launch_bounds__(256)
diag_op_kernel(const I* __restrict__ crow, const I* __restrict__ col, T* __restrict__ val, int64_t n, T lo, T hi, T scale) {
  const int lane16 = threadIdx.x & 15;
  for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; row < n; row += ((int64_t)gridDim.x * 256) >> 4) {
    const I beg = crow[row], end = crow[row + 1];
    for (I c = beg + lane16; c < end; c += 16)
      if (col[c] == (I)row) { const T v = val[c]; val[c] = (v < lo ? lo : (v > hi ? hi : v)) * scale; }
  }
}

inline int row_grid(int64_t n) {
  const int64_t nb = (n * 16 + 255) / 256;
  return (int)(nb < 1 ? 1 : (nb < 8192 ? nb : 8192));
}
template <class T, class I>
int prepare(const void* crow, const void* col, const void* val, const void* b, void* minv, void* x, void* r, void* z, void* p, void* scal,
            int64_t n, void* stream) {
  hipLaunchKernelGGL((prepare_kernel<T, I>), dim3(row_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const I*)crow,
                     (const I*)col, (const T*)val, (const T*)b, (T*)minv, (T*)x, (T*)r, (T*)z, (T*)p, (T*)scal, n);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
template <class T, class I>
int spmv(const void* crow, const void* col, const void* val, const void* p, const void* z, const void* minv, void* q, void* scal, void* it,
         int64_t n, double tol2, void* stream) {
  hipLaunchKernelGGL((spmv_kernel<T, I>), dim3(row_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const I*)crow,
                     (const I*)col, (const T*)val, (const T*)p, (const T*)z, (const T*)minv, (T*)q, (T*)scal, (int*)it, n, (T)tol2);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
template <class T>
int step(void* x, void* r, void* p, const void* q, void* z, const void* minv, void* scal, void* it, int64_t n, void* stream) {
  const int64_t nb = (n + 255) / 256;
  hipLaunchKernelGGL((step_kernel<T>), dim3((int)(nb < 2048 ? nb : 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (T*)x, (T*)r,
                     (T*)p, (const T*)q, (T*)z, (const T*)minv, (T*)scal, (int*)it, n);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
template <class T, class I>
int diag_op(const void* crow, const void* col, void* val, int64_t n, double lo, double hi, double scale, void* stream) {
  hipLaunchKernelGGL((diag_op_kernel<T, I>), dim3(row_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const I*)crow,
                     (const I*)col, (T*)val, n, (T)lo, (T)hi, (T)scale);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace csr
}  // namespace pplie

// idx64 != 0: crow / col are int64 (torch's sparse_csr), else int32
#define PPLIE_CSR(SFX, T)                                                                                                          \
  extern "C" int pplie_csr_pcg_prepare_##SFX(const void* crow, const void* col, const void* val, const void* b, void* minv, void* x, \
                                             void* r, void* z, void* p, void* scal, int64_t n, int idx64, void* stream) {          \
    if (n <= 0) return n == 0 ? pplie::PPLIE_OK : pplie::PPLIE_EBADARG;                                                            \
    if (!crow || !col || !val || !b || !minv || !x || !r || !z || !p || !scal) return pplie::PPLIE_EBADARG;                          \
    return idx64 ? pplie::csr::prepare<T, int64_t>(crow, col, val, b, minv, x, r, z, p, scal, n, stream)                           \
                 : pplie::csr::prepare<T, int>(crow, col, val, b, minv, x, r, z, p, scal, n, stream);                              \
  }                                                                                                                                \
  extern "C" int pplie_csr_pcg_spmv_##SFX(const void* crow, const void* col, const void* val, const void* p, const void* z,        \
                                          const void* minv, void* q, void* scal, void* it, int64_t n, double tol2, int idx64,      \
                                          void* stream) {                                                                          \
    if (n <= 0) return n == 0 ? pplie::PPLIE_OK : pplie::PPLIE_EBADARG;                                                            \
    if (!crow || !col || !val || !p || !z || !minv || !q || !scal || !it) return pplie::PPLIE_EBADARG;                              \
    return idx64 ? pplie::csr::spmv<T, int64_t>(crow, col, val, p, z, minv, q, scal, it, n, tol2, stream)                          \
                 : pplie::csr::spmv<T, int>(crow, col, val, p, z, minv, q, scal, it, n, tol2, stream);                             \
  }                                                                                                                                \
  extern "C" int pplie_csr_pcg_step_##SFX(void* x, void* r, void* p, const void* q, void* z, const void* minv, void* scal, void* it, \
                                          int64_t n, void* stream) {                                                               \
    if (n <= 0) return n == 0 ? pplie::PPLIE_OK : pplie::PPLIE_EBADARG;                                                            \
    if (!x || !r || !p || !q || !z || !minv || !scal || !it) return pplie::PPLIE_EBADARG;                                           \
    return pplie::csr::step<T>(x, r, p, q, z, minv, scal, it, n, stream);                                                          \
  }                                                                                                                                \
  extern "C" int pplie_csr_diag_op_##SFX(const void* crow, const void* col, void* val, int64_t n, double lo, double hi,           \
                                         double scale, int idx64, void* stream) {                                                  \
    if (n <= 0) return n == 0 ? pplie::PPLIE_OK : pplie::PPLIE_EBADARG;                                                            \
    if (!crow || !col || !val) return pplie::PPLIE_EBADARG;                                                                        \
    return idx64 ? pplie::csr::diag_op<T, int64_t>(crow, col, val, n, lo, hi, scale, stream)                                       \
                 : pplie::csr::diag_op<T, int>(crow, col, val, n, lo, hi, scale, stream);                                          \
  }
PPLIE_CSR(f32, float)
PPLIE_CSR(f64, double)
