// pcg_persist.hip -- the whole block-Jacobi PCG solve of a pose-graph LM step in ONE launch.
//
// On graphs of ~10^4 nodes (BASELINE's metric: "LM iters/sec, PGO 10k poses") an iteration of the two-launch scheme
// (graph.hip, pplie_pcg2_*) moves ~16 MB -- a few microseconds of HBM time -- but costs ~20 us: two dependent kernel
// launches inside a hipGraph, plus a host read-back every `check_every` iterations to test convergence.  Here a few
// dozen resident workgroups keep the iteration on the device: each owns a contiguous range of node rows, the two
// reductions an iteration needs are exchanged through one row of partial sums per workgroup and a grid barrier
// (every workgroup adds the rows up in the same order, so all of them hold the same alpha / beta / |r|^2 bits and take
// the same exit), and the loop ends in the iteration that meets the tolerance, as the reference's CG does
// (pypose/optim/solver.py:276-340 tests |r| <= tol |b| every iteration).
//
// Same arithmetic as pplie_pcg2_spmv / pplie_pcg2_step (the comment there derives beta from node-local products):
//   phase A   q = (D + HB) p over own rows;   partial { p.q, q.z, q.Binv q }           -- barrier 1
//             alpha = rho / p.q;  rho_rec = rho - 2 alpha q.z + alpha^2 q.Binv q;  beta = rho_rec / rho
//   phase B   x += alpha p;  r -= alpha q;  z = Binv r;  p = z + beta p over own rows;  partial { r.z, r.r }  -- barrier 2
//             rho = r.z;  stop when r.r <= tol^2 |b|^2
// (the two "barriers" are the tagged all-gathers of the partial sums themselves, see gather_tagged)
// Only p (the SpMV's gather) and the partial-sum rows cross workgroups: they go through agent-scope stores / loads and
// the barrier does no cache maintenance, so the blocks, the preconditioner and the owner-private vectors x, r, z, q stay
// in the owner's L1 / L2 from one iteration to the next.
#include "rowmap.h"
#include "gridsync.h"

namespace pplie {

constexpr int kPersistGridMax = 256;      // = PPLIE_PCG_PERSIST_GRID: rows of the partial-sum table
constexpr int kPersistBlock = 1024;       // 16 waves per workgroup: with ~64 workgroups every node of a 10^4-node graph has its
                                          // own lanes, so a phase is ONE pass of independent gathers instead of a serial chain

// sum over the workgroup, result in thread 0; every thread must call it
template <class T, int BLOCK> __device__ __forceinline__ T wg_total(T v) {
  __shared__ T part[BLOCK / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  T s = T(0);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) s += part[w];
  }
  __syncthreads();
  return s;
}

// Q workgroup totals with one pair of barriers (valid in thread 0)
template <class T, int Q, int BLOCK> __device__ __forceinline__ void wg_totals(T* v) {
  __shared__ T part[Q][BLOCK / 64];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_down(v[q], off, 64);
    if ((threadIdx.x & 63) == 0) part[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      T s = T(0);
#pragma unroll
      for (int w = 0; w < BLOCK / 64; ++w) s += part[q][w];
      v[q] = s;
    }
  }
  __syncthreads();
}

template <class T, int M>
__global__ void __launch_bounds__(kPersistBlock)
pcg_persist_kernel(const int* __restrict__ ptr, const int* __restrict__ other, const T* __restrict__ HB, const T* __restrict__ D,
                   const T* __restrict__ Binv, T* __restrict__ x, T* __restrict__ r, T* p, T* __restrict__ q, T* __restrict__ z,
                   u64* part /* [2][kPersistGridMax][4 values as tagged words] */, T* __restrict__ rr_hist, T* info /* [4] */, int* it_out,
                   T tol2, int maxiter, int cap, int64_t N) {
  constexpr int NPW = 64 / M;              // nodes per wave pass: M lanes per node
  constexpr int WV = kPersistBlock / 64;   // waves per workgroup
  __shared__ T sh[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane / M, i = lane % M;
  const bool lane_on = sub < NPW;
  const int64_t n0 = N * blockIdx.x / gridDim.x, n1 = N * (blockIdx.x + 1) / gridDim.x;
  constexpr int RW = 4 * (int)(sizeof(T) / 4);
  u64* partA = part + (size_t)blockIdx.x * RW;
  u64* partB = part + (size_t)(kPersistGridMax + blockIdx.x) * RW;
  const u64* tabA = part;
  const u64* tabB = part + (size_t)kPersistGridMax * RW;
  unsigned seq = 1;                        // the table was zeroed by the caller: tag 0 is never expected

  // ---- prologue: rho = r.z and |b|^2 = r.r of the initial residual (pplie_pcg_prepare left r = -g, z = Binv r, p = z)
  {
    T a0 = T(0), a1 = T(0);
    for (int64_t n = n0 + w * NPW + sub; n < n1; n += WV * NPW) {
      if (lane_on) {
        const T rv = r[n * M + i];
        a0 += rv * z[n * M + i];
        a1 += rv * rv;
      }
    }
    T v[2] = {a0, a1};
    wg_totals<T, 2, kPersistBlock>(v);
    if (threadIdx.x == 0) { put_tagged(partB, 0, v[0], seq); put_tagged(partB, 1, v[1], seq); }
  }
  T tot[4];
  int it = 0, flag = 0;                   // flag: 1 converged, 2 NaN, 3 a workgroup never arrived, 0 iteration limit
  if (!gather_tagged<T, 2>(tabB, gridDim.x, seq, tot, sh)) flag = 3;
  ++seq;
  T rho = tot[0];
  const T bn2 = tot[1];
  T rr = bn2;
  if (flag == 0 && bn2 == T(0)) flag = 1;
  while (flag == 0 && it < maxiter) {
    // ---- phase A: q = A p on own rows
    T a_pq = T(0), a_qz = T(0), a_qmq = T(0);
    for (int64_t nb = n0 + w * NPW; nb < n1; nb += WV * NPW) {
      const int64_t n = nb + sub;
      const bool act = lane_on && n < n1;
      T acc = T(0), pi = T(0);
      if (act) {
        T pv[M];
#pragma unroll
        for (int j = 0; j < M; ++j) pv[j] = xwg_load(p + n * M + j);
        pi = pv[i];
#pragma unroll
        for (int j = 0; j < M; ++j) acc += D[(n * M + i) * M + j] * pv[j];
        const int beg = ptr[n], end = ptr[n + 1];
        // four incidences at a time, every load of a chunk issued before the first use: the gather is a chain of two
        // dependent memory round trips (neighbour index, then its p row through the memory side), and a loop that
        // walked the incidences one pair per trip paid that chain four times per node (eight at a time spills at the 128 VGPRs a 1024-lane workgroup may use)
        for (int c = beg; c < end; c += 4) {
          int64_t o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = other[c + k < end ? c + k : beg];
          T hv[4][M], pw[4][M];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const T* h = HB + ((int64_t)(c + k < end ? c + k : beg) * M + i) * M;
            const T* pk = p + o[k] * M;
#pragma unroll
            for (int j = 0; j < M; ++j) { hv[k][j] = h[j]; pw[k][j] = xwg_load(pk + j); }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            T sk = T(0);
#pragma unroll
            for (int j = 0; j < M; ++j) sk += hv[k][j] * pw[k][j];
            acc += c + k < end ? sk : T(0);
          }
        }
        q[n * M + i] = acc;
      }
      T bq = T(0);
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const T qj = __shfl(acc, sub * M + j, 64);
        if (act) bq += Binv[(n * M + i) * M + j] * qj;
      }
      if (act) {
        a_pq += acc * pi;
        a_qz += acc * z[n * M + i];
        a_qmq += acc * bq;
      }
    }
    {
      T v[3] = {a_pq, a_qz, a_qmq};
      wg_totals<T, 3, kPersistBlock>(v);       // (its barrier also drains this workgroup's q stores)
      if (threadIdx.x == 0) { put_tagged(partA, 0, v[0], seq); put_tagged(partA, 1, v[1], seq); put_tagged(partA, 2, v[2], seq); }
    }
    if (!gather_tagged<T, 3>(tabA, gridDim.x, seq, tot, sh)) { flag = 3; break; }
    const T pq = tot[0], qz = tot[1], qmq = tot[2];
    const T alpha = pq != T(0) ? rho / pq : T(0);                 // p.q = 0 only once r = 0: stay put, no NaN
    T rho_rec = rho - T(2) * alpha * qz + alpha * alpha * qmq;
    if (rho_rec < T(0)) rho_rec = T(0);
    const T beta = rho != T(0) ? rho_rec / rho : T(0);
    // ---- phase B: vector updates on own rows
    T a_rho = T(0), a_rr = T(0);
    for (int64_t nb = n0 + w * NPW; nb < n1; nb += WV * NPW) {
      const int64_t n = nb + sub;
      const bool act = lane_on && n < n1;
      T re = T(0), pe = T(0);
      if (act) {
        re = r[n * M + i] - alpha * q[n * M + i];
        pe = xwg_load(p + n * M + i);
      }
      T ze = T(0);
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const T rj = __shfl(re, sub * M + j, 64);
        if (act) ze += Binv[(n * M + i) * M + j] * rj;
      }
      if (act) {
        x[n * M + i] += alpha * pe;
        r[n * M + i] = re;
        z[n * M + i] = ze;
        xwg_store(p + n * M + i, ze + beta * pe);
        a_rho += re * ze;
        a_rr += re * re;
      }
    }
    {
      T v[2] = {a_rho, a_rr};
      wg_totals<T, 2, kPersistBlock>(v);       // its __syncthreads drains every wave's p stores (vmcnt(0)) BEFORE the tag goes out:
      if (threadIdx.x == 0) { put_tagged(partB, 0, v[0], seq); put_tagged(partB, 1, v[1], seq); }   // whoever sees the tag sees p
    }
    if (!gather_tagged<T, 2>(tabB, gridDim.x, seq, tot, sh)) { flag = 3; break; }
    ++seq;
    rho = tot[0];
    rr = tot[1];
    if (blockIdx.x == 0 && threadIdx.x == 0 && it < cap) rr_hist[it] = rr;
    ++it;
    if (!(rr == rr)) flag = 2;
    else if (rr <= tol2 * bn2) flag = 1;
  }
  if (flag >= 2) {
    // a failed solve (NaN, or a workgroup that never arrived) hands back x = 0: the caller may have queued the parameter
    // update behind this launch and look at `info` only afterwards (one read-back per LM trial) -- Exp(0) p = p
    for (int64_t n = n0 + w * NPW + sub; n < n1; n += WV * NPW)
      if (lane_on) x[n * M + i] = T(0);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    info[0] = (T)it; info[1] = rr; info[2] = bn2; info[3] = (T)flag;
    it_out[0] = it;
  }
}

template <class T>
int pcg_persist(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x, void* r, void* p,
                void* q, void* z, void* part, void* bar, void* rr_hist, void* info, void* it, double tol, int maxiter, int cap,
                int grid, int64_t N, int m, void* stream) {
  if (N <= 0) return N == 0 ? PPLIE_OK : PPLIE_EBADARG;
  if (!ptr || !other || !HB || !D || !Binv || !x || !r || !p || !q || !z || !part || !bar || !rr_hist || !info || !it) return PPLIE_EBADARG;
  if (grid < 1 || grid > kPersistGridMax || maxiter < 0) return PPLIE_EBADARG;
  if (grid > N) grid = (int)N;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define LAUNCH(MM)                                                                                                             \
  hipLaunchKernelGGL((pcg_persist_kernel<T, MM>), dim3(grid), dim3(kPersistBlock), 0, st, (const int*)ptr, (const int*)other, (const T*)HB, \
                     (const T*)D, (const T*)Binv, (T*)x, (T*)r, (T*)p, (T*)q, (T*)z, (unsigned long long*)part, (T*)rr_hist,     \
                     (T*)info, (int*)it, (T)(tol * tol), maxiter, cap, N);
  if (m == 6) { LAUNCH(6) } else if (m == 7) { LAUNCH(7) } else if (m == 3) { LAUNCH(3) } else return PPLIE_EBADARG;
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace pplie

extern "C" int pplie_pcg_persist_f32(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x,
                                     void* r, void* p, void* q, void* z, void* part, void* bar, void* rr_hist, void* info, void* it,
                                     double tol, int maxiter, int cap, int grid, int64_t N, int m, void* stream) {
  return pplie::pcg_persist<float>(ptr, other, HB, D, Binv, x, r, p, q, z, part, bar, rr_hist, info, it, tol, maxiter, cap, grid, N, m, stream);
}
extern "C" int pplie_pcg_persist_f64(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x,
                                     void* r, void* p, void* q, void* z, void* part, void* bar, void* rr_hist, void* info, void* it,
                                     double tol, int maxiter, int cap, int grid, int64_t N, int m, void* stream) {
  return pplie::pcg_persist<double>(ptr, other, HB, D, Binv, x, r, p, q, z, part, bar, rr_hist, info, it, tol, maxiter, cap, grid, N, m, stream);
}
