// pcg_persist.hip -- the whole block-Jacobi PCG solve of a pose-graph LM step in ONE launch, ONE exchange per iteration.
//
// On graphs of ~10^4 nodes (BASELINE's metric: "LM iters/sec, PGO 10k poses") an iteration moves ~16 MB of cache-resident
// data -- a few microseconds -- so what an iteration costs is the latency of whatever crosses workgroups.  Round 2 paid two
// grid-wide exchanges per iteration (partial sums after the SpMV, partial sums + the hand-off of p after the vector update:
// ~15 us per iteration).  This version pays ONE:
//
//   * the search direction p is handed over as DATA-FLOW, not behind a barrier: every element of p is a 64-bit word
//     { iteration tag | value bits } in a double-buffered table, written with one agent-scope store and read with one
//     agent-scope load -- tag and payload cannot be seen apart, so a reader that finds the tag it expects has the value,
//     whatever the memory model says about ordering between different addresses.  The SpMV simply re-reads a neighbour's
//     row until its tag is this iteration's; neighbours finish their updates within a microsecond of each other.
//   * r.z and r.r of the CURRENT residual are accumulated locally during the previous vector update and ride in the same
//     exchange as the SpMV's products { p.q, q.z, q.Binv q }, so every scalar of the iteration comes out of one all-gather:
//         rho_k = r_k.z_k (exact),  |r_k|^2 (exact; the stop test, BEFORE the update, where the reference's CG tests it:
//         pypose/optim/solver.py:319),  alpha = rho_k / p.q,  rho_{k+1} = rho_k - 2 alpha q.z + alpha^2 q.Binv q,  beta.
//     (rho_{k+1} from node-local products is only used for beta; the next exchange replaces it by the exact r.z.)
//   * each lane owns ONE (node, component) for the whole solve: x, r, z, p, q and its rows of D and Binv live in registers;
//     per iteration a lane reads its rows of the off-diagonal blocks (L1/L2-resident) and its neighbours' p rows.
//
// The exchange itself is the tagged all-gather of gridsync.h (one row of partial sums per workgroup, two tables alternating
// with the iteration's parity, polled by one wave per quantity); every workgroup adds the rows up in the same order, so all
// of them hold the same alpha / beta / |r|^2 bits and take the same exit.
#include "rowmap.h"
#include "gridsync.h"

namespace pplie {

constexpr int kPersistGridMax = 256;      // = PPLIE_PCG_PERSIST_GRID: rows of the partial-sum tables
constexpr int kPersistBlock = 1024;       // 16 waves per workgroup
constexpr int kPersistQ = 5;              // quantities per exchange: p.q, q.z, q.Binv q, r.z, r.r
constexpr int kPersistSlots = 8;          // table row = 8 quantity slots (PPLIE_PCG_PERSIST_SLOTS)

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

// one value as NW tagged words
template <class T> __device__ __forceinline__ void put_value(u64* dst, T v, unsigned tag) {
  constexpr int NW = sizeof(T) / 4;
  unsigned w[NW];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (int k = 0; k < NW; ++k) xwg_store(dst + k, ((u64)tag << 32) | (u64)w[k]);
}
template <class T> __device__ __forceinline__ T get_value(const u64* src, unsigned tag, bool& ok) {
  constexpr int NW = sizeof(T) / 4;
  unsigned w[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const u64 v = xwg_load(src + k);
    ok = ok && (unsigned)(v >> 32) == tag;
    w[k] = (unsigned)v;
  }
  T out;
  __builtin_memcpy(&out, w, sizeof(T));
  return out;
}

// Sums of Q quantities over `rows` table rows; wave q polls quantity q (4 rows per lane at 256 rows, all loads of a round in
// flight together), the rows are added in a fixed order.  Every thread must call it; false on a timeout.
template <class T, int Q> __device__ __forceinline__ bool gather_rows(const u64* tab, int rows, unsigned tag, T out[Q]) {
  constexpr int NW = sizeof(T) / 4, RW = kPersistSlots * NW;
  __shared__ T tot_sh[Q];
  __shared__ int bad_sh;
  if (threadIdx.x == 0) bad_sh = 0;
  __syncthreads();
  const int wq = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wq < Q) {
    T a = T(0);
    bool all = true;
    for (int base = 0; base < rows; base += 256) {
      T v[4];
      bool done[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { done[k] = base + lane + 64 * k >= rows; v[k] = T(0); }
      for (long spin = 0; spin < (1L << 20); ++spin) {
        bool pending = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!done[k]) {
            bool ok = true;
            const T t = get_value<T>(tab + (size_t)(base + lane + 64 * k) * RW + wq * NW, tag, ok);
            if (ok) { v[k] = t; done[k] = true; } else pending = true;
          }
        }
        if (!pending) break;
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { all = all && done[k]; a += v[k]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) tot_sh[wq] = a;
    if (!__all(all) && lane == 0) bad_sh = 1;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < Q; ++q) out[q] = tot_sh[q];
  const bool ok = bad_sh == 0;
  __syncthreads();
  return ok;
}


// q_i += sum over this lane's incidences of (row i of the off-diagonal block) . (the neighbour's p row).  A lane fetches ONE
// element of each neighbour row -- its own component -- as a tagged word and the node's M lanes trade them by shuffle.  CH
// incidences per round: every tagged load of a round is in flight before the first is looked at, so a node of degree <= CH
// costs ONE memory round trip per iteration (the round-2 kernel walked the list four at a time, each round a chain of two
// dependent trips: neighbour index, then its row -- that chain, not the grid exchange, was most of its 15 us).  The loop is
// wave-uniform (runs to the largest degree in the wave, absent incidences masked) so the shuffles sit in uniform control
// flow.  `hb` / `nb`: this lane's first block / neighbour index -- in LDS when the workgroup's slice was staged there
// (LDS = true), else in global memory.
template <class T, int M, int CH, class HP, class NP>
__device__ __forceinline__ T spmv_rows(HP hb, NP nb, int deg, int maxdeg, int sub, int i, const u64* pin, unsigned own, unsigned tag,
                                       bool& stale) {
  constexpr int NW = sizeof(T) / 4;
  T acc = T(0);
  for (int c0 = 0; c0 < maxdeg; c0 += CH) {
    unsigned po[CH];                       // (word index in the table: < 2^32 for every graph the persistent solve takes)
    T pv[CH];
    // an absent incidence reads this lane's OWN element of p (always current: no extra wait) against a zeroed block row:
    // every load is unconditional, no divergent branches around them
#pragma unroll
    for (int q = 0; q < CH; ++q) po[q] = c0 + q < deg ? (unsigned)((nb[c0 + q] * M + i) * NW) : own;
    for (long spin = 0;; ++spin) {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < CH; ++q) pv[q] = get_value<T>(pin + po[q], tag, ok);
      if (__all(ok)) break;
      if (spin >= (1L << 20)) { stale = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const bool valid = c0 + q < deg;
      const T mask = valid ? T(1) : T(0);
      const int c = valid ? c0 + q : 0;
#pragma unroll
      for (int j = 0; j < M; ++j) acc += mask * hb[(size_t)c * M * M + j] * __shfl(pv[q], sub * M + j, 64);
    }
  }
  return acc;
}

template <class T, int M>
__global__ void __launch_bounds__(kPersistBlock)
pcg_persist_kernel(const int* __restrict__ ptr, const int* __restrict__ other, const T* __restrict__ HB, const T* __restrict__ D,
                   const T* __restrict__ Binv, T* __restrict__ x, const T* __restrict__ r, const T* __restrict__ z,
                   u64* part /* [2][kPersistGridMax][kPersistSlots values as tagged words] */,
                   u64* ptag /* [2][N * M values as tagged words] */, T* __restrict__ rr_hist, T* info /* [4] */, int* it_out,
                   T tol2, int maxiter, int cap, int64_t N, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  constexpr int CH = sizeof(T) == 4 ? 16 : 8;   // incidences per round of tagged loads
  constexpr int NPW = 64 / M;              // nodes per wave: M lanes per node
  constexpr int WV = kPersistBlock / 64;   // waves per workgroup
  constexpr int NW = sizeof(T) / 4, RW = kPersistSlots * NW;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane / M, i = lane % M;
  const int64_t n0 = N * blockIdx.x / gridDim.x, n1 = N * (blockIdx.x + 1) / gridDim.x;     // (host: n1 - n0 <= WV * NPW)
  const int64_t n = n0 + w * NPW + sub;
  const bool act = sub < NPW && n < n1;
  const size_t NM = (size_t)N * M * NW;

  // ---- this lane's (node, component) for the whole solve
  T dr[M], br[M];                          // rows i of the damped diagonal block and of its inverse
  T xe = T(0), re = T(0), ze = T(0), pe = T(0);
  int beg = 0, deg = 0;
  if (act) {
#pragma unroll
    for (int j = 0; j < M; ++j) { dr[j] = D[(n * M + i) * M + j]; br[j] = Binv[(n * M + i) * M + j]; }
    re = r[n * M + i];                     // pplie_pcg_prepare left r = -g, z = Binv r (= p_0), x = 0
    ze = z[n * M + i];
    pe = ze;
    beg = ptr[n];
    deg = ptr[n + 1] - beg;
    put_value<T>(ptag + (size_t)(n * M + i) * NW, pe, 1u);           // p_k carries tag k + 1, in table k & 1
  } else {
#pragma unroll
    for (int j = 0; j < M; ++j) { dr[j] = T(0); br[j] = T(0); }
  }
  int maxdeg = deg;                        // largest degree among this wave's nodes
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(maxdeg, off, 64);
    maxdeg = o > maxdeg ? o : maxdeg;
  }
  // ---- this workgroup's slice of the matrix -- the off-diagonal blocks of its nodes' incidences (contiguous: incidence
  // order is node order) and their neighbour indices -- is staged into LDS once and read from there in every iteration;
  // a slice that does not fit (dense neighbourhoods, few workgroups) stays in global memory (L2)
  const int c_lo = ptr[n0], c_cnt = ptr[n1] - c_lo;
  const bool in_lds = (size_t)c_cnt * (M * M * sizeof(T) + 4) <= (size_t)lds_bytes;
  T* hb_l = reinterpret_cast<T*>(dyn_lds);
  unsigned* nb_l = reinterpret_cast<unsigned*>(dyn_lds + (size_t)c_cnt * M * M * sizeof(T));
  if (in_lds) {
    const T* src = HB + (size_t)c_lo * M * M;
    for (int e = threadIdx.x; e < c_cnt * M * M; e += kPersistBlock) hb_l[e] = src[e];
    for (int e = threadIdx.x; e < c_cnt; e += kPersistBlock) nb_l[e] = (unsigned)other[c_lo + e];
  }
  __syncthreads();
  const unsigned own = (unsigned)(((act ? n : n0) * M + (act ? i : 0)) * NW);     // (an idle lane watches the workgroup's first element)
  const int lbeg = act ? beg - c_lo : 0;
  T bn2 = T(0), rr = T(0);
  int k = 0, flag = 0;                     // flag: 1 converged, 2 NaN, 3 a workgroup never arrived, 0 iteration limit
  for (;; ++k) {
    const unsigned tag = (unsigned)k + 1u;
    const u64* pin = ptag + (size_t)(k & 1) * NM;
    // ---- q = A p on this lane's row: own block from the node's lanes, neighbours' rows from the tagged table
    T acc = T(0);
    bool stale = false;
#pragma unroll
    for (int j = 0; j < M; ++j) acc += dr[j] * __shfl(pe, sub * M + j, 64);
    if (in_lds)
      acc += spmv_rows<T, M, CH>(hb_l + ((size_t)lbeg * M + i) * M, nb_l + lbeg, deg, maxdeg, sub, i, pin, own, tag, stale);
    else
      acc += spmv_rows<T, M, CH>(HB + ((size_t)beg * M + i) * M, other + beg, deg, maxdeg, sub, i, pin, own, tag, stale);
    T bq = T(0);
#pragma unroll
    for (int j = 0; j < M; ++j) bq += br[j] * __shfl(acc, sub * M + j, 64);
    T v[kPersistQ] = {acc * pe, acc * ze, acc * bq, re * ze, re * re};
    if (!act) {
#pragma unroll
      for (int q = 0; q < kPersistQ; ++q) v[q] = T(0);
    }
    wg_totals<T, kPersistQ, kPersistBlock>(v);
    u64* row = part + ((size_t)(k & 1) * kPersistGridMax + blockIdx.x) * RW;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < kPersistQ; ++q) put_value<T>(row + q * NW, v[q], tag);
    }
    T tot[kPersistQ];
    const bool arrived = gather_rows<T, kPersistQ>(part + (size_t)(k & 1) * kPersistGridMax * RW, gridDim.x, tag, tot);
    if (__syncthreads_or((int)stale) || !arrived) { flag = 3; break; }
    const T pq = tot[0], qz = tot[1], qmq = tot[2], rho = tot[3];
    rr = tot[4];
    if (k == 0) bn2 = rr;
    if (blockIdx.x == 0 && threadIdx.x == 0 && k < cap) rr_hist[k] = rr;
    if (!(rr == rr)) { flag = 2; break; }
    if (rr <= tol2 * bn2) { flag = 1; break; }                       // (also |b| = 0: x = 0 is the answer)
    if (k >= maxiter) break;
    const T alpha = pq != T(0) ? rho / pq : T(0);                   // p.q = 0 only once r = 0: stay put, no NaN
    T rho_next = rho - T(2) * alpha * qz + alpha * alpha * qmq;
    if (rho_next < T(0)) rho_next = T(0);
    const T beta = rho != T(0) ? rho_next / rho : T(0);
    // ---- vector update on this lane's element
    xe += alpha * pe;
    re -= alpha * acc;
    T zn = T(0);
#pragma unroll
    for (int j = 0; j < M; ++j) zn += br[j] * __shfl(re, sub * M + j, 64);
    ze = zn;
    pe = ze + beta * pe;
    if (act) put_value<T>(ptag + (size_t)((k + 1) & 1) * NM + (size_t)(n * M + i) * NW, pe, tag + 1u);
  }
  // a failed solve (NaN, or a workgroup that never arrived) hands back x = 0: the caller may have queued the parameter
  // update behind this launch and look at `info` only afterwards (one read-back per LM trial) -- Exp(0) p = p
  if (act) x[n * M + i] = flag >= 2 ? T(0) : xe;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    info[0] = (T)k; info[1] = rr; info[2] = bn2; info[3] = (T)flag;
    it_out[0] = k;
  }
}

// Dynamic LDS of a workgroup (the staged matrix slice); the most workgroups of this kernel the device holds at once
// (they spin on each other: all must be resident)
constexpr int kPersistLds = 128 * 1024;
template <class T, int M> static int persist_capacity(int& lds_bytes) {
  static int cap[16] = {0}, lds[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (cap[dev] == 0) {
    int cus = 0, per = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    lds[dev] = kPersistLds;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_persist_kernel<T, M>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kPersistLds) != hipSuccess) {
      (void)hipGetLastError();
      lds[dev] = 48 * 1024;                                      // (always available without the attribute)
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, pcg_persist_kernel<T, M>, kPersistBlock, lds[dev]) != hipSuccess) return 0;
    cap[dev] = cus * per > 0 ? cus * per : -1;
  }
  lds_bytes = lds[dev];
  return cap[dev] > 0 ? cap[dev] : 0;
}

template <class T>
int pcg_persist(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x, void* r, void* p,
                void* q, void* z, void* part, void* ptag, void* rr_hist, void* info, void* it, double tol, int maxiter, int cap,
                int grid, int64_t N, int m, void* stream) {
  if (N <= 0) return N == 0 ? PPLIE_OK : PPLIE_EBADARG;
  if (!ptr || !other || !HB || !D || !Binv || !x || !r || !z || !part || !ptag || !rr_hist || !info || !it) return PPLIE_EBADARG;
  if (grid < 1 || grid > kPersistGridMax || maxiter < 0) return PPLIE_EBADARG;
  (void)p; (void)q;                                              // (round-2 signature: p and q now live in registers)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define LAUNCH(MM)                                                                                                             \
  {                                                                                                                            \
    int lds_bytes = 0;                                                                                                         \
    const int resident = persist_capacity<T, MM>(lds_bytes);                                                                   \
    if (resident <= 0) return PPLIE_ECAPACITY;                                                                                 \
    if (grid > resident) grid = resident;                         /* fewer CUs than asked for: every workgroup must be resident */ \
    if (grid > N) grid = (int)N;                                                                                               \
    const int64_t per_wg = (kPersistBlock / 64) * (64 / MM);     /* one lane per (node, component): nodes one workgroup holds */ \
    if ((N + grid - 1) / grid > per_wg) return PPLIE_ECAPACITY;  /* too large for this device: use the two-launch iteration */  \
    hipLaunchKernelGGL((pcg_persist_kernel<T, MM>), dim3(grid), dim3(kPersistBlock), lds_bytes, st, (const int*)ptr, (const int*)other, \
                       (const T*)HB, (const T*)D, (const T*)Binv, (T*)x, (const T*)r, (const T*)z, (unsigned long long*)part,     \
                       (unsigned long long*)ptag, (T*)rr_hist, (T*)info, (int*)it, (T)(tol * tol), maxiter, cap, N, lds_bytes);  \
  }
  if (m == 6) LAUNCH(6) else if (m == 7) LAUNCH(7) else if (m == 3) LAUNCH(3) else return PPLIE_EBADARG;
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace pplie

extern "C" int pplie_pcg_persist_f32(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x,
                                     void* r, void* p, void* q, void* z, void* part, void* ptag, void* rr_hist, void* info, void* it,
                                     double tol, int maxiter, int cap, int grid, int64_t N, int m, void* stream) {
  return pplie::pcg_persist<float>(ptr, other, HB, D, Binv, x, r, p, q, z, part, ptag, rr_hist, info, it, tol, maxiter, cap, grid, N, m, stream);
}
extern "C" int pplie_pcg_persist_f64(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x,
                                     void* r, void* p, void* q, void* z, void* part, void* ptag, void* rr_hist, void* info, void* it,
                                     double tol, int maxiter, int cap, int grid, int64_t N, int m, void* stream) {
  return pplie::pcg_persist<double>(ptr, other, HB, D, Binv, x, r, p, q, z, part, ptag, rr_hist, info, it, tol, maxiter, cap, grid, N, m, stream);
}
