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
//   * each lane owns ONE (node, component) for the whole solve: x, r, z, p, q and its parts of D and Binv live in registers;
//     the workgroup's slice of the off-diagonal blocks (the incidences of its nodes: contiguous, incidence order is node
//     order) and their neighbour indices are staged into LDS once, transposed, and read from there in every iteration.
//   * the SpMV is organised by COLUMNS: lane (node, j) fetches component j of every neighbour's p -- one tagged load per
//     incidence, all of a node's (up to 16) in flight together, none for absent incidences -- multiplies it into column j
//     of the block (six contiguous LDS words) and accumulates the node's whole q row; the M lanes' partial rows are then
//     summed through a per-wave LDS transpose.  (Measured on MI355X, tools/micro/pingpong.hip: a hand-off between two
//     workgroups costs 0.3-0.5 us and an all-gather over 256 workgroups 2.4 us -- the 15 us of the round-2 iteration and the
//     11 us of this file's first version were mostly the walk over the incidence list: dependent round trips four
//     incidences at a time, then six cross-lane shuffles per incidence.)
//
// The exchange itself is a tagged all-gather (one row of partial sums per workgroup, two tables alternating with the
// iteration's parity, polled by one wave per quantity); every workgroup adds the rows up in the same order, so all of them
// hold the same alpha / beta / |r|^2 bits and take the same exit.  Two workgroup barriers per iteration.
#include <cstdlib>
#include "rowmap.h"
#include "gridsync.h"
#include "dpp.h"

namespace pplie {

// Denominators of alpha = rho / p.q and beta = rho' / rho below this are treated as zero (the step stays put): an iteration driven
// far past convergence reaches the denormal range (fp32: ~130 iterations at the two-level preconditioner's rate of 0.5 per
// iteration), where the fast reciprocal of a flushed denormal is inf and 0 * inf a NaN in every vector
template <class T> __device__ __forceinline__ constexpr T pcg_tiny() { return sizeof(T) == 4 ? T(1e-30) : T(1e-290); }

constexpr int kPersistGridMax = 256;      // = PPLIE_PCG_PERSIST_GRID: rows of the partial-sum tables
constexpr int kPersistBlock = 1024;       // 16 waves per workgroup
// polls of a word another RANK writes (another process, its launch not synchronised with this one's beyond a host barrier): four times
// the in-GPU limit -- a peer process held up for a second on a busy host is late, not lost (the one flaky failure of the multi-process
// test in round 6 was on a loaded box)
constexpr long kPeerSpins = 1L << 22;
constexpr int kPersistQ = 5;              // quantities per exchange: p.q, q.z, q.Binv q, r.z, r.r
constexpr int kPersistSlots = 8;          // table row = 8 quantity slots (PPLIE_PCG_PERSIST_SLOTS)
constexpr int kCoarseSlots = 24;          // row of the partial-sum table with the coarse sums (5 + 2 M <= 19 quantities)

// one value as NW tagged words.  SYS: the word crosses GPUs (peer-mapped memory over xGMI): system-scope accesses
template <class T, bool SYS = false> __device__ __forceinline__ void put_value(u64* dst, T v, unsigned tag) {
  constexpr int NW = sizeof(T) / 4;
  unsigned w[NW];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const u64 word = ((u64)tag << 32) | (u64)w[k];
    if (SYS) __hip_atomic_store(dst + k, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else xwg_store(dst + k, word);
  }
}
template <class T, bool SYS = false> __device__ __forceinline__ T get_value(const u64* src, unsigned tag, bool& ok) {
  constexpr int NW = sizeof(T) / 4;
  unsigned w[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const u64 v = SYS ? __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : xwg_load(src + k);
    ok = ok && (unsigned)(v >> 32) == tag;
    w[k] = (unsigned)v;
  }
  T out;
  __builtin_memcpy(&out, w, sizeof(T));
  return out;
}

// a[0..M) += sum over this lane's incidences c of (column i of block c) * (component i of the neighbour's p).  `h`: the first
// block's column i (element j at h[c * cs + j * js]), `nb`: the neighbour indices.  CH incidences per round: every tagged
// load of a round is in flight before the first is looked at; absent incidences (c >= deg) load nothing (exec-masked).
template <class T, int M, int CH, bool SYS, class HP, class NP>
__device__ __forceinline__ void spmv_cols(T* a, HP h, int cs, int js, NP nb, int deg, int maxdeg, int i, const u64* pin, unsigned tag,
                                          bool& stale) {
  constexpr int NW = sizeof(T) / 4;
  for (int c0 = 0; c0 < maxdeg; c0 += CH) {
    T pv[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) pv[q] = T(0);
    for (long spin = 0;; ++spin) {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < CH; ++q)
        if (c0 + q < deg) pv[q] = get_value<T, SYS>(pin + (unsigned)((nb[c0 + q] * M + i) * NW), tag, ok);
      if (__all(ok)) break;
      if (spin >= (SYS ? kPeerSpins : (1L << 20))) { stale = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      if (c0 + q < maxdeg) {                                                    // (wave-uniform)
        const bool valid = c0 + q < deg;
        const int c = valid ? c0 + q : 0;
#pragma unroll
        for (int j = 0; j < M; ++j) a[j] += (valid ? h[(size_t)c * cs + j * js] : T(0)) * pv[q];
      }
    }
  }
}

// Several GPUs, one process each (LM(group=, shard="nodes", exchange="p2p"), optim/nodeshard.py): every rank runs this kernel on
// the node rows it owns.  The hand-off table of p spans ALL nodes and exists once per rank; a rank stores its elements into
// every rank's copy (its own and, through peer-mapped pointers over xGMI, the others'), so that every gather stays a local
// read.  The dot products go two levels: the all-gather over this rank's workgroups as on one GPU, then workgroup 0 stores
// the rank's totals into every rank's small `rpart` table and everybody adds the `world` rows in rank order -- the same
// bits on every GPU.  Tags carry a per-solve epoch (the tables are not cleared between solves: a clear would race with a
// peer that is already a solve ahead).
constexpr int kPersistMaxWorld = 8;
struct PersistPeers {
  int world, rank;
  unsigned tag_base;               // (epoch & 0xffff) << 16
  long long row0, n_global;        // first owned node, nodes of the whole graph
  u64* ptag[kPersistMaxWorld];     // rank r's hand-off table  [2][n_global * M values as tagged words]
  u64* rpart[kPersistMaxWorld];    // rank r's rank-level sums [2][kPersistMaxWorld][kPersistSlots values as tagged words]
};

// static LDS of the exchange, double-buffered by the iteration's parity so that one barrier separates "written" from "read"
// and the next iteration's writes cannot overtake this one's reads
template <class T, int NQMAX = kPersistQ> struct PersistShared {
  T wave_part[2][NQMAX][kPersistBlock / 64];       // per-wave partial sums
  T total[2][NQMAX];                               // the all-gathered sums
  int bad[2];                                      // a poll timed out / a lane saw a stale vector element for too long
};

// per-lane constants of the solve: which (node, component) this lane owns and where its matrix slice sits
template <class T, int M> struct PersistLane {
  T dc[M], br[M];            // COLUMN i of the damped diagonal block, ROW i of its inverse
  int64_t n;                 // node
  int sub, i, beg, deg, lbeg, maxdeg;
  bool act, in_lds;
  T* tr;                     // this wave's transpose pad
  const T* hb_l;             // staged blocks (transposed), neighbour indices
  const unsigned* nb_l;
};

// q_i of this lane's node for the vector whose component i this lane holds (`ve`) and whose other elements are read from
// the tagged table `pin`:  diagonal block column, neighbour columns (spmv_cols), then the sum over the node's M lanes
template <class T, int M, int CH, bool SYS>
__device__ __forceinline__ T node_matvec(const PersistLane<T, M>& L, T ve, const T* HB, const int* other, const u64* pin, unsigned tag,
                                         bool& stale) {
  constexpr int NPW = 64 / M;
  T a[M];
#pragma unroll
  for (int j = 0; j < M; ++j) a[j] = L.dc[j] * ve;
  if (L.in_lds) spmv_cols<T, M, CH, SYS>(a, L.hb_l + ((size_t)L.lbeg * M + L.i) * M, M * M, 1, L.nb_l + L.lbeg, L.deg, L.maxdeg, L.i, pin, tag, stale);
  else spmv_cols<T, M, CH, SYS>(a, HB + (size_t)L.beg * M * M + L.i, M * M, M, other + L.beg, L.deg, L.maxdeg, L.i, pin, tag, stale);
  // the node's row = sum of its M lanes' partial rows: through this wave's LDS pad (a wave's LDS accesses execute in order)
  T acc = T(0);
  if (L.sub < NPW) {
#pragma unroll
    for (int j = 0; j < M; ++j) L.tr[(L.sub * M + L.i) * M + j] = a[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (L.sub < NPW) {
#pragma unroll
    for (int j = 0; j < M; ++j) acc += L.tr[(L.sub * M + j) * M + L.i];
  }
  __builtin_amdgcn_wave_barrier();                               // (the pad is rewritten by the next product)
  return acc;
}
// (Binv v)_i from the node's lanes
template <class T, int M> __device__ __forceinline__ T node_binv(const PersistLane<T, M>& L, T ve) {
  T s = T(0);
#pragma unroll
  for (int j = 0; j < M; ++j) s += L.br[j] * __shfl(ve, L.sub * M + j, 64);
  return s;
}
// wave-level sums of NQ quantities into the exchange's LDS (every lane calls; then ONE __syncthreads, then publish_row)
template <class T, int NQ, class SH> __device__ __forceinline__ void post_wave_sums(SH& sh, int par, T* v, bool act, bool stale) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // (DPP adds on the VALU: the 30 ds_bpermute of a shuffle tree queue behind the SpMV's reads on the one LDS pipe the 16 waves share)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (!act) v[q] = T(0);
    v[q] = wave_total63<T>(v[q]);
    if (lane == 63) sh.wave_part[par][q][w] = v[q];
  }
  if (stale) sh.bad[par] = 1;
}
template <class T, int NQ, class SH, int SLOTS = kPersistSlots>
__device__ __forceinline__ void publish_row(SH& sh, int par, u64* part, unsigned tag) {
  constexpr int NW = sizeof(T) / 4, RW = SLOTS * NW, WV = kPersistBlock / 64;
  if (threadIdx.x < NQ) {
    T sum = T(0);
#pragma unroll
    for (int ww = 0; ww < WV; ++ww) sum += sh.wave_part[par][threadIdx.x][ww];
    put_value<T>(part + ((size_t)par * kPersistGridMax + blockIdx.x) * RW + threadIdx.x * NW, sum, tag);
  }
}
// all-gather: wave q polls quantity q of every workgroup's row (all loads of a round in flight together) and adds them up in
// row order -- the same order, hence the same bits, in every workgroup.  Then ONE __syncthreads; totals in sh.total[par].
template <class T, int NQ, class SH, int SLOTS = kPersistSlots>
__device__ __forceinline__ void gather_rows(SH& sh, int par, const u64* part, unsigned tag, int first = 0, int stride = 1, int count = -1,
                                            int rows_per_table = kPersistGridMax) {
  // rows first, first + stride, ... (count of them; default: one per workgroup of the grid) of table `par`
  constexpr int NW = sizeof(T) / 4, RW = SLOTS * NW, WVS = kPersistBlock / 64;
  constexpr int PER = (NQ + WVS - 1) / WVS;                  // quantities per wave: w, w + 16, ... polled TOGETHER (one spin loop:
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;   //  a second quantity costs no second round of L2 latencies)
  if (count < 0) count = (int)gridDim.x;
  if (w < NQ) {
    const u64* tab = part + (size_t)par * rows_per_table * RW;
    T sum[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) sum[u] = T(0);
    bool all = true;
    for (int base = 0; base < count; base += 256) {
      T val[PER][4];
      bool done[PER][4];
#pragma unroll
      for (int u = 0; u < PER; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) { done[u][q] = base + lane + 64 * q >= count || w + u * WVS >= NQ; val[u][q] = T(0); }
      for (long spin = 0; spin < (1L << 20); ++spin) {
        bool pending = false;
#pragma unroll
        for (int u = 0; u < PER; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (!done[u][q]) {
              bool ok = true;
              const T t = get_value<T>(tab + (size_t)(first + stride * (base + lane + 64 * q)) * RW + (w + u * WVS) * NW, tag, ok);
              if (ok) { val[u][q] = t; done[u][q] = true; } else pending = true;
            }
          }
        if (!pending) break;
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int u = 0; u < PER; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) { all = all && done[u][q]; sum[u] += val[u][q]; }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      T t = sum[u];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
      if (lane == 0 && w + u * WVS < NQ) sh.total[par][w + u * WVS] = t;
    }
    if (!__all(all) && lane == 0) sh.bad[par] = 1;
  }
}

// Profiling clock of the ghost-zone kernel (tools/time_pcg_iter.py, cap < 0): kTickSlots accumulators of 10 ns wall-clock ticks +
// the previous reading, in LDS, touched by thread 0 of a clocked workgroup only (`clocked` is false on every production launch).
constexpr int kTickSlots = 14;
__device__ __forceinline__ unsigned* tick_store() {
  __shared__ unsigned tk_[kTickSlots + 3];
  return tk_;
}
__device__ __forceinline__ void tick(bool clocked, int slot) {
  if (clocked) {
    unsigned* tk = tick_store();
    const unsigned t_now = (unsigned)wall_clock64();
    tk[slot] += t_now - tk[kTickSlots];
    tk[kTickSlots] = t_now;
  }
}

// PAIRS: two fp32 partial sums per 64-bit word, each carrying a 2-bit tag in the two LOWEST MANTISSA BITS (a relative perturbation of
// 2.4e-7 of a dot product whose own summation error is larger).  Two bits are enough because the tables alternate with the
// iteration's parity and every workgroup rewrites its row at every use of a table: what a reader can find in a slot is the value
// of THIS use or of the PREVIOUS use of the same table, never an older one -- consecutive uses carry tags 1, 2, 3, 1, ... and 0 is
// "never written" (the tables are zeroed per solve).  Halves the words of the wide exchange (17 quantities -> 9 words): measured
// per-exchange cost is ~2.1 us + 0.19 us per polled word at 160 workgroups.
__device__ __forceinline__ unsigned pair_tag(int use) { return (unsigned)(use % 3) + 1u; }
__device__ __forceinline__ u64 pack_pair(float a, float b, unsigned t2) {
  const unsigned ua = (__float_as_uint(a) & ~3u) | t2, ub = (__float_as_uint(b) & ~3u) | t2;
  return ((u64)ub << 32) | (u64)ua;
}
__device__ __forceinline__ bool unpack_pair(u64 w, unsigned t2, float& a, float& b) {
  const unsigned ua = (unsigned)w, ub = (unsigned)(w >> 32);
  a = __uint_as_float(ua & ~3u);
  b = __uint_as_float(ub & ~3u);
  return (ua & 3u) == t2 && (ub & 3u) == t2;
}
// word j of a row = quantities (2 j, 2 j + 1); rows of SLOTS words (fp32: one word per slot)
// Table layout: ROW-major, word j of row b at tab[b * SLOTS + j] -- a row's words share one or two cache lines that ONE workgroup
// writes.  (Round 6 experiment, PPLIE_PAIRS_ROW_MAJOR=0: WORD-major, tab[j * kPairRows + b], so that a gathering wave reads 512
// contiguous bytes instead of 64 lines.  Measured on the 10 k-pose LM step: 0.306 ms against 0.263 -- the final gather went from
// 2.8 to 4.1 us per iteration.  A line then carries words of 8 - 16 workgroups written at different times: the partial-line
// write-throughs and the polls of that line collide.  One writer per line is what is fast; kept for the record.)
#ifndef PPLIE_PAIRS_ROW_MAJOR
#define PPLIE_PAIRS_ROW_MAJOR 1
#endif
constexpr int kPairRows = 256 + 8;       // = kHierRows (defined below): rows of one pair table
// (also tried, round 6: a row pitch of 512 B / 1 KB / 4 KB instead of 192 B, to spread the rows over more memory channels -- no
//  effect beyond the run-to-run noise of 1 %: the gather is not bound by where the rows live)
template <int SLOTS> __device__ __forceinline__ constexpr int pair_pitch() { return SLOTS; }
template <int SLOTS> __device__ __forceinline__ size_t pair_at(int row, int word) {
  return PPLIE_PAIRS_ROW_MAJOR ? (size_t)row * pair_pitch<SLOTS>() + word : (size_t)word * kPairRows + row;
}
template <int NQ, class SH, int SLOTS>
__device__ __forceinline__ void put_pairs(const float* vals_lds, u64* tab, int row, unsigned t2) {
  constexpr int NP = (NQ + 1) / 2;
  static_assert(NP <= SLOTS, "a table holds SLOTS words per row in either layout");
  if (threadIdx.x < NP) {
    const float a = vals_lds[2 * threadIdx.x], b = 2 * threadIdx.x + 1 < NQ ? vals_lds[2 * threadIdx.x + 1] : 0.f;
    xwg_store(tab + pair_at<SLOTS>(row, threadIdx.x), pack_pair(a, b, t2));
  }
}
// (Round 6, measured on the 10 k-pose LM step with four builds of this file side by side, tools/gpu_ab_lib.sh: the loop below waits
//  for each row's load before issuing the next -- three dependent round trips per poll round -- and is still the FASTEST form.
//  Issuing a round's loads together and re-reading every row until all are complete made the step 16 us slower (0.281 against
//  0.265 ms); issuing only the first round together, 3 - 5 us slower.  The polls of 160 workgroups x 9 waves share the memory side
//  with the stores they are waiting for: fewer, staggered polls win over fewer round trips.  A delay in front of the first poll
//  (s_sleep 8 / 16 / 32 / 64 x 64 clocks): 0 / 0 / +0.5 / +2.4 us per iteration -- the last row is there ~0.6 us after the gather
//  starts; the batched first round behind such a delay: still 3 - 6 % slower than the serial loop.)
template <int NQ, class SH, int SLOTS>
__device__ __forceinline__ void gather_pairs(SH& sh, int par, const u64* part, unsigned t2, int first, int stride, int count, int rows_per_table) {
  constexpr int NP = (NQ + 1) / 2, RW = SLOTS, LD = 3;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w < NP) {
    const u64* tab = part + (size_t)par * rows_per_table * pair_pitch<SLOTS>();
    float s0 = 0.f, s1 = 0.f;
    bool all = true;
    for (int base = 0; base < count; base += 64 * LD) {            // (LD rows per lane and round: 192 cover the usual grids of 160 - 192)
      float v0[LD], v1[LD];
      bool done[LD];
#pragma unroll
      for (int q = 0; q < LD; ++q) { done[q] = base + lane + 64 * q >= count; v0[q] = 0.f; v1[q] = 0.f; }
      for (long spin = 0; spin < (1L << 20); ++spin) {
        bool pending = false;
#pragma unroll
        for (int q = 0; q < LD; ++q) {
          if (!done[q]) {
            float a, b;
            if (unpack_pair(xwg_load(tab + pair_at<SLOTS>(first + stride * (base + lane + 64 * q), w)), t2, a, b)) { v0[q] = a; v1[q] = b; done[q] = true; }
            else pending = true;
          }
        }
        if (!pending) break;
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int q = 0; q < LD; ++q) { all = all && done[q]; s0 += v0[q]; s1 += v1[q]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    if (lane == 0) {
      sh.total[par][2 * w] = s0;
      if (2 * w + 1 < NQ) sh.total[par][2 * w + 1] = s1;
    }
    if (!__all(all) && lane == 0) sh.bad[par] = 1;
  }
}

// The second level of the two-level exchange by ONE wavefront (round 6): lane (g, word) = (lane / 8, lane % 8) fetches one word of group
// g's row -- G <= 8 rows of NP <= 8 words, one load instruction instead of one per word-wave -- and three shuffle steps add the groups
// (a fixed tree: the same bits in every workgroup).  The wave then holds every total and can go on to the iteration's scalars without a
// workgroup barrier in between (pcg_ghost_kernel: alpha, beta and the coarse part of z, which sixteen waves otherwise each work out).
template <int NQ, class SH, int SLOTS>
__device__ __forceinline__ void gather_groups_one_wave(SH& sh, int par, const u64* part, unsigned t2, int G, int rows_per_table) {
  constexpr int NP = (NQ + 1) / 2;
  static_assert(NP <= 8 && kPersistGridMax % 8 == 0, "lane = group * 8 + word");
  const int lane = threadIdx.x & 63;
  if ((threadIdx.x >> 6) == 0) {
    const int g = lane >> 3, wq = lane & 7;
    const u64* tab = part + (size_t)par * rows_per_table * pair_pitch<SLOTS>();
    float a = 0.f, b = 0.f;
    bool done = !(g < G && wq < NP);
    for (long spin = 0; spin < (1L << 20); ++spin) {
      if (!done) {
        float x, y;
        if (unpack_pair(xwg_load(tab + pair_at<SLOTS>(kPersistGridMax + g, wq)), t2, x, y)) { a = x; b = y; done = true; }
      }
      if (__all(done)) break;
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
    if (lane < NP) {
      sh.total[par][2 * lane] = a;
      if (2 * lane + 1 < NQ) sh.total[par][2 * lane + 1] = b;
    }
    if (!__all(done) && lane == 0) sh.bad[par] = 1;
  }
}

// TWO-LEVEL all-gather (the wide exchange of the two-level preconditioner: 5 + 2 M quantities).  Every workgroup polling every
// workgroup's row is grid x grid x NQ tagged loads per iteration -- all of them served by the memory side, the L2s of the eight
// XCDs are not coherent with each other: measured 7.4 us per exchange with 17 quantities at 256 workgroups against 3.1 us with 5.
// Here workgroup g < G (G = 8 groups) adds up the rows of the workgroups b = g (mod G) -- the ones the dispatcher places on its
// own XCD -- and publishes ONE group row; everybody then polls the G group rows: (grid / G + G) x NQ loads per workgroup instead
// of grid x NQ, two dependent hops instead of one.  Same bits everywhere: a group's sum has one order (its leader's), the final
// sum runs over the groups in order.  Table rows: [0, kPersistGridMax) workgroups, then kHierGroups group rows.
constexpr int kHierGroups = 8;
constexpr int kHierMinGrid = 224;        // grids from here on exchange in two levels
constexpr int kHierRows = kPersistGridMax + kHierGroups;
static_assert(kHierRows == kPairRows, "pair tables: kPairRows rows");
template <class T, int NQ, class SH, int SLOTS, bool ONEWAVE = false>
__device__ __forceinline__ void exchange_two_level(SH& sh, int par, u64* part, unsigned tag, int use, bool clocked = false) {
  constexpr int NW = sizeof(T) / 4, RW = SLOTS * NW, WV = kPersistBlock / 64;
  const int G = (int)gridDim.x < kHierGroups ? (int)gridDim.x : kHierGroups;
  if constexpr (sizeof(T) == 4) {
    // fp32: packed pairs (see above); `use` = how many times this table has been used before in this solve
    __shared__ float mine[NQ + 1];
    const unsigned t2 = pair_tag(use);
    // (round 6: thread j summing quantities 2 j, 2 j + 1 itself and storing the pair -- no `mine`, no workgroup barrier here -- measured
    //  SLOWER on the LM step, 0.264 against 0.260 ms: the barrier also holds the gathering waves back until the row is on its way)
    if (threadIdx.x < NQ) {
      float sum = 0.f;
#pragma unroll
      for (int ww = 0; ww < WV; ++ww) sum += sh.wave_part[par][threadIdx.x][ww];
      mine[threadIdx.x] = sum;
    }
    __syncthreads();
    put_pairs<NQ, SH, SLOTS>(mine, part + (size_t)par * kHierRows * pair_pitch<SLOTS>(), blockIdx.x, t2);
    tick(clocked, 5);
    if ((int)gridDim.x < kHierMinGrid) {
      gather_pairs<NQ, SH, SLOTS>(sh, par, part, t2, 0, 1, (int)gridDim.x, kHierRows);
      tick(clocked, 7);
      return;
    }
    if ((int)blockIdx.x < G) {
      const int members = ((int)gridDim.x - (int)blockIdx.x + G - 1) / G;
      gather_pairs<NQ, SH, SLOTS>(sh, par, part, t2, (int)blockIdx.x, G, members, kHierRows);
      __syncthreads();
      put_pairs<NQ, SH, SLOTS>(&sh.total[par][0], part + (size_t)par * kHierRows * pair_pitch<SLOTS>(), kPersistGridMax + blockIdx.x, t2);
      __syncthreads();
      tick(clocked, 6);
    }
    if constexpr (ONEWAVE) gather_groups_one_wave<NQ, SH, SLOTS>(sh, par, part, t2, G, kHierRows);
    else gather_pairs<NQ, SH, SLOTS>(sh, par, part, t2, kPersistGridMax, 1, G, kHierRows);
    tick(clocked, 7);
    return;
  }
  // (caller: post_wave_sums + __syncthreads done)  own row
  if (threadIdx.x < NQ) {
    T sum = T(0);
#pragma unroll
    for (int ww = 0; ww < WV; ++ww) sum += sh.wave_part[par][threadIdx.x][ww];
    put_value<T>(part + ((size_t)par * kHierRows + blockIdx.x) * RW + threadIdx.x * NW, sum, tag);
  }
  // Measured (tools/time_pcg_iter.py, 10 k nodes, 17 quantities): one level 8.6 us per iteration at 160 workgroups and 11.1 at 256;
  // two levels 9.5 at 160 and 9.2 at 256 -- the second hop costs what the smaller gather saves unless the grid is large.
  if ((int)gridDim.x < kHierMinGrid) {
    gather_rows<T, NQ, SH, SLOTS>(sh, par, part, tag, 0, 1, (int)gridDim.x, kHierRows);
    return;
  }
  if ((int)blockIdx.x < G) {                                   // leader of group blockIdx.x
    const int members = ((int)gridDim.x - (int)blockIdx.x + G - 1) / G;
    gather_rows<T, NQ, SH, SLOTS>(sh, par, part, tag, (int)blockIdx.x, G, members, kHierRows);
    __syncthreads();
    if (threadIdx.x < NQ)
      put_value<T>(part + ((size_t)par * kHierRows + kPersistGridMax + blockIdx.x) * RW + threadIdx.x * NW, sh.total[par][threadIdx.x], tag);
    __syncthreads();                                           // (sh.total is rewritten by the gather below)
  }
  gather_rows<T, NQ, SH, SLOTS>(sh, par, part, tag, kPersistGridMax, 1, G, kHierRows);
}

// (A PIPELINED recurrence -- Ghysels & Vanroose 2014: post the dot products before the matrix product, collect them after, so
// that the all-gather overlaps the neighbour exchange -- was built and measured in round 3: 10.5 us per iteration instead of
// 12.2, but its recurrences for Binv r and A Binv r lose the residual in fp32: it stalls above 1e-4 already at condition
// number 100, cf. profiles/r03/SUMMARY.md.  Not kept.)
// CZ (with XG): the two-level preconditioner of the ghost-zone kernel below (block-Jacobi + the gauge modes Z = 1_N (x) I_M; see the
// comment there) on the node-sharded multi-GPU solve: 2 M more sums per exchange at both levels (workgroups of this rank, then the
// ranks), one exchange before the first iteration for E = sum_n shift[n, :] and Z^T r_0.  Rows of kCoarseSlots values in `part` and
// in the peers' `rpart` tables.
template <class T, int M, bool XG, bool CZ = false>
__global__ void __launch_bounds__(kPersistBlock)
pcg_persist_kernel(const int* __restrict__ ptr, const int* __restrict__ other, const T* __restrict__ HB, const T* __restrict__ D,
                   const T* __restrict__ Binv, T* __restrict__ x, const T* __restrict__ r, const T* __restrict__ z,
                   u64* part /* [2][kPersistGridMax][kPersistSlots (CZ: kCoarseSlots) values as tagged words] */,
                   u64* ptag /* [2][N * M values as tagged words] (XG: peers.ptag[rank], all nodes of the graph) */,
                   T* __restrict__ rr_hist, T* info /* [4] */, int* it_out, T tol2, int maxiter, int cap, int64_t N, int lds_bytes,
                   PersistPeers peers, const T* __restrict__ shift = nullptr) {
  static_assert(!CZ || XG, "the two-level variant of this kernel is the multi-GPU one (one GPU: pcg_ghost_kernel<T, M, true>)");
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  constexpr int NQ = CZ ? kPersistQ + 2 * M : kPersistQ;
  constexpr int SLOTS = CZ ? kCoarseSlots : kPersistSlots;
  typedef PersistShared<T, NQ> SH;
  __shared__ SH sh;
  __shared__ T cz_pad[CZ ? kPersistBlock : 1];
  __shared__ T einv[CZ ? M : 1];
  constexpr int CH = sizeof(T) == 4 ? 16 : 8;   // incidences per round of tagged loads
  constexpr int NPW = 64 / M;              // nodes per wave: M lanes per node
  constexpr int WV = kPersistBlock / 64;   // waves per workgroup
  constexpr int NW = sizeof(T) / 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t n0 = N * blockIdx.x / gridDim.x, n1 = N * (blockIdx.x + 1) / gridDim.x;     // (host: n1 - n0 <= WV * NPW)
  const size_t NM = (size_t)(XG ? peers.n_global : N) * M * NW;     // one table of the hand-off
  const int64_t g0 = XG ? peers.row0 : 0;                          // global id of this rank's first node
  const unsigned tag0 = XG ? peers.tag_base : 0u;
  PersistLane<T, M> L;
  L.sub = lane / M;
  L.i = lane % M;
  L.n = n0 + w * NPW + L.sub;
  L.act = L.sub < NPW && L.n < n1;
  const int64_t n = L.n;
  const int i = L.i;
  const bool act = L.act;

  // ---- this lane's (node, component) for the whole solve
  T xe = T(0), re = T(0), ze = T(0);
  L.beg = 0;
  L.deg = 0;
#pragma unroll
  for (int j = 0; j < M; ++j) { L.dc[j] = T(0); L.br[j] = T(0); }
  if (act) {
#pragma unroll
    for (int j = 0; j < M; ++j) { L.dc[j] = D[(n * M + j) * M + i]; L.br[j] = Binv[(n * M + i) * M + j]; }
    re = r[n * M + i];                     // pplie_pcg_prepare left r = -g, z = Binv r, x = 0
    ze = z[n * M + i];
    L.beg = ptr[n];
    L.deg = ptr[n + 1] - L.beg;
  }
  // hand-off h of the search direction carries tag h + 1 (+ the solve's epoch), in table h & 1 -- of every rank
  auto hand_off = [&](T v, int table, unsigned tag) {
    if (!act) return;
    const size_t at = (size_t)table * NM + (size_t)((g0 + n) * M + i) * NW;
    if (XG) {
      for (int rk = 0; rk < peers.world; ++rk) put_value<T, true>(peers.ptag[rk] + at, v, tag);
    } else {
      put_value<T>(ptag + at, v, tag);
    }
  };
  if constexpr (!CZ) hand_off(ze, 0, tag0 + 1u);
  L.maxdeg = L.deg;                        // largest degree among this wave's nodes
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(L.maxdeg, off, 64);
    L.maxdeg = o > L.maxdeg ? o : L.maxdeg;
  }
  // ---- LDS: [ per-wave transpose pads | staged blocks (transposed) | staged neighbour indices ]
  L.tr = reinterpret_cast<T*>(dyn_lds) + (size_t)w * NPW * M * M;             // this wave's pad: [NPW][M (source lane j)][M (row)]
  constexpr size_t kPadBytes = (size_t)WV * NPW * M * M * sizeof(T);
  const int c_lo = ptr[n0], c_cnt = ptr[n1] - c_lo;
  L.in_lds = kPadBytes + (size_t)c_cnt * (M * M * sizeof(T) + 4) <= (size_t)lds_bytes;
  T* hb_l = reinterpret_cast<T*>(dyn_lds + kPadBytes);
  unsigned* nb_l = reinterpret_cast<unsigned*>(dyn_lds + kPadBytes + (size_t)c_cnt * M * M * sizeof(T));
  if (L.in_lds) {
    const T* src = HB + (size_t)c_lo * M * M;
    for (int e = threadIdx.x; e < c_cnt * M * M; e += kPersistBlock) {
      const int c = e / (M * M), ij = e % (M * M);
      hb_l[c * M * M + (ij % M) * M + ij / M] = src[e];              // transposed: [c][j][i] = H_c[i][j]
    }
    for (int e = threadIdx.x; e < c_cnt; e += kPersistBlock) nb_l[e] = (unsigned)other[c_lo + e];
  }
  L.hb_l = hb_l;
  L.nb_l = nb_l;
  if (threadIdx.x == 0) { sh.bad[0] = 0; sh.bad[1] = 0; }
  __syncthreads();
  L.lbeg = act ? L.beg - c_lo : 0;
  // sh.wave_part[par][qbase + c][w] = sum over this wave's nodes of `val` at component c (as in pcg_ghost_kernel)
  auto wave_comp_sums = [&](T val, int par_, int qbase) {
    T* pad = cz_pad + (CZ ? w * 64 : 0);
    pad[lane] = act ? val : T(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < M) {
      T sum = T(0);
#pragma unroll
      for (int s2 = 0; s2 < NPW; ++s2) sum += pad[s2 * M + lane];
      sh.wave_part[par_][qbase + lane][w] = sum;
    }
    __builtin_amdgcn_wave_barrier();
  };
  // second level of an exchange (XG): this rank's totals of QN quantities to every rank; everybody adds the `world` rows in rank order
  auto rank_exchange = [&](int par_, unsigned tag_, int QN) {
    constexpr int RW2 = SLOTS * NW;
    const size_t tab = (size_t)par_ * kPersistMaxWorld * RW2;
    if (blockIdx.x == 0 && (int)threadIdx.x < QN) {
      for (int rk = 0; rk < peers.world; ++rk)
        put_value<T, true>(peers.rpart[rk] + tab + (size_t)peers.rank * RW2 + threadIdx.x * NW, sh.total[par_][threadIdx.x], tag_);
    }
    __syncthreads();                                             // (everybody has read this rank's totals before they are replaced)
    if ((int)threadIdx.x < QN) {
      const u64* mine = peers.rpart[peers.rank] + tab + threadIdx.x * NW;
      T sum = T(0);
      bool all = true;
      for (int rk = 0; rk < peers.world; ++rk) {
        bool ok = false;
        T v = T(0);
        for (long spin = 0; spin < kPeerSpins && !ok; ++spin) {
          ok = true;
          v = get_value<T, true>(mine + (size_t)rk * RW2, tag_, ok);
          if (!ok) __builtin_amdgcn_s_sleep(1);
        }
        all = all && ok;
        sum += v;
      }
      sh.total[par_][threadIdx.x] = sum;
      if (!all) sh.bad[par_] = 1;
    }
    __syncthreads();
  };
  if constexpr (CZ) {
    // one exchange before the first iteration: E_i = sum_n shift[n, i] and (Z^T r_0)_i over ALL ranks' nodes (table 1, tag = the
    // epoch's base, which no iteration uses: iteration k carries tag0 + k + 1; nobody reaches iteration 1's rows of table 1
    // before everybody has left this exchange, because iteration 0's exchange waits for all)
    wave_comp_sums(act ? shift[n * M + i] : T(0), 1, 0);
    wave_comp_sums(re, 1, M);
    __syncthreads();
    publish_row<T, 2 * M, SH, SLOTS>(sh, 1, part, tag0);
    gather_rows<T, 2 * M, SH, SLOTS>(sh, 1, part, tag0);
    __syncthreads();
    rank_exchange(1, tag0, 2 * M);
    if (threadIdx.x < M) {
      const T e = sh.total[1][threadIdx.x];
      einv[threadIdx.x] = e > T(0) ? T(1) / e : T(0);
    }
    __syncthreads();
  }

  T bn2 = T(0), rr = T(0);
  int k = 0, flag = 0;                     // flag: 1 converged, 2 NaN, 3 a workgroup never arrived, 0 iteration limit
  // cap < 0 (tools/time_pcg_iter.py): thread 0 of the middle workgroup accumulates the wall-clock ticks (10 ns) of every
  // phase of the iteration and leaves them in the last 8 entries of rr_hist
  const bool prof = cap < 0;
  if (prof) cap = -cap;
  const bool clocked = prof && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
  unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, t_prev = clocked ? wall_clock64() : 0;
#define PPLIE_TICK(slot)                                       \
  if (clocked) {                                               \
    const unsigned long long t_now = wall_clock64();           \
    tk[slot] += t_now - t_prev;                                \
    t_prev = t_now;                                            \
  }
  {
    T pe = ze;                             // p_0 = z_0 (published above as hand-off 0)
    if constexpr (CZ) {
      pe += sh.total[1][M + i] * einv[i];  // p_0 = z_0 = Binv r_0 + Z (Z^T r_0 / E)
      __syncthreads();                     // (sh.total[1] is rewritten by iteration 1's exchange: far away, but bad[] is read below)
      hand_off(pe, 0, tag0 + 1u);
    }
    for (;; ++k) {
      const unsigned tag = tag0 + (unsigned)k + 1u;
      const int par = k & 1;
      bool stale = false;
      const T acc = node_matvec<T, M, CH, XG>(L, pe, HB, other, ptag + (size_t)par * NM, tag, stale);   // q = A p
      PPLIE_TICK(0)
      const T bq = node_binv<T, M>(L, acc);
      T v[kPersistQ] = {acc * pe, acc * ze, acc * bq, re * ze, re * re};
      post_wave_sums<T, kPersistQ>(sh, par, v, act, stale);
      if constexpr (CZ) {
        wave_comp_sums(acc, par, kPersistQ);                                       // Z^T q
        wave_comp_sums(re, par, kPersistQ + M);                                    // Z^T r
      }
      PPLIE_TICK(1)
      __syncthreads();                                                           // barrier 1
      PPLIE_TICK(2)
      publish_row<T, NQ, SH, SLOTS>(sh, par, part, tag);
      gather_rows<T, NQ, SH, SLOTS>(sh, par, part, tag);
      PPLIE_TICK(3)
      __syncthreads();                                                           // barrier 2
      PPLIE_TICK(4)
      if constexpr (CZ) {
        rank_exchange(par, tag, NQ);
      } else if (XG) {
        // second level: this rank's totals to every rank; everybody adds the `world` rows in rank order
        constexpr int RW2 = kPersistSlots * NW;
        const size_t tab = (size_t)par * kPersistMaxWorld * RW2;
        if (blockIdx.x == 0 && threadIdx.x < kPersistQ) {
          for (int rk = 0; rk < peers.world; ++rk)
            put_value<T, true>(peers.rpart[rk] + tab + (size_t)peers.rank * RW2 + threadIdx.x * NW, sh.total[par][threadIdx.x], tag);
        }
        __syncthreads();                                             // (everybody has read this rank's totals before they are replaced)
        if (threadIdx.x < kPersistQ) {
          const u64* mine = peers.rpart[peers.rank] + tab + threadIdx.x * NW;
          T sum = T(0);
          bool all = true;
          for (int rk = 0; rk < peers.world; ++rk) {
            bool ok = false;
            T v = T(0);
            for (long spin = 0; spin < kPeerSpins && !ok; ++spin) {
              ok = true;
              v = get_value<T, true>(mine + (size_t)rk * RW2, tag, ok);
              if (!ok) __builtin_amdgcn_s_sleep(1);
            }
            all = all && ok;
            sum += v;
          }
          sh.total[par][threadIdx.x] = sum;
          if (!all) sh.bad[par] = 1;
        }
        __syncthreads();
      }
      const T pq = sh.total[par][0], qz = sh.total[par][1], qmq = sh.total[par][2], rho_loc = sh.total[par][3];
      rr = sh.total[par][4];
      if (sh.bad[par]) { flag = 3; break; }
      if (k == 0) bn2 = rr;
      if (blockIdx.x == 0 && threadIdx.x == 0 && k < cap) rr_hist[k] = rr;
      if (!(rr == rr)) { flag = 2; break; }
      if (rr <= tol2 * bn2) { flag = 1; break; }                     // (also |b| = 0: x = 0 is the answer)
      if (k >= maxiter) break;
      T rho = rho_loc;                                               // (CZ: rho_loc = r.Binv r, the coarse part is added here)
      if constexpr (CZ) {
#pragma unroll
        for (int q = 0; q < M; ++q) { const T sr = sh.total[par][kPersistQ + M + q]; rho += sr * sr * einv[q]; }
      }
      const T alpha = pq > pcg_tiny<T>() ? rho / pq : T(0);                 // p.q = 0 only once r = 0: stay put, no NaN
      T rho_next = rho_loc - T(2) * alpha * qz + alpha * alpha * qmq;
      T cz = T(0);                                                   // this component's coarse part of the new z
      if constexpr (CZ) {
#pragma unroll
        for (int q = 0; q < M; ++q) {
          const T sp = sh.total[par][kPersistQ + M + q] - alpha * sh.total[par][kPersistQ + q];    // Z^T r' = Z^T r - alpha Z^T q
          rho_next += sp * sp * einv[q];
        }
        cz = (sh.total[par][kPersistQ + M + i] - alpha * sh.total[par][kPersistQ + i]) * einv[i];
      }
      if (rho_next < T(0)) rho_next = T(0);
      const T beta = rho > pcg_tiny<T>() ? rho_next / rho : T(0);
      xe += alpha * pe;
      re -= alpha * acc;
      ze = node_binv<T, M>(L, re);
      pe = (CZ ? ze + cz : ze) + beta * pe;
      hand_off(pe, (k + 1) & 1, tag + 1u);
      PPLIE_TICK(5)
    }
  }
#undef PPLIE_TICK
  if (clocked) {
#pragma unroll
    for (int q = 0; q < 6; ++q) rr_hist[cap - 8 + q] = (T)tk[q];
  }
  // a failed solve (NaN, or a workgroup that never arrived) hands back x = 0: the caller may have queued the parameter
  // update behind this launch and look at `info` only afterwards (one read-back per LM trial) -- Exp(0) p = p
  if (act) x[n * M + i] = flag >= 2 ? T(0) : xe;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    info[0] = (T)k; info[1] = rr; info[2] = bn2; info[3] = (T)flag;
    it_out[0] = k;
  }
}

// =====================================================================================================================
// GHOST-ZONE form: ONE grid-wide dependency per iteration.
//
// In the kernel above an iteration waits twice for the whole grid: for the all-gather of the dot products, and -- because with
// loop closures every workgroup borders on nearly every other -- for the neighbours' new search direction before the next
// product (measured: 5.7 + 4.2 of 11.9 us, tools/time_pcg_iter.py).  Here a workgroup also keeps r and p of its GHOST nodes
// (the neighbours it does not own) and advances them itself: what an owner publishes per iteration is q = A p of its nodes,
// at the same moment as its partial sums; a reader fetches its ghosts' q while it polls the all-gather, and then applies the
// very recurrence the owner applies --  r -= alpha q,  z = Binv r,  p = z + beta p  -- to its ghosts, same operands, same
// order, same bits.  The matrix product reads every p from LDS and never waits; the only grid-wide wait left is the
// all-gather, with the q exchange riding in its shadow.
//   lanes: a lane owns (node, component) of one OWNED node and of up to kGhostLayers ghost nodes (layer l, position as the
//   owned one); p of all local nodes (owned, then ghosts) sits in LDS, indexed by the local slot the host precomputed for every
//   incidence (optim/posegraph.py FusedPCG._ghost_map).
constexpr int kGhostLayers = 3;

// CZ ("coarse"): TWO-LEVEL preconditioner  M^-1 = blockdiag(A)^-1 + Z E^-1 Z^T  with Z = 1_N (x) I_M, the GAUGE modes of a pose
// graph of relative-pose edges.  With left perturbations a global motion Exp(xi) is the SAME tangent xi at every node, and
// J[e,0] = -J[e,1] makes J Z = 0 exactly: A Z = shift (.) Z (the damping / clamp part of the diagonal, pplie_pcg_prepare's `shift`)
// and E = Z^T A Z = diag(sum_n shift[n, :]) is DIAGONAL.  Those M directions carry eigenvalues ~ damping (1e-5 of the rest of the
// block-Jacobi-preconditioned spectrum, which starts at 0.06): they are what makes block-Jacobi PCG need 35 / 105 iterations in
// the later LM steps of BASELINE's 10 k-pose graph where the rest of the spectrum needs ~20 (measured on the host in fp64:
// 18 / 20 / 91 -> 18 / 20 / 26, and the solution's gauge component, 5e-3 of |x| with block-Jacobi at tol 1e-4, is exact).
// Cost: M more sums per exchange (Z^T q per component; Z^T r by its recurrence from the set-up exchange's Z^T r_0 -- pcg_persist_kernel
// still sums both, 2 M) and one exchange before the first iteration (E and Z^T r_0).
//   rho = r.Binv r + sum_i (Z^T r)_i^2 / E_i ;   z = Binv r + Z (Z^T r / E) ;  the recurrence value of rho_{k+1} (for beta only,
//   as before) uses Z^T r' = Z^T r - alpha Z^T q.  Any E > 0 gives an SPD preconditioner: the stop test |r| <= tol |b| is unchanged.

// The LM trial's tail in the solve's epilogue (round 6; SE3 pose graphs, M = 6; nodes == nullptr: none).  What the tail's first launch
// did -- pgo_tail_first_kernel, 9.6 us at 10 k nodes and a launch boundary -- needs nothing the workgroups do not hold when the loop ends:
//   gain terms   a = sum_e |J_e d|^2 = d^T H d  and  b = sum_e (J_e d).r_e = d^T g  (strategy.py:144, :261).  With A = H + diag(shift),
//                the solve's right-hand side r_0 = -g and its final residual r = r_0 - A d:   a = d.(r_0 - r) - sum shift d^2,
//                b = -d.r_0 -- node-local products of what every lane has in registers (d, r) or reads back once (r_0, shift),
//                exact up to the rounding of the recurrence residual; one partial pair per workgroup in gain_partial[2 b .. 2 b + 1]
//   retraction   nodes_n <- Exp(d_n) nodes_n, the old row to `backup` (lane 0 of each node's M lanes); state[3] counts it
template <class T> struct GhostTail {
  T* nodes;
  T* backup;
  T* gain_partial;
  unsigned long long* state;
  const T* shift;
};
template <class T, int M, bool CZ = false, bool PROF = false>
__global__ void __launch_bounds__(kPersistBlock)
pcg_ghost_kernel(const int* __restrict__ ptr, const int* __restrict__ slot, const T* __restrict__ HB, const T* __restrict__ D,
                 const T* __restrict__ Binv, T* __restrict__ x, const T* __restrict__ r, const T* __restrict__ z,
                 const int* __restrict__ gptr, const int* __restrict__ gids, u64* part, u64* qtag /* [2][N * M tagged values] */,
                 T* __restrict__ rr_hist, T* info, int* it_out, T tol2, int maxiter, int cap, int64_t N, int lds_bytes,
                 const T* __restrict__ shift = nullptr, const GhostTail<T>* __restrict__ tail_p = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  // cap < 0 (tools/time_pcg_iter.py): thread 0 of two workgroups -- the middle one (a plain member of the two-level exchange) and
  // workgroup 0 (a group leader) -- accumulates the wall-clock ticks (10 ns) of the phases; the clock starts with the kernel
  // (PROF: an instantiation of its own -- the production kernel carries none of this, not even the flag's SGPRs)
  if (cap < 0) cap = -cap;
  const bool clocked = PROF && (blockIdx.x == gridDim.x / 2 || blockIdx.x == 0) && threadIdx.x == 0;
  if (clocked) {
    unsigned* tk = tick_store();
    for (int q = 0; q < kTickSlots; ++q) tk[q] = 0u;
    tk[kTickSlots] = (unsigned)wall_clock64();
    tk[kTickSlots + 1] = tk[kTickSlots];                         // (the kernel's start, kept)
  }
  // (round 6) the exchange carries Z^T q only: Z^T r follows the recurrence Z^T r' = Z^T r - alpha Z^T q that beta already used -- every
  // workgroup applies it to the same gathered operands in the same order, so all hold the same bits -- from the set-up exchange's
  // Z^T r_0.  Six sums, six wave-level reductions and three of nine 64-bit words per table row less per iteration (M = 6).
  constexpr int NQ = CZ ? kPersistQ + M : kPersistQ;
  constexpr int SLOTS = CZ ? kCoarseSlots : kPersistSlots;
  typedef PersistShared<T, (CZ ? kPersistQ + 2 * M : kPersistQ)> SH;             // (the set-up exchange has 2 M quantities)
  __shared__ SH sh;
  __shared__ T zr_sh[2][CZ ? M : 1];                                  // Z^T r of the iteration (parity k & 1)
  // the iteration's scalars worked out ONCE, by the wave that gathers the second level of the exchange (fp32 two-level grids): what was
  // ~27 LDS reads and ~70 VALU instructions in each of the 16 waves of an issue-bound iteration is three reads behind barrier 2
  constexpr bool ONCE = CZ && sizeof(T) == 4;
  __shared__ T ab_sh[2][2], czv_sh[2][CZ ? M : 1];
  const bool hier = ONCE && (int)gridDim.x >= kHierMinGrid;
  // CZ: per-component sums over a wave's nodes go through a 64-element pad per wave (M lanes add NPW values each: two LDS round
  // trips) instead of M masked wave reductions per quantity
  __shared__ T cz_pad[CZ ? kPersistBlock : 1];
  constexpr int NPW = 64 / M, WV = kPersistBlock / 64, NW = sizeof(T) / 4, POS = WV * NPW;      // POS: node positions per layer
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane / M, i = lane % M;
  const int64_t n0 = N * blockIdx.x / gridDim.x, n1 = N * (blockIdx.x + 1) / gridDim.x;
  const int n_own = (int)(n1 - n0);
  const int pos = w * NPW + sub;                                   // this lane's node position in every layer
  const int64_t n = n0 + pos;
  const bool act = sub < NPW && pos < n_own;
  const size_t NM = (size_t)N * M * NW;
  const int g_lo = gptr[blockIdx.x], n_ghost = gptr[blockIdx.x + 1] - g_lo;
  const int c_lo = ptr[n0], c_cnt = ptr[n1] - c_lo;                // the slice's incidences

  // ---- owned element
  // (the node's row of D is not kept in registers: the owned nodes' diagonal blocks are staged behind the off-diagonal slice as one
  //  more "incidence" per node -- block D_n, far end the node itself -- so the diagonal term is formed by pass 1 of the SpMV with
  //  all 16 waves and six VGPRs per lane are free: this kernel sits at its 128-VGPR cap)
  T br[M], xe = T(0), re = T(0), ze = T(0), pe = T(0);
  int beg = 0, deg = 0;
#pragma unroll
  for (int j = 0; j < M; ++j) br[j] = T(0);
  if (act) {
#pragma unroll
    for (int j = 0; j < M; ++j) br[j] = Binv[(n * M + i) * M + j];
    re = r[n * M + i];
    ze = z[n * M + i];
    pe = ze;
    beg = ptr[n];
    deg = ptr[n + 1] - beg;
  }
  // ---- ghost elements: layer l holds ghost g_lo + l * POS + pos
  T gbr[kGhostLayers][M], gr[kGhostLayers], gp[kGhostLayers];
  int gnode[kGhostLayers];
  bool gact[kGhostLayers];
  // (the set-up's loads in LEVELS of dependency, issued level by level: [gptr, ptr of the slice and of the node, the owned rows] ->
  //  [the ghosts' ids, the block slice into LDS] -> [the ghosts' rows].  In program order -- owned, ghost ids, ghost rows, THEN the slice's
  //  bounds and blocks -- four dependent memory latencies followed each other: 6.8 us before the set-up exchange, tools/time_pcg_iter.py)
#pragma unroll
  for (int l = 0; l < kGhostLayers; ++l) {
    const int gi = l * POS + pos;
    gact[l] = sub < NPW && gi < n_ghost;
    gnode[l] = gact[l] ? gids[g_lo + gi] : 0;
    gr[l] = T(0);
    gp[l] = T(0);
  }
  // WAVE-UNIFORM activity (round 6): with 39 owned nodes of 160 positions only waves 0..3 of 16 hold owned elements, and the last ghost
  // layer is usually empty -- but the iteration below is issue-bound (4 waves per SIMD share its VALU slots), and branch-free code
  // runs the reductions, the 6-lane shuffles and the updates of all of them.  A wave without a single active lane in a layer skips
  // that layer's work on a scalar branch; its partial sums stay the zeros written here.
  const int ws = __builtin_amdgcn_readfirstlane(w);
  const bool wact = ws * NPW < n_own;
  bool wgl[kGhostLayers];
#pragma unroll
  for (int l = 0; l < kGhostLayers; ++l) wgl[l] = l * POS + ws * NPW < n_ghost;
  if (!wact && lane < (int)(sizeof(sh.wave_part[0]) / sizeof(sh.wave_part[0][0]))) {
    sh.wave_part[0][lane][w] = T(0);
    sh.wave_part[1][lane][w] = T(0);
  }
  // ---- LDS: [ p of the local nodes (owned, then ghost layers) | staged blocks, row-major | y = H_c p per incidence | local slots ]
  constexpr size_t kPBytes = (size_t)(1 + kGhostLayers) * POS * M * sizeof(T);
  T* p_l = reinterpret_cast<T*>(dyn_lds);
  const int c_all = c_cnt + n_own;                                 // staged blocks: the slice's incidences, then the owned nodes' D
  T* hb_l = reinterpret_cast<T*>(dyn_lds + kPBytes);
  T* yb = hb_l + (size_t)c_all * M * M;
  int* sl_l = reinterpret_cast<int*>(yb + (size_t)c_all * M);
  {                                                                // (host guarantees the slice fits: see pcg_ghost())
    const T* src = HB + (size_t)c_lo * M * M;
    const T* dsrc = D + (size_t)n0 * M * M;
    constexpr int VE = 16 / sizeof(T);                             // elements of a 16-byte word
    if constexpr ((M * M) % VE == 0) {
      // (blocks are M * M * sizeof(T) = a multiple of 16 bytes from 16-byte aligned bases: a quarter of the copy's instructions)
      typedef T V16 __attribute__((ext_vector_type(VE)));
      const V16* s4 = reinterpret_cast<const V16*>(src);
      V16* d4 = reinterpret_cast<V16*>(hb_l);
      for (int e = threadIdx.x; e < c_cnt * (M * M / VE); e += kPersistBlock) d4[e] = s4[e];
      const V16* sd4 = reinterpret_cast<const V16*>(dsrc);
      V16* dd4 = reinterpret_cast<V16*>(hb_l + (size_t)c_cnt * M * M);
      for (int e = threadIdx.x; e < n_own * (M * M / VE); e += kPersistBlock) dd4[e] = sd4[e];
    } else {
      for (int e = threadIdx.x; e < c_cnt * M * M; e += kPersistBlock) hb_l[e] = src[e];
      for (int e = threadIdx.x; e < n_own * M * M; e += kPersistBlock) hb_l[(size_t)c_cnt * M * M + e] = dsrc[e];
    }
    for (int e = threadIdx.x; e < c_cnt; e += kPersistBlock) sl_l[e] = slot[c_lo + e];
    for (int e = threadIdx.x; e < n_own; e += kPersistBlock) sl_l[c_cnt + e] = e;
  }
#pragma unroll
  for (int l = 0; l < kGhostLayers; ++l) {
#pragma unroll
    for (int j = 0; j < M; ++j) gbr[l][j] = gact[l] ? Binv[((size_t)gnode[l] * M + i) * M + j] : T(0);
    if (gact[l]) {
      gr[l] = r[(size_t)gnode[l] * M + i];
      gp[l] = z[(size_t)gnode[l] * M + i];                         // p_0 = z_0
    }
  }
  if (threadIdx.x == 0) { sh.bad[0] = 0; sh.bad[1] = 0; }
  tick(clocked, 9);                                            // slot 9: registers loaded, block slice staged (issued)
  __shared__ T einv[CZ ? M : 1];                                      // 1 / E_i (LDS: uniform values, not M registers per lane)
  // sh.wave_part[par][qbase + c][w] = sum over this wave's nodes of `val` at component c (lanes 0..M-1 store)
  auto wave_comp_sums = [&](T val, int par_, int qbase) {
    T* pad = cz_pad + (CZ ? w * 64 : 0);
    pad[lane] = act ? val : T(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < M) {
      T sum = T(0);
#pragma unroll
      for (int s2 = 0; s2 < NPW; ++s2) sum += pad[s2 * M + lane];
      sh.wave_part[par_][qbase + lane][w] = sum;
    }
    __builtin_amdgcn_wave_barrier();                                   // (the pad is rewritten by the next call)
  };
  if constexpr (CZ) {
    // one exchange before the first iteration: E_i = sum_n shift[n, i] and (Z^T r_0)_i, per component (table 1, a tag no iteration
    // uses; nobody reaches iteration 1's row of table 1 before everybody has left this gather: iteration 0's gather waits for all)
    __syncthreads();
    wave_comp_sums(act ? shift[n * M + i] : T(0), 1, 0);
    wave_comp_sums(re, 1, M);
    __syncthreads();
    exchange_two_level<T, 2 * M, SH, SLOTS>(sh, 1, part, 0x7fffffffu, 0);         // (use 0 of table 1)
    __syncthreads();
    if (threadIdx.x < M) {
      const T e = sh.total[1][threadIdx.x];
      einv[threadIdx.x] = e > T(0) ? T(1) / e : T(0);
      zr_sh[0][threadIdx.x] = sh.total[1][M + threadIdx.x];
    }
    __syncthreads();
    const T c0 = sh.total[1][M + i] * einv[i];
    pe += c0;                                                        // p_0 = z_0 = Binv r_0 + Z (Z^T r_0 / E)
#pragma unroll
    for (int l = 0; l < kGhostLayers; ++l) gp[l] += c0;
  }
  tick(clocked, 10);                                           // slot 10: the set-up exchange (CZ)
  if (act) p_l[pos * M + i] = pe;
#pragma unroll
  for (int l = 0; l < kGhostLayers; ++l)
    if (gact[l]) p_l[(n_own + l * POS + pos) * M + i] = gp[l];
  __syncthreads();
  const int lbeg = act ? beg - c_lo : 0;

  T bn2 = T(0), rr = T(0);
  int k = 0, flag = 0;
  // cap < 0 (tools/time_pcg_iter.py): thread 0 of the middle workgroup accumulates the wall-clock ticks (10 ns) of the phases
  tick(clocked, 8);                                            // slot 8: everything before the first iteration
  // (the accumulators live in LDS, not in registers: five 64-bit counters and the previous reading were 12 VGPRs of EVERY lane for the
  //  whole loop -- at this kernel's 128-VGPR cap they pushed twelve of the ghosts' Binv rows into scratch, reloaded one by one inside
  //  every iteration: <float, 6, CZ> spilled 24 VGPRs with them and spills 2 without, profiles/r05/kernel_resources.txt.  32-bit
  //  ticks: differences are taken modulo 2^32 (43 s).)
#define PPLIE_TICK(slot) tick(clocked, slot);
  unsigned t_start = 0u;
  if (clocked) t_start = tick_store()[kTickSlots + 1];
  for (;; ++k) {
    const unsigned tag = (unsigned)k + 1u;
    const int par = k & 1;
    // (profiling runs: when each of the first 64 passes of the middle workgroup began, in 10 ns ticks since the kernel's start)
    if (clocked && blockIdx.x != 0 && k < 64) rr_hist[cap - 128 + k] = (T)((unsigned)wall_clock64() - t_start);
    // ---- q = A p from LDS in two passes.  (1) INCIDENCE-parallel over all 16 waves: lane group g takes incidences g, g + POS, ...
    // and lane i of it forms component i of y_c = H_c p_far(c) (row i of the block, the neighbour's p: both contiguous);
    // (2) the node's lanes add the diagonal term and their node's y in incidence order.  (Was: the node's own 6 lanes walked its
    // incidences -- a chain of dependent LDS reads as long as the largest degree in the wave, on 4 of the 16 waves -- and a
    // transpose through LDS: 1.65 us of the 7.1 us iteration.)
    if (sub < NPW) {
      for (int c = pos; c < c_all; c += POS) {
        const T* h = hb_l + ((size_t)c * M + i) * M;
        const T* pp = p_l + (size_t)sl_l[c] * M;
        T y = T(0);
#pragma unroll
        for (int j = 0; j < M; ++j) y += h[j] * pp[j];
        yb[c * M + i] = y;
      }
    }
    __syncthreads();                                                             // barrier 0: every y is in place
    T acc = T(0);
    if (act) {
      acc = yb[(c_cnt + pos) * M + i];                                           // D_n p_n
#pragma unroll 4
      for (int c = 0; c < deg; ++c) acc += yb[(lbeg + c) * M + i];
    }
    if (act) put_value<T>(qtag + (size_t)par * NM + (size_t)(n * M + i) * NW, acc, tag);      // q of the owned nodes, for their readers
    PPLIE_TICK(0)
    if (wact) {
      T bq = T(0);
#pragma unroll
      for (int j = 0; j < M; ++j) bq += br[j] * __shfl(acc, sub * M + j, 64);
      T v[kPersistQ] = {acc * pe, acc * ze, acc * bq, re * ze, re * re};                     // (ze: the LOCAL part Binv r of z)
      post_wave_sums<T, kPersistQ>(sh, par, v, act, false);
      if constexpr (CZ) {
        wave_comp_sums(acc, par, kPersistQ);                                                 // Z^T q
      }
    }
    __syncthreads();                                                             // barrier 1
    PPLIE_TICK(1)
    if constexpr (!CZ) publish_row<T, NQ, SH, SLOTS>(sh, par, part, tag);
    // ---- the ghosts' q: issued now, needed after the all-gather.  Raw tagged words, UNCONDITIONAL loads (an absent ghost reads
    // node 0's word and is ignored), tags looked at after the exchange: nothing between here and the exchange's own loads waits
    // for these (with `gact[l] ? get_value(..) : 0` every layer's load was followed by its own s_waitcnt).
    u64 gw[kGhostLayers][NW];
    const u64* qt = qtag + (size_t)par * NM;
    const int i0 = sub < NPW ? i : 0;                            // (lanes beyond the last node of the wave read node 0's words)
#pragma unroll
    for (int l = 0; l < kGhostLayers; ++l)
#pragma unroll
      for (int kk = 0; kk < NW; ++kk) gw[l][kk] = wgl[l] ? xwg_load(qt + (gnode[l] * M + i0) * NW + kk) : 0ull;
    if constexpr (CZ) {
      exchange_two_level<T, NQ, SH, SLOTS, ONCE>(sh, par, part, tag, (k >> 1) + par, clocked);   // (table 1's use 0 was the set-up exchange)
      if constexpr (ONCE) {
        if (hier && w == 0) {
          // (wave 0 has just stored the totals itself: a wave's LDS accesses execute in order, no barrier between the stores and these reads)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const T pq1 = sh.total[par][0], qz1 = sh.total[par][1], qmq1 = sh.total[par][2], rho_loc1 = sh.total[par][3];
          T rho1 = rho_loc1;
#pragma unroll
          for (int q = 0; q < M; ++q) { const T sr = zr_sh[par][q]; rho1 += sr * sr * einv[q]; }
          const T alpha1 = pq1 > pcg_tiny<T>() ? rho1 / pq1 : T(0);
          T rho_next1 = rho_loc1 - T(2) * alpha1 * qz1 + alpha1 * alpha1 * qmq1;
#pragma unroll
          for (int q = 0; q < M; ++q) {
            const T sp = zr_sh[par][q] - alpha1 * sh.total[par][kPersistQ + q];
            rho_next1 += sp * sp * einv[q];
          }
          if (rho_next1 < T(0)) rho_next1 = T(0);
          const T beta1 = rho1 > pcg_tiny<T>() ? rho_next1 / rho1 : T(0);
          if (lane == 0) { ab_sh[par][0] = alpha1; ab_sh[par][1] = beta1; }
          if (lane < M) {
            const T zri = zr_sh[par][lane] - alpha1 * sh.total[par][kPersistQ + lane];
            czv_sh[par][lane] = zri * einv[lane];
            zr_sh[par ^ 1][lane] = zri;                            // (read behind barrier 3)
          }
        }
      }
    }
    else gather_rows<T, NQ, SH, SLOTS>(sh, par, part, tag);
    PPLIE_TICK(2)
    bool stale = false;
    for (long spin = 0;; ++spin) {
      unsigned miss = 0u;
#pragma unroll
      for (int l = 0; l < kGhostLayers; ++l)
#pragma unroll
        for (int kk = 0; kk < NW; ++kk) miss |= gact[l] ? ((unsigned)(gw[l][kk] >> 32) ^ tag) : 0u;
      if (miss == 0u) break;
      if (spin >= (1L << 20)) { stale = true; break; }
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int l = 0; l < kGhostLayers; ++l)
#pragma unroll
        for (int kk = 0; kk < NW; ++kk)
          if (wgl[l]) gw[l][kk] = xwg_load(qt + (gnode[l] * M + i0) * NW + kk);
    }
    T gq[kGhostLayers];
#pragma unroll
    for (int l = 0; l < kGhostLayers; ++l) {
      unsigned lo[NW];
#pragma unroll
      for (int kk = 0; kk < NW; ++kk) lo[kk] = (unsigned)gw[l][kk];
      T v;
      __builtin_memcpy(&v, lo, sizeof(T));
      gq[l] = gact[l] && !stale ? v : T(0);
    }
    if (stale) sh.bad[par] = 1;
    __syncthreads();                                                             // barrier 2
    PPLIE_TICK(3)
    rr = sh.total[par][4];
    if (sh.bad[par]) { flag = 3; break; }
    if (k == 0) bn2 = rr;
    if (blockIdx.x == 0 && threadIdx.x == 0 && k < cap) rr_hist[k] = rr;
    if (!(rr == rr)) { flag = 2; break; }
    if (rr <= tol2 * bn2) { flag = 1; break; }
    if (k >= maxiter) break;
    T alpha, beta, cz = T(0);                                          // cz: this component's coarse part of the new z
    if (hier) {
      alpha = ab_sh[par][0];
      beta = ab_sh[par][1];
      cz = czv_sh[par][CZ ? i : 0];
    } else {
      const T pq = sh.total[par][0], qz = sh.total[par][1], qmq = sh.total[par][2], rho_loc = sh.total[par][3];
      T rho = rho_loc;
      if constexpr (CZ) {
#pragma unroll
        for (int q = 0; q < M; ++q) { const T sr = zr_sh[par][q]; rho += sr * sr * einv[q]; }
      }
      alpha = pq > pcg_tiny<T>() ? rho / pq : T(0);
      T rho_next = rho_loc - T(2) * alpha * qz + alpha * alpha * qmq;
      if constexpr (CZ) {
#pragma unroll
        for (int q = 0; q < M; ++q) {
          const T sp = zr_sh[par][q] - alpha * sh.total[par][kPersistQ + q];      // Z^T r' = Z^T r - alpha Z^T q
          rho_next += sp * sp * einv[q];
        }
        const T zri = zr_sh[par][i] - alpha * sh.total[par][kPersistQ + i];
        cz = zri * einv[i];
        if (threadIdx.x < M) zr_sh[par ^ 1][i] = zri;                    // (read again behind barrier 3)
      }
      if (rho_next < T(0)) rho_next = T(0);
      beta = rho > pcg_tiny<T>() ? rho_next / rho : T(0);
    }
    // ---- the same update for the owned element and for the ghosts (identical operands, order and contraction: identical bits)
    if (wact) {
      xe += alpha * pe;
      re -= alpha * acc;
      T zn = T(0);
#pragma unroll
      for (int j = 0; j < M; ++j) zn += br[j] * __shfl(re, sub * M + j, 64);
      ze = zn;
      pe = (CZ ? ze + cz : ze) + beta * pe;
      if (act) p_l[pos * M + i] = pe;
    }
#pragma unroll
    for (int l = 0; l < kGhostLayers; ++l) {
      if (wgl[l]) {
        gr[l] -= alpha * gq[l];
        T zn = T(0);
#pragma unroll
        for (int j = 0; j < M; ++j) zn += gbr[l][j] * __shfl(gr[l], sub * M + j, 64);
        gp[l] = (CZ ? zn + cz : zn) + beta * gp[l];
        if (gact[l]) p_l[(n_own + l * POS + pos) * M + i] = gp[l];
      }
    }
    __syncthreads();                                                             // barrier 3: every local p is in place
    PPLIE_TICK(4)
  }
#undef PPLIE_TICK
  if (clocked) {
    // middle workgroup: the five phase slots at rr_hist[cap - 8 ..] (as before) and all kTickSlots at [cap - 48 ..]; workgroup 0
    // (a group leader of the two-level exchange): all slots at [cap - 32 ..]
    const unsigned* tk = tick_store();
    const bool mid = blockIdx.x == gridDim.x / 2;
    if (mid)
      for (int q = 0; q < 5; ++q) rr_hist[cap - 8 + q] = (T)tk[q];
    if (mid || gridDim.x / 2 != 0)
      for (int q = 0; q < kTickSlots; ++q) rr_hist[cap - (mid ? 48 : 32) + q] = (T)tk[q];
  }
  const T xo = flag >= 2 ? T(0) : xe;
  if (act) x[n * M + i] = xo;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    info[0] = (T)k; info[1] = rr; info[2] = bn2; info[3] = (T)flag;
    it_out[0] = k;
  }
  if constexpr (M == 6) {
    // (the five pointers sit in device memory and are read HERE: as kernel arguments they cost ten SGPRs for the whole loop -- at this
    //  kernel's register cap that spilled two VGPRs)
    if (tail_p) {                                                // (launch-uniform)
      const GhostTail<T> tail = *tail_p;
      T ta = T(0), tb = T(0);
      if (act && flag < 2) {
        const T r0e = r[n * M + i], she = tail.shift[n * M + i];
        ta = xo * (r0e - re - she * xo);
        tb = -xo * r0e;
      }
      ta = wave_total63<T>(ta);
      tb = wave_total63<T>(tb);
      __syncthreads();                                           // (every wave is out of the loop: the exchange's LDS is free)
      if (lane == 63) { sh.wave_part[0][0][w] = ta; sh.wave_part[0][1][w] = tb; }
      __syncthreads();
      if (threadIdx.x == 0) {
        T sa = T(0), sb = T(0);
#pragma unroll
        for (int ww = 0; ww < WV; ++ww) { sa += sh.wave_part[0][0][ww]; sb += sh.wave_part[0][1][ww]; }
        tail.gain_partial[2 * blockIdx.x] = sa;
        tail.gain_partial[2 * blockIdx.x + 1] = sb;
        if (blockIdx.x == 0) tail.state[3] += 1;                 // "the parameters were moved" (pgo_fused.hip pgo_retract_rows)
      }
      T d[7];
#pragma unroll
      for (int j = 0; j < 6; ++j) d[j] = __shfl(xo, (sub * M + j) & 63, 64);
      d[6] = T(0);
      if (act && i == 0) {
        T X[7], Ex[7], out[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) X[q] = tail.nodes[n * 7 + q];
        if (tail.backup) {
#pragma unroll
          for (int q = 0; q < 7; ++q) tail.backup[n * 7 + q] = X[q];
        }
        se3_exp<T>(d, Ex);
        se3_mul<T>(Ex, X, out);
#pragma unroll
        for (int q = 0; q < 7; ++q) tail.nodes[n * 7 + q] = out[q];
      }
    }
  }
}

// Dynamic LDS of a workgroup (the staged matrix slice); the most workgroups of this kernel the device holds at once
// (they spin on each other: all must be resident)
constexpr int kPersistLds = 152 * 1024;      // of the 160 KB a CU has (one 1024-lane workgroup per CU; ~1.5 KB are static)
template <class T, int M, bool XG, bool CZ = false> static int persist_capacity(int& lds_bytes) {
  static int cap[16] = {0}, lds[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (cap[dev] == 0) {
    int cus = 0, per = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    // (the two-level variant's static LDS -- wider exchange rows, the component-sum pad -- is 6.5 KB in fp32, 13 KB in fp64)
    lds[dev] = CZ ? kPersistLds - (sizeof(T) == 8 ? 12 : 6) * 1024 : kPersistLds;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_persist_kernel<T, M, XG, CZ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds[dev]) != hipSuccess) {
      (void)hipGetLastError();
      lds[dev] = 48 * 1024;                                      // (always available without the attribute)
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, pcg_persist_kernel<T, M, XG, CZ>, kPersistBlock, lds[dev]) != hipSuccess) return 0;
    cap[dev] = cus * per > 0 ? cus * per : -1;
  }
  lds_bytes = lds[dev];
  return cap[dev] > 0 ? cap[dev] : 0;
}

template <class T>
int pcg_persist(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x, const void* r, const void* z,
                void* part, void* ptag, void* rr_hist, void* info, void* it, double tol, int maxiter, int cap, int grid, int64_t N, int m,
                void* stream, const PersistPeers* peers, const void* shift = nullptr) {
  if (N <= 0) return N == 0 ? PPLIE_OK : PPLIE_EBADARG;
  if (!ptr || !other || !HB || !D || !Binv || !x || !r || !z || !part || !ptag || !rr_hist || !info || !it) return PPLIE_EBADARG;
  if (grid < 1 || grid > kPersistGridMax || maxiter < 0 || maxiter > 65534 || (shift && !peers)) return PPLIE_EBADARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  PersistPeers none = {};
#define LAUNCH2(MM, XG, CZ)                                                                                                    \
  {                                                                                                                            \
    int lds_bytes = 0;                                                                                                         \
    const int resident = persist_capacity<T, MM, XG, CZ>(lds_bytes);                                                           \
    if (resident <= 0 || (size_t)lds_bytes < (size_t)(kPersistBlock / 64) * (64 / MM) * MM * MM * sizeof(T)) return PPLIE_ECAPACITY; \
    if (grid > resident) grid = resident;                         /* fewer CUs than asked for: every workgroup must be resident */ \
    if (grid > N) grid = (int)N;                                                                                               \
    const int64_t per_wg = (kPersistBlock / 64) * (64 / MM);     /* one lane per (node, component): nodes one workgroup holds */ \
    if ((N + grid - 1) / grid > per_wg) return PPLIE_ECAPACITY;  /* too large for this device: use the two-launch iteration */  \
    hipLaunchKernelGGL((pcg_persist_kernel<T, MM, XG, CZ>), dim3(grid), dim3(kPersistBlock), lds_bytes, st, (const int*)ptr,   \
                       (const int*)other, (const T*)HB, (const T*)D, (const T*)Binv, (T*)x, (const T*)r, (const T*)z,            \
                       (unsigned long long*)part, (unsigned long long*)ptag, (T*)rr_hist, (T*)info, (int*)it, (T)(tol * tol),    \
                       maxiter, cap, N, lds_bytes, XG ? *peers : none, (const T*)shift);                                       \
  }
#define LAUNCH(MM) { if (peers && shift) LAUNCH2(MM, true, true) else if (peers) LAUNCH2(MM, true, false) else LAUNCH2(MM, false, false) }
  if (m == 6) LAUNCH(6) else if (m == 7) LAUNCH(7) else if (m == 3) LAUNCH(3) else return PPLIE_EBADARG;
#undef LAUNCH
#undef LAUNCH2
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

template <class T>
int pcg_persist_p2p(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x, const void* r,
                    const void* z, void* part, const void* const* ptag_peers, const void* const* rpart_peers, void* rr_hist, void* info,
                    void* it, double tol, int maxiter, int cap, int grid, int64_t n_own, int64_t row0, int64_t n_global, int world,
                    int rank, int epoch, int m, void* stream, const void* shift = nullptr) {
  if (world < 1 || world > kPersistMaxWorld || rank < 0 || rank >= world || !ptag_peers || !rpart_peers || row0 < 0 ||
      row0 + n_own > n_global)
    return PPLIE_EBADARG;
  PersistPeers pe = {};
  pe.world = world;
  pe.rank = rank;
  pe.tag_base = ((unsigned)epoch & 0xffffu) << 16;
  pe.row0 = row0;
  pe.n_global = n_global;
  for (int k = 0; k < world; ++k) {
    if (!ptag_peers[k] || !rpart_peers[k]) return PPLIE_EBADARG;
    pe.ptag[k] = (u64*)ptag_peers[k];
    pe.rpart[k] = (u64*)rpart_peers[k];
  }
  return pcg_persist<T>(ptr, other, HB, D, Binv, x, r, z, part, pe.ptag[rank], rr_hist, info, it, tol, maxiter, cap, grid, n_own, m, stream, &pe,
                        shift);
}

// ghost-zone solve: PPLIE_ECAPACITY (nothing launched) when a workgroup's slice or ghost set does not fit -- the caller then uses
// pplie_pcg_persist.  max_cnt / max_ghost: the largest incidence count and ghost count of any workgroup for THIS grid.
template <class T, int M, bool CZ> static int ghost_capacity(int& lds_bytes) {
  static int cap[16] = {0}, lds[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (cap[dev] == 0) {
    int cus = 0, per = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    // (the two-level variant's static LDS -- wider exchange rows, the component-sum pads -- is 6.5 KB in fp32, 13 KB in fp64)
    lds[dev] = (CZ && sizeof(T) == 8) ? kPersistLds - 8 * 1024 : kPersistLds;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_ghost_kernel<T, M, CZ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds[dev]) != hipSuccess) {
      (void)hipGetLastError();
      lds[dev] = 48 * 1024;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, pcg_ghost_kernel<T, M, CZ>, kPersistBlock, lds[dev]) != hipSuccess) return 0;
    cap[dev] = cus * per > 0 ? cus * per : -1;
  }
  lds_bytes = lds[dev];
  return cap[dev] > 0 ? cap[dev] : 0;
}

template <class T>
int pcg_ghost(const void* ptr, const void* slot, const void* HB, const void* D, const void* Binv, void* x, const void* r, const void* z,
              const void* gptr, const void* gids, void* part, void* qtag, void* rr_hist, void* info, void* it, double tol, int maxiter,
              int cap, int grid, int max_cnt, int max_ghost, int64_t N, int m, void* stream, const void* shift = nullptr,
              const GhostTail<T>* tail_in = nullptr) {
  if (N <= 0) return N == 0 ? PPLIE_OK : PPLIE_EBADARG;
  if (!ptr || !slot || !HB || !D || !Binv || !x || !r || !z || !gptr || !gids || !part || !qtag || !rr_hist || !info || !it) return PPLIE_EBADARG;
  if (grid < 1 || grid > kPersistGridMax || maxiter < 0 || max_cnt < 0 || max_ghost < 0) return PPLIE_EBADARG;
  if (tail_in && m != 6) return PPLIE_EBADARG;                  // (tail_in: DEVICE memory, five pointers -- see pplie_pcg_ghost_tail)
  const GhostTail<T>* tail = tail_in;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define LAUNCH(MM) { if (shift) LAUNCH2(MM, true) else LAUNCH2(MM, false) }
#define LAUNCH2(MM, CZ)                                                                                                        \
  {                                                                                                                            \
    int lds_bytes = 0;                                                                                                         \
    const int resident = ghost_capacity<T, MM, CZ>(lds_bytes);                                                                 \
    constexpr int POS = (kPersistBlock / 64) * (64 / MM);                                                                      \
    const size_t per = (size_t)((N + grid - 1) / grid);       /* (+ the owned nodes' diagonal blocks, staged behind the slice) */ \
    const size_t need = (size_t)(1 + kGhostLayers) * POS * MM * sizeof(T) +                                                    \
                        ((size_t)max_cnt + per) * (MM * MM * sizeof(T) + MM * sizeof(T) + 4);                                  \
    if (resident < grid || need > (size_t)lds_bytes || max_ghost > kGhostLayers * POS || (N + grid - 1) / grid > POS)           \
      return PPLIE_ECAPACITY;             /* (the grid is part of the host's ghost map: it cannot be shrunk here) */             \
    if (cap < 0 && MM == 6 && sizeof(T) == 4) {   /* (the profiling build exists for the metric's shape only: fp32, m = 6) */   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_ghost_kernel<float, 6, CZ, true>),                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return PPLIE_ELAUNCH;      \
      hipLaunchKernelGGL((pcg_ghost_kernel<float, 6, CZ, true>), dim3(grid), dim3(kPersistBlock), lds_bytes, st, (const int*)ptr, \
                         (const int*)slot, (const float*)HB, (const float*)D, (const float*)Binv, (float*)x, (const float*)r,     \
                         (const float*)z, (const int*)gptr, (const int*)gids, (unsigned long long*)part, (unsigned long long*)qtag, \
                           (float*)rr_hist, (float*)info, (int*)it, (float)(tol * tol), maxiter, cap, N, lds_bytes, (const float*)shift); \
    } else                                                                                                                     \
    hipLaunchKernelGGL((pcg_ghost_kernel<T, MM, CZ>), dim3(grid), dim3(kPersistBlock), lds_bytes, st, (const int*)ptr, (const int*)slot, \
                       (const T*)HB, (const T*)D, (const T*)Binv, (T*)x, (const T*)r, (const T*)z, (const int*)gptr,             \
                       (const int*)gids, (unsigned long long*)part, (unsigned long long*)qtag, (T*)rr_hist, (T*)info, (int*)it,   \
                       (T)(tol * tol), maxiter, cap, N, lds_bytes, (const T*)shift, tail);                                     \
  }
  if (m == 6) LAUNCH(6) else if (m == 7) LAUNCH(7) else if (m == 3) LAUNCH(3) else return PPLIE_EBADARG;
#undef LAUNCH
#undef LAUNCH2
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace pplie

extern "C" int pplie_pcg_persist_f32(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x,
                                     void* r, void* p, void* q, void* z, void* part, void* ptag, void* rr_hist, void* info, void* it,
                                     double tol, int maxiter, int cap, int grid, int64_t N, int m, void* stream) {
  (void)p; (void)q;                                              // (round-2 signature: p and q now live in registers)
  return pplie::pcg_persist<float>(ptr, other, HB, D, Binv, x, r, z, part, ptag, rr_hist, info, it, tol, maxiter, cap, grid, N, m, stream, nullptr);
}
extern "C" int pplie_pcg_persist_f64(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, void* x,
                                     void* r, void* p, void* q, void* z, void* part, void* ptag, void* rr_hist, void* info, void* it,
                                     double tol, int maxiter, int cap, int grid, int64_t N, int m, void* stream) {
  (void)p; (void)q;
  return pplie::pcg_persist<double>(ptr, other, HB, D, Binv, x, r, z, part, ptag, rr_hist, info, it, tol, maxiter, cap, grid, N, m, stream, nullptr);
}
#define PPLIE_P2P(SFX, T)                                                                                                         \
  extern "C" int pplie_pcg_persist_p2p_##SFX(const void* ptr, const void* other, const void* HB, const void* D, const void* Binv, \
                                             void* x, const void* r, const void* z, void* part, const void* const* ptag_peers,   \
                                             const void* const* rpart_peers, void* rr_hist, void* info, void* it, double tol,   \
                                             int maxiter, int cap, int grid, int64_t n_own, int64_t row0, int64_t n_global,     \
                                             int world, int rank, int epoch, int m, void* stream) {                             \
    return pplie::pcg_persist_p2p<T>(ptr, other, HB, D, Binv, x, r, z, part, ptag_peers, rpart_peers, rr_hist, info, it, tol,     \
                                     maxiter, cap, grid, n_own, row0, n_global, world, rank, epoch, m, stream);                  \
  }
PPLIE_P2P(f32, float)
PPLIE_P2P(f64, double)
// the same solve with the two-level (block-Jacobi + gauge modes) preconditioner: `shift` = pplie_pcg_prepare's output for the owned
// rows [n_own, m]; `part` must hold 2 x PPLIE_PCG_PERSIST_GRID x PPLIE_PCG_COARSE_SLOTS tagged values and every rank's `rpart` table
// 2 x 8 x PPLIE_PCG_COARSE_SLOTS (rows of COARSE_SLOTS values instead of PERSIST_SLOTS)
#define PPLIE_P2P_CZ(SFX, T)                                                                                                      \
  extern "C" int pplie_pcg_persist_p2p_coarse_##SFX(const void* ptr, const void* other, const void* HB, const void* D,            \
                                                    const void* Binv, const void* shift, void* x, const void* r, const void* z,   \
                                                    void* part, const void* const* ptag_peers, const void* const* rpart_peers,   \
                                                    void* rr_hist, void* info, void* it, double tol, int maxiter, int cap,        \
                                                    int grid, int64_t n_own, int64_t row0, int64_t n_global, int world, int rank, \
                                                    int epoch, int m, void* stream) {                                            \
    if (!shift) return pplie::PPLIE_EBADARG;                                                                                      \
    return pplie::pcg_persist_p2p<T>(ptr, other, HB, D, Binv, x, r, z, part, ptag_peers, rpart_peers, rr_hist, info, it, tol,     \
                                     maxiter, cap, grid, n_own, row0, n_global, world, rank, epoch, m, stream, shift);           \
  }
PPLIE_P2P_CZ(f32, float)
PPLIE_P2P_CZ(f64, double)
#define PPLIE_GHOST(SFX, T)                                                                                                       \
  extern "C" int pplie_pcg_ghost_##SFX(const void* ptr, const void* slot, const void* HB, const void* D, const void* Binv, void* x,  \
                                       const void* r, const void* z, const void* gptr, const void* gids, void* part, void* qtag,     \
                                       void* rr_hist, void* info, void* it, double tol, int maxiter, int cap, int grid, int max_cnt, \
                                       int max_ghost, int64_t N, int m, void* stream) {                                             \
    return pplie::pcg_ghost<T>(ptr, slot, HB, D, Binv, x, r, z, gptr, gids, part, qtag, rr_hist, info, it, tol, maxiter, cap, grid,    \
                               max_cnt, max_ghost, N, m, stream);                                                                   \
  }
PPLIE_GHOST(f32, float)
PPLIE_GHOST(f64, double)
// the same solve with the two-level (block-Jacobi + gauge modes) preconditioner: `shift` = pplie_pcg_prepare's output [N, m];
// `part` must hold 2 x PPLIE_PCG_PERSIST_GRID x PPLIE_PCG_COARSE_SLOTS tagged values (zeroed by the caller like the narrow table)
#define PPLIE_GHOST_CZ(SFX, T)                                                                                                    \
  extern "C" int pplie_pcg_ghost_coarse_##SFX(const void* ptr, const void* slot, const void* HB, const void* D, const void* Binv,  \
                                              const void* shift, void* x, const void* r, const void* z, const void* gptr,        \
                                              const void* gids, void* part, void* qtag, void* rr_hist, void* info, void* it,      \
                                              double tol, int maxiter, int cap, int grid, int max_cnt, int max_ghost, int64_t N,  \
                                              int m, void* stream) {                                                             \
    if (!shift) return pplie::PPLIE_EBADARG;                                                                                      \
    return pplie::pcg_ghost<T>(ptr, slot, HB, D, Binv, x, r, z, gptr, gids, part, qtag, rr_hist, info, it, tol, maxiter, cap, grid,    \
                               max_cnt, max_ghost, N, m, stream, shift);                                                            \
  }
PPLIE_GHOST_CZ(f32, float)
PPLIE_GHOST_CZ(f64, double)
// the solve with the LM trial's tail in its epilogue (GhostTail above; m = 6): shift_cz = the two-level preconditioner's shift or NULL
// (block-Jacobi); tail_args = FIVE pointers in DEVICE memory { nodes [N, 7], backup [N, 7] or 0, gain_partial [2 x grid] (the gain
// partials of pplie_pgo_trial_tail's `partial`, i.e. partial + PPLIE_PGO_PARTIALS), state (that entry's state block), shift [N, 6]
// (the damping shift of pplie_pcg_prepare) }
#define PPLIE_GHOST_TAIL(SFX, T)                                                                                                  \
  extern "C" int pplie_pcg_ghost_tail_##SFX(const void* ptr, const void* slot, const void* HB, const void* D, const void* Binv,    \
                                            const void* shift_cz, void* x, const void* r, const void* z, const void* gptr,        \
                                            const void* gids, void* part, void* qtag, void* rr_hist, void* info, void* it,        \
                                            double tol, int maxiter, int cap, int grid, int max_cnt, int max_ghost, int64_t N,    \
                                            int m, const void* tail_args, void* stream) {                                        \
    if (!tail_args || (reinterpret_cast<uintptr_t>(tail_args) & 7)) return pplie::PPLIE_EBADARG;                                  \
    return pplie::pcg_ghost<T>(ptr, slot, HB, D, Binv, x, r, z, gptr, gids, part, qtag, rr_hist, info, it, tol, maxiter, cap, grid,    \
                               max_cnt, max_ghost, N, m, stream, shift_cz, reinterpret_cast<const pplie::GhostTail<T>*>(tail_args)); \
  }
PPLIE_GHOST_TAIL(f32, float)
PPLIE_GHOST_TAIL(f64, double)

// Peer access for the hand-off tables of the multi-GPU solve: the kernel of GPU `device` stores into (and polls) tables that
// live in the memory of GPU `peer` (hipIpc-mapped by the caller).  Returns 0 when `device` can reach `peer`'s memory after the
// call (already enabled counts), PPLIE_ECAPACITY (-3) when the hardware offers no peer path -- the caller then keeps the
// RCCL exchange --, PPLIE_ELAUNCH (-2) on any other runtime error.  The calling thread's current device is left as it was.
extern "C" int pplie_enable_peer_access(int device, int peer) {
  using namespace pplie;
  if (device < 0 || peer < 0) return PPLIE_EBADARG;
  if (device == peer) return PPLIE_OK;
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess) { (void)hipGetLastError(); return PPLIE_ELAUNCH; }
  if (!can) return PPLIE_ECAPACITY;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return PPLIE_ELAUNCH; }
  int rc = PPLIE_OK;
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return PPLIE_ELAUNCH; }
  const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
  if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = PPLIE_ELAUNCH;
  (void)hipGetLastError();
  if (cur >= 0) (void)hipSetDevice(cur);
  return rc;
}
