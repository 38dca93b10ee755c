// pgo_fused.hip -- the pose-graph residual program of the reference (examples/module/pgo/pgo.py:15-25)
//
//     node1, node2 = nodes[edges[..., 0]], nodes[edges[..., 1]]
//     r_e = Log(Z_e^-1 * node1^-1 * node2)                         Z = the measured relative poses
//
// linearised per edge in one kernel.  Chaining the reference's backward rules (operation.py:385-395 SE3_Log,
// :905-908 SE3_Mul: Y_grad = g @ Adj(X), :992-998 SE3_Inv: X_grad = -g @ Adj(Y)) gives, with T = Z^-1 node1^-1,
//     d r / d node2 = Jl_inv(r) Adj(T)            d r / d node1 = -Jl_inv(r) Adj(T)
// (left tangents).  With Jl_inv(r) = [[Ji, Mq], [0, Ji]] (operation.py:68-75) and Adj(T) = [[R, tx R], [0, R]]
// (:202-210) the product is [[Ji R, Ji tx R + Mq R], [0, Ji R]]: three 3x3 products per edge.
// Algorithmic bytes per edge: 16 idx + 28 Z + 2 * 28 gathered nodes read,
// 24 residual + 2 * 144 Jacobian written = 412 B; the autograd route makes 6 backward sweeps through five ops.
#include "rowmap.h"
#include "robust.h"
#include "gridsync.h"

extern "C" int pplie_graph_gain_terms_f32(const void* J, const void* idx, const void* d, int ld, const void* R, void* partial,
                                          int64_t E, int dr, int m, int k, void* stream);      // csrc/graph.hip
extern "C" int pplie_graph_gain_terms_f64(const void* J, const void* idx, const void* d, int ld, const void* R, void* partial,
                                          int64_t E, int dr, int m, int k, void* stream);

namespace pplie {

// r = Log(Z^-1 n1^-1 n2);  T = Z^-1 n1^-1
template <class T> __device__ __forceinline__ void pgo_residual(const T* z, const T* n1, const T* n2, T* Tm, T* r) {
  T zi[7], n1i[7], u[7];
  se3_inv<T>(z, zi);
  se3_inv<T>(n1, n1i);
  se3_mul<T>(zi, n1i, Tm);
  se3_mul<T>(Tm, n2, u);
  se3_log<T>(u, r);
}

// LAP != 0: the edge's share of the normal equations leaves the same kernel (round 6; was pplie_graph_assemble_lap's first launch,
// csrc/graph.hip lap_blocks_kernel, which read the J blocks this kernel had just written back from memory -- 288 B per edge -- and
// formed every S_e twice, once per incidence).  With J_e0 = -J_e1 = -Jm every block of H touched by edge e is +-S_e, S_e = Jm^T Jm
// (symmetric; Jm = [[A, B], [0, A]] makes it 81 FMAs), and the gradient shares are -+Jm^T r_e: -S_e goes to BOTH incidence slots
// inc[e, 0], inc[e, 1] of HB (incidence = position of (edge, side) in the node-sorted CSR list; LAP = 1: full [6, 6] blocks,
// LAP = 2: packed upper triangles [21]), -+Jm^T r to the same slots of gg.  Same products in the same order as lap_blocks_kernel
// (zeros of Jm's lower-left block skipped).  Per edge: 24 B (R) + 288 B (J) + 2 x (84 | 144) B + 2 x 24 B written, nothing re-read.
// THREADS: lanes of the workgroup (a multiple of BLOCK = the edges of a tile): lanes beyond BLOCK compute nothing and carry their share of
// the copies between LDS and memory (the LAP builds run two waves: the tile's 4608 + 4608 words of J and HB leave in half the trips)
template <class T, int BLOCK, int LAP = 0, int THREADS = BLOCK>
__global__ void __launch_bounds__(THREADS)
pgo_linearize_kernel(const T* __restrict__ nodes, const int64_t* __restrict__ idx, const T* __restrict__ Z,
                     T* __restrict__ R, T* __restrict__ J, int64_t E, RobustParam<T> rk, const int* __restrict__ inc = nullptr,
                     T* __restrict__ HB = nullptr, T* __restrict__ gg = nullptr, unsigned long long* __restrict__ ctl = nullptr,
                     int64_t ctl_words = 0, const double* __restrict__ s_src = nullptr, double* __restrict__ s_dst = nullptr) {
  __shared__ __attribute__((aligned(16))) T lds[BLOCK * 72];
  __shared__ int inc_s[LAP ? BLOCK * 2 : 1];
  constexpr int SW = 27;                                          // staged per edge: 21 entries of S's upper triangle + 6 of Jm^T r
  if constexpr (LAP != 0) {
    // what pplie_pcg_begin did in a launch of its own, 5 us in front of every solve: clear the coming solve's control block and bring
    // the damping factor of the day from host-pinned memory (system-scope load) into the device scalar its set-up launch reads
    if (ctl) {
      const bool fetch = s_src && blockIdx.x == 0 && threadIdx.x == 0;
      double sv = 0.0;
      if (fetch) sv = __hip_atomic_load(s_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      for (int64_t k = (int64_t)blockIdx.x * THREADS + threadIdx.x; k < ctl_words; k += (int64_t)gridDim.x * THREADS) ctl[k] = 0ull;
      if (fetch) *s_dst = sv;
    }
  }
  const int64_t ntiles = (E + BLOCK - 1) / BLOCK;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t e0 = tile * BLOCK;
    const int64_t left = E - e0;
    const bool full = left >= BLOCK;
    const int rows = full ? BLOCK : (int)left;
    slab_g2s<T, THREADS, BLOCK * 7, true>(Z + e0 * 7, lds, rows * 7, full);
    __syncthreads();
    const int t = threadIdx.x;
    T r[6], Jm[36];
    if (t < rows) {
      const int64_t e = e0 + t;
      const int64_t i0 = idx[e * 2], i1 = idx[e * 2 + 1];
      T z[7], n1[7], n2[7], Tm[7];
      row_ld<7>(lds + t * 7, z);
#pragma unroll
      for (int k = 0; k < 7; ++k) { n1[k] = nodes[i0 * 7 + k]; n2[k] = nodes[i1 * 7 + k]; }
      pgo_residual<T>(z, n1, n2, Tm, r);
      V3<T> tau = v3(r), phi = v3(r + 3);
      const T th2 = norm2(phi);
      const RotCoef<T> kc = rot_coef(th2);
      const T F = rot_coef_F(th2);
      const V3<T> tt = v3(Tm), qv = v3(Tm + 3);
      const T qw = Tm[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        V3<T> e3 = v3<T>(c == 0 ? T(1) : T(0), c == 1 ? T(1) : T(0), c == 2 ? T(1) : T(0));
        V3<T> rc = adj_rotate(qv, qw, e3);                        // column c of R
        V3<T> a = jlinv_apply(F, phi, rc);                        // column c of Ji R
        V3<T> txr = cross(tt, rc);                                // column c of tx R
        V3<T> b = jlinv_apply(F, phi, txr - q_apply(kc, tau, phi, a));   // Ji tx R + Mq R, Mq = -Ji Q Ji
        Jm[0 * 6 + c] = a.x; Jm[1 * 6 + c] = a.y; Jm[2 * 6 + c] = a.z;
        Jm[3 * 6 + c] = T(0); Jm[4 * 6 + c] = T(0); Jm[5 * 6 + c] = T(0);
        Jm[0 * 6 + 3 + c] = b.x; Jm[1 * 6 + 3 + c] = b.y; Jm[2 * 6 + 3 + c] = b.z;
        Jm[3 * 6 + 3 + c] = a.x; Jm[4 * 6 + 3 + c] = a.y; Jm[5 * 6 + 3 + c] = a.z;
      }
      if (rk.kind != RK_NONE) {
        // robust kernel on the edge: residual and both blocks scaled by sqrt(rho'(|r|^2)) in registers (corrector.py:91-96)
        const T sc = robust_row_scale<T, 6>(rk, r);
#pragma unroll
        for (int k = 0; k < 6; ++k) r[k] *= sc;
#pragma unroll
        for (int k = 0; k < 36; ++k) Jm[k] *= sc;
      }
    }
    __syncthreads();                                              // every lane has read its Z row
    if (t < rows) row_st<6>(lds + t * 6, r);
    __syncthreads();
    slab_s2g<T, THREADS, BLOCK * 6, true>(lds, R + e0 * 6, rows * 6, full);
    __syncthreads();
    if (t < rows) {
#pragma unroll
      for (int k = 0; k < 36; ++k) { lds[t * 72 + k] = -Jm[k]; lds[t * 72 + 36 + k] = Jm[k]; }
    }
    __syncthreads();
    slab_s2g<T, THREADS, BLOCK * 72, true>(lds, J + e0 * 72, rows * 72, full);
    __syncthreads();
    if constexpr (LAP != 0) {
      if (t < rows) {
        T* st = lds + t * SW;                                     // (stride 27 words: conflict-free)
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int b = i; b < 6; ++b) {
            T a = T(0);
#pragma unroll
            for (int l = 0; l < 6; ++l)
              if (l < 3 || i >= 3) a += Jm[l * 6 + i] * Jm[l * 6 + b];     // (rows 3..5 of Jm are zero in columns 0..2)
            st[k++] = a;
          }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          T a = T(0);
#pragma unroll
          for (int l = 0; l < 6; ++l)
            if (l < 3 || i >= 3) a += Jm[l * 6 + i] * r[l];
          st[21 + i] = a;
        }
      }
      for (int q = t; q < rows * 2; q += THREADS) inc_s[q] = inc[e0 * 2 + q];
      __syncthreads();
      constexpr int RW = LAP == 2 ? 21 : 36;
      for (int q = t; q < rows * 2 * RW; q += THREADS) {            // consecutive lanes write consecutive words of an incidence's row
        const int row = q / RW, k = q - row * RW;
        int src = k;
        if constexpr (LAP == 1) {
          const int i = k / 6, b = k - i * 6;
          const int lo = i < b ? i : b, hi = i < b ? b : i;
          src = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
        }
        HB[(int64_t)inc_s[row] * RW + k] = -lds[(row >> 1) * SW + src];
      }
      for (int q = t; q < rows * 12; q += THREADS) {
        const int row = q / 6, k = q - row * 6;
        const T v = lds[(row >> 1) * SW + 21 + k];
        gg[(int64_t)inc_s[row] * 6 + k] = (row & 1) ? v : -v;     // (side 0: J_c = -Jm)
      }
      __syncthreads();
    }
  }
}

// residuals only + per-workgroup partial sums of |r|^2 (the Trivial-kernel loss, optimizer.py:118-125)
template <class T>
__device__ __forceinline__ void pgo_trial_pack(const T* partial, int nparts, const T* pcg_info, unsigned long long* state, double* out,
                                               bool coherent_loss, int ngain = -1);
// PACKLAST (the trial tail's second launch): the workgroup that arrives last at the ticket state[4] also does what a fourth launch
// did -- sums everybody's partials and reports (pgo_trial_pack).  The loss partials cross workgroups inside this launch: agent-scope
// stores and loads (csrc/gridsync.h xwg_*), the ticket's increment a release.
// PACKLAST: 0 = partial sums only; 1 = the last workgroup to arrive packs the trial's report (pgo_trial_pack); 2 = the last workgroup
// adds the partials (one wavefront, fixed order, in double) and stores the loss as ONE device scalar -- the model's loss in one launch
// (pplie_pgo_loss; was: a zero fill, this kernel and a torch reduction, three launches in front of a run's first LM trial)
template <class T, int BLOCK, int PACKLAST = 0>
__global__ void __launch_bounds__(BLOCK)
pgo_residual_kernel(const T* __restrict__ nodes, const int64_t* __restrict__ idx, const T* __restrict__ Z,
                    T* __restrict__ R /* or null */, T* __restrict__ partial /* [gridDim.x] */, int64_t E, RobustParam<T> rk,
                    const T* __restrict__ pcg_info = nullptr, unsigned long long* state = nullptr, double* out = nullptr, int ngain = -1,
                    T* __restrict__ loss_out = nullptr) {
  __shared__ __attribute__((aligned(16))) T lds[BLOCK * 7];
  T acc = T(0);
  const int64_t ntiles = (E + BLOCK - 1) / BLOCK;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t e0 = tile * BLOCK;
    const int64_t left = E - e0;
    const bool full = left >= BLOCK;
    const int rows = full ? BLOCK : (int)left;
    slab_g2s<T, BLOCK, BLOCK * 7, true>(Z + e0 * 7, lds, rows * 7, full);
    __syncthreads();
    const int t = threadIdx.x;
    T r[6];
    if (t < rows) {
      const int64_t e = e0 + t;
      const int64_t i0 = idx[e * 2], i1 = idx[e * 2 + 1];
      T z[7], n1[7], n2[7], Tm[7];
      row_ld<7>(lds + t * 7, z);
#pragma unroll
      for (int k = 0; k < 7; ++k) { n1[k] = nodes[i0 * 7 + k]; n2[k] = nodes[i1 * 7 + k]; }
      pgo_residual<T>(z, n1, n2, Tm, r);
      T x = T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) x += r[k] * r[k];
      acc += rk.kind != RK_NONE ? robust_rho<T>(rk, x) : x;        // the model's loss sum_e rho(|r_e|^2) (optimizer.py:118-125)
    }
    __syncthreads();
    if (R) {
      if (t < rows) row_st<6>(lds + t * 6, r);
      __syncthreads();
      slab_s2g<T, BLOCK, BLOCK * 6, true>(lds, R + e0 * 6, rows * 6, full);
      __syncthreads();
    }
  }
  T s = block_sum(acc);
  if constexpr (PACKLAST == 0) {
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
  } else {
    __shared__ int last_sh;
    if (threadIdx.x == 0) {
      xwg_store(partial + blockIdx.x, s);
      unsigned* ticket = reinterpret_cast<unsigned*>(state + 4);
      // (acquire as well: the last arriver's reads of the other workgroups' partials are ordered after their releases by the
      //  memory model, not only by the control dependency on the ticket's value -- ADVICE r05; <= 256 arrivals, nothing measurable)
      const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      last_sh = t == gridDim.x - 1 ? 1 : 0;
      if (last_sh) xwg_store(ticket, 0u);                          // (at rest again for the next execution)
    }
    __syncthreads();
    if constexpr (PACKLAST == 1) {
      if (last_sh && threadIdx.x < 64) pgo_trial_pack<T>(partial, (int)gridDim.x, pcg_info, state, out, true, ngain);
    } else {
      if (last_sh && threadIdx.x < 64) {
        double tot = 0.0;
        for (int q = threadIdx.x; q < (int)gridDim.x; q += 64) tot += (double)xwg_load(partial + q);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
        if (threadIdx.x == 0) loss_out[0] = (T)tot;
      }
    }
  }
}

constexpr int kPgoPartials = 1024;     // = PPLIE_PGO_PARTIALS in include/pplie.h
constexpr int kPackLastGrid = 256;     // residual grids up to this size let their last workgroup pack (pgo_trial_tail)

// ---------------------------------------------------------------------------------------------
// The tail of one captured LM trial (optim/pgograph.py): everything between the linear solve and the host's decision,
// TWO launches (three when the residual grid exceeds kPackLastGrid; four until round 5: at 10 k nodes each dependent launch costs
// ~4.5 us whatever it does) and no tensor ops --
//   first     gain + retract side by side (pgo_tail_first_kernel);   second   residual, whose last workgroup packs --
//   retract   nodes <- Exp(x_n) nodes_n   (lietensor.py:60-65, optimizer.py update_parameter); the old rows go to `backup` if given
//   residual  per edge at the candidate: per-workgroup partials of |r|^2, the trial's loss (optimizer.py:672; pgo_residual_kernel)
//   gain      JD_e = J_e0 x_i + J_e1 x_j = J_e1 (x_j - x_i): partials of sum JD.JD, sum JD.R  (strategy.py:144, :261; pgo_gain_antisym_kernel)
//   pack      one wavefront: the partials summed in index order (double), the solve's (iterations, |r|^2, |b|^2, flag) appended,
//             the loss stored into the caller's ring of loss scalars, and the 8 doubles {a, b, loss, its, rr, bn2, flag, seq}
//             stored with SYSTEM scope -- `out` may be host-pinned memory the host polls for `seq`, the last word written.
// `state` (device memory, EIGHT 64-bit words): {seq: incremented by every execution's last kernel, address of the loss ring (T*) or 0,
// its length, retractions: incremented by every execution's FIRST kernel, arrival ticket of the second launch (zero at rest), 3 reserved}.
// ---------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void pgo_retract_rows(int64_t block, T* __restrict__ nodes, const T* __restrict__ x,
                                                 T* __restrict__ backup /* or null */, int64_t N, unsigned long long* state) {
  const int64_t n = block * 256 + threadIdx.x;
  if (n >= N) return;
  if (n == 0) state[3] += 1;             // "the parameters were moved": what an error path needs to know before it restores them
  T d[7], X[7], E[7], out[7];
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = x[n * 6 + k];
  d[6] = T(0);
#pragma unroll
  for (int k = 0; k < 7; ++k) X[k] = nodes[n * 7 + k];
  if (backup) {
#pragma unroll
    for (int k = 0; k < 7; ++k) backup[n * 7 + k] = X[k];
  }
  se3_exp<T>(d, E);
  se3_mul<T>(E, X, out);
#pragma unroll
  for (int k = 0; k < 7; ++k) nodes[n * 7 + k] = out[k];
}

template <class T>
__device__ __forceinline__ void pgo_trial_pack(const T* partial, int nparts, const T* pcg_info, unsigned long long* state, double* out,
                                               bool coherent_loss, int ngain) {
  // (one wavefront) every lane sums the partials i = q, q + 64, ... and a shuffle tree adds the lanes: a fixed order, the same bits every replay
  // ngain: the number of gain-partial pairs when an earlier launch other than the tail's own wrote them (the solve's epilogue: one
  // pair per workgroup of ITS grid); < 0: as many as loss partials
  const int q = threadIdx.x;
  double loss = 0.0, a = 0.0, b = 0.0;
  for (int i = q; i < nparts; i += 64) loss += (double)(coherent_loss ? xwg_load(partial + i) : partial[i]);   // pgo_residual_kernel
  for (int i = q; i < (ngain < 0 ? nparts : ngain); i += 64) {
    a += (double)partial[kPgoPartials + 2 * i];                   // gain partials (an earlier launch): sum JD.JD
    b += (double)partial[kPgoPartials + 2 * i + 1];               //                    sum JD.R
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    loss += __shfl_down(loss, off, 64);
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  loss = __shfl(loss, 0, 64); a = __shfl(a, 0, 64); b = __shfl(b, 0, 64);
  const unsigned long long seq = state[0] + 1, ring = state[1], len = state[2];
  const T lossT = (T)loss;
  if (q == 0) {
    state[0] = seq;
    if (ring && len) reinterpret_cast<T*>(ring)[seq % len] = lossT;
  }
  // seven lanes store one word each (the stores overlap: `out` may be host memory, a round trip per store), then the
  // sequence number behind a system-scope fence
  const double v = q == 0 ? a : q == 1 ? b : q == 2 ? (double)lossT : q < 7 ? (double)pcg_info[q - 3] : 0.0;
  if (q < 7) __hip_atomic_store(out + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  if (q == 0) __hip_atomic_store(out + 7, (double)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the pack as a launch of its own (large grids: see pgo_trial_tail)
template <class T>
__global__ void __launch_bounds__(64)
pgo_trial_pack_kernel(const T* __restrict__ partial, int nparts, const T* __restrict__ pcg_info, unsigned long long* state, double* out,
                      int ngain = -1) {
  pgo_trial_pack<T>(partial, nparts, pcg_info, state, out, false, ngain);
}

// gain-ratio terms of the relative-pose program (strategy.py:144, :261): its two blocks per edge are opposite (J_e0 = -J_e1, see
// pgo_linearize_kernel), so JD_e = J_e1 (x_j - x_i) and only the second block is read -- 144 instead of 288 bytes per edge.
// M lanes per edge, lane i owns row i: the wave reads 10 consecutive blocks as one contiguous run (the generic graph_gain_kernel,
// one lane per edge, walks 288-byte records per lane: 2.4 TB/s at 4e5 edges).  partial[2 w], partial[2 w + 1] per workgroup w.
template <class T>
__device__ __forceinline__ void pgo_gain_antisym(int block, int nblocks, const T* __restrict__ J, const int64_t* __restrict__ idx,
                                                 const T* __restrict__ x, const T* __restrict__ R, T* __restrict__ partial, int64_t E) {
  constexpr int M = 6, NPW = 64 / M;
  const int lane = threadIdx.x & 63;
  const int sub = lane / M, i = lane - sub * M;
  const int64_t wave = ((int64_t)block * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)nblocks * 256) >> 6;
  T a1 = T(0), a2 = T(0);
  for (int64_t base = wave * NPW; base < E; base += nwaves * NPW) {
    const bool act = sub < NPW && base + sub < E;
    const int64_t e = act ? base + sub : E - 1;                    // (clamped: every load below is unconditional)
    const int64_t n0 = idx[e * 2], n1 = idx[e * 2 + 1];
    const T dv = x[n1 * M + i] - x[n0 * M + i];
    const T* Jr = J + ((e * 2 + 1) * M + i) * M;
    T row[M];
#pragma unroll
    for (int j = 0; j < M; ++j) row[j] = Jr[j];
    const T rv = R[e * M + i];
    T jd = T(0);
#pragma unroll
    for (int j = 0; j < M; ++j) jd += row[j] * __shfl(dv, (sub * M + j) & 63, 64);
    if (act) { a1 += jd * jd; a2 += jd * rv; }
  }
  T s1 = block_sum(a1);
  T s2 = block_sum(a2);
  if (threadIdx.x == 0) { partial[block * 2] = s1; partial[block * 2 + 1] = s2; }
}
// The tail's FIRST launch: the two pieces of work that only read the solve's step x and nothing of each other -- the gain terms
// (workgroups 0 .. gain_blocks - 1) and the retraction of the parameters (the rest, 256 nodes each)
template <class T>
__global__ void __launch_bounds__(256)
pgo_tail_first_kernel(int gain_blocks, const T* __restrict__ J, const int64_t* __restrict__ idx, const T* __restrict__ x,
                      const T* __restrict__ R, T* __restrict__ gain_partial, int64_t E, T* __restrict__ nodes,
                      T* __restrict__ backup /* or null */, int64_t N, unsigned long long* state) {
  if ((int)blockIdx.x < gain_blocks) pgo_gain_antisym<T>((int)blockIdx.x, gain_blocks, J, idx, x, R, gain_partial, E);
  else pgo_retract_rows<T>((int64_t)blockIdx.x - gain_blocks, nodes, x, backup, N, state);
}

template <class T>
int pgo_trial_tail(void* nodes, void* backup, const void* idx, const void* Z, const void* J, const void* R, const void* x, const void* pcg_info,
                   void* partial, void* state, void* out, int64_t N, int64_t E, void* stream) {
  if (N <= 0 || E <= 0) return PPLIE_EBADARG;
  if (!nodes || !idx || !Z || !J || !R || !x || !pcg_info || !partial || !state || !out || !aligned16(Z)) return PPLIE_EBADARG;
  if (reinterpret_cast<uintptr_t>(state) & 7) return PPLIE_EBADARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int BLOCK = 256;
  const int64_t nt = (E + BLOCK - 1) / BLOCK;
  const int grid = (int)(nt < kPgoPartials ? nt : kPgoPartials);             // (= the number of partials of every kind)
  T* part = (T*)partial;
  const int64_t rb = (N + 255) / 256;
  if (rb + grid >= (int64_t)1 << 31) return PPLIE_EBADARG;
  hipLaunchKernelGGL((pgo_tail_first_kernel<T>), dim3((unsigned)(grid + rb)), dim3(256), 0, st, grid, (const T*)J, (const int64_t*)idx,
                     (const T*)x, (const T*)R, part + kPgoPartials, E, (T*)nodes, (T*)backup, N, (unsigned long long*)state);
  // The last-arriving workgroup packs only while the arrivals are few: every workgroup's ticket is a release on ONE address -- 157
  // of them (10 k nodes / 40 k edges) cost about what the pack's own launch did (residual + pack 8.8 -> 9.1 us, one launch less), 1024
  // (4e5 edges) turned an 11.6 us residual kernel into 41.5 us (profiles/r05/lm_pgo_100k_kernel_stats.csv).
  if (grid <= kPackLastGrid) {
    hipLaunchKernelGGL((pgo_residual_kernel<T, BLOCK, 1>), dim3(grid), dim3(BLOCK), 0, st, (const T*)nodes, (const int64_t*)idx,
                       (const T*)Z, (T*)nullptr, part, E, RobustParam<T>{RK_NONE, T(0), T(0)}, (const T*)pcg_info,
                       (unsigned long long*)state, (double*)out);
  } else {
    hipLaunchKernelGGL((pgo_residual_kernel<T, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, (const T*)nodes, (const int64_t*)idx,
                       (const T*)Z, (T*)nullptr, part, E, RobustParam<T>{RK_NONE, T(0), T(0)});
    hipLaunchKernelGGL((pgo_trial_pack_kernel<T>), dim3(1), dim3(64), 0, st, (const T*)part, grid, (const T*)pcg_info,
                       (unsigned long long*)state, (double*)out);
  }
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

// what is left of the tail when the solve's epilogue has retracted the parameters and left the gain partials (csrc/pcg_persist.hip
// GhostTail): the candidate loss and the report -- ONE launch (two beyond kPackLastGrid residual workgroups)
template <class T>
int pgo_trial_tail_after_solve(const void* nodes, const void* idx, const void* Z, const void* pcg_info, void* partial, void* state, void* out,
                               int64_t E, int ngain, void* stream) {
  if (E <= 0 || ngain < 1 || ngain > kPgoPartials) return PPLIE_EBADARG;
  if (!nodes || !idx || !Z || !pcg_info || !partial || !state || !out || !aligned16(Z)) return PPLIE_EBADARG;
  if (reinterpret_cast<uintptr_t>(state) & 7) return PPLIE_EBADARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int BLOCK = 256;
  const int64_t nt = (E + BLOCK - 1) / BLOCK;
  const int grid = (int)(nt < kPgoPartials ? nt : kPgoPartials);
  T* part = (T*)partial;
  if (grid <= kPackLastGrid) {
    hipLaunchKernelGGL((pgo_residual_kernel<T, BLOCK, 1>), dim3(grid), dim3(BLOCK), 0, st, (const T*)nodes, (const int64_t*)idx,
                       (const T*)Z, (T*)nullptr, part, E, RobustParam<T>{RK_NONE, T(0), T(0)}, (const T*)pcg_info,
                       (unsigned long long*)state, (double*)out, ngain);
  } else {
    hipLaunchKernelGGL((pgo_residual_kernel<T, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, (const T*)nodes, (const int64_t*)idx,
                       (const T*)Z, (T*)nullptr, part, E, RobustParam<T>{RK_NONE, T(0), T(0)});
    hipLaunchKernelGGL((pgo_trial_pack_kernel<T>), dim3(1), dim3(64), 0, st, (const T*)part, grid, (const T*)pcg_info,
                       (unsigned long long*)state, (double*)out, ngain);
  }
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

template <class T>
int pgo_linearize(const void* nodes, const void* idx, const void* Z, void* R, void* J, int64_t E, void* stream, int kind = 0,
                  double p0 = 0, double p1 = 0, const void* inc = nullptr, void* HB = nullptr, void* gg = nullptr, int pack = 0,
                  void* ctl = nullptr, int64_t ctl_bytes = 0, const void* s_src = nullptr, void* s_dst = nullptr) {
  if (E < 0 || kind < 0 || kind > RK_TOLERANT) return PPLIE_EBADARG;
  if (E == 0) return PPLIE_OK;
  if (!nodes || !idx || !Z || !R || !J || !aligned16(Z) || !aligned16(R) || !aligned16(J)) return PPLIE_EBADARG;
  if ((inc || HB || gg) && !(inc && HB && gg)) return PPLIE_EBADARG;
  if (inc && E >= ((int64_t)1 << 30)) return PPLIE_EBADARG;      // (incidence slots are int32)
  if (ctl && (!inc || ctl_bytes < 0 || (ctl_bytes & 7) || (reinterpret_cast<uintptr_t>(ctl) & 7) || (s_src && !s_dst))) return PPLIE_EBADARG;
  constexpr int BLOCK = 64;
  int64_t nt = (E + BLOCK - 1) / BLOCK;
  int grid = (int)(nt < (1 << 20) ? nt : (1 << 20));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const RobustParam<T> rk{kind, (T)p0, (T)p1};
  if (!inc)
    hipLaunchKernelGGL((pgo_linearize_kernel<T, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, (const T*)nodes, (const int64_t*)idx,
                       (const T*)Z, (T*)R, (T*)J, E, rk);
  else if (pack)
    hipLaunchKernelGGL((pgo_linearize_kernel<T, BLOCK, 2, 2 * BLOCK>), dim3(grid), dim3(2 * BLOCK), 0, st, (const T*)nodes, (const int64_t*)idx,
                       (const T*)Z, (T*)R, (T*)J, E, rk, (const int*)inc, (T*)HB, (T*)gg, (unsigned long long*)ctl, ctl_bytes / 8,
                       (const double*)s_src, (double*)s_dst);
  else
    hipLaunchKernelGGL((pgo_linearize_kernel<T, BLOCK, 1, 2 * BLOCK>), dim3(grid), dim3(2 * BLOCK), 0, st, (const T*)nodes, (const int64_t*)idx,
                       (const T*)Z, (T*)R, (T*)J, E, rk, (const int*)inc, (T*)HB, (T*)gg, (unsigned long long*)ctl, ctl_bytes / 8,
                       (const double*)s_src, (double*)s_dst);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
template <class T>
int pgo_residual_launch(const void* nodes, const void* idx, const void* Z, void* R, void* partial, int64_t E, void* stream,
                        int kind = 0, double p0 = 0, double p1 = 0) {
  if (kind < 0 || kind > RK_TOLERANT) return PPLIE_EBADARG;
  if (E <= 0) return E == 0 ? PPLIE_OK : PPLIE_EBADARG;
  if (!nodes || !idx || !Z || !partial || !aligned16(Z) || (R && !aligned16(R))) return PPLIE_EBADARG;
  constexpr int BLOCK = 256;
  int64_t nt = (E + BLOCK - 1) / BLOCK;
  int grid = (int)(nt < kPgoPartials ? nt : kPgoPartials);
  hipLaunchKernelGGL((pgo_residual_kernel<T, BLOCK>), dim3(grid), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)nodes, (const int64_t*)idx, (const T*)Z, (T*)R, (T*)partial, E, RobustParam<T>{kind, (T)p0, (T)p1});
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
// the model's loss sum_e rho(|r_e|^2) as one device scalar in ONE launch; ws: [kPgoPartials] partials of T followed (8-byte aligned) by
// eight 64-bit words of which word 4 is the arrival ticket (zero at rest) -- pplie_pgo_loss_ws_bytes() says how much
template <class T>
int pgo_loss(const void* nodes, const void* idx, const void* Z, void* ws, void* loss, int64_t E, void* stream, int kind, double p0, double p1) {
  if (kind < 0 || kind > RK_TOLERANT || E < 0) return PPLIE_EBADARG;
  if (!loss || (E > 0 && (!nodes || !idx || !Z || !ws || !aligned16(Z) || (reinterpret_cast<uintptr_t>(ws) & 7)))) return PPLIE_EBADARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (E == 0) return hipMemsetAsync(loss, 0, sizeof(T), st) == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
  constexpr int BLOCK = 256;
  int64_t nt = (E + BLOCK - 1) / BLOCK;
  int grid = (int)(nt < kPgoPartials ? nt : kPgoPartials);
  unsigned long long* state = reinterpret_cast<unsigned long long*>((char*)ws + sizeof(T) * kPgoPartials);
  hipLaunchKernelGGL((pgo_residual_kernel<T, BLOCK, 2>), dim3(grid), dim3(BLOCK), 0, st, (const T*)nodes, (const int64_t*)idx, (const T*)Z,
                     (T*)nullptr, (T*)ws, E, RobustParam<T>{kind, (T)p0, (T)p1}, (const T*)nullptr, state, (double*)nullptr, -1, (T*)loss);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace pplie

extern "C" int pplie_pgo_loss_ws_bytes(void) { return 8 * pplie::kPgoPartials + 64; }
extern "C" int pplie_pgo_loss_f32(const void* nodes, const void* idx, const void* Z, void* ws, void* loss, int64_t E, int kind, double p0,
                                  double p1, void* stream) {
  return pplie::pgo_loss<float>(nodes, idx, Z, ws, loss, E, stream, kind, p0, p1);
}
extern "C" int pplie_pgo_loss_f64(const void* nodes, const void* idx, const void* Z, void* ws, void* loss, int64_t E, int kind, double p0,
                                  double p1, void* stream) {
  return pplie::pgo_loss<double>(nodes, idx, Z, ws, loss, E, stream, kind, p0, p1);
}

// the same two entries with a robust kernel (kind: PPLIE_ROBUST_* of include/pplie.h; p0 = delta, Tolerant: p0 = a, p1 = b):
// the linearisation returns the CORRECTED residuals and blocks sqrt(rho'(|r|^2)) (r, J) of corrector.py:91-96, the residual
// entry the partial sums of rho(|r_e|^2)
extern "C" int pplie_pgo_linearize_robust_f32(const void* nodes, const void* idx, const void* Z, void* R, void* J, int64_t E, int kind,
                                              double p0, double p1, void* stream) {
  return pplie::pgo_linearize<float>(nodes, idx, Z, R, J, E, stream, kind, p0, p1);
}
extern "C" int pplie_pgo_linearize_robust_f64(const void* nodes, const void* idx, const void* Z, void* R, void* J, int64_t E, int kind,
                                              double p0, double p1, void* stream) {
  return pplie::pgo_linearize<double>(nodes, idx, Z, R, J, E, stream, kind, p0, p1);
}
extern "C" int pplie_pgo_residual_robust_f32(const void* nodes, const void* idx, const void* Z, void* R, void* partial, int64_t E, int kind,
                                             double p0, double p1, void* stream) {
  return pplie::pgo_residual_launch<float>(nodes, idx, Z, R, partial, E, stream, kind, p0, p1);
}
extern "C" int pplie_pgo_residual_robust_f64(const void* nodes, const void* idx, const void* Z, void* R, void* partial, int64_t E, int kind,
                                             double p0, double p1, void* stream) {
  return pplie::pgo_residual_launch<double>(nodes, idx, Z, R, partial, E, stream, kind, p0, p1);
}

// linearisation + the edges' shares of the normal equations in one launch (pgo_linearize_kernel LAP): inc [E, 2] int32 = the incidence
// slot of (edge, side) in the node-sorted list of pplie_graph_assemble_lap (the inverse of its `blk`), HB [2 E, 36 | pack: 21] and
// gg [2 E, 6] as that entry's first launch leaves them; pplie_graph_lap_diag then finishes the assembly.  ctl != NULL: the launch also
// does pplie_pcg_begin's work for the solve that follows (clears ctl_bytes of its control block, s_src -> s_dst)
extern "C" int pplie_pgo_linearize_lap_f32(const void* nodes, const void* idx, const void* Z, void* R, void* J, const void* inc, void* HB,
                                           void* gg, int64_t E, int pack, int kind, double p0, double p1, void* ctl, int64_t ctl_bytes,
                                           const void* s_src, void* s_dst, void* stream) {
  if (!inc) return pplie::PPLIE_EBADARG;
  return pplie::pgo_linearize<float>(nodes, idx, Z, R, J, E, stream, kind, p0, p1, inc, HB, gg, pack, ctl, ctl_bytes, s_src, s_dst);
}
extern "C" int pplie_pgo_linearize_lap_f64(const void* nodes, const void* idx, const void* Z, void* R, void* J, const void* inc, void* HB,
                                           void* gg, int64_t E, int pack, int kind, double p0, double p1, void* ctl, int64_t ctl_bytes,
                                           const void* s_src, void* s_dst, void* stream) {
  if (!inc) return pplie::PPLIE_EBADARG;
  return pplie::pgo_linearize<double>(nodes, idx, Z, R, J, E, stream, kind, p0, p1, inc, HB, gg, pack, ctl, ctl_bytes, s_src, s_dst);
}
extern "C" int pplie_pgo_linearize_f32(const void* nodes, const void* idx, const void* Z, void* R, void* J, int64_t E, void* stream) {
  return pplie::pgo_linearize<float>(nodes, idx, Z, R, J, E, stream);
}
extern "C" int pplie_pgo_linearize_f64(const void* nodes, const void* idx, const void* Z, void* R, void* J, int64_t E, void* stream) {
  return pplie::pgo_linearize<double>(nodes, idx, Z, R, J, E, stream);
}
extern "C" int pplie_pgo_residual_f32(const void* nodes, const void* idx, const void* Z, void* R, void* partial, int64_t E, void* stream) {
  return pplie::pgo_residual_launch<float>(nodes, idx, Z, R, partial, E, stream);
}
extern "C" int pplie_pgo_residual_f64(const void* nodes, const void* idx, const void* Z, void* R, void* partial, int64_t E, void* stream) {
  return pplie::pgo_residual_launch<double>(nodes, idx, Z, R, partial, E, stream);
}
extern "C" int pplie_pgo_trial_tail_after_solve_f32(const void* nodes, const void* idx, const void* Z, const void* pcg_info, void* partial,
                                                    void* state, void* out, int64_t E, int ngain, void* stream) {
  return pplie::pgo_trial_tail_after_solve<float>(nodes, idx, Z, pcg_info, partial, state, out, E, ngain, stream);
}
extern "C" int pplie_pgo_trial_tail_after_solve_f64(const void* nodes, const void* idx, const void* Z, const void* pcg_info, void* partial,
                                                    void* state, void* out, int64_t E, int ngain, void* stream) {
  return pplie::pgo_trial_tail_after_solve<double>(nodes, idx, Z, pcg_info, partial, state, out, E, ngain, stream);
}
extern "C" int pplie_pgo_trial_tail_f32(void* nodes, void* backup, const void* idx, const void* Z, const void* J, const void* R, const void* x,
                                        const void* pcg_info, void* partial, void* state, void* out, int64_t N, int64_t E,
                                        void* stream) {
  return pplie::pgo_trial_tail<float>(nodes, backup, idx, Z, J, R, x, pcg_info, partial, state, out, N, E, stream);
}
extern "C" int pplie_pgo_trial_tail_f64(void* nodes, void* backup, const void* idx, const void* Z, const void* J, const void* R, const void* x,
                                        const void* pcg_info, void* partial, void* state, void* out, int64_t N, int64_t E,
                                        void* stream) {
  return pplie::pgo_trial_tail<double>(nodes, backup, idx, Z, J, R, x, pcg_info, partial, state, out, N, E, stream);
}
