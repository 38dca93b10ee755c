// graph.hip -- gather-structured (pose-graph) normal equations for Levenberg-Marquardt.
//
// A residual row e depends on K gathered parameter rows idx[e][0..K-1] (pose-graph edge: the
// two node poses; reference model examples/module/pgo/pgo.py:15-25).  With per-edge Jacobian
// blocks J[e][k] (DR x M) and an optional per-edge weight W[e] (DR x DR) the Gauss-Newton matrix
// H = sum_e J_e^T W_e J_e is block sparse (M x M blocks at (idx[e][k], idx[e][l])).  It is never
// formed: the reference's dense A = J^T W J (optimizer.py:655-657) would be [N*7]^2.
//
//   pplie_graph_assemble : block diagonal of H (for the diagonal clamp / damping / block-Jacobi
//                          preconditioner) and the gradient J^T W r, by atomic scatter-add
//   pplie_graph_spmv     : y += H p, matrix-free, one lane per edge
//
// Layout: J [E][K][DR][M] (edge-major, so a tile of edges is one contiguous slab -> staged to
// LDS with dwordx4 like every other kernel here), W [E][DR][DR], idx [E][K] int64,
// node vectors [N][M].  Scatter uses hardware fp atomics (-munsafe-fp-atomics).
#include "rowmap.h"

namespace pplie {

template <class T, int DR, int M, int K, bool HAS_W, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
graph_spmv_kernel(const T* __restrict__ J, const T* __restrict__ W, const int64_t* __restrict__ idx,
                  const T* __restrict__ p, T* __restrict__ y, int64_t E) {
  constexpr int JW = K * DR * M, WW = HAS_W ? DR * DR : 0;
  __shared__ __attribute__((aligned(16))) T lds[BLOCK * (JW + WW)];
  T* sJ = lds;
  T* sW = lds + BLOCK * JW;
  const int64_t ntiles = (E + BLOCK - 1) / BLOCK;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t e0 = tile * BLOCK;
    const int64_t left = E - e0;
    const bool full = left >= BLOCK;
    const int rows = full ? BLOCK : (int)left;
    slab_g2s<T, BLOCK, BLOCK * JW, true>(J + e0 * JW, sJ, rows * JW, full);
    if constexpr (HAS_W) slab_g2s<T, BLOCK, BLOCK * WW, true>(W + e0 * WW, sW, rows * WW, full);
    __syncthreads();
    const int t = threadIdx.x;
    if (t < rows) {
      const int64_t e = e0 + t;
      int64_t nid[K];
#pragma unroll
      for (int k = 0; k < K; ++k) nid[k] = idx[e * K + k];
      T q[DR];
#pragma unroll
      for (int i = 0; i < DR; ++i) q[i] = T(0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        T pk[M];
#pragma unroll
        for (int j = 0; j < M; ++j) pk[j] = p[nid[k] * M + j];
#pragma unroll
        for (int i = 0; i < DR; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) q[i] += sJ[t * JW + (k * DR + i) * M + j] * pk[j];
      }
      T wq[DR];
      if constexpr (HAS_W) {
#pragma unroll
        for (int i = 0; i < DR; ++i) {
          T a = T(0);
#pragma unroll
          for (int l = 0; l < DR; ++l) a += sW[t * WW + i * DR + l] * q[l];
          wq[i] = a;
        }
      } else {
#pragma unroll
        for (int i = 0; i < DR; ++i) wq[i] = q[i];
      }
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < M; ++j) {
          T a = T(0);
#pragma unroll
          for (int i = 0; i < DR; ++i) a += sJ[t * JW + (k * DR + i) * M + j] * wq[i];
          atomicAdd(y + nid[k] * M + j, a);
        }
    }
    __syncthreads();
  }
}

// Bdiag[idx[e][k]] += J_k^T W J_k ; grad[idx[e][k]] += J_k^T W r
template <class T, int DR, int M, int K, bool HAS_W, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
graph_assemble_kernel(const T* __restrict__ J, const T* __restrict__ W, const T* __restrict__ R,
                      const int64_t* __restrict__ idx, T* __restrict__ Bdiag, T* __restrict__ grad,
                      T* __restrict__ H12 /* [E,M,M] = J_0^T W J_1, or null */, int64_t E) {
  constexpr int JW = K * DR * M, WW = HAS_W ? DR * DR : 0;
  __shared__ __attribute__((aligned(16))) T lds[BLOCK * (JW + WW)];
  T* sJ = lds;
  T* sW = lds + BLOCK * JW;
  const int64_t ntiles = (E + BLOCK - 1) / BLOCK;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t e0 = tile * BLOCK;
    const int64_t left = E - e0;
    const bool full = left >= BLOCK;
    const int rows = full ? BLOCK : (int)left;
    slab_g2s<T, BLOCK, BLOCK * JW, true>(J + e0 * JW, sJ, rows * JW, full);
    if constexpr (HAS_W) slab_g2s<T, BLOCK, BLOCK * WW, true>(W + e0 * WW, sW, rows * WW, full);
    __syncthreads();
    const int t = threadIdx.x;
    if (t < rows) {
      const int64_t e = e0 + t;
      T r[DR];
#pragma unroll
      for (int i = 0; i < DR; ++i) r[i] = R[e * DR + i];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int64_t n = idx[e * K + k];
        const T* Jk = sJ + t * JW + k * DR * M;
        T WJ[DR * M];   // W J_k  (or J_k)
        T Wr[DR];
#pragma unroll
        for (int i = 0; i < DR; ++i) {
          if constexpr (HAS_W) {
            T a = T(0);
#pragma unroll
            for (int l = 0; l < DR; ++l) a += sW[t * WW + i * DR + l] * r[l];
            Wr[i] = a;
#pragma unroll
            for (int j = 0; j < M; ++j) {
              T b = T(0);
#pragma unroll
              for (int l = 0; l < DR; ++l) b += sW[t * WW + i * DR + l] * Jk[l * M + j];
              WJ[i * M + j] = b;
            }
          } else {
            Wr[i] = r[i];
#pragma unroll
            for (int j = 0; j < M; ++j) WJ[i * M + j] = Jk[i * M + j];
          }
        }
        if (K == 2 && k == 1 && H12) {   // off-diagonal block of this edge: J_0^T (W J_1), stored once
          const T* J0 = sJ + t * JW;
#pragma unroll
          for (int a = 0; a < M; ++a)
#pragma unroll
            for (int b = 0; b < M; ++b) {
              T s = T(0);
#pragma unroll
              for (int i = 0; i < DR; ++i) s += J0[i * M + a] * WJ[i * M + b];
              H12[(e * M + a) * M + b] = s;
            }
        }
#pragma unroll
        for (int a = 0; a < M; ++a) {
#pragma unroll
          for (int b = 0; b < M; ++b) {
            T s = T(0);
#pragma unroll
            for (int i = 0; i < DR; ++i) s += Jk[i * M + a] * WJ[i * M + b];
            atomicAdd(Bdiag + (n * M + a) * M + b, s);
          }
          T s = T(0);
#pragma unroll
          for (int i = 0; i < DR; ++i) s += Jk[i * M + a] * Wr[i];
          atomicAdd(grad + n * M + a, s);
        }
      }
    }
    __syncthreads();
  }
}

template <class T, int DR, int M, int K>
int graph_spmv_launch(const void* J, const void* W, const void* idx, const void* p, void* y, int64_t E, void* stream) {
  if (E <= 0) return E == 0 ? PPLIE_OK : PPLIE_EBADARG;
  if (!J || !idx || !p || !y || !aligned16(J) || (W && !aligned16(W))) return PPLIE_EBADARG;
  constexpr int BLOCK = 64;
  int64_t nt = (E + BLOCK - 1) / BLOCK;
  int grid = (int)(nt < (1 << 30) ? nt : (1 << 30));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (W)
    hipLaunchKernelGGL((graph_spmv_kernel<T, DR, M, K, true, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, (const T*)J,
                       (const T*)W, (const int64_t*)idx, (const T*)p, (T*)y, E);
  else
    hipLaunchKernelGGL((graph_spmv_kernel<T, DR, M, K, false, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, (const T*)J,
                       (const T*)nullptr, (const int64_t*)idx, (const T*)p, (T*)y, E);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
template <class T, int DR, int M, int K>
int graph_assemble_launch(const void* J, const void* W, const void* R, const void* idx, void* B, void* g, void* H12,
                          int64_t E, void* stream) {
  if (E <= 0) return E == 0 ? PPLIE_OK : PPLIE_EBADARG;
  if (!J || !R || !idx || !B || !g || !aligned16(J) || (W && !aligned16(W))) return PPLIE_EBADARG;
  constexpr int BLOCK = 64;
  int64_t nt = (E + BLOCK - 1) / BLOCK;
  int grid = (int)(nt < (1 << 30) ? nt : (1 << 30));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (W)
    hipLaunchKernelGGL((graph_assemble_kernel<T, DR, M, K, true, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, (const T*)J,
                       (const T*)W, (const T*)R, (const int64_t*)idx, (T*)B, (T*)g, (T*)H12, E);
  else
    hipLaunchKernelGGL((graph_assemble_kernel<T, DR, M, K, false, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, (const T*)J,
                       (const T*)nullptr, (const T*)R, (const int64_t*)idx, (T*)B, (T*)g, (T*)H12, E);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

// supported shapes: (DR, M, K) = (6,6,2) SE3 pose graph, (7,7,2) Sim3, (3,3,2) SO3, (6,6,1)/(3,3,1) priors
#define PPLIE_GRAPH_SHAPES(X) X(6, 6, 2) X(7, 7, 2) X(3, 3, 2) X(6, 6, 1) X(3, 3, 1)

template <class T>
int graph_spmv_dispatch(int dr, int m, int k, const void* J, const void* W, const void* idx, const void* p, void* y,
                        int64_t E, void* stream) {
#define X(A, B, C) \
  if (dr == A && m == B && k == C) return graph_spmv_launch<T, A, B, C>(J, W, idx, p, y, E, stream);
  PPLIE_GRAPH_SHAPES(X)
#undef X
  return PPLIE_EBADARG;
}
template <class T>
int graph_assemble_dispatch(int dr, int m, int k, const void* J, const void* W, const void* R, const void* idx, void* B,
                            void* g, void* H12, int64_t E, void* stream) {
#define X(A, B_, C) \
  if (dr == A && m == B_ && k == C) return graph_assemble_launch<T, A, B_, C>(J, W, R, idx, B, g, H12, E, stream);
  PPLIE_GRAPH_SHAPES(X)
#undef X
  return PPLIE_EBADARG;
}
}  // namespace pplie

extern "C" int pplie_graph_spmv_f32(const void* J, const void* W, const void* idx, const void* p, void* y, int64_t E,
                                    int dr, int m, int k, void* stream) {
  return pplie::graph_spmv_dispatch<float>(dr, m, k, J, W, idx, p, y, E, stream);
}
extern "C" int pplie_graph_spmv_f64(const void* J, const void* W, const void* idx, const void* p, void* y, int64_t E,
                                    int dr, int m, int k, void* stream) {
  return pplie::graph_spmv_dispatch<double>(dr, m, k, J, W, idx, p, y, E, stream);
}
extern "C" int pplie_graph_assemble_f32(const void* J, const void* W, const void* R, const void* idx, void* Bdiag,
                                        void* grad, void* H12, int64_t E, int dr, int m, int k, void* stream) {
  return pplie::graph_assemble_dispatch<float>(dr, m, k, J, W, R, idx, Bdiag, grad, H12, E, stream);
}
extern "C" int pplie_graph_assemble_f64(const void* J, const void* W, const void* R, const void* idx, void* Bdiag,
                                        void* grad, void* H12, int64_t E, int dr, int m, int k, void* stream) {
  return pplie::graph_assemble_dispatch<double>(dr, m, k, J, W, R, idx, Bdiag, grad, H12, E, stream);
}

// ---------------------------------------------------------------------------------------------
// Fused vector kernels of the block-Jacobi PCG iteration (optim/posegraph.py).  One iteration is
//   y = 0 ; spmv(y += H p) ; dot_shift ; update ; direction ; rotate
// five small launches on N*m-element vectors, captured into a hipGraph by the host (the loop is
// launch-bound: 10^5 nodes x 6 floats = 2.4 MB per vector).  Scalars live on the device:
//   scal[0] = rho = r.z   scal[1] = p.q   scal[2] = rho_new   scal[3] = r.r   (all of type T)
// ---------------------------------------------------------------------------------------------
namespace pplie {

// q += shift * p (elementwise) ; scal[1] += p . q
template <class T> __global__ void __launch_bounds__(256) pcg_dot_shift_kernel(T* q, const T* p, const T* shift, T* scal, int64_t n) {
  T acc = T(0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T qi = q[i] + shift[i] * p[i];
    q[i] = qi;
    acc += p[i] * qi;
  }
  T s = block_sum(acc);
  if (threadIdx.x == 0) atomicAdd(scal + 1, s);
}

// alpha = rho / (p.q); x += alpha p; r -= alpha q; z = Binv r (per node, m x m); rho_new += r.z; rr += r.r
// One lane per vector element (node n, row i): its own x/r/p/q element is a coalesced access, the
// node's other r/q elements and Binv row i are 4m-byte contiguous reads shared within the node's lanes.
template <class T> __global__ void __launch_bounds__(256)
pcg_update_kernel(T* x, T* r, const T* p, const T* q, T* z, const T* Binv, T* scal, int64_t N, int m) {
  const T alpha = scal[1] != T(0) ? scal[0] / scal[1] : T(0);   // p.q = 0 only once r = 0: stay put, no NaN
  T a1 = T(0), a2 = T(0);
  const int64_t total = N * m;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t nidx = e / m;
    const int i = (int)(e - nidx * m);
    x[e] += alpha * p[e];
    T s = T(0), ri = T(0);
    for (int j = 0; j < m; ++j) {
      const T rj = r[nidx * m + j] - alpha * q[nidx * m + j];
      if (j == i) ri = rj;
      s += Binv[e * m + j] * rj;
    }
    z[e] = s;
    a1 += ri * s;
    a2 += ri * ri;
    // r is read by the other lanes of this node in the same pass: written in a second sweep below
  }
  __syncthreads();
  T s1 = block_sum(a1);
  T s2 = block_sum(a2);
  if (threadIdx.x == 0) { atomicAdd(scal + 2, s1); atomicAdd(scal + 3, s2); }
}
// second half of the update: r -= alpha q (kept separate so every lane of stage 1 sees the old r)
template <class T> __global__ void __launch_bounds__(256) pcg_residual_kernel(T* r, const T* q, const T* scal_prev, int64_t n) {
  const T alpha = scal_prev[1] != T(0) ? scal_prev[0] / scal_prev[1] : T(0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) r[i] -= alpha * q[i];
}

// p = z + (rho_new / rho) p
template <class T> __global__ void __launch_bounds__(256) pcg_direction_kernel(T* p, const T* z, const T* scal, int64_t n) {
  const T beta = scal[0] != T(0) ? scal[2] / scal[0] : T(0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = z[i] + beta * p[i];
}

// rho <- rho_new ; history[it] = r.r ; clear the accumulators ; ++it
template <class T> __global__ void pcg_rotate_kernel(T* scal, T* rr_hist, int* it, int cap) {
  int k = *it;
  if (k < cap) rr_hist[k] = scal[3];
  scal[0] = scal[2];
  scal[1] = T(0); scal[2] = T(0); scal[3] = T(0);
  *it = k + 1;
}

template <class T>
int pcg_vector_step(int stage, void* x, void* r, void* p, void* q, void* z, const void* Binv, const void* shift, void* scal,
                    void* rr_hist, void* it, int cap, int64_t N, int m, void* stream) {
  if (N <= 0 || m <= 0 || m > 8) return PPLIE_EBADARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = N * m;
  // few workgroups: each ends in one atomicAdd on a shared scalar, and the vectors are L2-sized
  int g1 = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
  switch (stage) {
    case 0: hipLaunchKernelGGL((pcg_dot_shift_kernel<T>), dim3(g1), dim3(256), 0, st, (T*)q, (const T*)p, (const T*)shift, (T*)scal, n); break;
    case 1:
      hipLaunchKernelGGL((pcg_update_kernel<T>), dim3(g1), dim3(256), 0, st, (T*)x, (T*)r, (const T*)p, (const T*)q, (T*)z, (const T*)Binv, (T*)scal, N, m);
      hipLaunchKernelGGL((pcg_residual_kernel<T>), dim3(g1), dim3(256), 0, st, (T*)r, (const T*)q, (const T*)scal, n);
      break;
    case 2: hipLaunchKernelGGL((pcg_direction_kernel<T>), dim3(g1), dim3(256), 0, st, (T*)p, (const T*)z, (const T*)scal, n); break;
    case 3: hipLaunchKernelGGL((pcg_rotate_kernel<T>), dim3(1), dim3(1), 0, st, (T*)scal, (T*)rr_hist, (int*)it, cap); break;
    default: return PPLIE_EBADARG;
  }
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace pplie

extern "C" int pplie_pcg_stage_f32(int stage, void* x, void* r, void* p, void* q, void* z, const void* Binv, const void* shift,
                                   void* scal, void* rr_hist, void* it, int cap, int64_t N, int m, void* stream) {
  return pplie::pcg_vector_step<float>(stage, x, r, p, q, z, Binv, shift, scal, rr_hist, it, cap, N, m, stream);
}
extern "C" int pplie_pcg_stage_f64(int stage, void* x, void* r, void* p, void* q, void* z, const void* Binv, const void* shift,
                                   void* scal, void* rr_hist, void* it, int cap, int64_t N, int m, void* stream) {
  return pplie::pcg_vector_step<double>(stage, x, r, p, q, z, Binv, shift, scal, rr_hist, it, cap, N, m, stream);
}

// ---------------------------------------------------------------------------------------------
// Node-parallel block-sparse SpMV (no atomics, deterministic):  q_n = D_n p_n + sum_inc HB[blk] p[other]
// over the incidences of node n (CSR: ptr[N+1], blk[nnz] = 2*edge + side, other[nnz] = the node at
// the far end).  HB[2e] = H12[e], HB[2e+1] = H12[e]^T, D_n = diagonal block with the LM damping
// already folded in.  M lanes cooperate on one node (lane i owns output row i), so a block row
// is one contiguous 4M-byte read per lane and a block is one contiguous 4M^2-byte read per node
// group.  Also accumulates scal[1] += p.q (the PCG step length needs it next).
// ---------------------------------------------------------------------------------------------
namespace pplie {
template <class T, int M>
__global__ void __launch_bounds__(256)
graph_bsr_spmv_kernel(const int* __restrict__ ptr, const int* __restrict__ blk, const int* __restrict__ other,
                      const T* __restrict__ HB, const T* __restrict__ D, const T* __restrict__ p, T* __restrict__ q,
                      T* __restrict__ scal, int64_t N) {
  constexpr int NPW = 64 / M;                       // nodes per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / M, i = lane % M;
  const bool active_lane = sub < NPW;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  T acc_dot = T(0);
  for (int64_t base = wave * NPW; base < N; base += nwaves * NPW) {
    const int64_t n = base + sub;
    if (active_lane && n < N) {
      T pv[M];
#pragma unroll
      for (int j = 0; j < M; ++j) pv[j] = p[n * M + j];
      T acc = T(0);
#pragma unroll
      for (int j = 0; j < M; ++j) acc += D[(n * M + i) * M + j] * pv[j];
      const int beg = ptr[n], end = ptr[n + 1];
      for (int c = beg; c < end; ++c) {
        const int64_t b = blk[c];
        const int64_t o = other[c];
        const T* row = HB + (b * M + i) * M;
        const T* po = p + o * M;
#pragma unroll
        for (int j = 0; j < M; ++j) acc += row[j] * po[j];
      }
      q[n * M + i] = acc;
      acc_dot += acc * pv[i];
    }
  }
  T s = block_sum(acc_dot);
  if (threadIdx.x == 0) atomicAdd(scal + 1, s);
}

template <class T>
int bsr_spmv_launch(const void* ptr, const void* blk, const void* other, const void* HB, const void* D, const void* p, void* q,
                    void* scal, int64_t N, int m, void* stream) {
  if (N <= 0) return N == 0 ? PPLIE_OK : PPLIE_EBADARG;
  if (!ptr || !blk || !other || !HB || !D || !p || !q || !scal) return PPLIE_EBADARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define LAUNCH(MM)                                                                                                   \
  {                                                                                                                  \
    int64_t waves = (N + (64 / MM) - 1) / (64 / MM);                                                                 \
    int64_t blocks = (waves + 3) / 4;                                                                                \
    int grid = (int)(blocks < 4096 ? blocks : 4096);                                                                 \
    hipLaunchKernelGGL((graph_bsr_spmv_kernel<T, MM>), dim3(grid), dim3(256), 0, st, (const int*)ptr, (const int*)blk, \
                       (const int*)other, (const T*)HB, (const T*)D, (const T*)p, (T*)q, (T*)scal, N);               \
  }
  if (m == 6) LAUNCH(6) else if (m == 7) LAUNCH(7) else if (m == 3) LAUNCH(3) else return PPLIE_EBADARG;
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace pplie

extern "C" int pplie_graph_bsr_spmv_f32(const void* ptr, const void* blk, const void* other, const void* HB, const void* D,
                                        const void* p, void* q, void* scal, int64_t N, int m, void* stream) {
  return pplie::bsr_spmv_launch<float>(ptr, blk, other, HB, D, p, q, scal, N, m, stream);
}
extern "C" int pplie_graph_bsr_spmv_f64(const void* ptr, const void* blk, const void* other, const void* HB, const void* D,
                                        const void* p, void* q, void* scal, int64_t N, int m, void* stream) {
  return pplie::bsr_spmv_launch<double>(ptr, blk, other, HB, D, p, q, scal, N, m, stream);
}
