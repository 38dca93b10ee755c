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
                      const int64_t* __restrict__ idx, T* __restrict__ Bdiag, T* __restrict__ grad, int64_t E) {
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
int graph_assemble_launch(const void* J, const void* W, const void* R, const void* idx, void* B, void* g, int64_t E,
                          void* stream) {
  if (E <= 0) return E == 0 ? PPLIE_OK : PPLIE_EBADARG;
  if (!J || !R || !idx || !B || !g || !aligned16(J) || (W && !aligned16(W))) return PPLIE_EBADARG;
  constexpr int BLOCK = 64;
  int64_t nt = (E + BLOCK - 1) / BLOCK;
  int grid = (int)(nt < (1 << 30) ? nt : (1 << 30));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (W)
    hipLaunchKernelGGL((graph_assemble_kernel<T, DR, M, K, true, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, (const T*)J,
                       (const T*)W, (const T*)R, (const int64_t*)idx, (T*)B, (T*)g, E);
  else
    hipLaunchKernelGGL((graph_assemble_kernel<T, DR, M, K, false, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, (const T*)J,
                       (const T*)nullptr, (const T*)R, (const int64_t*)idx, (T*)B, (T*)g, E);
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
                            void* g, int64_t E, void* stream) {
#define X(A, B_, C) \
  if (dr == A && m == B_ && k == C) return graph_assemble_launch<T, A, B_, C>(J, W, R, idx, B, g, E, stream);
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
                                        void* grad, int64_t E, int dr, int m, int k, void* stream) {
  return pplie::graph_assemble_dispatch<float>(dr, m, k, J, W, R, idx, Bdiag, grad, E, stream);
}
extern "C" int pplie_graph_assemble_f64(const void* J, const void* W, const void* R, const void* idx, void* Bdiag,
                                        void* grad, int64_t E, int dr, int m, int k, void* stream) {
  return pplie::graph_assemble_dispatch<double>(dr, m, k, J, W, R, idx, Bdiag, grad, E, stream);
}
