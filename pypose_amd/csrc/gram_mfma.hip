// gram_mfma.hip -- normal equations of a BATCH of dense Jacobian blocks on the matrix cores.
//
// BASELINE north_star: "MFMA only for the dense batched J stack when residual dimension is large".  The reference forms
// A = J^T W J and b = J^T W R on one dense matrix (pypose/optim/optimizer.py:655-657); for B independent problems the block
// linearisation keeps J as [B, d_res, d_par] blocks (optim/blocks.py).  Small residuals (d_res <= 7) stay in registers
// (csrc/lm_blocks.hip); LARGE ones -- many stacked residuals per problem: a pose observing dozens of points, a trajectory
// segment against a long measurement vector -- are a tall-skinny Gram product per problem,
//
//     [ A  g ]     [ J^T ]
//     [ g^T . ]  = [ R^T ] [ J  R ]          (d_par + 1 <= 16 columns, d_res rows)
//
// which is exactly one 16 x 16 accumulator tile of  v_mfma_f32_16x16x4_f32  (v_mfma_f64_16x16x4_f64 in double): per
// instruction a wave consumes FOUR rows of [J | R], and because the left operand is the transpose of the right one, lane l
// supplies the SAME register as A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]: one load per lane per step, no
// transposition, no LDS.  One wavefront per problem; the corner of the tile is R^T R, the problem's loss, for free.
// The kernel is HBM-bound (it reads d_res (d_par + 1) words per problem and writes d_par (d_par + 1)): the matrix cores are
// what keeps a dense 512 x 7 block from costing 512 x 56 VALU FMAs per problem.
#include "rowmap.h"

namespace pplie {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <class T> struct GramAcc;
template <> struct GramAcc<float> {
  typedef f32x4 V;
  static __device__ __forceinline__ V mma(float a, V c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) * 4 + reg; }      // C/D map of the f32 16x16 tiles
};
template <> struct GramAcc<double> {
  typedef f64x4 V;
  static __device__ __forceinline__ V mma(double a, V c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }      // (f64 uses its own map)
};

constexpr int kGramWaves = 4;        // problems per workgroup

template <class T>
__global__ void __launch_bounds__(64 * kGramWaves)
block_gram_mfma_kernel(const T* __restrict__ J, const T* __restrict__ R, T* __restrict__ A, T* __restrict__ g, T* __restrict__ rr,
                       int64_t n, int dr, int dp) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * kGramWaves + (threadIdx.x >> 6);
  if (b >= n) return;                                      // (whole waves: MFMA needs EXEC all ones)
  const int col = lane & 15, k = lane >> 4;
  const T* Jb = J + (size_t)b * dr * dp;
  const T* Rb = R + (size_t)b * dr;
  typename GramAcc<T>::V acc = {T(0), T(0), T(0), T(0)};
  // four steps (16 rows) per trip, every load of a trip issued before the first product
  int r0 = 0;
  for (; r0 + 16 <= dr; r0 += 16) {
    T v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = r0 + 4 * s + k;
      v[s] = col < dp ? Jb[(size_t)row * dp + col] : (col == dp ? Rb[row] : T(0));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = GramAcc<T>::mma(v[s], acc);
  }
  for (; r0 < dr; r0 += 4) {
    const int row = r0 + k;
    const T v = row < dr ? (col < dp ? Jb[(size_t)row * dp + col] : (col == dp ? Rb[row] : T(0))) : T(0);
    acc = GramAcc<T>::mma(v, acc);
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int i = GramAcc<T>::row(lane, reg);
    const T c = acc[reg];
    if (i < dp && col < dp) A[((size_t)b * dp + i) * dp + col] = c;
    if (i < dp && col == dp) g[(size_t)b * dp + i] = c;
    if (i == dp && col == dp && rr) rr[b] = c;
  }
}

template <class T>
int block_gram_mfma(const void* J, const void* R, void* A, void* g, void* rr, int64_t n, int dr, int dp, void* stream) {
  if (n < 0 || dr < 1 || dp < 1 || dp > 15) return PPLIE_EBADARG;
  if (n == 0) return PPLIE_OK;
  if (!J || !R || !A || !g) return PPLIE_EBADARG;
  const int64_t grid = (n + kGramWaves - 1) / kGramWaves;
  hipLaunchKernelGGL((block_gram_mfma_kernel<T>), dim3((unsigned)grid), dim3(64 * kGramWaves), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)J, (const T*)R, (T*)A, (T*)g, (T*)rr, n, dr, dp);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace pplie

extern "C" int pplie_block_gram_mfma_f32(const void* J, const void* R, void* A, void* g, void* rr, int64_t n, int dr, int dp, void* stream) {
  return pplie::block_gram_mfma<float>(J, R, A, g, rr, n, dr, dp, stream);
}
extern "C" int pplie_block_gram_mfma_f64(const void* J, const void* R, void* A, void* g, void* rr, int64_t n, int dr, int dp, void* stream) {
  return pplie::block_gram_mfma<double>(J, R, A, g, rr, n, dr, dp, stream);
}
