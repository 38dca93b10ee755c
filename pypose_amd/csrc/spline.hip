// spline.hip -- the two interpolators of pypose/function/spline.py as single kernels.
//
// bspline (spline.py:105-225): cumulative cubic B-spline on SE3.  A query in segment i uses the four poses
// P_i..P_{i+3}:   T(u) = P_i * ((Exp(w0(u) xi_0) * Exp(w1(u) xi_1)) * Exp(w2(u) xi_2)),   xi_j = Log(P_{i+j}^-1 P_{i+j+1}),
// with weights w(u) = M [1 u u^2 u^3]^T (the 3x4 matrix at spline.py:208-210), u = k * interval, k = 0..K-1, plus ONE
// closing pose per trajectory evaluated at u = 1 on the last segment (:221-224).  The reference runs this as
// gather -> Inv -> Mul -> Log -> scale -> Exp -> 3 Mul -> Mul -> cat with every intermediate in HBM (≈ 25 row
// tensors per output pose); here a workgroup owns a tile of consecutive OUTPUT rows: the segments under the tile
// put their three twists in LDS once (3/K Logs per output instead of 3), every lane then evaluates its pose in
// registers and the tile leaves through one coalesced slab store.
// Algorithmic bytes per output pose: 28 written + 28/K read (each control pose is needed by four segments but is
// fetched from HBM once; the other three uses hit L2).
//
// chspline (spline.py:4-102): cubic Hermite interpolation of plain points [.., N, C] with finite-difference
// tangents.  The per-sample basis values hh[M,4] and segment indices depend on the sample times only (shared by
// every trajectory and channel); the host passes them in, the kernel blends.
#include "rowmap.h"

namespace pplie {

template <class T, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
se3_bspline_kernel(const T* __restrict__ data, const T* __restrict__ w /* [3][K+1] */, T* __restrict__ out,
                   int64_t nb, int64_t N, int64_t K) {
  constexpr int MAXSEG = BLOCK / 2 + 2;          // K >= 2 output rows per segment
  constexpr int SP = 25;                         // 3 twists + the segment's first pose; odd pitch: conflict-free
  __shared__ __attribute__((aligned(16))) T s_seg[MAXSEG * SP];
  __shared__ __attribute__((aligned(16))) T s_out[BLOCK * 7];
  const int64_t nseg = N - 3;                    // segments per trajectory
  const int64_t L = nseg * K + 1;                // output poses per trajectory
  const int64_t total = nb * L;
  const int64_t ntiles = (total + BLOCK - 1) / BLOCK;
  const uint32_t Lu = (uint32_t)L, Ku = (uint32_t)K, nsegu = (uint32_t)nseg;      // L < 2^31 (checked by the launcher)
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * BLOCK;
    const int64_t left = total - row0;
    const bool full = left >= BLOCK;
    const int rows = full ? BLOCK : (int)left;
    const int64_t tb = row0 / L;                                  // one 64-bit division per tile (wave-uniform) ...
    const uint32_t tr = (uint32_t)(row0 - tb * L);
    // ... and 32-bit ones per lane: row0 + d sits tr + d rows into trajectory tb (tr + d < 2^31 + 256)
    auto locate = [&](uint32_t d, int64_t& b, uint32_t& s, uint32_t& k) {
      const uint32_t q = (tr + d) / Lu;
      const uint32_t r = (tr + d) - q * Lu;
      b = tb + q;
      s = r / Ku;
      if (s > nsegu - 1) s = nsegu - 1;            // the closing pose belongs to the last segment, k == K
      k = r - s * Ku;
    };
    int64_t b0, b1;
    uint32_t s0, k0, s1, k1;
    locate(0, b0, s0, k0);
    locate((uint32_t)(rows - 1), b1, s1, k1);
    const int64_t g0 = b0 * nseg + s0;
    const int segs = (int)(b1 * nseg + s1 - g0) + 1;
    for (int j = threadIdx.x; j < segs; j += BLOCK) {
      const uint32_t sj = s0 + (uint32_t)j, qj = sj / nsegu;        // segment j of the tile, counted from (b0, s0)
      const int64_t b = b0 + qj, s = sj - qj * nsegu;
      const T* p = data + (b * N + s) * 7;
      T P[4][7];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 7; ++c) P[a][c] = p[a * 7 + c];
      T* dst = s_seg + j * SP;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        T inv[7], rel[7], xi[6];
        se3_inv<T>(P[a], inv);
        se3_mul<T>(inv, P[a + 1], rel);
        se3_log<T>(rel, xi);
#pragma unroll
        for (int c = 0; c < 6; ++c) dst[a * 6 + c] = xi[c];
      }
#pragma unroll
      for (int c = 0; c < 7; ++c) dst[18 + c] = P[0][c];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < rows) {
      int64_t b;
      uint32_t sg, k;
      locate((uint32_t)t, b, sg, k);
      const T* src = s_seg + (int)(b * nseg + sg - g0) * SP;
      T A[3][7];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const T wa = w[a * (K + 1) + k];
        T xi[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) xi[c] = src[a * 6 + c] * wa;
        se3_exp<T>(xi, A[a]);
      }
      T P0[7], m01[7], m[7], o[7];
#pragma unroll
      for (int c = 0; c < 7; ++c) P0[c] = src[18 + c];
      se3_mul<T>(A[0], A[1], m01);
      se3_mul<T>(m01, A[2], m);
      se3_mul<T>(P0, m, o);
      row_st<7>(s_out + t * 7, o);
    }
    __syncthreads();
    slab_s2g<T, BLOCK, BLOCK * 7, true>(s_out, out + row0 * 7, rows * 7, full);
    // the next tile's segment pass writes s_seg only; s_out is rewritten after that tile's barrier
  }
}

// out[b, o, c] = hh[o,0] p[i] + hh[o,1] m[i] + hh[o,2] p[i+1] + hh[o,3] m[i+1],  i = seg[o],
// m = finite-difference tangents (unit knot spacing): one-sided at the two ends, mean of both sides inside.
// One lane per sample o (its basis row and segment are loaded once), looping over the trajectories of its
// blockIdx.y slice and the C channels: no index division, each lane writes C consecutive values, a wave a
// contiguous 64 C run.
template <class T, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
chspline_kernel(const T* __restrict__ pts, const T* __restrict__ hh, const int64_t* __restrict__ seg,
                T* __restrict__ out, int64_t nb, int64_t N, int64_t C, int64_t M) {
  const int64_t o = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (o >= M) return;
  const int64_t i = seg[o];
  const T h0 = hh[o * 4], h1 = hh[o * 4 + 1], h2 = hh[o * 4 + 2], h3 = hh[o * 4 + 3];
  const bool has_left = i > 0, has_right = i + 2 < N;
  for (int64_t b = blockIdx.y; b < nb; b += gridDim.y) {
    const T* p = pts + (b * N + i) * C;
    T* dst = out + (b * M + o) * C;
    for (int64_t c = 0; c < C; ++c) {
      const T p0 = p[c], p1 = p[C + c];
      const T d = p1 - p0;
      const T m0 = has_left ? (d + (p0 - p[c - C])) / T(2) : d;
      const T m1 = has_right ? ((p[2 * C + c] - p1) + d) / T(2) : d;
      T v = h0 * p0;
      v += h1 * m0;
      v += h2 * p1;
      v += h3 * m1;
      dst[c] = v;
    }
  }
}

template <class T>
int se3_bspline(const void* data, const void* w, void* out, int64_t nb, int64_t N, int64_t K, void* stream) {
  if (nb < 0 || N < 4 || K < 2 || (N - 3) * K + 1 >= ((int64_t)1 << 31) - 256) return PPLIE_EBADARG;
  if (nb == 0) return PPLIE_OK;
  if (!data || !w || !out || !aligned16(out)) return PPLIE_EBADARG;
  constexpr int BLOCK = 256;
  const int64_t total = nb * ((N - 3) * K + 1);
  const int64_t nt = (total + BLOCK - 1) / BLOCK;
  const int grid = (int)(nt < (1 << 20) ? nt : (1 << 20));
  hipLaunchKernelGGL((se3_bspline_kernel<T, BLOCK>), dim3(grid), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)data, (const T*)w, (T*)out, nb, N, K);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

template <class T>
int chspline(const void* pts, const void* hh, const void* seg, void* out, int64_t nb, int64_t N, int64_t C, int64_t M, void* stream) {
  if (nb < 0 || N < 2 || C < 1 || M < 0) return PPLIE_EBADARG;
  if (nb == 0 || M == 0) return PPLIE_OK;
  if (!pts || !hh || !seg || !out) return PPLIE_EBADARG;
  constexpr int BLOCK = 256;
  const int64_t nt = (M + BLOCK - 1) / BLOCK;
  if (nt > 0x7fffffff) return PPLIE_EBADARG;
  const int64_t want = (int64_t)16384 / nt + 1;                     // enough workgroups to fill the chip, then loop
  const int gy = (int)(nb < want ? nb : (want < 65535 ? want : 65535));
  hipLaunchKernelGGL((chspline_kernel<T, BLOCK>), dim3((unsigned)nt, (unsigned)gy), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)pts, (const T*)hh, (const int64_t*)seg, (T*)out, nb, N, C, M);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}
}  // namespace pplie

extern "C" int pplie_se3_bspline_f32(const void* data, const void* w, void* out, int64_t nb, int64_t N, int64_t K, void* stream) {
  return pplie::se3_bspline<float>(data, w, out, nb, N, K, stream);
}
extern "C" int pplie_se3_bspline_f64(const void* data, const void* w, void* out, int64_t nb, int64_t N, int64_t K, void* stream) {
  return pplie::se3_bspline<double>(data, w, out, nb, N, K, stream);
}
extern "C" int pplie_chspline_f32(const void* pts, const void* hh, const void* seg, void* out, int64_t nb, int64_t N, int64_t C, int64_t M, void* stream) {
  return pplie::chspline<float>(pts, hh, seg, out, nb, N, C, M, stream);
}
extern "C" int pplie_chspline_f64(const void* pts, const void* hh, const void* seg, void* out, int64_t nb, int64_t N, int64_t C, int64_t M, void* stream) {
  return pplie::chspline<double>(pts, hh, seg, out, nb, N, C, M, stream);
}
