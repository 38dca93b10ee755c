// robust.hip -- robust re-weighting of residual rows and their Jacobian rows in one pass.
//
// Reference: pypose/optim/corrector.py:69-96 (FastTriggs) and :132-167 (Triggs) compute rho'(|r_i|^2) by autograd -- a graph
// over kernel(x).sum() built and differentiated every step -- then `s * R` and `s.expand_as(R).reshape(-1, 1) * J` with
// element-wise launches; the kernels themselves (pypose/optim/kernel.py) are 3-6 element-wise launches each, Huber with two
// boolean-mask writes (a host synchronisation).  Here:
//   pplie_robust_scale_rows   R [n, dr], J [n, w] -> Rout = s R, Jout = s J with s_i = sqrt(rho'(|R_i|^2)) from robust.h's closed
//                             forms: ONE launch, J read once and written once (Jout may be J), whatever the inner layout of a
//                             row of J (dense rows, [dr, cols] blocks, [K, dr, m] edge blocks: the scale is per row i)
//   pplie_robust_rho          out = rho(x) element-wise (the model's loss sum_i rho(|r_i|^2), optimizer.py:118-125)
// HBM-bound: 2 x 4 w + 8 dr bytes per row.
#include "rowmap.h"
#include "robust.h"

namespace pplie {

constexpr int kMaxDr = 64;

template <class T, int VEC>
__global__ void __launch_bounds__(256)
robust_scale_rows_kernel(const T* __restrict__ R, T* __restrict__ Rout, const T* J, T* Jout, int64_t n, int dr, int64_t w,
                         RobustParam<T> rk) {
  // one thread per VEC consecutive scalars of J (VEC divides w: they share their row); the thread of a row's first scalars also
  // writes the row of Rout
  const int64_t wv = w / VEC, total = n * wv;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t row = t / wv;
    const T* r = R + row * dr;
    T x = T(0);
    for (int k = 0; k < dr; ++k) x += r[k] * r[k];
    const T s = pp_sqrt(robust_rho1<T>(rk, x));
    const int64_t e = t * VEC;
    T v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = J[e + k];
#pragma unroll
    for (int k = 0; k < VEC; ++k) Jout[e + k] = s * v[k];
    if (t - row * wv == 0) {
      for (int k = 0; k < dr; ++k) Rout[row * dr + k] = s * r[k];
    }
  }
}

template <class T>
__global__ void __launch_bounds__(256)
robust_rho_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t n, RobustParam<T> rk) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) out[t] = robust_rho<T>(rk, x[t]);
}

template <class T>
int robust_scale_rows(const void* R, void* Rout, const void* J, void* Jout, int64_t n, int dr, int64_t w, int kind, double p0, double p1,
                      void* stream) {
  if (n < 0 || dr <= 0 || dr > kMaxDr || w <= 0 || kind < 0 || kind > RK_TOLERANT) return PPLIE_EBADARG;
  if (n == 0) return PPLIE_OK;
  if (!R || !Rout || !J || !Jout || R == Rout) return PPLIE_EBADARG;       // (Rout must be a separate buffer: rows of R are re-read)
  const RobustParam<T> rk{kind, (T)p0, (T)p1};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool v4 = w % 4 == 0 && aligned16(J) && aligned16(Jout);
  const int64_t threads = n * (v4 ? w / 4 : w);
  const int64_t nb = (threads + 255) / 256;
  const unsigned grid = (unsigned)(nb < (1 << 22) ? nb : (1 << 22));
  if (v4)
    hipLaunchKernelGGL((robust_scale_rows_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T*)R, (T*)Rout, (const T*)J, (T*)Jout, n, dr,
                       w, rk);
  else
    hipLaunchKernelGGL((robust_scale_rows_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const T*)R, (T*)Rout, (const T*)J, (T*)Jout, n, dr,
                       w, rk);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

template <class T> int robust_rho_launch(const void* x, void* out, int64_t n, int kind, double p0, double p1, void* stream) {
  if (n < 0 || kind < 0 || kind > RK_TOLERANT) return PPLIE_EBADARG;
  if (n == 0) return PPLIE_OK;
  if (!x || !out) return PPLIE_EBADARG;
  const int64_t nb = (n + 255) / 256;
  hipLaunchKernelGGL((robust_rho_kernel<T>), dim3((unsigned)(nb < (1 << 22) ? nb : (1 << 22))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const T*)x, (T*)out, n, RobustParam<T>{kind, (T)p0, (T)p1});
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

}  // namespace pplie

extern "C" int pplie_robust_scale_rows_f32(const void* R, void* Rout, const void* J, void* Jout, int64_t n, int dr, int64_t w, int kind,
                                           double p0, double p1, void* stream) {
  return pplie::robust_scale_rows<float>(R, Rout, J, Jout, n, dr, w, kind, p0, p1, stream);
}
extern "C" int pplie_robust_scale_rows_f64(const void* R, void* Rout, const void* J, void* Jout, int64_t n, int dr, int64_t w, int kind,
                                           double p0, double p1, void* stream) {
  return pplie::robust_scale_rows<double>(R, Rout, J, Jout, n, dr, w, kind, p0, p1, stream);
}
extern "C" int pplie_robust_rho_f32(const void* x, void* out, int64_t n, int kind, double p0, double p1, void* stream) {
  return pplie::robust_rho_launch<float>(x, out, n, kind, p0, p1, stream);
}
extern "C" int pplie_robust_rho_f64(const void* x, void* out, int64_t n, int kind, double p0, double p1, void* stream) {
  return pplie::robust_rho_launch<double>(x, out, n, kind, p0, p1, stream);
}
