/* A C caller of the drop-in boundary: no Python, no torch -- only the HIP runtime and include/pplie.h.
 *   hipcc tests/c_abi/abi_smoke.c -Iinclude -Lpypose_amd/lib -lpplie -Wl,-rpath,$PWD/pypose_amd/lib -o abi_smoke
 * Allocates device buffers, runs SE3 Exp -> Log -> Inv/Mul and one block Cholesky solve on the NULL stream, checks the
 * round trips on the host, and exercises the status codes.  Prints "pplie C ABI OK" and exits 0. */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pplie.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_PP(x) do { int c_ = (x); if (c_ != 0) { fprintf(stderr, "pplie status %d at %s:%d\n", c_, __FILE__, __LINE__); return 3; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f; }

int main(void) {
  const int64_t n = 100003;                  /* not a multiple of the tile size */
  unsigned seed = 12345u;
  float* hx = (float*)malloc(n * 6 * sizeof(float));
  float* hy = (float*)malloc(n * 6 * sizeof(float));
  float* hI = (float*)malloc(n * 7 * sizeof(float));
  for (int64_t i = 0; i < n * 6; ++i) hx[i] = frand(&seed);     /* |phi| <= sqrt(3) < pi: Log(Exp(x)) == x */
  float *dx, *dX, *dy, *dXi, *dI;
  CHECK_HIP(hipMalloc((void**)&dx, n * 6 * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&dX, n * 7 * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&dy, n * 6 * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&dXi, n * 7 * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&dI, n * 7 * sizeof(float)));
  CHECK_HIP(hipMemcpy(dx, hx, n * 6 * sizeof(float), hipMemcpyHostToDevice));
  CHECK_PP(pplie_se3_exp_fwd_f32(dx, NULL, NULL, dX, NULL, n, NULL));
  CHECK_PP(pplie_se3_log_fwd_f32(dX, NULL, NULL, dy, NULL, n, NULL));
  CHECK_PP(pplie_se3_inv_fwd_f32(dX, NULL, NULL, dXi, NULL, n, NULL));
  CHECK_PP(pplie_se3_mul_fwd_f32(dX, dXi, NULL, dI, NULL, n, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(hy, dy, n * 6 * sizeof(float), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(hI, dI, n * 7 * sizeof(float), hipMemcpyDeviceToHost));
  double e1 = 0, e2 = 0;
  for (int64_t i = 0; i < n * 6; ++i) { double d = fabs((double)hy[i] - hx[i]); if (d > e1) e1 = d; }
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < 7; ++k) { double d = fabs((double)hI[i * 7 + k] - (k == 6 ? 1.0 : 0.0)); if (d > e2) e2 = d; }
  printf("Log(Exp(x)) - x: %.3g   X * X^-1 - identity: %.3g\n", e1, e2);
  if (!(e1 < 2e-5 && e2 < 2e-6)) { fprintf(stderr, "round trip out of tolerance\n"); return 4; }
  /* block Cholesky: A x = -g with A = diag(2..7), g = 1 */
  {
    const int64_t nb = 1000;
    float* hA = (float*)calloc(nb * 36, sizeof(float));
    float* hg = (float*)malloc(nb * 6 * sizeof(float));
    float* hs = (float*)malloc(nb * 6 * sizeof(float));
    for (int64_t b = 0; b < nb; ++b)
      for (int i = 0; i < 6; ++i) { hA[b * 36 + i * 7] = 2.0f + i; hg[b * 6 + i] = 1.0f; }
    float *dA, *dg, *ds;
    CHECK_HIP(hipMalloc((void**)&dA, nb * 36 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dg, nb * 6 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&ds, nb * 6 * sizeof(float)));
    CHECK_HIP(hipMemcpy(dA, hA, nb * 36 * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dg, hg, nb * 6 * sizeof(float), hipMemcpyHostToDevice));
    CHECK_PP(pplie_block_chol_solve_f32(dA, dg, ds, nb, 6, NULL));
    CHECK_HIP(hipMemcpy(hs, ds, nb * 6 * sizeof(float), hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < nb; ++b)
      for (int i = 0; i < 6; ++i)
        if (fabs(hs[b * 6 + i] + 1.0 / (2.0 + i)) > 1e-6) { fprintf(stderr, "chol_solve wrong at %ld,%d: %g\n", (long)b, i, hs[b * 6 + i]); return 5; }
  }
  /* status codes: n = 0 is a no-op, negative n and missing buffers are PPLIE_EBADARG, nothing throws */
  if (pplie_se3_exp_fwd_f32(dx, NULL, NULL, dX, NULL, 0, NULL) != PPLIE_OK) return 6;
  if (pplie_se3_exp_fwd_f32(dx, NULL, NULL, dX, NULL, -1, NULL) != PPLIE_EBADARG) return 7;
  if (pplie_se3_exp_fwd_f32(NULL, NULL, NULL, dX, NULL, n, NULL) != PPLIE_EBADARG) return 8;
  if (pplie_se3_mul_fwd_f32(dX, NULL, NULL, dI, NULL, n, NULL) != PPLIE_EBADARG) return 9;
  printf("pplie C ABI OK\n");
  return 0;
}
