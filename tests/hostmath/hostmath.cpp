// hostmath.cpp -- TEST INFRASTRUCTURE.  Compiles pypose_amd/csrc/lie_math.h (the exact per-row
// arithmetic the HIP kernels run) with g++ so the arithmetic can be checked against the oracle
// and the reference's golden vectors in the CPU-only build container.  Never loaded by the
// product (pypose_amd has no CPU path); only tests/test_hostmath.py builds and loads it.
#include <stdint.h>
#include "lie_math.h"

using namespace pplie;

#define HM_1_1(NAME, FN, I0, O0)                                                                         \
  template <class T> static void NAME(const T* a, const T*, const T*, T* o, T*, int64_t n) {             \
    for (int64_t i = 0; i < n; ++i) FN<T>(a + i * I0, o + i * O0);                                       \
  }
#define HM_2_1(NAME, FN, I0, I1, O0)                                                                     \
  template <class T> static void NAME(const T* a, const T* b, const T*, T* o, T*, int64_t n) {           \
    for (int64_t i = 0; i < n; ++i) FN<T>(a + i * I0, b + i * I1, o + i * O0);                           \
  }
#define HM_2_2(NAME, FN, I0, I1, O0, O1)                                                                 \
  template <class T> static void NAME(const T* a, const T* b, const T*, T* o, T* p, int64_t n) {         \
    for (int64_t i = 0; i < n; ++i) FN<T>(a + i * I0, b + i * I1, o + i * O0, p + i * O1);               \
  }
#define HM_3_2(NAME, FN, I0, I1, I2, O0, O1)                                                             \
  template <class T> static void NAME(const T* a, const T* b, const T* c, T* o, T* p, int64_t n) {       \
    for (int64_t i = 0; i < n; ++i) FN<T>(a + i * I0, b + i * I1, c + i * I2, o + i * O0, p + i * O1);   \
  }
#define HM_EXPORT(SYM, NAME)                                                                                     \
  extern "C" void hm_##SYM##_f32(const float* a, const float* b, const float* c, float* o, float* p, int64_t n) { \
    NAME<float>(a, b, c, o, p, n);                                                                               \
  }                                                                                                              \
  extern "C" void hm_##SYM##_f64(const double* a, const double* b, const double* c, double* o, double* p,        \
                                 int64_t n) {                                                                    \
    NAME<double>(a, b, c, o, p, n);                                                                              \
  }

#define HM_GROUP(g, DA, DG)                                        \
  HM_1_1(k_##g##_exp_fwd, g##_exp, DA, DG)                         \
  HM_2_1(k_##g##_exp_bwd, g##_exp_bwd, DA, DG, DA)                 \
  HM_1_1(k_##g##_log_fwd, g##_log, DG, DA)                         \
  HM_2_1(k_##g##_log_bwd, g##_log_bwd, DA, DA, DG)                 \
  HM_1_1(k_##g##_inv_fwd, g##_inv, DG, DG)                         \
  HM_2_1(k_##g##_inv_bwd, g##_inv_bwd, DG, DG, DG)                 \
  HM_2_1(k_##g##_mul_fwd, g##_mul, DG, DG, DG)                     \
  HM_2_2(k_##g##_mul_bwd, g##_mul_bwd, DG, DG, DG, DG)             \
  HM_2_1(k_##g##_act_fwd, g##_act, DG, 3, 3)                       \
  HM_3_2(k_##g##_act_bwd, g##_act_bwd, DG, 3, 3, DG, 3)            \
  HM_2_1(k_##g##_act4_fwd, g##_act4, DG, 4, 4)                     \
  HM_3_2(k_##g##_act4_bwd, g##_act4_bwd, DG, 4, 4, DG, 4)          \
  HM_2_1(k_##g##_adj_fwd, g##_adj, DG, DA, DA)                     \
  HM_3_2(k_##g##_adj_bwd, g##_adj_bwd, DG, DA, DA, DG, DA)         \
  HM_2_1(k_##g##_adjt_fwd, g##_adjt, DG, DA, DA)                   \
  HM_3_2(k_##g##_adjt_bwd, g##_adjt_bwd, DG, DA, DA, DG, DA)       \
  HM_2_1(k_##g##_jinvp_fwd, g##_jinvp, DG, DA, DA)                 \
  HM_3_2(k_##g##_jinvp_bwd, g##_jinvp_bwd, DG, DA, DA, DG, DA)     \
  HM_EXPORT(g##_exp_fwd, k_##g##_exp_fwd) HM_EXPORT(g##_exp_bwd, k_##g##_exp_bwd)       \
  HM_EXPORT(g##_log_fwd, k_##g##_log_fwd) HM_EXPORT(g##_log_bwd, k_##g##_log_bwd)       \
  HM_EXPORT(g##_inv_fwd, k_##g##_inv_fwd) HM_EXPORT(g##_inv_bwd, k_##g##_inv_bwd)       \
  HM_EXPORT(g##_mul_fwd, k_##g##_mul_fwd) HM_EXPORT(g##_mul_bwd, k_##g##_mul_bwd)       \
  HM_EXPORT(g##_act_fwd, k_##g##_act_fwd) HM_EXPORT(g##_act_bwd, k_##g##_act_bwd)       \
  HM_EXPORT(g##_act4_fwd, k_##g##_act4_fwd) HM_EXPORT(g##_act4_bwd, k_##g##_act4_bwd)   \
  HM_EXPORT(g##_adj_fwd, k_##g##_adj_fwd) HM_EXPORT(g##_adj_bwd, k_##g##_adj_bwd)       \
  HM_EXPORT(g##_adjt_fwd, k_##g##_adjt_fwd) HM_EXPORT(g##_adjt_bwd, k_##g##_adjt_bwd)   \
  HM_EXPORT(g##_jinvp_fwd, k_##g##_jinvp_fwd) HM_EXPORT(g##_jinvp_bwd, k_##g##_jinvp_bwd)

HM_GROUP(so3, 3, 4)
HM_GROUP(se3, 6, 7)
HM_GROUP(sim3, 7, 8)
HM_GROUP(rxso3, 4, 5)
HM_1_1(k_so3_jr_fwd, so3_jr, 3, 9)
HM_EXPORT(so3_jr_fwd, k_so3_jr_fwd)
HM_2_1(k_so3_jr_bwd, so3_jr_bwd, 3, 9, 3)
HM_EXPORT(so3_jr_bwd, k_so3_jr_bwd)
