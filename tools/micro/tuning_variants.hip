// tuning_variants.hip -- launch-shape variants of the SE3 Exp/Log kernels and a plain
// dwordx4 copy, exported for tools/tune_rowmap.py (A/B measurements on the GPU box).
// Not part of the product API (not declared in include/pplie.h; symbols are pplie_var_*).
#include "lie_ops.h"

namespace pplie {
PPLIE_OP_1_1(Var_se3_exp, se3_exp, 6, 7)
PPLIE_OP_1_1(Var_se3_log, se3_log, 7, 6)

template <class Op>
int var_dispatch(int path, int rpt, int grid_cap, const void* x, void* y, int64_t n, void* stream) {
  if (path == 1) return launch_rowmap_direct<float, Op>(x, nullptr, nullptr, y, nullptr, n, stream, grid_cap);
  switch (rpt) {
    case 1: return launch_rowmap<float, Op, 1, 256>(x, nullptr, nullptr, y, nullptr, n, stream, grid_cap);
    case 2: return launch_rowmap<float, Op, 2, 256>(x, nullptr, nullptr, y, nullptr, n, stream, grid_cap);
    case 4: return launch_rowmap<float, Op, 4, 256>(x, nullptr, nullptr, y, nullptr, n, stream, grid_cap);
    case 8: return launch_rowmap<float, Op, 8, 256>(x, nullptr, nullptr, y, nullptr, n, stream, grid_cap);
  }
  return PPLIE_EBADARG;
}

__global__ void __launch_bounds__(256) copy16_kernel(const raw16* __restrict__ src, raw16* __restrict__ dst, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
}  // namespace pplie

extern "C" int pplie_var_se3_exp_f32(int path, int rpt, int grid_cap, const void* x, void* y, int64_t n, void* stream) {
  return pplie::var_dispatch<pplie::Var_se3_exp<float>>(path, rpt, grid_cap, x, y, n, stream);
}
extern "C" int pplie_var_se3_log_f32(int path, int rpt, int grid_cap, const void* x, void* y, int64_t n, void* stream) {
  return pplie::var_dispatch<pplie::Var_se3_log<float>>(path, rpt, grid_cap, x, y, n, stream);
}

// general (multi-slab) variants: rpt in {1,2,4} x block in {128,256}
namespace pplie {
PPLIE_OP_2_1(Var_se3_exp_bwd, se3_exp_bwd, 6, 7, 6)
PPLIE_OP_2_1(Var_se3_log_bwd, se3_log_bwd, 6, 6, 7)
PPLIE_OP_2_1(Var_se3_mul_fwd, se3_mul, 7, 7, 7)
PPLIE_OP_2_2(Var_se3_mul_bwd, se3_mul_bwd, 7, 7, 7, 7)
PPLIE_OP_3_2(Var_se3_act_bwd, se3_act_bwd, 7, 3, 3, 7, 3)
PPLIE_OP_2_1(Var_se3_jinvp_fwd, se3_jinvp, 7, 6, 6)
PPLIE_OP_2_1(Var_se3_adj_fwd, se3_adj, 7, 6, 6)
PPLIE_OP_3_2(Var_se3_jinvp_bwd, se3_jinvp_bwd, 7, 6, 6, 7, 6)
PPLIE_OP_1_1(Var_sim3_exp_fwd, sim3_exp, 7, 8)
PPLIE_OP_1_1(Var_sim3_log_fwd, sim3_log, 8, 7)
PPLIE_OP_2_1(Var_sim3_exp_bwd, sim3_exp_bwd, 7, 8, 7)
PPLIE_OP_2_1(Var_sim3_log_bwd, sim3_log_bwd, 7, 7, 8)
PPLIE_OP_1_1(Var_se3_exp_fwd, se3_exp, 6, 7)
PPLIE_OP_1_1(Var_se3_log_fwd, se3_log, 7, 6)
PPLIE_OP_2_1(Var_se3_act_fwd, se3_act, 7, 3, 3)
PPLIE_OP_2_1(Var_se3_inv_bwd, se3_inv_bwd, 7, 7, 7)
PPLIE_OP_1_1(Var_so3_exp_fwd, so3_exp, 3, 4)
PPLIE_OP_1_1(Var_so3_log_fwd, so3_log, 4, 3)
PPLIE_OP_1_1(Var_rxso3_exp_fwd, rxso3_exp, 4, 5)
PPLIE_OP_1_1(Var_rxso3_log_fwd, rxso3_log, 5, 4)
PPLIE_OP_2_1(Var_rxso3_mul_fwd, rxso3_mul, 5, 5, 5)
template <class Op, class float_t = float>
int var_general(int rpt, int block, const void* a, const void* b, const void* c, void* o, void* p, int64_t n, void* st) {
  // block: 256 / 128 = unrolled rows ; 1256 / 1128 = rolled rows (one row's registers at a time)
  if (block == 256) {
    if (rpt == 1) return launch_rowmap<float_t, Op, 1, 256>(a, b, c, o, p, n, st);
    if (rpt == 2) return launch_rowmap<float_t, Op, 2, 256>(a, b, c, o, p, n, st);
    if (rpt == 4) return launch_rowmap<float_t, Op, 4, 256>(a, b, c, o, p, n, st);
  } else if (block == 128) {
    if (rpt == 1) return launch_rowmap<float_t, Op, 1, 128>(a, b, c, o, p, n, st);
    if (rpt == 2) return launch_rowmap<float_t, Op, 2, 128>(a, b, c, o, p, n, st);
    if (rpt == 4) return launch_rowmap<float_t, Op, 4, 128>(a, b, c, o, p, n, st);
  } else if (block == 1256) {
    if (rpt == 2) return launch_rowmap<float_t, Op, 2, 256, true>(a, b, c, o, p, n, st);
    if (rpt == 4) return launch_rowmap<float_t, Op, 4, 256, true>(a, b, c, o, p, n, st);
  } else if (block == 1128) {
    if (rpt == 2) return launch_rowmap<float_t, Op, 2, 128, true>(a, b, c, o, p, n, st);
    if (rpt == 4) return launch_rowmap<float_t, Op, 4, 128, true>(a, b, c, o, p, n, st);
  }
  return PPLIE_EBADARG;
}
}  // namespace pplie
#define PPLIE_VAR_GENERAL(NAME)                                                                                      \
  extern "C" int pplie_var_##NAME##_f32(int rpt, int block, const void* a, const void* b, const void* c, void* o,    \
                                        void* p, int64_t n, void* st) {                                              \
    return pplie::var_general<pplie::Var_##NAME<float>>(rpt, block, a, b, c, o, p, n, st);                           \
  }
#define PPLIE_VAR_GENERAL64(NAME)                                                                                    \
  extern "C" int pplie_var_##NAME##_f64(int rpt, int block, const void* a, const void* b, const void* c, void* o,    \
                                        void* p, int64_t n, void* st) {                                              \
    return pplie::var_general<pplie::Var_##NAME<double>, double>(rpt, block, a, b, c, o, p, n, st);                  \
  }
PPLIE_VAR_GENERAL(so3_exp_fwd)
PPLIE_VAR_GENERAL(rxso3_mul_fwd)
PPLIE_VAR_GENERAL64(so3_exp_fwd)
PPLIE_VAR_GENERAL64(so3_log_fwd)
PPLIE_VAR_GENERAL64(se3_exp_fwd)
PPLIE_VAR_GENERAL64(se3_log_fwd)
PPLIE_VAR_GENERAL64(sim3_exp_fwd)
PPLIE_VAR_GENERAL64(sim3_log_fwd)
PPLIE_VAR_GENERAL64(rxso3_exp_fwd)
PPLIE_VAR_GENERAL64(rxso3_log_fwd)
PPLIE_VAR_GENERAL(se3_exp_bwd)
PPLIE_VAR_GENERAL(se3_log_bwd)
PPLIE_VAR_GENERAL(se3_mul_fwd)
PPLIE_VAR_GENERAL(se3_mul_bwd)
PPLIE_VAR_GENERAL(se3_act_bwd)
PPLIE_VAR_GENERAL(se3_jinvp_fwd)
PPLIE_VAR_GENERAL(se3_adj_fwd)
PPLIE_VAR_GENERAL(se3_jinvp_bwd)
PPLIE_VAR_GENERAL(sim3_exp_fwd)
PPLIE_VAR_GENERAL(sim3_log_fwd)
PPLIE_VAR_GENERAL(sim3_exp_bwd)
PPLIE_VAR_GENERAL(sim3_log_bwd)
PPLIE_VAR_GENERAL(se3_exp_fwd)
PPLIE_VAR_GENERAL(se3_log_fwd)
PPLIE_VAR_GENERAL(se3_act_fwd)
PPLIE_VAR_GENERAL(se3_inv_bwd)

// device-copy ceiling: nbytes must be a multiple of 16
extern "C" int pplie_var_copy(const void* src, void* dst, int64_t nbytes, int grid, void* stream) {
  hipLaunchKernelGGL(pplie::copy16_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     static_cast<const pplie::raw16*>(src), static_cast<pplie::raw16*>(dst), nbytes / 16);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
