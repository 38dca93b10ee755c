// lie_ops.h -- op functors (lie_math.h row functions wrapped for rowmap.h) and the macro
// that exports one group's full op set through the C ABI declared in include/pplie.h.
#pragma once
#include "rowmap.h"

// PPLIE_DEFINE_GROUP_OPS defines functors Op_<g>_<op> for a group with algebra width DA and group
// width DG; PPLIE_EXPORT_GROUP exports them (TileOf specialisations go in between).
#define PPLIE_DEFINE_GROUP_OPS(g, DA, DG)                                               \
  namespace pplie {                                                              \
  /* optimizer update p.add_(d) = Exp(d[:DA]) * p (reference lietensor.py:60-65), d zero-padded to DG */ \
  template <class S> PP_HD void g##_retract(const S* d, const S* X, S* out) {    \
    S E[DG];                                                                     \
    g##_exp<S>(d, E);                                                            \
    g##_mul<S>(E, X, out);                                                       \
  }                                                                              \
  PPLIE_OP_2_1(Op_##g##_retract, g##_retract, DG, DG, DG)                        \
  PPLIE_OP_1_1(Op_##g##_exp_fwd, g##_exp, DA, DG)                                \
  PPLIE_OP_2_1(Op_##g##_exp_bwd, g##_exp_bwd, DA, DG, DA)                        \
  PPLIE_OP_1_1(Op_##g##_log_fwd, g##_log, DG, DA)                                \
  PPLIE_OP_2_1(Op_##g##_log_bwd, g##_log_bwd, DA, DA, DG)                        \
  PPLIE_OP_1_1(Op_##g##_inv_fwd, g##_inv, DG, DG)                                \
  PPLIE_OP_2_1(Op_##g##_inv_bwd, g##_inv_bwd, DG, DG, DG)                        \
  PPLIE_OP_2_1(Op_##g##_mul_fwd, g##_mul, DG, DG, DG)                            \
  PPLIE_OP_2_2(Op_##g##_mul_bwd, g##_mul_bwd, DG, DG, DG, DG)                    \
  PPLIE_OP_2_1(Op_##g##_act_fwd, g##_act, DG, 3, 3)                              \
  PPLIE_OP_3_2(Op_##g##_act_bwd, g##_act_bwd, DG, 3, 3, DG, 3)                   \
  PPLIE_OP_2_1(Op_##g##_act4_fwd, g##_act4, DG, 4, 4)                            \
  PPLIE_OP_3_2(Op_##g##_act4_bwd, g##_act4_bwd, DG, 4, 4, DG, 4)                 \
  PPLIE_OP_2_1(Op_##g##_adj_fwd, g##_adj, DG, DA, DA)                            \
  PPLIE_OP_3_2(Op_##g##_adj_bwd, g##_adj_bwd, DG, DA, DA, DG, DA)                \
  PPLIE_OP_2_1(Op_##g##_adjt_fwd, g##_adjt, DG, DA, DA)                          \
  PPLIE_OP_3_2(Op_##g##_adjt_bwd, g##_adjt_bwd, DG, DA, DA, DG, DA)              \
  PPLIE_OP_2_1(Op_##g##_jinvp_fwd, g##_jinvp, DG, DA, DA)                        \
  PPLIE_OP_3_2(Op_##g##_jinvp_bwd, g##_jinvp_bwd, DG, DA, DA, DG, DA)            \
  }

#define PPLIE_EXPORT_GROUP(g)                                                    \
  PPLIE_EXPORT(pplie_##g##_retract, pplie::Op_##g##_retract)                     \
  PPLIE_EXPORT(pplie_##g##_exp_fwd, pplie::Op_##g##_exp_fwd)                     \
  PPLIE_EXPORT(pplie_##g##_exp_bwd, pplie::Op_##g##_exp_bwd)                     \
  PPLIE_EXPORT(pplie_##g##_log_fwd, pplie::Op_##g##_log_fwd)                     \
  PPLIE_EXPORT(pplie_##g##_log_bwd, pplie::Op_##g##_log_bwd)                     \
  PPLIE_EXPORT(pplie_##g##_inv_fwd, pplie::Op_##g##_inv_fwd)                     \
  PPLIE_EXPORT(pplie_##g##_inv_bwd, pplie::Op_##g##_inv_bwd)                     \
  PPLIE_EXPORT(pplie_##g##_mul_fwd, pplie::Op_##g##_mul_fwd)                     \
  PPLIE_EXPORT(pplie_##g##_mul_bwd, pplie::Op_##g##_mul_bwd)                     \
  PPLIE_EXPORT(pplie_##g##_act_fwd, pplie::Op_##g##_act_fwd)                     \
  PPLIE_EXPORT(pplie_##g##_act_bwd, pplie::Op_##g##_act_bwd)                     \
  PPLIE_EXPORT(pplie_##g##_act4_fwd, pplie::Op_##g##_act4_fwd)                   \
  PPLIE_EXPORT(pplie_##g##_act4_bwd, pplie::Op_##g##_act4_bwd)                   \
  PPLIE_EXPORT(pplie_##g##_adj_fwd, pplie::Op_##g##_adj_fwd)                     \
  PPLIE_EXPORT(pplie_##g##_adj_bwd, pplie::Op_##g##_adj_bwd)                     \
  PPLIE_EXPORT(pplie_##g##_adjt_fwd, pplie::Op_##g##_adjt_fwd)                   \
  PPLIE_EXPORT(pplie_##g##_adjt_bwd, pplie::Op_##g##_adjt_bwd)                   \
  PPLIE_EXPORT(pplie_##g##_jinvp_fwd, pplie::Op_##g##_jinvp_fwd)                 \
  PPLIE_EXPORT(pplie_##g##_jinvp_bwd, pplie::Op_##g##_jinvp_bwd)                 \
  /* the Function backwards with a broadcast cotangent (rowmap.h GB): first node of op(x).sum().backward() */ \
  PPLIE_EXPORT_GB(pplie_##g##_exp_bwd, pplie::Op_##g##_exp_bwd)                  \
  PPLIE_EXPORT_GB(pplie_##g##_log_bwd, pplie::Op_##g##_log_bwd)                  \
  PPLIE_EXPORT_GB(pplie_##g##_inv_bwd, pplie::Op_##g##_inv_bwd)                  \
  PPLIE_EXPORT_GB(pplie_##g##_mul_bwd, pplie::Op_##g##_mul_bwd)                  \
  PPLIE_EXPORT_GB(pplie_##g##_act_bwd, pplie::Op_##g##_act_bwd)                  \
  PPLIE_EXPORT_GB(pplie_##g##_act4_bwd, pplie::Op_##g##_act4_bwd)                \
  PPLIE_EXPORT_GB(pplie_##g##_adj_bwd, pplie::Op_##g##_adj_bwd)                  \
  PPLIE_EXPORT_GB(pplie_##g##_adjt_bwd, pplie::Op_##g##_adjt_bwd)
