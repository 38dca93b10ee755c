// lie_se3.hip -- C-ABI entry points of the se3 / SE3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP_OPS(se3, 6, 7)
// tile shapes measured on MI355X at 10M rows (profiles/r01/tune_rowmap_v2.json, tune_general.json)
namespace pplie {
PPLIE_TILE(Op_se3_log_fwd, 4)
PPLIE_TILE(Op_se3_exp_bwd, 4)
PPLIE_TILE(Op_se3_log_bwd, 2)
// (round 4, tools/tune_general.py at 10 M rows: jinvp_fwd 135.0 us as 256 x 1 -> 127.0 as 128 x 4 rolled; adj_fwd 124.9 -> 121.0 as 256 x 2)
PPLIE_TILE_EX(Op_se3_jinvp_fwd, 4, 128, true)
PPLIE_TILE(Op_se3_adj_fwd, 2)
// (second sweep, profiles/r04/tune_general_all.json: 128-lane workgroups for the two-slab ops with 7-wide outputs, 3-6 %)
PPLIE_TILE_EX(Op_se3_jinvp_bwd, 1, 128, false)
PPLIE_TILE_EX(Op_se3_mul_fwd, 4, 128, false)
PPLIE_TILE_EX(Op_se3_inv_bwd, 4, 128, false)
PPLIE_TILE(Op_se3_adjt_fwd, 2)
PPLIE_TILE(Op_se3_act_fwd, 1)
// (round 5, fp64: profiles/r05/tune_general_f64.json)
PPLIE_TILE64(Op_se3_log_fwd, 2, 128, false)        // 0.2480 -> 0.1634 ms
}
PPLIE_EXPORT_GROUP(se3)
