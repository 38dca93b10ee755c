// lie_se3.hip -- C-ABI entry points of the se3 / SE3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP_OPS(se3, 6, 7)
// tile shapes measured on MI355X at 10M rows (profiles/r01/tune_rowmap_v2.json, tune_general.json)
namespace pplie {
PPLIE_TILE(Op_se3_log_fwd, 4)
PPLIE_TILE(Op_se3_exp_bwd, 4)
PPLIE_TILE(Op_se3_log_bwd, 2)
}
PPLIE_EXPORT_GROUP(se3)
