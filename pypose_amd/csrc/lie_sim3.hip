// lie_sim3.hip -- C-ABI entry points of the sim3 / SIM3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP_OPS(sim3, 7, 8)
// tile shapes measured at 10 M rows (profiles/r04/tune_general_all.json)
namespace pplie {
PPLIE_TILE_EX(Op_sim3_exp_bwd, 4, 128, false)
PPLIE_TILE_EX(Op_sim3_log_bwd, 4, 128, false)
}
PPLIE_EXPORT_GROUP(sim3)
