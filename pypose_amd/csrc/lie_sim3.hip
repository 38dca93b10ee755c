// lie_sim3.hip -- C-ABI entry points of the sim3 / SIM3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP_OPS(sim3, 7, 8)
// tile shapes measured at 10 M rows (profiles/r04/tune_general_all.json)
namespace pplie {
PPLIE_TILE_EX(Op_sim3_exp_bwd, 4, 128, false)
PPLIE_TILE_EX(Op_sim3_log_bwd, 4, 128, false)
// (round 5, fp64: profiles/r05/tune_general_f64.json)
PPLIE_TILE64(Op_sim3_exp_fwd, 2, 128, false)       // 0.2977 -> 0.2198 ms
PPLIE_TILE64(Op_sim3_log_fwd, 2, 256, false)       // 0.3114 -> 0.2606
}
PPLIE_EXPORT_GROUP(sim3)
