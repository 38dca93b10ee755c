// lie_rxso3.hip -- C-ABI entry points of the rxso3 / RXSO3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP_OPS(rxso3, 4, 5)
// tile shapes measured at 10 M rows (round 5: profiles/r05/tune_general_f32.json, tune_general_f64.json)
namespace pplie {
PPLIE_TILE_EX(Op_rxso3_mul_fwd, 4, 128, true)      // fp32 0.1104 -> 0.0949 ms
PPLIE_TILE64(Op_rxso3_log_fwd, 2, 128, true)       // fp64 0.1734 -> 0.1139
PPLIE_TILE64(Op_rxso3_exp_fwd, 1, 128, false)      // fp64 0.1173 -> 0.1137
}
PPLIE_EXPORT_GROUP(rxso3)
