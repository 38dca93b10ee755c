// lie_so3.hip -- C-ABI entry points of the so3 / SO3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP_OPS(so3, 3, 4)
// tile shapes measured at 10 M rows (round 5: profiles/r05/tune_general_f32.json, tune_general_f64.json)
namespace pplie {
PPLIE_TILE(Op_so3_exp_fwd, 4)                      // fp32 0.0500 -> 0.0459 ms
PPLIE_TILE64(Op_so3_exp_fwd, 2, 256, false)        // fp64 0.1122 -> 0.0885 ms
PPLIE_TILE64(Op_so3_log_fwd, 2, 256, false)        // fp64 0.0947 -> 0.0879
}
PPLIE_EXPORT_GROUP(so3)
// so3.Jr (reference lietensor.py:343-351): [N,3] -> [N,9] row-major right Jacobians
namespace pplie { PPLIE_OP_1_1(Op_so3_jr_fwd, so3_jr, 3, 9) }
PPLIE_EXPORT(pplie_so3_jr_fwd, pplie::Op_so3_jr_fwd)
namespace pplie { PPLIE_OP_2_1(Op_so3_jr_bwd, so3_jr_bwd, 3, 9, 3) }
PPLIE_EXPORT(pplie_so3_jr_bwd, pplie::Op_so3_jr_bwd)
