// lie_so3.hip -- C-ABI entry points of the so3 / SO3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP_OPS(so3, 3, 4)
PPLIE_EXPORT_GROUP(so3)
// so3.Jr (reference lietensor.py:343-351): [N,3] -> [N,9] row-major right Jacobians
namespace pplie { PPLIE_OP_1_1(Op_so3_jr_fwd, so3_jr, 3, 9) }
PPLIE_EXPORT(pplie_so3_jr_fwd, pplie::Op_so3_jr_fwd)
namespace pplie { PPLIE_OP_2_1(Op_so3_jr_bwd, so3_jr_bwd, 3, 9, 3) }
PPLIE_EXPORT(pplie_so3_jr_bwd, pplie::Op_so3_jr_bwd)
