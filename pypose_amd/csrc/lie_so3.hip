// lie_so3.hip -- C-ABI entry points of the so3 / SO3 op set (include/pplie.h).
#include "lie_ops.h"
// last argument: rows per lane of the fp32 log_fwd tile (tuned on MI355X, profiles/r01)
PPLIE_DEFINE_GROUP(so3, 3, 4, 2)
// so3.Jr (reference lietensor.py:343-351): [N,3] -> [N,9] row-major right Jacobians
namespace pplie { PPLIE_OP_1_1(Op_so3_jr_fwd, so3_jr, 3, 9) }
PPLIE_EXPORT(pplie_so3_jr_fwd, pplie::Op_so3_jr_fwd)
