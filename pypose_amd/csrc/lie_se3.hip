// lie_se3.hip -- C-ABI entry points of the se3 / SE3 op set (include/pplie.h).
#include "lie_ops.h"
// last argument: rows per lane of the fp32 log_fwd tile (tuned on MI355X, profiles/r01)
PPLIE_DEFINE_GROUP(se3, 6, 7, 4)
