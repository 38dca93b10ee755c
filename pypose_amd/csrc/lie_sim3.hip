// lie_sim3.hip -- C-ABI entry points of the sim3 / SIM3 op set (include/pplie.h).
#include "lie_ops.h"
// last argument: rows per lane of the fp32 log_fwd tile (tuned on MI355X, profiles/r01)
PPLIE_DEFINE_GROUP(sim3, 7, 8, 2)
