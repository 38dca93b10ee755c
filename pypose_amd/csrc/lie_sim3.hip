// lie_sim3.hip -- C-ABI entry points of the sim3 / SIM3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP_OPS(sim3, 7, 8)
PPLIE_EXPORT_GROUP(sim3)
