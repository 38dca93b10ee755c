// lie_se3.hip -- C-ABI entry points of the se3 / SE3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP(se3, 6, 7)
