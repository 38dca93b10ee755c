// lie_rxso3.hip -- C-ABI entry points of the rxso3 / RXSO3 op set (include/pplie.h).
#include "lie_ops.h"
PPLIE_DEFINE_GROUP_OPS(rxso3, 4, 5)
PPLIE_EXPORT_GROUP(rxso3)
