"""pypose_amd -- MI355X-native (gfx950) implementation of PyPose's batched LieTensor hot path.

Same Python surface as ``pypose`` for the path it covers (``LieTensor``, ``SO3 .. rxso3``,
``randn_*``, ``identity_*``, ``Exp/Log/Inv/Mul/Act/Adj/AdjT/Jinvp/Jr``, ``optim.LM/GN``,
``module.IMUPreintegrator``); all arithmetic runs in hand-written HIP kernels behind a C ABI
(``include/pplie.h``, ``pypose_amd/lib/libpplie.so``).  There is no CPU compute path.
"""
from ._version import __version__
from .lietensor import LieTensor, Parameter, SO3, so3, SE3, se3, Sim3, sim3, RxSO3, rxso3
from .lietensor import randn_like, randn_SE3, randn_SO3, randn_so3, randn_se3
from .lietensor import randn_Sim3, randn_sim3, randn_RxSO3, randn_rxso3
from .lietensor import identity_like, identity_SO3, identity_so3, identity_SE3, identity_se3
from .lietensor import identity_Sim3, identity_sim3, identity_RxSO3, identity_rxso3
from .lietensor import add, add_, mul, Exp, Log, Inv, Mul, Retr, Act, Adj, AdjT, Jinvp, Jr
from .lietensor import SO3_type, so3_type, SE3_type, se3_type
from .lietensor import Sim3_type, sim3_type, RxSO3_type, rxso3_type
from .lietensor import tensor, translation, rotation, scale, matrix, euler, vec2skew
from .lietensor.lietensor import retain_ltype
from .lietensor import mat2SO3, mat2SE3, mat2Sim3, mat2RxSO3, from_matrix, euler2SO3, quat2unit
from .basics import pm, cumops, cummul, cumprod, cumops_, cummul_, cumprod_
from . import autograd
from . import optim
from . import module
from . import io
from . import function
from . import testing
from . import func
from .function import cart2homo, homo2cart, point2pixel, pixel2point, reprojerr, is_lietensor, is_SE3, hasnan, chspline, bspline, svdtf, svdstf, \
    bvv, bmv, bvmv
from . import metric
from .module.loss import geodesic_loss
