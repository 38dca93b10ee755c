from .lietensor import LieTensor, Parameter, retain_ltype
from .lietensor import SO3_type, so3_type, SE3_type, se3_type, Sim3_type, sim3_type, RxSO3_type, rxso3_type
from .utils import (randn_like, randn_SE3, randn_SO3, randn_so3, randn_se3, randn_Sim3, randn_sim3, randn_RxSO3,
                    randn_rxso3, identity_like, identity_SO3, identity_so3, identity_SE3, identity_se3, identity_Sim3,
                    identity_sim3, identity_RxSO3, identity_rxso3, SO3, so3, SE3, se3, Sim3, sim3, RxSO3, rxso3,
                    Exp, Log, Inv, Mul, Retr, Act, Adj, AdjT, Jinvp, Jr, tensor, translation, rotation, scale, matrix,
                    euler)
from .basics import vec2skew, add, add_, mul
from .convert import mat2SO3, mat2SE3, mat2Sim3, mat2RxSO3, from_matrix, euler2SO3, quat2unit
