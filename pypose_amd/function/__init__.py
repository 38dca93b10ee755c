from .geometry import cart2homo, homo2cart, point2pixel, pixel2point, reprojerr, svdtf, svdstf
from .checking import is_lietensor, is_SE3, hasnan
from .spline import chspline, bspline
from .linalg import bvv, bmv, bvmv
