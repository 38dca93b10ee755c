from .geometry import cart2homo, homo2cart, point2pixel, pixel2point, reprojerr
