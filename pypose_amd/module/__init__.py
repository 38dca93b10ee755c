from .imu_preintegrator import IMUPreintegrator
from .loss import GeodesicLoss, geodesic_loss
