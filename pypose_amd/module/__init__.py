from .imu_preintegrator import IMUPreintegrator
