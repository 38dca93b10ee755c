"""IMU pre-integration on the MI355X -- the workflow of the reference's IMUPreintegrator (pypose/module/
imu_preintegrator.py) with `import pypose_amd as pp`: B sequences x F steps of (dt, gyro, acc) -> rotation, velocity,
position and the 9x9 covariance, one fused kernel for the state and one for the covariance.

    python examples/imu.py --batch 4096 --steps 1024
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pypose_amd as pp


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--no-cov", dest="cov", action="store_false")
    a = ap.parse_args(argv)
    B, F, dev = a.batch, a.steps, a.device
    torch.manual_seed(0)
    dt = torch.full((B, F, 1), 0.005, device=dev)
    gyro = 0.1 * torch.randn(B, F, 3, device=dev)
    acc = torch.randn(B, F, 3, device=dev) + torch.tensor([0., 0., 9.81], device=dev)
    integrator = pp.module.IMUPreintegrator(prop_cov=a.cov, reset=True).to(dev)
    out = integrator(dt=dt, gyro=gyro, acc=acc)                       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = integrator(dt=dt, gyro=gyro, acc=acc)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"{B} x {F} steps in {ms:.3f} ms = {B * F / ms * 1e3:.3g} steps/s; rot {tuple(out['rot'].shape)}, "
          f"vel {tuple(out['vel'].shape)}, pos {tuple(out['pos'].shape)}" + (f", cov {tuple(out['cov'].shape)}" if a.cov else ""))
    return out


if __name__ == "__main__":
    main()
