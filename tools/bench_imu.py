"""C5 (BASELINE configs[4]): IMUPreintegrator 4096 sequences x 1024 steps, fp32 -- `python tools/bench_imu.py [B F]`."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
B, F = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 1024)
dev = "cuda"


def med_ms(f, reps=20, inner=4):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            f()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    return sorted(ts)[len(ts) // 2]


torch.manual_seed(0)
dt = torch.full((B, F, 1), 0.005, device=dev)
gyro = 0.1 * torch.randn(B, F, 3, device=dev)
acc = torch.randn(B, F, 3, device=dev) + torch.tensor([0., 0., 9.81], device=dev)
for cov in (False, True):
    integ = pp.module.IMUPreintegrator(prop_cov=cov, reset=True).to(dev)
    ms = med_ms(lambda: integ(dt=dt, gyro=gyro, acc=acc))
    print(json.dumps({"B": B, "F": F, "prop_cov": cov, "ms": round(ms, 4), "steps_per_s": B * F / ms * 1e3,
                      "GBps_on_68B_per_step": B * F * 68 / ms / 1e6}))
