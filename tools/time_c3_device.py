"""C3 (InvNet LM, B problems) through the device-resident step: steps/s with asynchronous steps, per-step host time.
python tools/time_c3_device.py [B] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import InvNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dtype = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else torch.float32
torch.manual_seed(0)
init = pp.randn_SE3(B, device="cuda", dtype=dtype)
inp = pp.randn_SE3(B, device="cuda", dtype=dtype)
net = InvNet(init.clone())
opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
import gc


def reset():
    net.pose.data.copy_(init.tensor())
    if hasattr(opt, "loss"):
        del opt.loss


reset()
rows = []
for _ in range(5):
    loss = opt.step(inp)
    rows.append((float(loss), opt.reject_count, opt.__dict__.get('_trials')))
print("path", opt.linearization, "per-step (loss, rejects, trials):", rows)
gc.collect(); gc.freeze()
for steps in (2, 3):
    for R in (1, 20, 200):
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(R):
                reset()
                for _ in range(steps):
                    loss = opt.step(inp)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            best = min(best, (t2 - t0) / (steps * R))
        print(f"{R} repetitions x {steps} steps, one sync at the end: {best * 1e6:.1f} us/step  ({1 / best:.0f} steps/s, "
              f"{84.0 * B / best / 8e12 * 100:.1f}% of 8 TB/s on 84 B); host enqueue {(t1 - t0) / (steps * R) * 1e6:.1f} us/step")
reset()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2):
    v = float(opt.step(inp))
print(f"synchronous (float(loss) per step): {(time.perf_counter() - t0) / 2 * 1e6:.1f} us/step")
