"""The reference's own eager ops on the MI355X: PyPose (oracle/_ref) is device-agnostic aten
code, so `pp.randn_se3(B, device='cuda').Exp().Log()` runs through PyTorch-ROCm as a chain of ~100 small kernels with
[B,3,3]..[B,6,6] temporaries.  Timed next to pypose_amd's two kernels on the same inputs.  python tools/ref_eager_gpu.py [B]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_loader
import pypose_amd as pa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rpp = ref_loader.load()
torch.manual_seed(0)
x = torch.randn(B, 6, device="cuda") * 0.5


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {"rows": B}
with torch.no_grad():
    xr, xa = rpp.se3(x), pa.se3(x)
    out["reference_eager_ms"] = timed(lambda: xr.Exp().Log()) * 1e3
    out["pypose_amd_ms"] = timed(lambda: xa.Exp().Log(), reps=20) * 1e3
    d = (xr.Exp().Log().tensor() - xa.Exp().Log().tensor()).abs().max().item()
out["reference_eager_pairs_per_s"] = B / out["reference_eager_ms"] * 1e3
out["pypose_amd_pairs_per_s"] = B / out["pypose_amd_ms"] * 1e3
out["speedup"] = out["reference_eager_ms"] / out["pypose_amd_ms"]
out["max_abs_difference"] = d
print(json.dumps(out))
