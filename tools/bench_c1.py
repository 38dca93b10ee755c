"""C1 (BASELINE configs[0]): pp.randn_se3(1024).Exp().Log() forward + backward -- plumbing latency on the GPU."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
for B in (1024, 1_000_000):
    x = pp.randn_se3(B, device="cuda", requires_grad=True)
    def fwd_bwd():
        x.grad = None
        x.Exp().Log().tensor().sum().backward()
    def fwd():
        with torch.no_grad():
            return x.Exp().Log()
    out = {"B": B}
    for name, f in (("fwd_us", fwd), ("fwd_bwd_us", fwd_bwd)):
        for _ in range(20): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(200): f()
        torch.cuda.synchronize(); out[name] = round((time.perf_counter() - t) / 200 * 1e6, 1)
    print(json.dumps(out))
