"""Wall-clock breakdown of the default (non-static) LM step on InvNet at B problems: where the traced forward spends its time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd import _C
from pypose_amd.lietensor import operation as OP, lietensor as LT
from pypose_amd.optim import optimizer as O, fused as F
from tests.optim_models import InvNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
torch.manual_seed(0)
net = InvNet(pp.randn_SE3(B, device="cuda")); inp = pp.randn_SE3(B, device="cuda")
init = net.pose.detach().clone()
T = {}


def timed(owner, name, label):
    f = getattr(owner, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[label] = T.get(label, 0.0) + time.perf_counter() - t
    setattr(owner, name, w)


timed(O, "_linearize", "linearize")
timed(F, "try_fused", "try_fused")
timed(F, "match_se3inv", "match")
timed(_C, "row_op", "row_op")
timed(OP, "_launch", "_launch")
timed(LT.LieTensor, "__matmul__", "matmul")
timed(LT.LieTensor, "Log", "Log")
timed(O.RobustModel, "forward", "model.forward")
timed(F.Se3InvLinearization, "run_trials", "run_trials")
timed(F.Se3InvLinearization, "__init__", "lin.__init__")
opt = pp.optim.LM(net, strategy=pp.optim.strategy.Adaptive(damping=1e-6))


def run(k):
    net.pose.data.copy_(init)
    if hasattr(opt, "loss"):
        del opt.loss
    for _ in range(k):
        opt.step(inp)


run(3); torch.cuda.synchronize(); T.clear()
import gc
gc.collect(); gc.freeze()          # a gen-2 collection with torch loaded is a 40-70 ms pause: keep it out of the loop
t0 = time.perf_counter()
for _ in range(20):
    run(3)
torch.cuda.synchronize()
print(f"ms/step {(time.perf_counter() - t0) / 60 * 1e3:.3f}")
# one repetition, piece by piece (synchronised after each piece)
def piece(label, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"    {label:28s} {(time.perf_counter() - t) * 1e3:.3f} ms")
for rep in range(2):
    piece("pose.data.copy_(init)", lambda: net.pose.data.copy_(init))
    piece("del opt.loss", lambda: delattr(opt, "loss") if hasattr(opt, "loss") else None)
    for i in range(3):
        piece(f"step {i} (damping {opt.param_groups[0]['damping']:.1e}, rejects before {getattr(opt, 'reject_count', 0)})", lambda: opt.step(inp))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {k:16s} {v / 60 * 1e3:.3f} ms/step")
