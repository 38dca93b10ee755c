"""BAL-scale synthetic bundle adjustment: LM step time and host profile (`python tools/prof_ba.py [host]`)."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from tests.optim_models import Reproj
from tests.test_ba_gpu import synthetic_ba
dtype = torch.float64 if "f64" in sys.argv else torch.float32
args, (K0, C0, P0) = synthetic_ba(257, 65_132, 4, dtype)
model = Reproj(K0, C0, P0)
solver = pp.optim.solver.Cholesky() if "chol" in sys.argv else pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(model, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), reject=30)
l0 = float(opt.model.loss(args, None).detach())
opt.step(args)
torch.cuda.synchronize()
pr = cProfile.Profile() if "host" in sys.argv else None
t0 = time.perf_counter()
if pr:
    pr.enable()
losses = []
for _ in range(3):
    losses.append(float(opt.step(args)))
torch.cuda.synchronize()
if pr:
    pr.disable()
print(opt.linearization, "s/step", (time.perf_counter() - t0) / 3, "pcg its", getattr(solver, "iterations", None), "loss", l0, losses)
if pr:
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
    print(s.getvalue()[:6000])
