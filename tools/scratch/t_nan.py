import sys, os; sys.path.insert(0,'/root/repo')
import torch, pypose_amd as pp
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph
from pypose_amd.optim import posegraph
N,E=100_000,400_000
edges, rel, init = _synthetic_graph(N, E, torch.float32)
graph = PoseGraph(init)
solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
orig = posegraph.FusedPCG.solve
def wrapped(self, lin, b, shift, Binv, Bd, tol, maxiter, group):
    x, done = orig(self, lin, b, shift, Binv, Bd, tol, maxiter, group)
    nan = lambda t: bool(torch.isnan(t).any())
    print('   solve: its', done, 's', lin.s, 'nan b/shift/Binv/Bd/x/HB', nan(b), nan(shift), nan(Binv), nan(Bd), nan(x), nan(lin.HB),
          'rr0', float(self.rr_hist[0]), 'rr_last', float(self.rr_hist[done-1]), 'bn2', float((b*b).sum()), 'it', self.it.tolist())
    return x, done
posegraph.FusedPCG.solve = wrapped
for k in range(8):
    l = opt.step((edges, rel)); print(k, float(l), opt.reject_count, opt.param_groups[0]['damping'])
