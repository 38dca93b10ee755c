"""The multi-GPU persistent PCG (pplie_pcg_persist_p2p, csrc/pcg_persist.hip; LM(group=, shard="nodes", exchange="p2p")) with its
`world` ranks emulated on ONE device: every "rank" is a launch on its own stream over its own node rows, the peers' tables are
plain pointers of the same process -- the kernel protocol (p stored into every rank's hand-off table, two-level tagged sums,
epoch-tagged tables that are never cleared) is exactly what runs over xGMI with peer-mapped pointers.  Against the one-rank
solve of the same system."""
import pytest
import torch

import pypose_amd as pp
from pypose_amd import _C
from pypose_amd.optim import fused as F, nodeshard as NS, posegraph as G
from tests.optim_models import PoseGraph
from tests.test_optim_gpu import _synthetic_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _system(N, E, dtype):
    edges, rel, init = _synthetic_graph(N, E, dtype)
    graph = PoseGraph(init.clone())
    solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
    opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    opt.step((edges, rel))
    prog = opt._structure_cache["program"][3]
    with torch.no_grad():
        lin = F._pgo_linearization(opt, prog, None, graph.nodes, True)
        lin.build_normal_equations(1e-6, 1e32)
        lin.damp(1e-4)
        wsp = next(iter(opt._pcg_workspaces.values()))
    return lin, wsp


@pytest.mark.parametrize("dtype,tol,atol", [(torch.float32, 1e-5, 2e-4), (torch.float64, 1e-10, 1e-8)])
@pytest.mark.parametrize("world", [2, 4])
def test_emulated_ranks_reproduce_the_one_rank_solve(world, dtype, tol, atol):
    N, E, m = 3000, 12000, 6
    lin, wsp = _system(N, E, dtype)
    with torch.no_grad():
        x_ref, its_ref = wsp.solve(lin, lin.s, lin.dmin, lin.dmax, tol, 2000, None)          # one rank (prepare + persistent solve)
        D, Binv = wsp.D.clone(), wsp.Binv.clone()
        # the right-hand side as pplie_pcg_prepare leaves it: r = -g, z = Binv r
        r = (-lin.g).contiguous()
        z = (Binv @ r.unsqueeze(-1)).squeeze(-1).contiguous()
        ptr, blk, other = lin.csr()
        HB = lin.HB
    ranks = [NS.P2PRank(N, m, dtype, torch.device(DEV)) for _ in range(world)]
    ptag_ptrs = [rk.ptag.data_ptr() for rk in ranks]
    rpart_ptrs = [rk.rpart.data_ptr() for rk in ranks]
    xs, keep = [], []
    for epoch in (1, 2):                                             # twice: the tables are not cleared between solves
        streams = [torch.cuda.Stream() for _ in range(world)]
        torch.cuda.synchronize()
        xs = []
        for rank in range(world):
            chunk, a, b = NS._bounds(N, world, rank)
            lo, hi = int(ptr[a]), int(ptr[b])
            ops = dict(ptr=(ptr[a:b + 1] - lo).to(torch.int32).contiguous(), other=other[lo:hi].contiguous(), HB=HB[lo:hi].contiguous(),
                       D=D[a:b].contiguous(), Binv=Binv[a:b].contiguous(), x=torch.zeros(b - a, m, dtype=dtype, device=DEV),
                       r=r[a:b].contiguous(), z=z[a:b].contiguous())
            keep.append(ops)
            with torch.cuda.stream(streams[rank]):
                code = NS.persist_p2p_launch(ranks[rank], ptag_ptrs, rpart_ptrs, tol=tol, maxiter=2000, grid=128 // world,
                                             row0=a, n_global=N, world=world, rank=rank, epoch=epoch, m=m, **ops)
            assert code == 0
            xs.append(ops["x"])
        torch.cuda.synchronize()
        infos = [rk.info.tolist() for rk in ranks]
        assert all(i[3] == 1.0 for i in infos), infos                # converged everywhere
        assert len({i[0] for i in infos}) == 1                       # ... in the same iteration
        assert abs(infos[0][0] - its_ref) <= 2, (infos[0][0], its_ref)
        x = torch.cat(xs, 0)
        assert float((x - x_ref).abs().max()) <= atol * max(1.0, float(x_ref.abs().max())), float((x - x_ref).abs().max())
