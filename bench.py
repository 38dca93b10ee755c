"""bench.py -- BASELINE.json's metric on its configs[1] workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload ("se3_explog_b10m"): batched SE3 Exp -> Log forward, B = 10,000,000 rows of fp32 per GPU
(BASELINE.json configs[1]; inputs `pp.randn_se3(B)`, seed = rank), through the public API
(``x.Exp()`` then ``X.Log()``: two HIP kernels per step, the group element X is materialised in
HBM as in the reference).  One step = one Exp+Log pass over the whole batch; value = SE3
Exp+Log pairs per second summed over all ranks (weak scaling: rows are independent, each rank
owns its own B rows, no data-path collective).

Printed JSON line (rank 0): the driver contract + "roofline" (dominant kernel, HIP-event timed
inside the timed region) + "cpu_baseline" (the oracle -- a numpy port of the reference's
algorithm -- timed on this box's host cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# torch is imported inside main(): the cpu_baseline leg spawns one worker per host core and every
# worker re-imports this file, which must stay light (numpy only).
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_ROW = {"se3_exp_fwd": 24 + 28, "se3_log_fwd": 28 + 24}    # 52 B/row each


def _cpu_worker(args):
    import numpy as np
    from oracle import lie_np
    seed, n = args
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = np.concatenate([rng.standard_normal((n, 3)), d * rng.standard_normal((n, 1))], -1).astype(np.float32)
    t0 = time.perf_counter()
    X = lie_np.se3_exp_fwd(x)[0]
    y = lie_np.se3_log_fwd(X)[0]
    return time.perf_counter() - t0, float(y[0, 0])


def cpu_baseline(budget_s: float = 12.0):
    """Oracle (numpy port of operation.py's se3_Exp / SE3_Log) on all host cores, bounded sample."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    chunk = 250_000
    ctx = mp.get_context("spawn")
    rows = 0
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(i, 1000) for i in range(cores)])       # warm the workers
        t0 = time.perf_counter()
        rnd = 0
        while time.perf_counter() - t0 < budget_s and rows < 10_000_000:
            pool.map(_cpu_worker, [(1000 + rnd * cores + i, chunk) for i in range(cores)])
            rows += chunk * cores
            rnd += 1
        wall = time.perf_counter() - t0
    return {"value": rows / wall, "unit": "SE3 Exp+Log pairs/s", "cores": cores, "kind": "port",
            "sample": f"{rows} rows of the same fp32 workload (oracle/lie_np.py se3_exp_fwd+se3_log_fwd, "
                      f"{cores} processes x {chunk}-row chunks, {wall:.1f} s wall)"}


def pgo_lm_rate(dev, nodes=10_000, edges=40_000, steps=3, reps=5):
    """Second half of BASELINE.json's metric: LM iterations/s on a synthetic pose graph
    (SURVEY.md section 8d C4 generator: chain + random loop closures, sigma 0.01 edge noise, sigma 0.05
    initial error; the reference's own PoseGraph model, examples/module/pgo/pgo.py:15-25; PCG tol 1e-4 /
    maxiter 250 as examples/module/ba, TrustRegion(radius=1e4) as pgo.py:67).  Not part of `value`."""
    import torch
    import pypose_amd as pp

    class PoseGraph(torch.nn.Module):
        def __init__(self, init):
            super().__init__()
            self.nodes = pp.Parameter(init)

        def forward(self, e, poses):
            n1, n2 = self.nodes[e[..., 0]], self.nodes[e[..., 1]]
            return (poses.Inv() @ n1.Inv() @ n2).Log().tensor()

    g = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    gt = pp.cumprod(pp.randn_SE3(nodes, sigma=0.3, device=dev), dim=0, left=False)
    chain = torch.stack([torch.arange(nodes - 1), torch.arange(1, nodes)], -1)
    extra = torch.randint(0, nodes, (edges - (nodes - 1), 2), generator=g)
    extra[:, 1] = torch.where(extra[:, 0] == extra[:, 1], (extra[:, 1] + 1) % nodes, extra[:, 1])
    e = torch.cat([chain, extra], 0).to(dev)
    rel = gt[e[:, 0]].Inv() @ gt[e[:, 1]] @ pp.randn_SE3(edges, sigma=0.01, device=dev)
    init = gt @ pp.randn_SE3(nodes, sigma=0.05, device=dev)
    graph = PoseGraph(init.clone())
    solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
    opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4))
    l0 = float(graph(e, rel).detach().square().sum())
    # every repetition restarts from the same initial estimate and takes the `steps` LM steps that do the
    # actual descent (this problem reaches its noise floor in 3-4); repetition 0 (structure probe, kernel
    # verification, hipGraph capture) is untimed; the rate is the median repetition
    times, losses, its = [], [], []
    for rep in range(reps + 1):
        graph.nodes.data.copy_(init.tensor())
        if hasattr(opt, "loss"):
            del opt.loss
        opt.param_groups[0].update(opt.strategy.defaults)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses, its = [], []
        for _ in range(steps):
            losses.append(opt.step((e, rel)))
            its.append(solver.iterations)
        torch.cuda.synchronize()
        if rep:
            times.append((time.perf_counter() - t0) / steps)
    dt = sorted(times)[len(times) // 2]
    out = {"metric": "LM iters/sec (PGO 10k poses)", "value": 1.0 / dt, "unit": "LM steps/s", "nodes": nodes, "edges": edges,
           "path": opt.linearization, "initial_loss": l0, "losses": [float(l) for l in losses], "pcg_iterations": its,
           "steps_per_repetition": steps, "repetitions_ms_per_step": [round(t * 1e3, 3) for t in times]}
    # the same with LM(static=True): the caller's promise that the model's residual program does not change between
    # steps lets the optimizer skip re-deriving it from a traced forward (reported separately; `value` is the default)
    opt2 = pp.optim.LM(PoseGraph(init.clone()), solver=pp.optim.solver.PCG(tol=1e-4, maxiter=250),
                       strategy=pp.optim.strategy.TrustRegion(radius=1e4), static=True)
    g2, times2 = opt2.model.model, []
    for rep in range(reps + 1):
        g2.nodes.data.copy_(init.tensor())
        if hasattr(opt2, "loss"):
            del opt2.loss
        opt2.param_groups[0].update(opt2.strategy.defaults)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = opt2.step((e, rel))
        torch.cuda.synchronize()
        if rep:
            times2.append((time.perf_counter() - t0) / steps)
    out["static_model_value"] = 1.0 / sorted(times2)[len(times2) // 2]
    out["static_model_final_loss"] = float(last)
    return out


def pgo_sharded_lm_rate(dev, rank, world, nodes=100_000, edges=400_000, steps=3, reps=3):
    """BASELINE configs[3]: pose-graph LM, 100k SE3 nodes / 400k relative-pose edges, the EDGES sharded over the
    ranks (rank r owns edges r::world, nodes replicated) -- `LM(group=...)`, SURVEY.md section 8(e).  Each rank
    linearises its edges; the per-edge blocks are all-gathered once per LM step and every rank assembles J^T J / J^T r
    and solves (no collective inside the PCG) while the blocks fit one GPU, else diag / gradient are all-reduced per
    step and J^T J p per PCG iteration (RCCL over xGMI); the loss is a one-number all-reduce per trial.  Same generator, solver and strategy as
    `pgo_lm_rate`.  Collective: every rank calls this; rank 0's figures are reported.  Not part of `value`."""
    import torch
    import torch.distributed as dist
    import pypose_amd as pp

    class PoseGraph(torch.nn.Module):
        def __init__(self, init):
            super().__init__()
            self.nodes = pp.Parameter(init)

        def forward(self, e, poses):
            n1, n2 = self.nodes[e[..., 0]], self.nodes[e[..., 1]]
            return (poses.Inv() @ n1.Inv() @ n2).Log().tensor()

    g = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    gt = pp.cumprod(pp.randn_SE3(nodes, sigma=0.3, device=dev), dim=0, left=False)
    chain = torch.stack([torch.arange(nodes - 1), torch.arange(1, nodes)], -1)
    extra = torch.randint(0, nodes, (edges - (nodes - 1), 2), generator=g)
    extra[:, 1] = torch.where(extra[:, 0] == extra[:, 1], (extra[:, 1] + 1) % nodes, extra[:, 1])
    e = torch.cat([chain, extra], 0).to(dev)
    rel = (gt[e[:, 0]].Inv() @ gt[e[:, 1]] @ pp.randn_SE3(edges, sigma=0.01, device=dev)).tensor().contiguous()
    init = (gt @ pp.randn_SE3(nodes, sigma=0.05, device=dev)).tensor().contiguous()
    for t in (e, rel, init):                      # one problem on every rank, whatever the device RNGs produced
        dist.broadcast(t, src=0)
    e_mine, rel_mine = e[rank::world].contiguous(), pp.SE3(rel[rank::world].contiguous())
    graph = PoseGraph(pp.SE3(init.clone()))
    solver = pp.optim.solver.PCG(tol=1e-4, maxiter=250)
    opt = pp.optim.LM(graph, solver=solver, strategy=pp.optim.strategy.TrustRegion(radius=1e4), group=dist.group.WORLD)
    times, losses, its = [], [], []
    for rep in range(reps + 1):                   # repetition 0 (structure probe, kernel verification) is untimed
        graph.nodes.data.copy_(init)
        if hasattr(opt, "loss"):
            del opt.loss
        opt.param_groups[0].update(opt.strategy.defaults)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        losses, its = [], []
        for _ in range(steps):
            losses.append(float(opt.step((e_mine, rel_mine))))
            its.append(solver.iterations)
        torch.cuda.synchronize()
        if rep:
            times.append((time.perf_counter() - t0) / steps)
    dt = sorted(times)[len(times) // 2]
    return {"metric": "LM iters/sec, pose graph 100k nodes / 400k edges, edges sharded over the ranks (BASELINE configs[3])",
            "value": 1.0 / dt, "unit": "LM steps/s", "n_gpus": world, "nodes": nodes, "edges": edges,
            "edges_per_rank": int(e_mine.shape[0]), "path": opt.linearization, "losses": losses, "pcg_iterations": its,
            "replicated_solve": bool(getattr(opt, "_last_replicated", False)),
            "collectives_per_step": ("all-gather of the per-edge residuals / Jacobian blocks / indices (E*(6+72)*4 B + E*16 B), "
                                     "loss scalar per trial, broadcast of the nodes" if getattr(opt, "_last_replicated", False) else
                                     "1 all-reduce of N*(36+6) floats + 1 of N*6 floats per PCG iteration + scalars"),
            "repetitions_ms_per_step": [round(t * 1e3, 3) for t in times]}


def invnet_lm_rate(dev, B=1_000_000, steps=3, reps=5):
    """BASELINE configs[2]: LM on the reference's README InvNet, B independent SE3 problems, fp32
    (SURVEY.md section 8d C3).  Each repetition restarts from the same random initial poses and takes `steps`
    LM steps (the problem converges in 2-3); LM steps/s over the best repetition.  Not part of `value`."""
    import torch
    import pypose_amd as pp

    class InvNet(torch.nn.Module):
        def __init__(self, init):
            super().__init__()
            self.pose = pp.Parameter(init)

        def forward(self, input):
            return (self.pose @ input).Log().tensor()

    torch.manual_seed(0)
    init = pp.randn_SE3(B, device=dev)
    inp = pp.randn_SE3(B, device=dev)
    net = InvNet(init.clone())
    opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
    l0 = float(net(inp).detach().square().sum())
    best, loss = float("inf"), None
    for rep in range(reps + 1):                               # repetition 0 = structure probe / verification, untimed
        net.pose.data.copy_(init.tensor())
        if hasattr(opt, "loss"):
            del opt.loss
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = opt.step(inp)
        torch.cuda.synchronize()
        if rep:
            best = min(best, (time.perf_counter() - t0) / steps)
    out = {"metric": "LM iters/sec (InvNet SE3, 1M independent problems)", "value": 1.0 / best, "unit": "LM steps/s",
           "problems": B, "problem_steps_per_s": B / best, "path": opt.linearization, "initial_loss": l0,
           "final_loss": float(loss), "algorithmic_bytes_per_problem_step": 84,
           "hbm_fraction_of_8TBps": 84.0 * B / best / 8e12}
    net2 = InvNet(init.clone())                                # LM(static=True): see pgo_lm_rate
    opt2 = pp.optim.LM(net2, strategy=pp.optim.strategy.Constant(damping=1e-4), static=True)
    best2 = float("inf")
    for rep in range(reps + 1):
        net2.pose.data.copy_(init.tensor())
        if hasattr(opt2, "loss"):
            del opt2.loss
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            opt2.step(inp)
        torch.cuda.synchronize()
        if rep:
            best2 = min(best2, (time.perf_counter() - t0) / steps)
    out["static_model_value"] = 1.0 / best2
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU (BASELINE configs[1]: 10M)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip lm_pgo / lm_invnet (clean rocprof kernel statistics)")
    a = ap.parse_args()
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ      # torch.distributed.run (also with one rank)
    if launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import pypose_amd as pp
    from pypose_amd import _C
    assert _C._test_backend is None

    B = a.rows
    torch.manual_seed(rank)
    x = pp.randn_se3(B, device=dev)          # [B,6] fp32, resident in HBM before timing starts

    def step():
        X = x.Exp()
        return X.Log()

    for _ in range(a.warmup):
        y = step()
    torch.cuda.synchronize()

    # HIP events bracket both kernels on every `EV`-th step of the timed region (default: every step).  Each record is
    # a packet in the stream: on every step they cost ~3% of `value`, but bracketing only some steps makes exactly
    # those launches slower (92-96 us instead of 90 for Log), so the per-kernel figure is taken on all of them
    EV = max(1, int(os.environ.get("PPLIE_BENCH_EVENT_STRIDE", "1")))
    ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(3)] for k in range(0, a.steps, EV)}

    def barrier():
        if launched:
            import torch.distributed as dist
            dist.barrier()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        e = ev.get(k)
        if e is None:
            X = x.Exp()
            y = X.Log()
        else:
            e[0].record()
            X = x.Exp()
            e[1].record()
            y = X.Log()
            e[2].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if launched:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()

    out = None
    if rank == 0:
        ms_exp = sum(e[0].elapsed_time(e[1]) for e in ev.values()) / len(ev)
        ms_log = sum(e[1].elapsed_time(e[2]) for e in ev.values()) / len(ev)
        dom, ms_dom = ("se3_log_fwd", ms_log) if ms_log >= ms_exp else ("se3_exp_fwd", ms_exp)
        achieved = B * BYTES_PER_ROW[dom] / (ms_dom * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get(dom)
        out = {
            "metric": "batched SE3 Exp+Log ops/sec (BASELINE metric, first half; second half under lm_pgo)",
            "value": world * B * a.steps / elapsed,
            "unit": "SE3 Exp+Log pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "se3_explog_b10m (BASELINE configs[1]: batched SE3 Exp then Log, fp32, forward)",
                       "rows_per_gpu": B, "parallelism": f"rows sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "kernel": f"rowmap_lds_kernel<{dom}> (pplie_{dom}_f32)",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": B * BYTES_PER_ROW[dom],
                         "avg_launch_ms": ms_dom, "timed_launches": len(ev), "other_kernel_ms": {"se3_exp_fwd": ms_exp, "se3_log_fwd": ms_log}},
        }
        if world == 1 and not a.no_secondary:
            import gc
            gc.collect()
            gc.freeze()                   # (a gen-2 collection with torch loaded is a 40-70 ms pause)
            for key, fn in (("lm_pgo", pgo_lm_rate), ("lm_invnet", invnet_lm_rate)):
                try:
                    out[key] = fn(dev)
                except Exception as e:    # never lose the headline line over a secondary figure
                    out[key] = {"error": repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    sharded = launched and not a.no_secondary and (world > 1 or os.environ.get("PPLIE_BENCH_SHARDED_PGO") == "1")
    if sharded:
        # every rank takes part; a watchdog keeps the headline line if a collective wedges (nothing in the timed
        # region above depends on this)
        import threading
        finished, printed = threading.Event(), threading.Lock()

        def emit(extra):
            if printed.acquire(blocking=False) and rank == 0:
                out["lm_pgo_sharded"] = extra
                print(json.dumps(out), flush=True)

        def watchdog():
            if not finished.wait(float(os.environ.get("PPLIE_BENCH_SHARDED_TIMEOUT", "120"))):
                emit({"error": "timed out"})
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            import gc
            gc.collect()
            gc.freeze()                   # (a gen-2 collection with torch loaded is a 40-70 ms pause)
            extra = pgo_sharded_lm_rate(dev, rank, world)
        except Exception as e:
            extra = {"error": repr(e)}
        finished.set()
        emit(extra)
    elif rank == 0:
        print(json.dumps(out), flush=True)
    if launched:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
