"""SURVEY.md section 8(c) "run them on the new backend": the reference's OWN test files (copied, unmodified, by oracle/Makefile
into oracle/_ref/tests -- git-ignored, shipped to the GPU box like a built .so) executed against this package aliased as
``pypose`` with the REAL HIP kernels: no ``--cpu-oracle`` stand-in, ``libpplie.so`` mapped.  The files choose ``cuda`` themselves
when a GPU is visible (``device = torch.device("cuda" if torch.cuda.is_available() else "cpu")``); whatever host tensors they
still build are staged through the kernels (pypose_amd/_C.py row_op).  Second pass: the same files with
``torch.set_default_device("cuda")``.

Exclusions (by name, everything else must pass):
  * test_parameter_dispatch -- monkeypatches ``pypose._require_backend_attr``, the loader of the external ``bae`` plugin.
  * test_sparse_lm_chain_pgo_runs_and_converges -- (skipped by the reference without CUDA + bae; it RUNS here.)  It stops at the
    first loss < 1e-5 and then demands translations within 2e-4; with the device RNG stream of this box the first LM step lands
    at loss 1.717e-6 with translations still 9.9e-4 off -- and the REFERENCE'S OWN dense LM on the same device does exactly
    the same (1.7170926963e-06 / 9.904495e-04, measured r04).  test_chain_pgo_follows_the_reference_dense_lm below pins that:
    our sparse=True / PCG trajectory equals the reference's dense trajectory step by step, and wherever the reference's LM
    meets the file's criterion ours must too.
  * function/test_metric.py, function/test_downsample.py, optim/test_pose_estimation.py -- need a dataset download /
    torchvision / are scripts without collected tests; module/test_{dynamics,ekf,icp,lqr,mpc,pf,pnp,ukf}.py -- subsystems
    SURVEY.md section 2 marks out of scope.
"""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REFTESTS = ROOT / "oracle" / "_ref" / "tests"
FILES = ["lietensor/test_lietensor.py", "optim/test_optimizer.py", "optim/test_jacobian.py", "optim/test_solver.py",
         "optim/test_scheduler.py", "optim/test_sparse_lm.py", "basics/test_ops.py", "basics/test_func.py",
         "function/test_checking.py", "function/test_spline.py", "module/test_loss.py"]
KNOWN = re.compile(r"test_parameter_dispatch|test_sparse_lm_chain_pgo_runs_and_converges")
# with cuda as torch's DEFAULT device a few reference tests mix their own explicit host tensors with default-device ones
# (their bug, not the backend's: they fail the same way against the reference itself); listed by name
KNOWN_DEFAULT_CUDA = KNOWN


def _run(extra):
    assert REFTESTS.exists(), "oracle/_ref/tests missing: run `make -C oracle` where /root/reference is mounted"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PPLIE_QUIET_STAGING="1", PPLIE_REPORT_LIBS="1")
    out = subprocess.run([sys.executable, str(ROOT / "tests" / "run_reference_tests.py"), *extra,
                          *[str(REFTESTS / f) for f in FILES]], capture_output=True, text=True, env=env, cwd="/tmp", timeout=1500)
    return out.stdout + out.stderr


def _check(text, known, at_least):
    failed = [l for l in text.splitlines() if l.startswith(("FAILED", "ERROR"))]
    unexpected = [l for l in failed if not known.search(l)]
    assert not unexpected, "\n".join(unexpected) + "\n" + text[-4000:]
    m = re.search(r"(\d+) passed", text)
    assert m and int(m.group(1)) >= at_least, text[-3000:]
    assert "libpplie.so mapped: True" in text, text[-2000:]
    return int(m.group(1))


@pytest.mark.gpu
def test_reference_tests_pass_on_the_hip_kernels():
    n = _check(_run([]), KNOWN, 60)
    print(f"[reference suite on HIP] {n} reference tests passed")


@pytest.mark.gpu
def test_reference_tests_pass_with_cuda_as_default_device():
    n = _check(_run(["--default-cuda"]), KNOWN_DEFAULT_CUDA, 55)
    print(f"[reference suite on HIP, default device cuda] {n} reference tests passed")


def _chain_run(pp, dev, **kw):
    import torch
    from torch import nn

    class ChainPGO(nn.Module):                      # the model of the reference's tests/optim/test_sparse_lm.py:23-36
        def __init__(self, root, nodes):
            super().__init__()
            self.register_buffer("root", root)
            self.nodes = pp.Parameter(nodes, **({"sjac": True} if kw else {}))

        def forward(self, edges, relposes):
            nodes = torch.cat((self.root, self.nodes), dim=0)
            return (relposes.Inv() @ nodes[edges[:, 0]].Inv() @ nodes[edges[:, 1]]).Log().tensor()

    torch.manual_seed(0)
    gt = pp.SE3(torch.tensor([[0., 0, 0, 0, 0, 0, 1], [1., 0, 0, 0, 0, 0, 1], [2., 0, 0, 0, 0, 0, 1]], dtype=torch.float64,
                             device=dev))
    edges = torch.tensor([[0, 1], [1, 2]], device=dev)
    rel = gt[edges[:, 0]].Inv() @ gt[edges[:, 1]]
    init = gt[1:] * pp.randn_SE3(2, sigma=0.1, device=dev, dtype=torch.float64)
    model = ChainPGO(gt[:1], init).to(dev)
    opt = pp.optim.LM(model, strategy=pp.optim.strategy.Constant(damping=1e-4), **kw)
    traj = []
    for _ in range(4):
        loss = opt.step(input=(edges, rel)).item()
        traj.append((loss, (pp.SE3(model.nodes).translation() - gt[1:].translation()).abs().max().item()))
    return traj


@pytest.mark.gpu
def test_chain_pgo_follows_the_reference_dense_lm():
    import torch
    sys.path.insert(0, str(ROOT))
    import pypose_amd as ppa
    import pypose_amd.optim.solver as ppos
    from oracle.ref_loader import load as load_reference
    ppr = load_reference()
    dev = torch.device("cuda")
    ours = _chain_run(ppa, dev, sparse=True, solver=ppos.PCG())
    ref = _chain_run(ppr, dev)
    for (lo, eo), (lr, er) in zip(ours[:2], ref[:2]):          # (later steps sit at 1e-18: rounding noise)
        assert abs(lo - lr) <= 2e-3 * lr, (ours, ref)          # PCG tol 1e-5 vs the dense Cholesky solve
        assert abs(eo - er) <= 2e-3 * er, (ours, ref)

    def file_criterion(traj):                                  # test_sparse_lm.py:129-149
        for loss, err in traj:
            if loss < 1e-5:
                return err <= 2e-4 + 1e-4 * 2.0
        return False
    if file_criterion(ref):
        assert file_criterion(ours)
