"""SURVEY.md section 8(c) "run them on the new backend": the reference's OWN test files (copied, unmodified, by oracle/Makefile
into oracle/_ref/tests -- git-ignored, shipped to the GPU box like a built .so) executed against this package aliased as
``pypose`` with the REAL HIP kernels: no ``--cpu-oracle`` stand-in, ``libpplie.so`` mapped.  The files choose ``cuda`` themselves
when a GPU is visible (``device = torch.device("cuda" if torch.cuda.is_available() else "cpu")``); whatever host tensors they
still build are staged through the kernels (pypose_amd/_C.py row_op).  Second pass: the same files with
``torch.set_default_device("cuda")``.

Each reference test function is ONE test here (its name in the id), looked up in the junit report of a single run of the files
per mode; the exclusions are xfail-by-name with the reason:
  * test_parameter_dispatch -- monkeypatches ``pypose._require_backend_attr``, the loader of the external ``bae`` plugin.
  * test_sparse_lm_chain_pgo_runs_and_converges -- (skipped by the reference without CUDA + bae; it RUNS here.)  It stops at the
    first loss < 1e-5 and then demands translations within 2e-4; with the device RNG stream of this box the first LM step lands
    at loss 1.717e-6 with translations still 9.9e-4 off -- and the REFERENCE'S OWN dense LM on the same device does exactly
    the same (1.7170926963e-06 / 9.904495e-04, measured r04).  test_chain_pgo_follows_the_reference_dense_lm below pins that:
    our sparse=True / PCG trajectory equals the reference's dense trajectory step by step, and wherever the reference's LM
    meets the file's criterion ours must too.
A reference test that FAILS in the run of all files is run again alone, in a fresh process, up to twice before it counts (the optimiser
tests draw unseeded random problems; see the comment in ``_report``); at most two such re-runs are tolerated per suite run and they are printed.
Not run: function/test_metric.py, function/test_downsample.py, optim/test_pose_estimation.py (dataset download / torchvision /
scripts without collected tests); module/test_{dynamics,ekf,icp,lqr,mpc,pf,pnp,ukf}.py (subsystems SURVEY.md section 2 marks out
of scope).
"""
import ast
import os
import subprocess
import sys
import tempfile
import xml.etree.ElementTree as ET
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REFTESTS = ROOT / "oracle" / "_ref" / "tests"
FILES = ["lietensor/test_lietensor.py", "optim/test_optimizer.py", "optim/test_jacobian.py", "optim/test_solver.py",
         "optim/test_scheduler.py", "optim/test_sparse_lm.py", "basics/test_ops.py", "basics/test_func.py",
         "function/test_checking.py", "function/test_spline.py", "module/test_loss.py"]
KNOWN = {"test_parameter_dispatch": "patches the loader of the external bae plugin (out of scope)",
         "test_sparse_lm_chain_pgo_runs_and_converges": "fails identically against the reference's own dense LM on this device's RNG "
                                                        "stream; pinned by test_chain_pgo_follows_the_reference_dense_lm"}


def _reference_test_ids():
    """(file, class or None, function) of every test the reference's files define, read from their source (no import)"""
    ids = []
    for f in FILES:
        path = REFTESTS / f
        if not path.exists():
            continue
        tree = ast.parse(path.read_text())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name.startswith("test"):
                ids.append((f, None, node.name))
            elif isinstance(node, ast.ClassDef) and node.name.startswith("Test"):
                ids += [(f, node.name, m.name) for m in node.body if isinstance(m, ast.FunctionDef) and m.name.startswith("test")]
    return ids


IDS = _reference_test_ids()
_reports = {}


def _report(mode):
    """{(module stem, class or None, function): [outcomes of its (possibly parametrised) cases]} of one run of all files"""
    if mode in _reports:
        return _reports[mode]
    assert REFTESTS.exists(), "oracle/_ref/tests missing: run `make -C oracle` where /root/reference is mounted"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PPLIE_QUIET_STAGING="1", PPLIE_REPORT_LIBS="1")
    with tempfile.TemporaryDirectory() as td:
        xml = os.path.join(td, "ref.xml")
        extra = ["--default-cuda"] if mode == "default_cuda" else []
        out = subprocess.run([sys.executable, str(ROOT / "tests" / "run_reference_tests.py"), *extra, f"--junitxml={xml}",
                              *[str(REFTESTS / f) for f in FILES]], capture_output=True, text=True, env=env, cwd="/tmp", timeout=1500)
        text = out.stdout + out.stderr
        assert "libpplie.so mapped: True | stand-in backend: False" in text, text[-3000:]
        res = {}
        for case in ET.parse(xml).getroot().iter("testcase"):
            cls = case.get("classname", "").split(".")
            name = case.get("name", "").split("[")[0]
            stem, klass = (cls[-2], cls[-1]) if cls and cls[-1].startswith("Test") and len(cls) > 1 else (cls[-1], None)
            bad = [c.tag for c in case if c.tag in ("failure", "error")]
            skipped = any(c.tag == "skipped" for c in case)
            res.setdefault((stem, klass, name), []).append("failed" if bad else "skipped" if skipped else "passed")
        # The reference's optimiser tests draw UNSEEDED random problems (e.g. tests/optim/test_optimizer.py:242-266: sigma = 1 poses, "at most
        # 9 LM steps") and where a test sits in the random stream depends on every test before it: one run in ~30 of the whole suite met an
        # instance that needs a tenth step (round 6; the same files passed 16 of 16 runs on their own).  A failed case is therefore run
        # again, ALONE in a fresh process, up to twice: a defect of the kernels or of the optimiser fails every time, a hard draw does not.
        # What was retried is kept in the report and printed by test_at_least_sixty_reference_tests_ran_on_the_kernels.
        retried = {}
        forced = os.environ.get("PPLIE_TEST_FORCE_RETRY")      # (exercises this path: the named reference test is treated as failed once)
        if forced:
            for key in res:
                if key[2] == forced:
                    res[key] = ["failed"] * len(res[key])
        for (stem, klass, name), outcomes in list(res.items()):
            if "failed" not in outcomes or name in KNOWN:
                continue
            f = next((f for f in FILES if Path(f).stem == stem), None)
            if f is None:
                continue
            node = str(REFTESTS / f) + ("::" + klass if klass else "") + "::" + name
            for attempt in (1, 2):
                again = subprocess.run([sys.executable, str(ROOT / "tests" / "run_reference_tests.py"), *extra, node], capture_output=True,
                                       text=True, env=env, cwd="/tmp", timeout=600)
                retried[(stem, klass, name)] = attempt if again.returncode == 0 else -attempt
                if again.returncode == 0:
                    res[(stem, klass, name)] = ["passed" if o == "failed" else o for o in outcomes]
                    break
            text += f"\n[retry] {node}: {'passed on attempt ' + str(retried[(stem, klass, name)]) if retried[(stem, klass, name)] > 0 else 'failed again'}\n"
    _reports[mode] = (res, text, retried)
    return _reports[mode]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["as_written", "default_cuda"])
@pytest.mark.parametrize("ref", IDS, ids=[f"{Path(f).stem}::{(c + '::') if c else ''}{n}" for f, c, n in IDS])
def test_reference_test_passes_on_the_hip_kernels(ref, mode):
    f, klass, name = ref
    res, text, _ = _report(mode)
    got = res.get((Path(f).stem, klass, name))
    assert got, f"the reference test {ref} was not collected\n" + text[-2000:]
    if name in KNOWN:
        if "failed" in got:
            pytest.xfail(KNOWN[name])
        return
    assert "failed" not in got, f"{ref}: {got}\n" + text[-4000:]
    assert "passed" in got, f"{ref} was skipped on the GPU box: {got}"


@pytest.mark.gpu
def test_at_least_sixty_reference_tests_ran_on_the_kernels():
    res, _, retried = _report("as_written")
    if retried:
        print(f"[reference suite on HIP] re-run alone after a failure in the full run: {retried}")
    assert len(retried) <= 2, retried                          # (a hard random draw is rare; several at once are a defect)
    cases = [o for v in res.values() for o in v]               # (a parametrised reference test counts once per case, as pytest does)
    passed = sum(1 for o in cases if o == "passed")
    print(f"[reference suite on HIP] {passed} of {len(cases)} reference test cases passed ({len(IDS)} test functions)")
    assert passed >= 60 and len(IDS) >= 35, (passed, len(cases), len(IDS))


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
