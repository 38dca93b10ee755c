"""The reference's loop at BASELINE configs[3] (100k nodes / 400k edges) in fp32 with tight linear solves, recorded once:
oracle/ref_restate.pgo_lm (every formula a function of the reference package, pinned to the real optimizer's trajectories by
tests/test_ref_restate.py) on the seed-0 problem rounded to fp32, CG tol 1e-7 / maxiter 1500.  ~90 s of host time, which is why
tests/test_fullsize_parity_gpu.py reads the result from tests/golden/pgo100k_fp32_ref.json instead of recomputing it in every
GPU session (it recomputes when the problem's checksum differs from the recorded one).

    python tests/golden/make_pgo100k_golden.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_restate          # noqa: E402


def checksum(edges, rel, init):
    return [float(edges.double().sum()), float(rel.abs().sum()), float(init.abs().sum())]


def run(edges, rel, init):
    ref = ref_restate.pgo_lm(init.float(), edges, rel.float(), 3, radius=1e4, tol=1e-7, maxiter=1500)
    return {k: ref[k] for k in ("loss", "damping", "reject")}


if __name__ == "__main__":
    edges, rel, init = ref_restate.pose_graph_problem(100_000, 400_000, seed=0, dtype=torch.float64)
    out = {"checksum": checksum(edges, rel, init), "settings": "fp32, CG tol 1e-7, maxiter 1500, TrustRegion(radius=1e4), 3 steps",
           "torch": torch.__version__, **run(edges, rel, init)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pgo100k_fp32_ref.json")
    json.dump(out, open(path, "w"), indent=1)
    print(out)
