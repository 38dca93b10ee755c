"""Where does the fp32-vs-oracle error of the non-autograd Lie ops exceed 1e-5?  (the 100k-row gate was a 99.99 %
quantile.)  For every op and three seeds: count of rows above 1e-5 and, for the worst ones, the rotation angles of the operands --
the data the explicit mask of tests/test_lie_parity_gpu.py::test_random_100k_fp32_vs_oracle_fp64 is chosen from.
    python tools/probe_parity_tail.py > gpurun_out/parity_tail.json"""
import json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import lie_np
from tests.golden_util import AUTOGRAD_OPS, row_rel_err
from tests.test_lie_parity_gpu import _random_inputs, run_hip, rotation_angles

out = {}
for name in sorted(lie_np.OPS):
    if name in AUTOGRAD_OPS:
        continue
    rec = []
    for k in range(3):
        rng = np.random.default_rng(zlib.crc32(name.encode()) + k)
        ins = _random_inputs(name, 100_003, np.float32, rng)
        refs = lie_np.OPS[name](*[a.astype(np.float64) for a in ins])
        outs = run_hip(name, ins)
        th = rotation_angles(name, ins)
        for j, (o, r) in enumerate(zip(outs, refs)):
            e, ok = row_rel_err(o, r)
            bad = np.nonzero(e > 1e-5)[0]
            worst = bad[np.argsort(-e[bad])][:6]
            rec.append({"seed": k, "out": j, "max": float(e.max()), "n_bad": int(bad.size),
                        "worst": [{"e": float(e[i]), "theta": [round(float(t[ok][i]), 5) for t in th],
                                   "ref_norm": float(np.linalg.norm(r[ok][i]))} for i in worst]})
    out[name] = rec
    print(json.dumps({name: rec}), flush=True)
