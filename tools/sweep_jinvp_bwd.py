"""which rows of se3_jinvp_bwd sit above 1e-4?  Host build of lie_math.h (tests/hostmath: the kernels'
arithmetic, no GPU needed) on the inputs of tests/test_lie_parity_gpu.py::test_random_100k_fp32_vs_oracle_fp64 over many
seeds; the rows above the gate are dumped with theta, |tau|, |p| and compared with (a) the central-difference oracle,
(b) the reference's own fp64 autograd (when /root/reference or oracle/_ref is importable) and (c) the same kernel
arithmetic in fp64 on the same fp32 inputs (what rounding inside the sweep costs).

    python tools/sweep_jinvp_bwd.py [--seeds 20] [--n 100003] [--op se3_jinvp_bwd]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lie_np                                    # noqa: E402
from tests.golden_util import row_rel_err, well_conditioned_rows   # noqa: E402
from tests.hostmath_util import hostmath_op                  # noqa: E402
from tests.test_lie_parity_gpu import _random_inputs         # noqa: E402


def reference_autograd(name, ins):
    """the reference package's fp64 autograd through X.Jinvp(p) (lietensor.py:422-429), or None"""
    for path in ("/root/reference", os.path.join(ROOT, "oracle", "_ref")):
        if os.path.isdir(os.path.join(path, "pypose")):
            sys.path.insert(0, path)
            break
    else:
        return None
    import torch
    import pypose as pp
    g = name.split("_")[0]
    T = {"so3": pp.SO3, "se3": pp.SE3, "sim3": pp.Sim3, "rxso3": pp.RxSO3}[g]
    X = T(torch.from_numpy(ins[0].astype(np.float64))).requires_grad_(True)
    p = torch.from_numpy(ins[1].astype(np.float64)).requires_grad_(True)
    out = X.Jinvp(p)
    out = out.tensor() if hasattr(out, "tensor") else out
    gX, gp = torch.autograd.grad(out, (X, p), torch.from_numpy(ins[2].astype(np.float64)))
    gX = gX.tensor() if hasattr(gX, "tensor") else gX
    return gX.numpy(), gp.numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=20)
    ap.add_argument("--n", type=int, default=100_003)
    ap.add_argument("--op", default="se3_jinvp_bwd")
    ap.add_argument("--gate", type=float, default=1e-4)
    ap.add_argument("--theta-min", type=float, default=1e-3)
    a = ap.parse_args()
    name = a.op
    worst = []
    for seed in range(a.seeds):
        rng = np.random.default_rng(seed)
        ins = _random_inputs(name, a.n, np.float32, rng)
        keep = well_conditioned_rows(name, ins, theta_min=a.theta_min)
        ins = [x[keep] for x in ins]
        refs = lie_np.OPS[name](*[x.astype(np.float64) for x in ins])
        outs = hostmath_op(name, ins)
        outs64 = hostmath_op(name, [x.astype(np.float64) for x in ins])
        x = lie_np.OPS[name.split("_")[0] + "_log_fwd"](ins[0].astype(np.float64))[0]
        th = np.linalg.norm(x[:, 3:6] if name.startswith(("se3", "sim3")) else x[:, :3], axis=-1)
        tau = np.linalg.norm(x[:, :3], axis=-1)
        for k, (o, o64, r) in enumerate(zip(outs, outs64, refs)):
            e, _ = row_rel_err(o, r)
            e64, _ = row_rel_err(o64, r)
            bad = np.nonzero(e > a.gate)[0]
            print(f"seed {seed:2d} out{k}: n={len(e)} max={e.max():.2e} q9999={np.quantile(e, 0.9999):.2e} above={len(bad)}"
                  f" | kernel-arithmetic-in-fp64 vs oracle: max={e64.max():.2e} q9999={np.quantile(e64, 0.9999):.2e}")
            for i in bad[np.argsort(-e[bad])][:6]:
                worst.append((seed, k, int(i), float(e[i]), float(e64[i]), float(th[i]), float(tau[i]),
                              float(np.linalg.norm(ins[1][i])), float(np.linalg.norm(ins[2][i]))))
        if seed == 0 and worst:
            rows = sorted({w[2] for w in worst if w[0] == 0})
            sub = [x[rows] for x in ins]
            ref = reference_autograd(name, sub)
            if ref is not None:
                o = [y[rows] for y in outs]
                o64 = [y[rows] for y in outs64]
                orc = [y[rows] for y in refs]
                for k in range(len(o)):
                    print(f"  seed 0, {len(rows)} worst rows, out{k}: fp32 kernel vs REFERENCE autograd fp64: "
                          f"{row_rel_err(o[k], ref[k][:, :o[k].shape[1]])[0].max():.2e}; fp64 kernel arithmetic vs reference: "
                          f"{row_rel_err(o64[k], ref[k][:, :o[k].shape[1]])[0].max():.2e}; central-difference oracle vs reference: "
                          f"{row_rel_err(orc[k], ref[k][:, :o[k].shape[1]])[0].max():.2e}")
    print("\nworst rows: seed out row  err(fp32 kernel)  err(fp64 arithmetic)  theta  |tau|  |p|  |g|")
    for w in sorted(worst, key=lambda w: -w[3])[:40]:
        print("  %2d %d %6d  %.2e  %.2e  theta=%.5f (pi-theta=%.2e)  tau=%.3g  p=%.3g  g=%.3g" %
              (w[0], w[1], w[2], w[3], w[4], w[5], np.pi - w[5], w[6], w[7], w[8]))


if __name__ == "__main__":
    main()
