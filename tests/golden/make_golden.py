"""Generate tests/golden/lie_golden.npz from the REAL reference (pypose at /root/reference).

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_golden.py

For every op of every group and both dtypes it stores the inputs (seeded random rows + an
adversarial set: theta in {0, <=eps, ~eps, 1e-4 .. pi .. 2pi}, w<0, w=0, |v|=0, un-normalised
quaternions, sigma ~ 0) and what the reference's autograd.Function returns for forward and --
through torch.autograd.grad with a stored cotangent -- backward.  The oracle (oracle/lie_np.py)
is pinned against this file by tests/test_oracle_golden.py; the HIP kernels are compared with it
in tests/test_lie_parity_gpu.py.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
import pypose as pp  # noqa: E402  (the reference)
from pypose.lietensor import operation as op  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lie_golden.npz")
GROUPS = {"so3": (3, 4), "se3": (6, 7), "sim3": (7, 8), "rxso3": (4, 5)}
NRAND = 96


def adversarial_phi(dtype):
    eps = torch.finfo(dtype).eps
    thetas = [0.0, eps * 1e-4, eps * 0.5, eps, eps * 2, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.1, 0.2499, 0.2501, 1.0,
              1.4999, 1.5001, 2.0, 3.0, np.pi - 1e-3, np.pi, np.pi + 1e-3, 4.0, 2 * np.pi - 1e-2, 6.0]
    axes = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0.6, 0.0, 0.8], [1 / 3 ** 0.5] * 3, [-0.36, 0.48, -0.8]], dtype=torch.float64)
    rows = [axes[i % len(axes)] * t for i, t in enumerate(thetas)]
    return torch.stack(rows).to(dtype)


def make_inputs(g, dtype, gen):
    da, dg = GROUPS[g]
    phi_adv = adversarial_phi(dtype)
    na = len(phi_adv)

    def rnd(*s):
        return torch.randn(*s, dtype=dtype, generator=gen)

    # algebra elements
    rand_alg = {"so3": pp.randn_so3, "se3": pp.randn_se3, "sim3": pp.randn_sim3, "rxso3": pp.randn_rxso3}[g]
    x_rand = rand_alg(NRAND, dtype=dtype).tensor()
    sig_adv = torch.tensor([0.0, 1e-9, 1e-4, 0.5, -1.0, 2.0, 1e-20, -1e-3], dtype=dtype).repeat(na // 8 + 1)[:na, None]
    if g == "so3":
        x_adv = phi_adv
    elif g == "se3":
        x_adv = torch.cat([rnd(na, 3), phi_adv], -1)
    elif g == "sim3":
        x_adv = torch.cat([rnd(na, 3), phi_adv, sig_adv], -1)
    else:
        x_adv = torch.cat([phi_adv, sig_adv], -1)
    x = torch.cat([x_rand, x_adv], 0)
    a = torch.cat([rand_alg(NRAND, dtype=dtype).tensor(), x_adv.flip(0)], 0)

    # group elements: Exp of the algebra rows, then quaternion corner cases
    X = pp.LieTensor(x, ltype=getattr(pp, g + "_type")).Exp().tensor().clone()
    Y = pp.LieTensor(a, ltype=getattr(pp, g + "_type")).Exp().tensor().clone()
    q0 = {"so3": 0, "se3": 3, "sim3": 3, "rxso3": 0}[g]
    k = NRAND  # start of adversarial block
    X[k + 0, q0:q0 + 4] = torch.tensor([0, 0, 0, 1.0], dtype=dtype)          # identity
    X[k + 1, q0:q0 + 4] = torch.tensor([0, 0, 0, -1.0], dtype=dtype)         # -identity (v = 0, w < 0)
    X[k + 2, q0:q0 + 4] = torch.tensor([1.0, 0, 0, 0], dtype=dtype)          # w = 0 exactly
    X[k + 3, q0:q0 + 4] = torch.tensor([0.6, 0, 0.8, 1e-12], dtype=dtype)    # |w| <= eps (fp32), > eps (fp64)
    X[k + 4, q0:q0 + 4] = torch.tensor([0, -0.8, 0.6, -1e-20], dtype=dtype)  # |w| <= eps, negative
    X[k + 5, q0:q0 + 4] = torch.tensor([1e-9, 2e-9, -1e-9, 1.0], dtype=dtype)  # tiny v
    X[k + 6, q0:q0 + 4] = torch.tensor([1e-20, 0, 0, 1.0], dtype=dtype)      # v <= eps both dtypes
    X[k + 7, q0:q0 + 4] *= -1                                                 # sign-flipped quaternion (w<0 long way)
    X[k + 8, q0:q0 + 4] *= 1.1                                                # un-normalised
    X[k + 9, q0:q0 + 4] = torch.tensor([0.5, 0.5, 0.5, -0.5], dtype=dtype)   # w < 0
    p3 = rnd(len(x), 3)
    p4 = torch.cat([rnd(len(x), 3), torch.ones(len(x), 1, dtype=dtype) * (1 + 0.5 * rnd(len(x), 1))], -1)
    return dict(x=x, a=a, X=X, Y=Y, p3=p3, p4=p4, g_alg=rnd(len(x), da), g_grp=rnd(len(x), dg), g3=rnd(len(x), 3),
                g4=rnd(len(x), 4))


def fn(g, name):
    cap = {"so3": "SO3", "se3": "SE3", "sim3": "Sim3", "rxso3": "RxSO3"}[g]
    return {"exp": getattr(op, g + "_Exp"), "log": getattr(op, cap + "_Log"), "inv": getattr(op, cap + "_Inv"),
            "mul": getattr(op, cap + "_Mul"), "act": getattr(op, cap + "_Act"), "act4": getattr(op, cap + "_Act4"),
            "adj": getattr(op, cap + "_AdjXa"), "adjt": getattr(op, cap + "_AdjTXa")}[name]


def run(F, ins, g):
    ins = [t.clone().requires_grad_(True) for t in ins]
    out = F.apply(*ins)
    grads = torch.autograd.grad(out, ins, g, allow_unused=True)
    return out.detach(), [gr.detach() for gr in grads]


def main():
    store = {}
    for dtype, dname in ((torch.float64, "f64"), (torch.float32, "f32")):
        for g in GROUPS:
            gen = torch.Generator().manual_seed(20260923 + len(g))
            torch.manual_seed(7 + len(g))   # pp.randn_* take no generator
            I = make_inputs(g, dtype, gen)
            for k, v in I.items():
                store[f"{dname}/{g}/in/{k}"] = v.numpy()
            spec = {"exp": ((I["x"],), I["g_grp"]), "log": ((I["X"],), I["g_alg"]), "inv": ((I["X"],), I["g_grp"]),
                    "mul": ((I["X"], I["Y"]), I["g_grp"]), "act": ((I["X"], I["p3"]), I["g3"]),
                    "act4": ((I["X"], I["p4"]), I["g4"]), "adj": ((I["X"], I["a"]), I["g_alg"]),
                    "adjt": ((I["X"], I["a"]), I["g_alg"])}
            for name, (ins, cot) in spec.items():
                out, grads = run(fn(g, name), ins, cot)
                store[f"{dname}/{g}_{name}_fwd/out0"] = out.numpy()
                for i, gr in enumerate(grads):
                    store[f"{dname}/{g}_{name}_bwd/out{i}"] = gr.numpy()
            # Jinvp forward (lietensor.py:257-264 etc.)
            XL = pp.LieTensor(I["X"], ltype=getattr(pp, {"so3": "SO3", "se3": "SE3", "sim3": "Sim3", "rxso3": "RxSO3"}[g] + "_type"))
            store[f"{dname}/{g}_jinvp_fwd/out0"] = XL.Jinvp(I["a"]).tensor().numpy()
            # Jinvp backward: plain autograd through so3_Jl_inv / calcQ and <Group>_Log (no custom Function)
            Xr = pp.LieTensor(I["X"].clone().requires_grad_(True), ltype=XL.ltype)
            pr = I["a"].clone().requires_grad_(True)
            out = Xr.Jinvp(pr).tensor()
            gX, gp = torch.autograd.grad(out, [Xr, pr], I["g_alg"], allow_unused=True)
            store[f"{dname}/{g}_jinvp_bwd/out0"] = gX.detach().numpy()
            store[f"{dname}/{g}_jinvp_bwd/out1"] = gp.detach().numpy()
            if g == "so3":
                store[f"{dname}/so3_jr_fwd/out0"] = pp.so3(I["x"]).Jr().reshape(-1, 9).numpy()
                xr = I["x"].clone().requires_grad_(True)
                Gm = torch.randn(len(xr), 9, dtype=dtype, generator=gen)
                store[f"{dname}/so3/in/g9"] = Gm.numpy()
                (gx,) = torch.autograd.grad(pp.so3(xr).Jr().reshape(-1, 9), xr, Gm)
                store[f"{dname}/so3_jr_bwd/out0"] = gx.detach().numpy()
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {len(store)} arrays, {os.path.getsize(OUT) / 1e6:.2f} MB, pypose {pp.__version__}, torch {torch.__version__}")


if __name__ == "__main__":
    main()
