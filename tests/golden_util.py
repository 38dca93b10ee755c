"""Access to tests/golden/lie_golden.npz (outputs of the real reference, see make_golden.py)."""
import os

import numpy as np

from oracle import lie_np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lie_golden.npz")


def load_golden():
    return dict(np.load(_PATH))


def golden_case(G, dname, name):
    """-> (inputs tuple, expected outputs tuple) for ABI op ``name`` ('se3_exp_bwd', ...)."""
    if name == "so3_jr_fwd":
        return (G[f"{dname}/so3/in/x"],), (G[f"{dname}/so3_jr_fwd/out0"],)
    if name == "so3_jr_bwd":
        return (G[f"{dname}/so3/in/x"], G[f"{dname}/so3/in/g9"]), (G[f"{dname}/so3_jr_bwd/out0"],)
    g, o = name.split("_", 1)
    I = lambda k: G[f"{dname}/{g}/in/{k}"]
    F = lambda op: G[f"{dname}/{g}_{op}_fwd/out0"]
    ins = {
        "exp_fwd": lambda: (I("x"),), "exp_bwd": lambda: (I("x"), I("g_grp")),
        "log_fwd": lambda: (I("X"),), "log_bwd": lambda: (F("log"), I("g_alg")),
        "inv_fwd": lambda: (I("X"),), "inv_bwd": lambda: (F("inv"), I("g_grp")),
        "mul_fwd": lambda: (I("X"), I("Y")), "mul_bwd": lambda: (I("X"), I("g_grp")),
        "act_fwd": lambda: (I("X"), I("p3")), "act_bwd": lambda: (I("X"), F("act"), I("g3")),
        "act4_fwd": lambda: (I("X"), I("p4")), "act4_bwd": lambda: (I("X"), F("act4"), I("g4")),
        "adj_fwd": lambda: (I("X"), I("a")), "adj_bwd": lambda: (I("X"), F("adj"), I("g_alg")),
        "adjt_fwd": lambda: (I("X"), I("a")), "adjt_bwd": lambda: (I("X"), I("a"), I("g_alg")),
        "jinvp_fwd": lambda: (I("X"), I("a")), "jinvp_bwd": lambda: (I("X"), I("a"), I("g_alg")),
    }[o]()
    nout = len(lie_np.op_signature(name)[1])
    outs = tuple(G[f"{dname}/{name}/out{i}"] for i in range(nout))
    return ins, outs


def row_rel_err(got, ref):
    """max over rows of |got-ref|_2 / max(|ref|_2, tiny), ignoring rows where ref is not finite / huge."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    ok = np.isfinite(ref).all(-1) & (np.abs(ref).max(-1) < 1e30)
    num = np.linalg.norm(got[ok] - ref[ok], axis=-1)
    den = np.maximum(np.linalg.norm(ref[ok], axis=-1), 1e-300)
    den = np.where(den < 1e-30, 1.0, den)   # rows whose reference is ~0: absolute error
    e = num / den
    e = np.where(np.isnan(e), np.inf, e)
    return e, ok


# Ops the reference obtains by PLAIN autograd through closed-form coefficient expressions
# (Jinvp: lietensor.py:257-264 ...; so3.Jr: :343-351).  Their reference derivatives lose all digits
# as theta -> 0 (and are NaN at theta = 0), so they are compared on well-conditioned rows only.
AUTOGRAD_OPS = {"so3_jinvp_bwd", "se3_jinvp_bwd", "sim3_jinvp_bwd", "rxso3_jinvp_bwd", "so3_jr_bwd"}


def well_conditioned_rows(name, ins, theta_min=1e-2):
    """row mask: rotation angle of the linearisation point >= theta_min (all rows for other ops)."""
    n = ins[0].shape[0]
    if name not in AUTOGRAD_OPS:
        return np.ones(n, dtype=bool)
    if name == "so3_jr_bwd":
        return np.linalg.norm(ins[0].astype(np.float64), axis=-1) >= theta_min
    g = name.split("_")[0]
    x = lie_np.OPS[f"{g}_log_fwd"](ins[0].astype(np.float64))[0]
    phi = x if g == "so3" else (x[:, :3] if g == "rxso3" else x[:, 3:6])
    return np.linalg.norm(phi, axis=-1) >= theta_min
