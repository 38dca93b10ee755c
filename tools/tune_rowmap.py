"""GPU A/B of rowmap launch shapes for SE3 Exp/Log (fp32) + device copy ceiling.

Run on the GPU box: python tools/tune_rowmap.py [N]  -> gpurun_out/tune_rowmap.json
"""
import ctypes, json, sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypose_amd import _C

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
lib = _C.tune_library()
VARSIG = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
fexp = lib.symbol("pplie_var_se3_exp_f32", VARSIG)
flog = lib.symbol("pplie_var_se3_log_f32", VARSIG)
fcopy = lib.symbol("pplie_var_copy", [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p])

torch.manual_seed(0)
ax = torch.randn(N, 3, device=dev); ax = ax / ax.norm(dim=-1, keepdim=True) * torch.randn(N, 1, device=dev)
x = torch.cat([torch.randn(N, 3, device=dev), ax], -1).contiguous()
X = torch.empty(N, 7, device=dev)
y = torch.empty(N, 6, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def ref_exp(x):
    x = x.double(); tau, phi = x[:, :3], x[:, 3:]
    th = phi.norm(dim=-1, keepdim=True); th2 = th * th
    B = (1 - th.cos()) / th2; C = (th - th.sin()) / (th2 * th)
    a = torch.linalg.cross(phi, tau); t = tau + B * a + C * torch.linalg.cross(phi, a)
    return torch.cat([t, phi * (th / 2).sin() / th, (th / 2).cos()], -1)

def timeit(f, reps=40):
    """median of per-launch HIP-event durations (ms)"""
    for _ in range(5): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]

res = {"N": N, "device": torch.cuda.get_device_name(0), "variants": []}
# correctness of the production entry points
(Xp,) = _C.row_op("se3_exp_fwd", [x], [7])
(yp,) = _C.row_op("se3_log_fwd", [Xp], [6])
R = ref_exp(x[:100000])
err = ((Xp[:100000].double() - R).norm(dim=-1) / R.norm(dim=-1)).max().item()
rt = ((yp.double() - x.double()).norm(dim=-1) / x.double().norm(dim=-1))
small = ax.norm(dim=-1) < 3.0
res["exp_max_rowrel_err_vs_fp64"] = err
res["roundtrip_max_rowrel_err_theta_lt_3"] = rt[small].max().item()
print("exp err", err, "roundtrip", res["roundtrip_max_rowrel_err_theta_lt_3"], flush=True)

for nbytes, label in ((N * 24, "24B/row"), (N * 28, "28B/row")):
    src = torch.empty(nbytes // 4, device=dev); dst = torch.empty_like(src)
    for grid in (2048, 4096, 16384):
        ms = timeit(lambda: fcopy(src.data_ptr(), dst.data_ptr(), nbytes // 16 * 16, grid, st))
        res["variants"].append({"op": "copy", "label": label, "grid": grid, "ms": ms, "GBps": 2 * nbytes / ms / 1e6})
        print(res["variants"][-1], flush=True)
    ms = timeit(lambda: dst.copy_(src))
    res["variants"].append({"op": "torch_copy", "label": label, "ms": ms, "GBps": 2 * nbytes / ms / 1e6})
    print(res["variants"][-1], flush=True)

for op, fn, src, dst in (("exp", fexp, x, X), ("log", flog, Xp, y)):
    for path, rpt in ((0, 1), (0, 2), (0, 4), (0, 8), (1, 0)):
        for cap in (4096, 1 << 30):
            code = fn(path, rpt, cap, src.data_ptr(), dst.data_ptr(), N, st)
            assert code == 0, code
            ms = timeit(lambda: fn(path, rpt, cap, src.data_ptr(), dst.data_ptr(), N, st))
            ok = torch.equal(dst, Xp if op == "exp" else yp)
            rec = {"op": op, "path": "lds" if path == 0 else "direct", "rpt": rpt, "grid_cap": cap, "ms": ms,
                   "GBps": N * 52 / ms / 1e6, "rows_per_s": N / ms * 1e3, "bitexact_vs_default": ok}
            res["variants"].append(rec); print(rec, flush=True)
json.dump(res, open("gpurun_out/tune_rowmap.json", "w"), indent=1)
