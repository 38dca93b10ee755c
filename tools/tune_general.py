"""A/B of (rows-per-lane, block size) for multi-slab row ops at 10M rows -> gpurun_out/tune_general.json"""
import ctypes, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypose_amd import _C
N = 10_000_000
dev = torch.device("cuda:0")
lib = _C.tune_library()
SIG = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ops = {"se3_exp_bwd": ((6, 7), (6,)), "se3_log_bwd": ((6, 6), (7,)), "se3_mul_fwd": ((7, 7), (7,)),
       "se3_mul_bwd": ((7, 7), (7, 7)), "se3_act_bwd": ((7, 3, 3), (7, 3)), "se3_jinvp_fwd": ((7, 6), (6,)), "se3_adj_fwd": ((7, 6), (6,)),
       "se3_jinvp_bwd": ((7, 6, 6), (7, 6)), "sim3_exp_fwd": ((7,), (8,)), "sim3_log_fwd": ((8,), (7,)), "sim3_exp_bwd": ((7, 8), (7,)),
       "sim3_log_bwd": ((7, 7), (8,)), "se3_exp_fwd": ((6,), (7,)), "se3_log_fwd": ((7,), (6,)), "se3_act_fwd": ((7, 3), (3,)),
       "se3_inv_bwd": ((7, 7), (7,))}
F64 = "--f64" in sys.argv
sys.argv = [a for a in sys.argv if a != "--f64"]
if F64:      # the fp64 rows of bench.py's ops_10m table that sit below 0.65 (+ the two that do not, as controls)
    ops = {"so3_exp_fwd": ((3,), (4,)), "so3_log_fwd": ((4,), (3,)), "se3_exp_fwd": ((6,), (7,)), "se3_log_fwd": ((7,), (6,)),
           "sim3_exp_fwd": ((7,), (8,)), "sim3_log_fwd": ((8,), (7,)), "rxso3_exp_fwd": ((4,), (5,)), "rxso3_log_fwd": ((5,), (4,))}
else:
    ops.update({"so3_exp_fwd": ((3,), (4,)), "rxso3_mul_fwd": ((5, 5), (5,))})
if len(sys.argv) > 1:
    ops = {k: v for k, v in ops.items() if k in sys.argv[1:]}
DT = torch.float64 if F64 else torch.float32
ES = 8 if F64 else 4

def med_ms(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]

res = []
for name, (iw, ow) in ops.items():
    fn = lib.symbol(f"pplie_var_{name}_{'f64' if F64 else 'f32'}", SIG)
    import pypose_amd as pp
    ins = [torch.randn(N, w, device=dev, dtype=DT) for w in iw]
    g_ = name.split("_")[0]
    kind = name.split("_", 1)[1]
    alg = getattr(pp, "randn_" + g_)
    grp = getattr(pp, "randn_" + {"so3": "SO3", "se3": "SE3", "sim3": "Sim3", "rxso3": "RxSO3"}[g_])
    # realistic inputs (both coefficient branches; unit quaternions for the group-valued operands)
    if kind in ("exp_fwd", "exp_bwd", "log_bwd"):
        ins[0] = alg(N, device=dev, dtype=DT).tensor().contiguous()
    else:
        ins[0] = grp(N, device=dev, dtype=DT).tensor().contiguous()
    if kind in ("mul_fwd",):
        ins[1] = grp(N, device=dev, dtype=DT).tensor().contiguous()
    outs = [torch.empty(N, w, device=dev, dtype=DT) for w in ow]
    P = lambda l, k: l[k].data_ptr() if k < len(l) else None
    for block in (128, 256, 1128, 1256):
        for rpt in ((1, 2, 4) if block < 1000 else (2, 4)):
            call = lambda: fn(rpt, block, P(ins, 0), P(ins, 1), P(ins, 2), P(outs, 0), P(outs, 1), N, st)
            assert call() == 0
            ms = med_ms(call)
            nb = ES * N * (sum(iw) + sum(ow))
            rec = {"op": name, "block": block, "rpt": rpt, "ms": ms, "GBps": nb / ms / 1e6}
            res.append(rec); print(rec, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/tune_general" + ("_f64" if F64 else "") + ".json", "w"), indent=1)
