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
if len(sys.argv) > 1:
    ops = {k: v for k, v in ops.items() if k in sys.argv[1:]}

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
    fn = lib.symbol(f"pplie_var_{name}_f32", SIG)
    import pypose_amd as pp
    ins = [torch.randn(N, w, device=dev) for w in iw]
    if name in ("se3_exp_bwd", "se3_log_bwd", "se3_exp_fwd"):
        ins[0] = pp.randn_se3(N, device=dev).tensor().contiguous()      # realistic angle distribution (both coefficient branches)
    if name in ("se3_jinvp_fwd", "se3_adj_fwd", "se3_jinvp_bwd", "se3_log_fwd", "se3_act_fwd", "se3_inv_bwd"):
        ins[0] = pp.randn_SE3(N, device=dev).tensor().contiguous()
    if name in ("sim3_exp_fwd", "sim3_exp_bwd", "sim3_log_bwd"):
        ins[0] = pp.randn_sim3(N, device=dev).tensor().contiguous()
    if name == "sim3_log_fwd":
        ins[0] = pp.randn_Sim3(N, device=dev).tensor().contiguous()
    outs = [torch.empty(N, w, device=dev) for w in ow]
    P = lambda l, k: l[k].data_ptr() if k < len(l) else None
    for block in (128, 256, 1128, 1256):
        for rpt in ((1, 2, 4) if block < 1000 else (2, 4)):
            call = lambda: fn(rpt, block, P(ins, 0), P(ins, 1), P(ins, 2), P(outs, 0), P(outs, 1), N, st)
            assert call() == 0
            ms = med_ms(call)
            nb = 4 * N * (sum(iw) + sum(ow))
            rec = {"op": name, "block": block, "rpt": rpt, "ms": ms, "GBps": nb / ms / 1e6}
            res.append(rec); print(rec, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/tune_general.json", "w"), indent=1)
