"""Per-kernel HBM-roofline table at B = 10M rows (fp32) + C5 IMU timing.  -> gpurun_out/bench_ops.json"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd import _C

GROUPS = {"so3": (3, 4), "se3": (6, 7), "sim3": (7, 8), "rxso3": (4, 5)}      # (algebra width, group width)


def op_signature(name):
    """(input widths, output widths) of row op ``name`` as the C ABI takes them (include/pplie.h)."""
    g, o = name.split("_", 1)
    da, dg = GROUPS[g]
    return {"exp_fwd": ((da,), (dg,)), "exp_bwd": ((da, dg), (da,)), "log_fwd": ((dg,), (da,)), "log_bwd": ((da, da), (dg,)),
            "inv_fwd": ((dg,), (dg,)), "inv_bwd": ((dg, dg), (dg,)), "mul_fwd": ((dg, dg), (dg,)), "mul_bwd": ((dg, dg), (dg, dg)),
            "act_fwd": ((dg, 3), (3,)), "act_bwd": ((dg, 3, 3), (dg, 3)), "adj_fwd": ((dg, da), (da,)), "adjt_fwd": ((dg, da), (da,)),
            "jinvp_fwd": ((dg, da), (da,)), "jinvp_bwd": ((dg, da, da), (dg, da))}[o]


dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
torch.manual_seed(0)


def med_ms(f, reps=20, inner=8):
    """median over `reps` of (HIP-event time of `inner` back-to-back launches) / inner: the queue never
    drains between the events, so host launch latency is not billed to the kernel"""
    for _ in range(5): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        f(); a.record()
        for _ in range(inner): f()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) / inner for a, b in ev)
    return ts[len(ts) // 2]

rows = []
ops = ["se3_exp_fwd", "se3_log_fwd", "se3_exp_bwd", "se3_log_bwd", "se3_adj_fwd", "se3_adjt_fwd", "se3_mul_fwd", "se3_mul_bwd",
       "se3_inv_fwd", "se3_inv_bwd", "se3_act_fwd", "se3_act_bwd", "se3_jinvp_fwd", "se3_jinvp_bwd", "so3_exp_fwd", "so3_log_fwd", "so3_mul_fwd",
       "sim3_exp_fwd", "sim3_log_fwd", "sim3_exp_bwd", "sim3_log_bwd", "rxso3_exp_fwd", "rxso3_log_fwd"]
gens = {"so3": (pp.randn_so3, pp.randn_SO3), "se3": (pp.randn_se3, pp.randn_SE3), "sim3": (pp.randn_sim3, pp.randn_Sim3),
        "rxso3": (pp.randn_rxso3, pp.randn_RxSO3)}
for name in ops:
    g = name.split("_")[0]
    da, dg = GROUPS[g]
    iw, ow = op_signature(name)
    alg, grp = gens[g]
    kind = name.split("_", 1)[1]
    def mk(w, first):
        if first and kind in ("exp_fwd", "exp_bwd", "log_bwd"):
            return alg(N, device=dev).tensor().contiguous()
        if first:
            return grp(N, device=dev).tensor().contiguous()
        if w == dg and kind in ("mul_fwd",):
            return grp(N, device=dev).tensor().contiguous()
        return torch.randn(N, w, device=dev)
    ins = [mk(w, i == 0) for i, w in enumerate(iw)]
    ms = med_ms(lambda: _C.row_op(name, ins, ow))
    nbytes = 4 * N * (sum(iw) + sum(ow))
    rows.append({"op": name, "ms": ms, "bytes_per_row": 4 * (sum(iw) + sum(ow)), "GBps": nbytes / ms / 1e6,
                 "frac_of_8TBps": nbytes / ms / 1e6 / 8000, "rows_per_s": N / ms * 1e3})
    print(rows[-1], flush=True)
    del ins
out = {"N": N, "ops": rows}

# C5: IMU 4096 x 1024 fp32 (fused route)
B, F = 4096, 1024
dt = torch.full((B, F, 1), 0.005, device=dev)
gyro = 0.1 * torch.randn(B, F, 3, device=dev)
acc = torch.randn(B, F, 3, device=dev) + torch.tensor([0, 0, 9.81], device=dev)
for cov in (True, False):
    m = pp.module.IMUPreintegrator(reset=True, prop_cov=cov).to(dev)
    ms = med_ms(lambda: m(dt, gyro, acc), reps=6, inner=3)
    out[f"c5_imu_cov{int(cov)}"] = {"ms": ms, "steps_per_s": B * F / ms * 1e3, "GBps_68B_per_step": B * F * 68 / ms / 1e6}
    print("c5", cov, out[f"c5_imu_cov{int(cov)}"], flush=True)
# scan alone: SO3 [4096, 1025, 4]
X = pp.randn_SO3(B, F + 1, device=dev)
ms = med_ms(lambda: pp.cumprod_(X, dim=1, left=False), reps=10, inner=4)
out["scan_so3_4096x1025"] = {"ms": ms, "GBps": B * (F + 1) * 32 / ms / 1e6}
print("scan", out["scan_so3_4096x1025"], flush=True)
json.dump(out, open("gpurun_out/bench_ops.json", "w"), indent=1)
