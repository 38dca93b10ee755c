"""Time pp.bspline / pp.chspline on the GPU: single kernel vs the composition of batched Lie kernels.
    python tools/bench_spline.py [--nb 4096] [--n 256] [--interval 0.1]"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import pypose_amd as pp  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=4096)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--interval", type=float, default=0.1)
    a = ap.parse_args()
    dev = "cuda:0"
    out = {}
    for dtype, name in ((torch.float32, "f32"), (torch.float64, "f64")):
        data = pp.randn_SE3(a.nb, a.n, sigma=0.5, dtype=dtype, device=dev)
        res = pp.bspline(data, a.interval)
        rows = res.numel() // 7
        K = (rows // a.nb - 1) // (a.n - 3)
        t = timed(lambda: pp.bspline(data, a.interval))
        w = 4 if dtype == torch.float32 else 8
        algo = rows * 7 * w + data.numel() * w
        entry = {"out_poses": rows, "K": K, "ms": t * 1e3, "poses_per_s": rows / t, "algorithmic_GBps": algo / t / 1e9}
        if a.nb * a.n <= 1 << 19:
            par = pp.Parameter(data)
            with torch.no_grad():
                from pypose_amd.function import spline as _sp
                wts = _sp._bspline_weights(a.interval, dtype, data.device)
                tc = timed(lambda: _sp._bspline_composed(data, wts), reps=5)
            entry["composed_ms"] = tc * 1e3
        out["bspline_" + name] = entry
        pts = torch.randn(a.nb, a.n, 3, dtype=dtype, device=dev)
        r2 = pp.chspline(pts, a.interval)
        t2 = timed(lambda: pp.chspline(pts, a.interval))
        out["chspline_" + name] = {"out_points": r2.numel() // 3, "ms": t2 * 1e3,
                                   "algorithmic_GBps": (r2.numel() + pts.numel()) * w / t2 / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
