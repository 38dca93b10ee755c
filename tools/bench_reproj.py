"""Reprojection linearisation at bundle-adjustment scale -- `python tools/bench_reproj.py [observations]`:
the fused kernel (residual + closed-form blocks, 84 B read + 80 B written per observation) against the same blocks from two
batched backward sweeps of the unfused composition (SE3_Act kernel + tensor algebra), and the LM step that uses each."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd import _C
from pypose_amd.optim import blocks as _blocks

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = "cuda"


def med_ms(f, reps=10):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


torch.manual_seed(0)
X = pp.randn_SE3(E, sigma=0.3, device=dev).tensor().contiguous()
p = (torch.randn(E, 3, device=dev) + torch.tensor([0, 0, 6.0], device=dev)).contiguous()
K = torch.tensor([[500.0, 0, 320], [0, 500, 240], [0, 0, 1]], device=dev)
cam = torch.cat([K.reshape(1, 9).expand(E, 9), torch.randn(E, 2, device=dev)], -1).contiguous()
out = {"observations": E}
ms = med_ms(lambda: _C.row_op("se3_reproj_lin", [X, p, cam], (2, 18)))
out["fused_lin_ms"] = ms
out["fused_lin_GBps_on_164B"] = E * 164 * 4 / 4 / ms / 1e6
ms = med_ms(lambda: _C.row_op("se3_reproj_fwd", [X, p, cam], (2,)))
out["fused_fwd_ms"] = ms
out["fused_fwd_GBps_on_92B"] = E * 92 / ms / 1e6


def sweeps():
    Xp = pp.SE3(X).requires_grad_(True)
    pr = p.clone().requires_grad_(True)
    r = pp.homo2cart(Xp.Act(pr) @ K.mT) - cam[:, 9:]
    return _blocks.jacobian_blocks([r], [Xp, pr])


out["autograd_blocks_ms"] = med_ms(sweeps, reps=5)
print(json.dumps(out))
