"""call the trial-tail entry standalone (no capture), fp32 and fp64, sync after each"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd import _C
from pypose_amd.optim.pgograph import _TAIL_SIG
dev = torch.device("cuda:0")
for dtype, sfx in ((torch.float32, "_f32"), (torch.float64, "_f64")):
    N, E = 40, 110
    torch.manual_seed(0)
    nodes = pp.randn_SE3(N, dtype=dtype, device=dev).tensor().contiguous()
    Z = pp.randn_SE3(E, dtype=dtype, device=dev).tensor().contiguous()
    idx = torch.randint(0, N, (E, 2), device=dev)
    J = torch.randn(E, 2, 6, 6, dtype=dtype, device=dev)
    R = torch.randn(E, 6, dtype=dtype, device=dev)
    x = 0.01 * torch.randn(N, 6, dtype=dtype, device=dev)
    info = torch.tensor([5.0, 1e-3, 1.0, 1.0], dtype=dtype, device=dev)
    partial = torch.empty(3 * 1024, dtype=dtype, device=dev)
    state = torch.zeros(3, dtype=torch.int64, device=dev)
    ring = torch.zeros(16, dtype=dtype, device=dev)
    state[1:].copy_(torch.tensor([ring.data_ptr(), 16]))
    out = torch.zeros(8, dtype=torch.float64).pin_memory()
    fn = _C.library().symbol("pplie_pgo_trial_tail" + sfx, _TAIL_SIG)
    torch.cuda.synchronize()
    print(sfx, "launch", flush=True)
    code = fn(nodes.data_ptr(), None, idx.data_ptr(), Z.data_ptr(), J.data_ptr(), R.data_ptr(), x.data_ptr(), info.data_ptr(),
              partial.data_ptr(), state.data_ptr(), out.data_ptr(), N, E, _C.stream_ptr(dev))
    torch.cuda.synchronize()
    print(sfx, "code", code, "out", out.tolist(), "ring", ring[:3].tolist(), "state", state.tolist(), flush=True)
