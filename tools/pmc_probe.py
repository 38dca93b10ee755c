"""Launch the copy / SE3 Exp / SE3 Log kernels a few times at 10M rows so that rocprofv3 --pmc
can attribute FETCH_SIZE / WRITE_SIZE per dispatch (tools/gpu_pmc.sh post-processes the CSVs)."""
import ctypes, sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypose_amd import _C
import pypose_amd as pp

N = 10_000_000
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = pp.randn_se3(N, device=dev)
lib = _C.library()
fcopy = lib.symbol("pplie_var_copy", [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p])
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
src = torch.empty(N * 6, device=dev); dst = torch.empty_like(src)
for _ in range(3):
    fcopy(src.data_ptr(), dst.data_ptr(), N * 24, 16384, st)     # calibration: 240 MB read + 240 MB written
    X = x.Exp()
    y = X.Log()
torch.cuda.synchronize()
print("done")
