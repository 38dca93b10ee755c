"""Build one of tools/micro/*.hip with the library's own flags (so a micro that includes a csrc/*.hip file gets the same code):
    python tools/build_micro.py imu_cov_chain        -> tools/micro/build/imu_cov_chain   (cross-compiles for gfx950; runs on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pypose_amd.build import CFLAGS, HIPCC
name = sys.argv[1]
src = os.path.join(ROOT, "tools", "micro", name + ".hip")
out = os.path.join(ROOT, "tools", "micro", "build", name)
os.makedirs(os.path.dirname(out), exist_ok=True)
flags = [f for f in CFLAGS if f != "-fPIC"]
cmd = [HIPCC, *flags, "-I", os.path.join(ROOT, "pypose_amd", "csrc"), src, "-o", out]
print(" ".join(cmd))
subprocess.run(cmd, check=True)
