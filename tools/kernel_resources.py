"""Per-kernel register / scratch figures of one translation unit (no GPU needed):
    python tools/kernel_resources.py pypose_amd/csrc/pcg_persist.hip [name-filter]
(hipcc -Rpass-analysis=kernel-resource-usage, demangled, one line per kernel)"""
import re
import subprocess
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pypose_amd.build import CFLAGS, HIPCC

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run([HIPCC, *CFLAGS, "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for mangled, name in zip(rows, names):
    if flt in name:
        r = rows[mangled]
        short = re.sub(r"\(.*", "", name)
        print(f"{short:90s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):3d} spill {r.get('VGPRs Spill', -1):4d} scratch {r.get('ScratchSize', -1):4d} occ {r.get('Occupancy', -1)} LDS {r.get('LDS Size', -1)}")
