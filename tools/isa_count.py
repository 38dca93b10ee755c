"""Static instruction mix of the gfx950 kernels in one HIP translation unit (no GPU needed).

    python tools/isa_count.py pypose_amd/csrc/lm_step.hip [substring-of-kernel-name]

Compiles with the library's flags to device assembly and counts, per kernel, VALU / packed VALU / transcendental /
SALU / LDS / global-memory instructions plus the VGPR count -- the numbers the VALU-issue estimates in DESIGN.md use
(a wave64 VALU instruction occupies its SIMD for 4 cycles; straight-line row kernels execute nearly all of them).
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pypose_amd.build import CFLAGS, HIPCC  # noqa: E402


def main():
    src = Path(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        flags = [f for f in CFLAGS if f != "-fPIC"]
        subprocess.run([HIPCC, *flags, "--cuda-device-only", "-S", str(src), "-o", str(out)], check=True,
                       stderr=subprocess.DEVNULL)
        text = out.read_text().splitlines()
    name, rows, cur = None, {}, None
    for line in text:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            cur = rows.setdefault(name, {"valu": 0, "pk": 0, "trans": 0, "salu": 0, "lds": 0, "vmem": 0, "dp": 0, "branch": 0, "vgpr": None})
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+\.vgpr_count:\s+(\d+)", line) or re.match(r"^; NumVgprs: (\d+)", line)
        if m and cur["vgpr"] is None:
            cur["vgpr"] = int(m.group(1))
        m = re.match(r"^\s+([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            cur["valu"] += 1
            if op.startswith("v_pk_"):
                cur["pk"] += 1
            if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_", op):
                cur["trans"] += 1
            if op.endswith("_f64"):
                cur["dp"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
            if op.startswith("s_cbranch") or op.startswith("s_branch"):
                cur["branch"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
    for k, v in rows.items():
        if pat in k and v["valu"]:
            print(f"{k[:110]}\n    " + "  ".join(f"{a}={b}" for a, b in v.items()))


if __name__ == "__main__":
    main()
