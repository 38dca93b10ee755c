"""Static check of the device assembly for loads that are waited for right where they are issued (no GPU needed):

    python tools/waitcnt_scan.py [file.hip ...] [--min N] [--match substring]

For every kernel of every translation unit under pypose_amd/csrc it counts the global / buffer loads that are followed, within three
instructions and before the next load, by `s_waitcnt vmcnt(0)`.  That pattern is what `ok ? p[k] : d` on an input stream compiles
to (a branch around the load, a wait at its join), what a launch-uniform `if (ptr)` around a load compiles to, and what a select
placed right behind a load causes: a "prefetch" made of such loads is a series of memory round trips (DESIGN.md section 3.5,
"Round 4, second half").  Polling loops of the persistent kernels show up too -- there the pattern is the point."""
import glob
import os
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pypose_amd.build import CFLAGS, HIPCC  # noqa: E402


def kernels(asm):
    name, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
        elif line.startswith(".Lfunc_end") and name:
            yield name, body
            name = None
        elif name:
            body.append(line)


def count(body):
    ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", "."))]
    is_load = lambda l: l.startswith(("global_load", "buffer_load"))
    n = 0
    for i, l in enumerate(ins):
        if is_load(l):
            for m in ins[i + 1:i + 4]:
                if m.startswith("s_waitcnt") and "vmcnt(0)" in m:
                    n += 1
                    break
                if is_load(m):
                    break
    return n, sum(1 for l in ins if is_load(l)), sum(1 for l in ins if l.startswith("s_cbranch"))


def main():
    args = sys.argv[1:]
    minimum, match, files = 3, "", []
    while args:
        a = args.pop(0)
        if a == "--min":
            minimum = int(args.pop(0))
        elif a == "--match":
            match = args.pop(0)
        else:
            files.append(a)
    files = files or sorted(glob.glob(str(ROOT / "pypose_amd" / "csrc" / "*.hip")))
    flags = [f for f in CFLAGS if f != "-fPIC"]
    for src in files:
        out = subprocess.run([HIPCC, *flags, "--cuda-device-only", "-S", src, "-o", "-"], check=True, capture_output=True, text=True).stdout
        rows = []
        for name, body in kernels(out):
            n, loads, br = count(body)
            if n >= minimum:
                rows.append((n, loads, br, name))
        for n, loads, br, name in sorted(rows, reverse=True):
            pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if match in pretty:
                print(f"{os.path.basename(src)}: {n} of {loads} loads waited for at once, {br} branches: {pretty[:150]}")


if __name__ == "__main__":
    main()
