"""Build the HIP library (gfx950) in-tree: pypose_amd/lib/libpplie.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).

    python -m pypose_amd.build [--force] [-j N]
"""
from __future__ import annotations

import argparse
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
OBJDIR = PKG / "build"
LIBNAME = "libpplie.so"
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# fp32 a/b and sqrt lower to v_rcp/v_sqrt (<= 2.5 ulp) instead of the ~10-instruction correctly
# rounded sequences: the kernels sit close enough to the VALU roof for that to matter.
# -fapprox-func: a / b lowers to v_rcp_f32 + v_mul (2 instructions, <= 2 ulp) instead of the 8-instruction
# frexp / ldexp range-safe sequence, sqrtf to a bare v_sqrt_f32 -- every division on the hot paths is guarded
# against tiny divisors by the reference's own eps switches.  -fno-slp-vectorize: the SLP vectoriser pairs fp32
# operations into v_pk_* and pays more v_mov to assemble the register pairs than it saves (LM trial row: 1456 -> 1272
# VALU; with both flags 1090; tools/isa_count.py).
CFLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
          "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics", "-fapprox-func", "-fno-slp-vectorize",
          f"-I{CSRC}", f"-I{PKG.parent / 'include'}"]


# Per-file opt-out of the two ulp-for-speed flags: a translation unit whose first 20 lines contain
#     // pplie-build: precise
# is compiled with correctly rounded fp32 division / square root and without -fapprox-func (e.g. a future kernel whose fp64
# helpers or whose parity needs the last ulp).  No file of this tree asks for it today: parity is green with the fast forms.
_FAST = ("-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fapprox-func")


def _flags_for(src: Path):
    try:
        head = "".join(src.open().readlines()[:20])
    except OSError:
        head = ""
    if "pplie-build: precise" in head:
        return [f for f in CFLAGS if f not in _FAST]
    return CFLAGS


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _digest(src: Path) -> str:
    h = hashlib.sha1()
    h.update(" ".join(_flags_for(src)).encode())
    for p in [src, *sorted(CSRC.glob("*.h")), *sorted((PKG.parent / "include").glob("*.h"))]:
        h.update(p.read_bytes())
    return h.hexdigest()


def _compile(src: Path, force: bool) -> Path:
    obj = OBJDIR / (src.stem + ".o")
    stamp = OBJDIR / (src.stem + ".sha1")
    dig = _digest(src)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj
    cmd = [HIPCC, *_flags_for(src), "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig)
    return obj


def build(force: bool = False, jobs: int | None = None, verbose: bool = True) -> Path:
    """Compile every csrc/*.hip for gfx950 and link lib/libpplie.so. Returns the .so path."""
    OBJDIR.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    srcs = _sources()
    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    out = LIBDIR / LIBNAME
    newest = max(o.stat().st_mtime for o in objs)
    if force or not out.exists() or out.stat().st_mtime < newest:
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(out), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[pypose_amd.build] {out} ({out.stat().st_size >> 10} KiB, {len(objs)} objects)")
    return out


TORCH_EXT = "pplie_torch_ext"


def build_torch_ext(force: bool = False, verbose: bool = True):
    """Compile csrc_torch/pplie_autograd.cpp (native autograd nodes for the row operators: the dispatch layer between torch's
    autograd engine and libpplie.so, no kernels of its own) against the installed torch into lib/pplie_torch_ext.so.  Needs no
    GPU.  Returns the path, or None if this torch cannot build extensions (the Python Functions then carry the autograd)."""
    src = PKG / "csrc_torch" / "pplie_autograd.cpp"
    out = LIBDIR / (TORCH_EXT + ".so")
    stamp = OBJDIR / (TORCH_EXT + ".sha1")
    try:
        import torch
        from torch.utils import cpp_extension
    except Exception as e:                          # pragma: no cover
        print(f"[pypose_amd.build] torch extension not built: {e}", file=sys.stderr)
        return None
    h = hashlib.sha1()
    h.update(src.read_bytes())
    h.update(torch.__version__.encode())
    dig = h.hexdigest()
    OBJDIR.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    if not force and out.exists() and stamp.exists() and stamp.read_text() == dig:
        return out
    bdir = OBJDIR / "torch_ext"
    bdir.mkdir(exist_ok=True)
    cpp_extension.load(name=TORCH_EXT, sources=[str(src)], build_directory=str(bdir), extra_cflags=["-O2"], with_cuda=True,
                       is_python_module=False, verbose=False)
    built = bdir / (TORCH_EXT + ".so")
    out.write_bytes(built.read_bytes())
    stamp.write_text(dig)
    if verbose:
        print(f"[pypose_amd.build] {out} ({out.stat().st_size >> 10} KiB)")
    return out


TUNE_SRC = PKG.parent / "tools" / "micro" / "tuning_variants.hip"
TUNE_LIB = "libpplie_tune.so"


def build_tune(force: bool = False, verbose: bool = True):
    """tools/micro/tuning_variants.hip -> lib/libpplie_tune.so: launch-shape variants of the row kernels for the A/B tools
    (tools/tune_*.py, tools/pmc_probe.py).  Measurement scaffolding: NOT linked into libpplie.so (round 4 shipped it inside the
    product binary), not declared in include/pplie.h, loaded only by those tools (``_C.tune_library()``)."""
    if not TUNE_SRC.exists():
        return None
    OBJDIR.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    out = LIBDIR / TUNE_LIB
    stamp = OBJDIR / "tuning_variants.sha1"
    dig = _digest(TUNE_SRC)
    if not force and out.exists() and stamp.exists() and stamp.read_text() == dig:
        return out
    r = subprocess.run([HIPCC, *CFLAGS, "-shared", str(TUNE_SRC), "-o", str(out)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {TUNE_SRC.name}:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig)
    if verbose:
        print(f"[pypose_amd.build] {out} ({out.stat().st_size >> 10} KiB)")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=None)
    a = ap.parse_args()
    try:
        build(force=a.force, jobs=a.j)
        build_torch_ext(force=a.force)
        build_tune(force=a.force)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
