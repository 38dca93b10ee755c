"""The boundary is a C ABI: tests/c_abi/abi_smoke.c is a plain C program (gcc, no Python, no torch) that links
libpplie.so + the HIP runtime, drives Exp / Log / Inv / Mul / block Cholesky through include/pplie.h and checks the
status codes.  It must compile everywhere; on a GPU box it must also run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c")
OUT = os.path.join(ROOT, "tests", "c_abi", "_build", "abi_smoke")
LIB = os.path.join(ROOT, "pypose_amd", "lib")


def _compile():
    if not os.path.exists(os.path.join(LIB, "libpplie.so")):
        from pypose_amd import build
        build.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           SRC, "-L" + LIB, "-lpplie", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-o", OUT]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return OUT


def test_c_caller_compiles_and_links_against_the_header():
    exe = _compile()
    assert os.access(exe, os.X_OK)
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libpplie.so" in ldd and "libtorch" not in ldd and "python" not in ldd.lower()


@pytest.mark.gpu
def test_c_caller_runs_on_the_gpu():
    exe = _compile()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    assert "pplie C ABI OK" in res.stdout
