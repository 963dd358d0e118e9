"""The C ABI driven from plain C++ (HIP runtime only, no Python / torch in the data path): tests/native/capi_native.cpp
is built with hipcc (host code; cross-compiles without a GPU) and, on a GPU box, run against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "capi_native.cpp")
EXE = os.path.join(ROOT, "tests", "native", "capi_native")


def build():
    from multigrid_amd import build as mgx_build
    from oracle import binding as ob
    lib = mgx_build.build_lib()
    olib = ob.build()
    deps = [SRC, lib, olib, os.path.join(ROOT, "include", "mgx.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    subprocess.check_call([mgx_build.hipcc(), "-O2", "-std=c++17", f"-I{ROOT}/include", SRC, "-o", EXE,
                           f"-L{os.path.dirname(lib)}", "-lmgx", f"-L{os.path.dirname(olib)}", "-lmgx_oracle",
                           f"-Wl,-rpath,{os.path.dirname(lib)}", f"-Wl,-rpath,{os.path.dirname(olib)}"])
    return EXE


def test_native_harness_builds():
    assert os.path.exists(build())


@pytest.mark.gpu
def test_native_harness_matches_oracle():
    out = subprocess.run([build(), "3001", "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "native capi ok" in out.stdout, out.stdout + out.stderr
