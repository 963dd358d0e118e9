"""The compiled operator library (csrc/mgx_torch.cpp -> lib/libmgx_torch.so) from plain C++: tests/native/torch_native.cpp
dlopen()s it and calls the ops through the c10 dispatcher -- no Python interpreter in the process."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "torch_native.cpp")
EXE = os.path.join(ROOT, "tests", "native", "torch_native")


def build():
    import torch
    from torch.utils import cpp_extension as ce
    from multigrid_amd import build as mgx_build
    lib = mgx_build.build_torch_lib()
    deps = [SRC, lib, os.path.join(ROOT, "include", "mgx.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE, lib
    tl = ce.library_paths()[0]
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
                           *[f"-I{p}" for p in ce.include_paths()], "-I/opt/rocm/include", f"-I{ROOT}/include", SRC, "-o", EXE,
                           "-Wl,--no-as-needed", f"-L{tl}", "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip",
                           f"-L{os.path.dirname(lib)}", "-lmgx", "-ldl",
                           f"-Wl,-rpath,{tl}", f"-Wl,-rpath,{os.path.dirname(lib)}"])
    return EXE, lib


def test_compiled_ops_register_without_python():
    exe, lib = build()
    out = subprocess.run([exe, lib], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "schemas ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_compiled_ops_run_from_cpp_on_the_gpu():
    exe, lib = build()
    out = subprocess.run([exe, lib, "gpu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "torch native ok" in out.stdout, out.stdout + out.stderr
