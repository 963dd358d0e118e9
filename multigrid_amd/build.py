"""Build libmgx.so (the HIP kernels + C ABI) in-tree for gfx950.

`python -m multigrid_amd.build` or `multigrid_amd.build.build_lib()`.  hipcc cross-compiles without a GPU.
The .so stays in the source tree (multigrid_amd/lib/) so that it travels with the repo snapshot to the GPU box
and shows up as in-tree native code when loaded.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRCS = [os.path.join(HERE, "csrc", "mgx_kernels.hip"), os.path.join(HERE, "csrc", "mgx_aux.hip")]
DEPS = SRCS + [os.path.join(HERE, "csrc", "mgx_rules.h"), os.path.join(ROOT, "include", "mgx.h")]
LIB = os.path.join(HERE, "lib", "libmgx.so")
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-Wall", f"-I{os.path.join(ROOT, 'include')}", *SRCS, "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
