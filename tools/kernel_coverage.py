#!/usr/bin/env python3
"""Launched-vs-shipped: which of the gfx950 kernels libmgx.so carries does the GPU test suite actually launch?

    python tools/kernel_coverage.py run   [pytest args ...]   (GPU box) rocprofv3 --kernel-trace over `pytest -m gpu`, then `merge`
    python tools/kernel_coverage.py merge                     gpurun_out/kcov/**/kernel_trace.csv -> profiles/kernels_launched.txt
    python tools/kernel_coverage.py diff                      (CPU) shipped symbols of lib/libmgx.so vs profiles/kernels_launched.txt

An instantiation nobody launched is untested code on the product path.  `profiles/kernels_launched.txt` is the
committed list -- one demangled kernel name per line with its launch count -- and tests/test_kernel_coverage.py (CPU) fails when
the library carries a kernel that is neither on it nor on the short, argued list of profiles/kernels_unlaunched_ok.txt.

Names are compared after normalisation (c++filt's and rocprofv3's demanglers differ in spacing and in how they print the argument
list): `mgx_fused::mgx_fused_kernel<7,1,false,true,...>`.
"""
from __future__ import annotations

import collections
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "gpurun_out", "kcov")
LAUNCHED = os.path.join(ROOT, "profiles", "kernels_launched.txt")
UNLAUNCHED_OK = os.path.join(ROOT, "profiles", "kernels_unlaunched_ok.txt")


def cxxfilt(names: list[str]) -> list[str]:
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
            if len(out) == len(names):
                return out
        except (OSError, subprocess.CalledProcessError):
            continue
    return names


def norm(name: str) -> str:
    """`void ns::kernel<a, b>(args) [clone .kd]` -> `ns::kernel<a,b>`."""
    name = name.strip().strip('"')
    name = re.sub(r"\s*\[clone[^\]]*\]", "", name)
    if name.endswith(".kd"):
        name = name[:-3]
    name = re.sub(r"^void\s+", "", name)
    depth, cut = 0, len(name)                     # drop the trailing argument list: the last top-level '(' outside <...>
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not name.startswith("(anonymous", i):
            cut = i
            break
    name = name[:cut]
    return re.sub(r"\s+", "", name).replace("(anonymousnamespace)::", "")


def shipped(lib: str | None = None) -> dict[str, str]:
    """{normalised name: mangled symbol} of every gfx950 kernel in the library."""
    import kernel_resources as kr
    lib = lib or os.path.join(ROOT, "multigrid_amd", "lib", "libmgx.so")
    syms = [k[".name"] for k in kr.library_kernels(lib)]
    return {norm(d): m for d, m in zip(cxxfilt(syms), syms)}


def launched_from_traces(root: str = OUT) -> collections.Counter:
    c: collections.Counter = collections.Counter()
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                c[norm(row["Kernel_Name"])] += 1
    return c


def read_list(path: str) -> dict[str, str]:
    out = {}
    if os.path.exists(path):
        for line in open(path):
            line = line.rstrip("\n")
            if line and not line.startswith("#"):
                name, _, rest = line.partition("\t")
                out[name] = rest
    return out


def merge():
    c = launched_from_traces()
    ship = shipped()
    ours = {n: k for n, k in c.items() if n in ship}
    os.makedirs(os.path.dirname(LAUNCHED), exist_ok=True)
    with open(LAUNCHED, "w") as fh:
        fh.write("# kernels of libmgx.so launched by `python -m pytest tests -m gpu` under rocprofv3 --kernel-trace (tools/kernel_coverage.py run):\n"
                 "# normalised demangled name <tab> launches.  tests/test_kernel_coverage.py compares the library's symbol table with it.\n")
        for n in sorted(ours):
            fh.write(f"{n}\t{ours[n]}\n")
    import shutil
    os.makedirs(OUT, exist_ok=True)
    shutil.copy(LAUNCHED, os.path.join(OUT, "kernels_launched.txt"))      # (gpurun_out/ is what travels back from the GPU box)
    print(f"{len(ours)} distinct mgx kernels launched ({sum(ours.values())} launches) -> {LAUNCHED}")


def diff(verbose: bool = True):
    ship, got, ok = shipped(), read_list(LAUNCHED), read_list(UNLAUNCHED_OK)
    missing = sorted(n for n in ship if n not in got and n not in ok)
    stale = sorted(n for n in got if n not in ship)
    if verbose:
        print(f"shipped {len(ship)}, launched (committed list) {len(got)}, argued exceptions {len(ok)}")
        print(f"shipped but never launched: {len(missing)}")
        for n in missing:
            print("  " + n)
        print(f"on the list but no longer shipped: {len(stale)}")
    return missing, stale


def run(pytest_args):
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", OUT, "-o", "k", "--",
           sys.executable, "-m", "pytest", "tests", "-q", "-m", "gpu", "-p", "no:cacheprovider", *pytest_args]
    print(" ".join(cmd), flush=True)
    rc = subprocess.run(cmd, cwd=ROOT, env=env).returncode
    print("pytest under rocprofv3 exited with", rc)
    merge()
    # (the traces are large: only the merged list travels back)
    for f in glob.glob(os.path.join(OUT, "**", "*.csv"), recursive=True):
        os.remove(f)
    return rc


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "diff"
    if what == "run":
        sys.exit(run(sys.argv[2:]))
    elif what == "merge":
        merge()
    else:
        missing, _ = diff()
        sys.exit(1 if missing else 0)
