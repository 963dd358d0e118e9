"""Build libmgx.so (the HIP kernels + C ABI) in-tree for gfx950.

`python -m multigrid_amd.build [--force] [--debug-knobs]` or `multigrid_amd.build.build_lib()`.  hipcc cross-compiles
without a GPU.  The .so stays in the source tree (multigrid_amd/lib/) so that it travels with the repo snapshot to the
GPU box and shows up as in-tree native code when loaded.

Staleness is decided by CONTENT: the SHA-256 of every source the library is compiled from plus the compile command is
stored beside the library (`<lib>.srchash`); a tree copied with fresh or stale mtimes neither rebuilds needlessly nor
silently reuses a library built from other sources.

`--debug-knobs` builds lib/libmgx_dbg.so with -DMGX_DEBUG_KNOBS=1: the profiling tools' variant that also exports
mgx_debug_skip_phases / mgx_debug_set_envs_per_wavefront / mgx_debug_set_waves_per_workgroup.  The product library
never has them; `MGX_LIBMGX=<path>` makes multigrid_amd load another build and bench.py then marks its line invalid.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
VIEWS = (3, 5, 7, 9, 11, 13, 15)
#: translation units: (object name, source, extra defines).  The fused kernel is instantiated in one unit per view size so
#: that the units compile in parallel (a single unit took a minute).
UNITS = ([("mgx_kernels", os.path.join(CSRC, "mgx_kernels.hip"), ()), ("mgx_aux", os.path.join(CSRC, "mgx_aux.hip"), ()),
          ("mgx_layout_gen", os.path.join(CSRC, "mgx_layout_gen.hip"), ())]
         + [(f"mgx_fused_v{v}", os.path.join(CSRC, "mgx_fused_inst.hip"), (f"MGX_INST_V={v}",)) for v in VIEWS])
SRCS = sorted({u[1] for u in UNITS})
DEPS = SRCS + [os.path.join(CSRC, "mgx_fused.h"), os.path.join(CSRC, "mgx_fused_body.inc"), os.path.join(CSRC, "mgx_layout_gen.h"), os.path.join(CSRC, "mgx_rules.h"),
               os.path.join(ROOT, "include", "mgx.h")]
LIB = os.path.join(HERE, "lib", "libmgx.so")
LIB_DBG = os.path.join(HERE, "lib", "libmgx_dbg.so")
LIB_CHK = os.path.join(HERE, "lib", "libmgx_chk.so")
LIB_TS = os.path.join(HERE, "lib", "libmgx_ts.so")
LIB_SPANS = os.path.join(HERE, "lib", "libmgx_spans.so")
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def flags(defines=()) -> list[str]:
    # (hipcc 7.2's greedy register allocator crashes at -O3 on the persistent kernels of the three largest views when the bounds
    # checks or the debug knobs are compiled in; those units of those builds are compiled at -O2, which does not -- so the checked
    # build carries EVERY instantiation the product ships.  The product library is -O3 throughout.)
    big_view = any(d.startswith("MGX_INST_V=") and int(d.split("=")[1]) >= 11 for d in defines)
    instrumented = any(d.split("=")[0] in ("MGX_BOUNDS_CHECK", "MGX_DEBUG_KNOBS") for d in defines)
    return [f"--offload-arch={ARCH}", "-O2" if big_view and instrumented else "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
            "-Wall", *[f"-D{d}" for d in defines]]


def source_hash(defines=()) -> str:
    h = hashlib.sha256()
    h.update(" ".join(flags(defines)).encode())
    for d in DEPS:
        h.update(os.path.relpath(d, ROOT).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stale(lib: str = LIB, defines=()) -> bool:
    if not os.path.exists(lib):
        return True
    try:
        with open(lib + ".srchash") as fh:
            return fh.read().strip() != source_hash(defines)
    except OSError:
        return True


def build_lib(force: bool = False, verbose: bool = False, lib: str = LIB, defines=()) -> str:
    if not force and not stale(lib, defines):
        return lib
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    objdir = os.path.join(ROOT, "build", os.path.basename(lib) + ".obj")
    os.makedirs(objdir, exist_ok=True)
    cc, inc = hipcc(), f"-I{os.path.join(ROOT, 'include')}"

    def compile_unit(unit):
        name, src, extra = unit
        obj = os.path.join(objdir, name + ".o")
        cmd = [cc, *flags(tuple(defines) + tuple(extra)), inc, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    # -DMGX_SINGLE_TU=1 (tools' builds with kernel-side globals, e.g. -DMGX_TIMESTAMPS=1): mgx_kernels.hip includes the
    # per-view instantiations itself
    units = [u for u in UNITS if not (u[0].startswith("mgx_fused_v") and "MGX_SINGLE_TU=1" in defines)]
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_unit, units))
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(lib + ".tmp", lib)
    with open(lib + ".srchash", "w") as fh:
        fh.write(source_hash(defines) + "\n")
    return lib


LIB_TORCH = os.path.join(HERE, "lib", "libmgx_torch.so")
TORCH_SRC = os.path.join(CSRC, "mgx_torch.cpp")


def _torch_build_cmd(out: str) -> list[str]:
    import torch
    from torch.utils import cpp_extension as ce
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", TORCH_SRC, "-o", out]
    cmd += [f"-L{p}" for p in ce.library_paths()] + ["-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip",
                                                      f"-L{os.path.dirname(LIB)}", "-lmgx", "-Wl,-rpath,$ORIGIN",
                                                      f"-L{rocm}/lib", "-lamdhip64"]
    cmd += [f"-Wl,-rpath,{p}" for p in ce.library_paths()]      # (a C++ host that dlopen()s the library finds torch's own)
    return cmd


def build_torch_lib(force: bool = False, verbose: bool = False) -> str:
    """lib/libmgx_torch.so: the compiled TORCH_LIBRARY(mgx) / TORCH_LIBRARY_IMPL(mgx, CUDA) operator library
    (csrc/mgx_torch.cpp: host code only, g++ against the installed torch's headers) over libmgx.so, which it finds beside
    itself ($ORIGIN).  Staleness by content: the source, include/mgx.h, the compile command and the torch version."""
    import torch
    build_lib()                                                    # (links against libmgx.so)
    h = hashlib.sha256()
    h.update((" ".join(_torch_build_cmd("out")) + torch.__version__).encode())
    for d in (TORCH_SRC, os.path.join(ROOT, "include", "mgx.h")):
        with open(d, "rb") as fh:
            h.update(fh.read())
    digest = h.hexdigest()
    try:
        with open(LIB_TORCH + ".srchash") as fh:
            fresh = os.path.exists(LIB_TORCH) and fh.read().strip() == digest
    except OSError:
        fresh = False
    if fresh and not force:
        return LIB_TORCH
    cmd = _torch_build_cmd(LIB_TORCH + ".tmp")
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB_TORCH + ".tmp", LIB_TORCH)
    with open(LIB_TORCH + ".srchash", "w") as fh:
        fh.write(digest + "\n")
    return LIB_TORCH


def build_debug_lib(force: bool = False, verbose: bool = False, extra_defines=()) -> str:
    return build_lib(force, verbose, LIB_DBG, ("MGX_DEBUG_KNOBS=1", *extra_defines))


def build_checked_lib(force: bool = False, verbose: bool = False) -> str:
    """lib/libmgx_chk.so: every computed LDS address of the fused kernel asserted inside its wavefront's slice
    (-DMGX_BOUNDS_CHECK=1); tests/test_checked_build.py runs the soak on it."""
    return build_lib(force, verbose, LIB_CHK, ("MGX_BOUNDS_CHECK=1",))


def build_timestamps_lib(force: bool = False, verbose: bool = False) -> str:
    """lib/libmgx_ts.so: -DMGX_TIMESTAMPS=1 (every wavefront records its begin / end in s_memrealtime ticks, one block of
    records per launch; one wavefront records the shader clock at every phase marker) + the debug knobs, for
    tools/span_probe.py, tools/stamp_probe.py and tools/chain_overlap.py.  Built on demand, never by build()."""
    return build_lib(force, verbose, LIB_TS, ("MGX_DEBUG_KNOBS=1", "MGX_TIMESTAMPS=1", "MGX_SINGLE_TU=1", "MGX_NO_BIG_PERSIST=1"))


def build_spans_lib(force: bool = False, verbose: bool = False, only_v: int = 7) -> str:
    """lib/libmgx_spans.so: the PRODUCT kernels + two s_memrealtime reads and two stores per wavefront (-DMGX_SPANS=1; no debug
    knobs, no phase stamps, no extra waits), for tools/chain_overlap.py: the launches' timelines as the kernels themselves saw
    them.  One view size only (a single translation unit: the records are a kernel-side global)."""
    return build_lib(force, verbose, LIB_SPANS, ("MGX_SPANS=1", "MGX_SINGLE_TU=1", f"MGX_ONLY_V={only_v}"))


if __name__ == "__main__":
    force, verbose = "--force" in sys.argv, True
    extra = tuple(a[2:] for a in sys.argv[1:] if a.startswith("-D"))
    if "--torch" in sys.argv:
        print(build_torch_lib(force, verbose))
    elif "--spans" in sys.argv:
        print(build_spans_lib(force, verbose))
    elif "--timestamps" in sys.argv:
        print(build_timestamps_lib(force, verbose))
    elif "--checked" in sys.argv:
        print(build_checked_lib(force, verbose))
    elif "--debug-knobs" in sys.argv:
        print(build_debug_lib(force, verbose, extra))
    else:
        print(build_lib(force, verbose))
