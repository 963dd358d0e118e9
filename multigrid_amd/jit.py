"""Shape specialisation for ANY env shape, compiled at run time (include/mgx.h: MgxShapeKey).

libmgx.so carries shape-specialised instantiations of the step kernel for the shapes BASELINE.json names; the reference registers 17
env ids (multigrid/envs/__init__.py:38-52) and users define their own (`_gen_grid`), and in the latency regime -- a launch is a lone
wavefront's instruction chain -- a compile-time (W, H, A, envs per wavefront) is worth 10-14 % of a step.  `ensure_shape(spec,
batch)` asks the library for the launch geometry of (spec, batch) (mgx_shape_key), compiles csrc/mgx_fused.h for exactly that
geometry with hipRTC (the same headers the library was built from, `-DMGX_JIT_SHAPE=...`; ~2 s), caches the code object by content
hash under lib/jit/ and registers it (mgx_shape_register).  Launches of the plain step with that geometry then run it -- same
results bit for bit, it is the same source.  Without libhiprtc the generic kernel keeps running (returns "unavailable").

    BatchedMultiGridEnv(spec, batch, device, specialise=True)      # or MGX_JIT=1 in the environment, or env.specialise()
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

from . import _lib, build
from .spec import EnvSpec

CACHE_DIR = os.path.join(os.path.dirname(_lib.PRODUCT_LIB_PATH), "jit")
_SOURCES = ("mgx_fused.h", "mgx_fused_body.inc", "mgx_rules.h", "mgx_layout_gen.h")
_OPTIONS = ["--offload-arch=" + build.ARCH, "-O3", "-std=c++17", "-ffp-contract=off"]


class MgxShapeKey(C.Structure):
    """include/mgx.h: struct MgxShapeKey."""
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "num_agents", "envs_per_wavefront", "hooks", "view_size", "dma", "stream",
                                          "kernel_args_bytes", "built_in", "registered")]


def _bind():
    L = _lib.lib()
    L.mgx_shape_key.restype = C.c_int
    L.mgx_shape_key.argtypes = [C.POINTER(_lib.MgxSpecC), C.c_int64, C.POINTER(MgxShapeKey)]
    L.mgx_shape_register.restype = C.c_int
    L.mgx_shape_register.argtypes = [C.POINTER(MgxShapeKey), C.c_void_p, C.c_size_t]
    return L


def shape_key(spec: EnvSpec, batch: int) -> MgxShapeKey:
    key = MgxShapeKey()
    sc = spec.to_c()
    _lib.check(_bind().mgx_shape_key(C.byref(sc), batch, C.byref(key)), "mgx_shape_key")
    return key


def source_for(key: MgxShapeKey) -> str:
    """The translation unit: the library's own kernel body, instantiated for one launch geometry (csrc/mgx_fused.h: kShapes'
    MGX_JIT_SHAPE entry), as two extern "C" kernels -- the plain step without / with the fused auto-reset."""
    b = lambda v: "true" if v else "false"
    init = f"{key.width}, {key.height}, {key.num_agents}, {key.envs_per_wavefront}, {b(key.hooks)}, {key.view_size}, {b(key.dma)}, {b(key.stream)}"
    kern = """
extern "C" __global__ __launch_bounds__(kMaxThreads) void %s(const KernelArgs a) {
    constexpr int V = %d, MODE = 1, GRP = kGroup, SHAPE = kNumShapes - 1;
    constexpr bool HOOKS = %s, AR = %s, OH = false, GEN = false, STREAM = %s, DMA = %s, C8 = false, B3 = false;
#include "mgx_fused_body.inc"
}
"""
    return (f"#define MGX_JIT_SHAPE {init}\n#include \"mgx_fused.h\"\nnamespace mgx_fused {{\n"
            "extern \"C\" __device__ const int mgx_jit_kernel_args_bytes = (int)sizeof(KernelArgs);\n"
            + kern % ("mgx_jit_step", key.view_size, b(key.hooks), "false", b(key.stream), b(key.dma))
            + kern % ("mgx_jit_step_ar", key.view_size, b(key.hooks), "true", b(key.stream), b(key.dma))
            + "}  // namespace mgx_fused\n")


_TOOLCHAIN = None


def _toolchain_tag() -> str:
    """What identifies the compiler behind a cached code object: the ROCm tree's version file and libhiprtc's own file name
    (hiprtcVersion needs the library mapped, which this process avoids: see compile_shape)."""
    global _TOOLCHAIN
    if _TOOLCHAIN is None:
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        tag = [os.path.realpath(rocm)]
        for rel in (".info/version", "lib/libhiprtc.so"):
            p = os.path.join(rocm, rel)
            try:
                tag.append(open(p).read().strip() if rel.startswith(".info") else os.path.realpath(p))
            except OSError:
                tag.append("?")
        _TOOLCHAIN = "|".join(tag)
    return _TOOLCHAIN


def _content_hash(src: str) -> str:
    h = hashlib.sha256()
    h.update(src.encode())
    h.update(" ".join(_OPTIONS).encode())
    h.update(_toolchain_tag().encode())               # (a code object of another hipRTC / ROCm is not reused)
    for name in _SOURCES:
        with open(os.path.join(build.CSRC, name), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(build.ROOT, "include", "mgx.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:24]


def hiprtc_available() -> bool:
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if os.path.exists(os.path.join(rocm, "lib", "libhiprtc.so")):
        return True
    try:
        C.CDLL("libhiprtc.so")
        return True
    except OSError:
        return False


def compile_shape(key: MgxShapeKey) -> bytes:
    """hipRTC: source_for(key) -> gfx950 code object, in a process of its own (_jit_worker.py: this process has PyTorch's bundled
    comgr mapped, which may be older than the compiler the library was built with).  Raises RuntimeError with the compiler's log."""
    import subprocess
    import sys
    import tempfile
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    opts = _OPTIONS + [f"-I{build.CSRC}", f"-I{os.path.join(build.ROOT, 'include')}"]
    with tempfile.TemporaryDirectory() as tmp:
        src_path, out_path = os.path.join(tmp, "mgx_jit.hip"), os.path.join(tmp, "mgx_jit.co")
        with open(src_path, "w") as fh:
            fh.write(source_for(key))
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.path.join(rocm, "lib") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_jit_worker.py"), src_path, out_path,
                              *opts], capture_output=True, text=True, env=env, timeout=600)
        if out.returncode != 0:
            raise RuntimeError(f"hipRTC could not compile the shape kernel (exit {out.returncode}):\n{out.stderr[-4000:]}")
        with open(out_path, "rb") as fh:
            return fh.read()


def code_object_for(key: MgxShapeKey) -> bytes:
    """Cached by the content of everything it is compiled from."""
    path = os.path.join(CACHE_DIR, _content_hash(source_for(key)) + ".co")
    try:
        with open(path, "rb") as fh:
            return fh.read()
    except OSError:
        pass
    code = compile_shape(key)
    try:
        os.makedirs(CACHE_DIR, exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(code)
        os.replace(tmp, path)
    except OSError:
        pass                                     # (a read-only install: compile again next time)
    return code


def ensure_shape(spec: EnvSpec, batch: int, device=None, latency_only: bool = True) -> str:
    """Make the step of (spec, batch) on `device` (default: the current one) run a shape-specialised kernel.
    Returns "built-in" | "registered" (already) | "compiled" | "not-latency" (a throughput launch: gains ~1 %, left alone unless
    latency_only=False) | "unavailable" (no hipRTC: the generic kernel runs)."""
    import torch
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        key = shape_key(spec, batch)
        if key.built_in:
            return "built-in"
        if key.registered:
            return "registered"
        if latency_only and not key.dma:
            return "not-latency"
        if not hiprtc_available():
            return "unavailable"
        try:
            code = code_object_for(key)
        except (RuntimeError, OSError, subprocess.SubprocessError) as e:     # (a hipRTC that cannot compile the kernel, or one that
            # ran into the worker's timeout: the generic instantiation keeps running)
            import warnings
            warnings.warn(f"multigrid_amd.jit: {str(e)[:300]}")
            return "unavailable"
        buf = C.create_string_buffer(code, len(code))
        _lib.check(_bind().mgx_shape_register(C.byref(key), buf, len(code)), "mgx_shape_register")
        return "compiled"
