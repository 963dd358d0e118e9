#!/usr/bin/env python3
"""Register / scratch budget of every gfx950 kernel in a built library, from the code objects themselves.

    python tools/kernel_resources.py [lib.so] [--filter SUBSTR]

Carves the clang offload bundles out of the library's `.hip_fatbin` section, reads the AMDGPU metadata note of every gfx950 code
object (`llvm-readelf --notes`) and lists, per kernel: VGPRs, AGPRs, SGPRs, spilled SGPRs / VGPRs, scratch bytes
(`.private_segment_fixed_size`), static LDS.  `tests/test_kernel_budget.py` (CPU) asserts the budget on lib/libmgx.so.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys

LLVM_BIN = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path: str, arch: str = "gfx950") -> list[bytes]:
    """The `arch` ELF code objects bundled in lib_path's .hip_fatbin section."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        data = open(fat, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        p = m.start()
        (n,) = struct.unpack_from("<Q", data, p + 24)
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if arch in triple and size > 0:
                out.append(data[p + off:p + off + size])
    return out


def kernels_of(elf: bytes) -> list[dict]:
    """Per-kernel metadata of one code object (the amdhsa.kernels list of its NT_AMDGPU_METADATA note)."""
    import tempfile

    import yaml
    with tempfile.NamedTemporaryFile(suffix=".elf") as fh:
        fh.write(elf)
        fh.flush()
        txt = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", fh.name], text=True)
    lines = txt.splitlines()
    start = next(i for i, l in enumerate(lines) if l.strip() == "---")
    end = next((i for i in range(start + 1, len(lines)) if lines[i].strip() == "..."), len(lines))
    meta = yaml.safe_load("\n".join(lines[start + 1:end]))
    return [{k: v for k, v in kern.items() if k != ".args"} for kern in (meta.get("amdhsa.kernels") or [])]


def demangle(names: list[str]) -> list[str]:
    try:
        out = subprocess.run([os.path.join(LLVM_BIN, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True)
        res = out.stdout.splitlines()
        return res if len(res) == len(names) else names
    except OSError:
        return names


def library_kernels(lib_path: str) -> list[dict]:
    ks = []
    for elf in code_objects(lib_path):
        ks += kernels_of(elf)
    for k, d in zip(ks, demangle([k[".name"] for k in ks])):
        k["demangled"] = d
    return ks


def main():
    argv = sys.argv[1:]
    flt = ""
    if "--filter" in argv:
        i = argv.index("--filter")
        flt = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multigrid_amd", "lib", "libmgx.so")
    ks = [k for k in library_kernels(lib) if flt in k["demangled"]]
    print(f"# {lib}: {len(ks)} kernels")
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'sspill':>6} {'vspill':>6} {'scratch':>7} {'lds':>6}  kernel")
    for k in sorted(ks, key=lambda k: k["demangled"]):
        print(f"{k.get('.vgpr_count', 0):5d} {k.get('.agpr_count', 0):5d} {k.get('.sgpr_count', 0):5d} {k.get('.sgpr_spill_count', 0):6d} "
              f"{k.get('.vgpr_spill_count', 0):6d} {k.get('.private_segment_fixed_size', 0):7d} {k.get('.group_segment_fixed_size', 0):6d}  "
              f"{k['demangled'].replace('mgx_fused::', '').replace('void ', '')[:150]}")


if __name__ == "__main__":
    main()
