#!/usr/bin/env python3
"""Static instruction counts per phase of the fused kernel (profiling aid).
Compiles mgx_kernels.hip with -DMGX_MARKERS=1 to assembly and counts VALU / SALU / LDS / VMEM instructions between the
phase markers of one kernel instantiation.  Usage: python tools/isa_phase_count.py [V] [MODE] [HOOKS]   (default 7 1 0)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = sys.argv[1] if len(sys.argv) > 1 else "7"
MODE = sys.argv[2] if len(sys.argv) > 2 else "1"
HOOKS = sys.argv[3] if len(sys.argv) > 3 else "0"
extra = sys.argv[4:]
out = os.path.join(tempfile.gettempdir(), "mgx_markers.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DMGX_MARKERS=1",
                       f"-I{ROOT}/include", "-S", "--cuda-device-only", f"-DMGX_INST_V={V}", *extra,
                       f"{ROOT}/multigrid_amd/csrc/mgx_fused_inst.hip", "-o", out])
want = f"mgx_fused_kernelILi{V}ELi{MODE}ELb{HOOKS}E" + (os.environ.get("MGX_KERNEL_SUFFIX", ""))   # e.g. "Lb1ELb0ELb0ELb0EEEv": AR, OH, GEN, STREAM
cur = None
phase = "pre"
counts = collections.OrderedDict()
for line in open(out):
    t = line.strip()
    m = re.match(r"^(_Z\w+):", t)
    if m:
        cur = m.group(1); phase = "pre"; continue
    if cur is None or want not in cur:
        continue
    if t.startswith(".Lfunc_end"):
        cur = None; continue
    m = re.match(r"^; MGX_MARK (\w+)", t)
    if m:
        phase = m.group(1); continue
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
            else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    counts.setdefault(phase, collections.Counter())[kind] += 1
tot = collections.Counter()
print(f"{'phase':10s} {'valu':>6s} {'salu':>6s} {'lds':>6s} {'vmem':>6s}")
for ph, c in counts.items():
    print(f"{ph:10s} {c['valu']:6d} {c['salu']:6d} {c['lds']:6d} {c['vmem']:6d}")
    tot.update(c)
print(f"{'total':10s} {tot['valu']:6d} {tot['salu']:6d} {tot['lds']:6d} {tot['vmem']:6d}")
