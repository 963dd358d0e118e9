"""The C-ABI library loads without a GPU and exports every symbol include/mgx.h declares; argument validation
(which happens before any launch) returns the documented codes.  No compute is attempted here."""
import ctypes as C
import os
import re

import pytest

from multigrid_amd import EnvSpec, _lib
from multigrid_amd.spec import MgxSpecC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgx_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported():
    names = declared_functions()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    L = _lib.lib()
    for n in names:
        assert getattr(L, n) is not None
    assert L.mgx_abi_version() == _lib.ABI_VERSION == 11
    assert _lib.error_string(0) == "ok" and "action" in _lib.error_string(-2)


def test_header_struct_matches_python_mirror():
    src = open(os.path.join(ROOT, "include", "mgx.h")).read()
    body = re.search(r"typedef struct MgxSpec \{(.*?)\} MgxSpec;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"int32_t\s+(\w+);", body)
    assert fields == [n for n, _ in MgxSpecC._fields_]
    assert C.sizeof(MgxSpecC) == 4 * len(fields)


def test_argument_validation_codes():
    L = _lib.lib()
    info = _lib.MgxLaunchInfo()
    ok = EnvSpec(16, 16, 4).to_c()
    assert L.mgx_launch_info(C.byref(ok), 4096, C.byref(info)) == _lib.OK
    assert info.threads_per_workgroup % 64 == 0 and info.workgroups > 0 and info.lds_bytes <= 160 * 1024
    assert info.envs_per_wavefront * ok.num_agents <= 64
    # the shape-specialised instantiations (ABI 7): the shapes BASELINE.json names, at the launch geometry of the latency regime
    assert info.fixed_shape == 1 and info.envs_per_wavefront == 4                    # Empty-16x16 x 4, 4096 envs (C2)
    assert L.mgx_launch_info(C.byref(ok), 16384, C.byref(info)) == _lib.OK and info.fixed_shape == 2 and info.envs_per_wavefront == 8
    assert L.mgx_launch_info(C.byref(ok), 65536, C.byref(info)) == _lib.OK and info.fixed_shape == 0   # throughput family
    bup = EnvSpec(11, 6, 2, env_kind="blockedunlockpickup", joint_reward=True).to_c()
    assert L.mgx_launch_info(C.byref(bup), 16384, C.byref(info)) == _lib.OK and info.fixed_shape == 3 and info.envs_per_wavefront == 8
    other = EnvSpec(15, 16, 4).to_c()
    assert L.mgx_launch_info(C.byref(other), 4096, C.byref(info)) == _lib.OK and info.fixed_shape == 0
    c5 = EnvSpec(64, 64, 16, 9, max_steps=16384).to_c()
    assert L.mgx_launch_info(C.byref(c5), 32768, C.byref(info)) == _lib.OK and info.fixed_shape == 4 and info.envs_per_wavefront == 1
    assert L.mgx_launch_info(C.byref(c5), 4096, C.byref(info)) == _lib.OK and info.fixed_shape == 0     # (fits the Infinity Cache: not streamed)
    v5 = EnvSpec(16, 16, 4, 5).to_c()
    assert L.mgx_launch_info(C.byref(v5), 4096, C.byref(info)) == _lib.OK and info.fixed_shape == 0
    bad = EnvSpec(16, 16, 4).to_c()
    bad.view_size = 6                                                   # even view: agent.py:78
    assert L.mgx_launch_info(C.byref(bad), 16, C.byref(info)) == _lib.ERR_INVALID_ARGUMENT
    bad.view_size = 17
    assert L.mgx_launch_info(C.byref(bad), 16, C.byref(info)) == _lib.ERR_UNSUPPORTED
    many = EnvSpec(16, 16, 4).to_c()
    many.num_agents = 33
    assert L.mgx_launch_info(C.byref(many), 16, C.byref(info)) == _lib.ERR_UNSUPPORTED
    # NULL tensors are rejected before anything is launched
    assert L.mgx_gen_obs(C.byref(ok), 8, None, None, None, None, None) == _lib.ERR_INVALID_ARGUMENT
    assert L.mgx_step(C.byref(ok), 8, *([None] * 13)) == _lib.ERR_INVALID_ARGUMENT
    assert L.mgx_step_autoreset(C.byref(ok), 8, None, *([None] * 13)) == _lib.ERR_INVALID_ARGUMENT
    ar = _lib.MgxAutoReset(0, 0, None, None, None, None, None)
    assert L.mgx_rollout_autoreset(C.byref(ok), 8, 4, C.byref(ar), *([None] * 13)) == _lib.ERR_INVALID_ARGUMENT
    # an empty batch is a no-op
    assert L.mgx_gen_obs(C.byref(ok), 0, None, None, None, None, None) == _lib.OK
    with pytest.raises(_lib.MgxError):
        _lib.check(_lib.ERR_UNSUPPORTED, "x")


def test_launch_geometry_for_baseline_configs():
    for spec, batch in [(EnvSpec(16, 16, 4, max_steps=1024), 4096), (EnvSpec(11, 6, 2, env_kind="blockedunlockpickup"), 16384),
                        (EnvSpec(16, 16, 4, max_steps=1024), 65536), (EnvSpec(64, 64, 16, view_size=9), 32768 // 8)]:
        li = _lib.launch_info(spec, batch)
        assert li["envs_per_wavefront"] >= 1
        assert li["workgroups"] * li["envs_per_workgroup"] >= batch


def test_ctypes_structs_match_the_header(tmp_path):
    """Every struct the Python binding declares (multigrid_amd/_lib.py) has the size and the field offsets the C compiler gives
    include/mgx.h: an ABI drift between the two would otherwise only show as wrong results on the GPU."""
    import shutil
    import subprocess
    import sys
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    structs = {"MgxSpec": _lib.MgxSpecC, "MgxLaunchInfo": _lib.MgxLaunchInfo, "MgxAutoReset": _lib.MgxAutoReset,
               "MgxGenStage": _lib.MgxGenStage, "MgxLayoutGen": _lib.MgxLayoutGen, "MgxStepArgs": _lib.MgxStepArgs,
               "MgxPersistent": _lib.MgxPersistent}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mgx.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu", sizeof({name}));')
        for f, _ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({name}, {f}));')
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([cc, "-std=c11", f"-I{os.path.join(root, 'include')}", str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    assert len(out) == len(structs)
    for line, (name, cls) in zip(out, structs.items()):
        got = line.split()
        assert got[0] == name
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert [int(x) for x in got[1:]] == want, f"{name}: header {got[1:]} vs ctypes {want}"
