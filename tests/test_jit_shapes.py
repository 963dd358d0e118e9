"""Shape specialisation at run time (include/mgx.h: MgxShapeKey; multigrid_amd/jit.py): for shapes the library has no built-in
instantiation for -- the reference registers 17 env ids (multigrid/envs/__init__.py:38-52), BASELINE.json names three shapes -- the
step kernel is compiled with hipRTC from the library's own headers for the launch geometry of (spec, batch) and registered; the
launches then run it.  Same source, same results: every step against the oracle and against the generic instantiation."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import multigrid_amd as mg
from multigrid_amd import _lib, jit
from multigrid_amd.batched import BatchedMultiGridEnv
from oracle import binding as ob
from tests import util


def test_the_translation_unit_names_its_geometry():
    """CPU: the key of a (spec, batch), and the source compiled for it."""
    spec = mg.spec_for("MultiGrid-Empty-8x8-v0", agents=2)
    key = jit.shape_key(spec, 4096)
    assert (key.width, key.height, key.num_agents, key.view_size, key.hooks, key.dma, key.built_in) == (8, 8, 2, 7, 0, 1, 0)
    assert key.envs_per_wavefront == _lib.launch_info(spec, 4096)["envs_per_wavefront"] and key.kernel_args_bytes > 300
    src = jit.source_for(key)
    assert f"#define MGX_JIT_SHAPE 8, 8, 2, {key.envs_per_wavefront}, false, 7, true, false" in src
    assert 'void mgx_jit_step(' in src and 'void mgx_jit_step_ar(' in src and "mgx_jit_kernel_args_bytes" in src
    c2 = jit.shape_key(mg.spec_for("MultiGrid-Empty-16x16-v0", agents=4), 4096)
    assert c2.built_in == 1                                         # BASELINE.json's shape: compiled into the library


@pytest.mark.skipif(not jit.hiprtc_available(), reason="no libhiprtc")
def test_hiprtc_compiles_the_kernel_without_a_gpu(tmp_path):
    """hipRTC cross-compiles (like hipcc): the code object has both kernels, no scratch, no spills."""
    import subprocess
    key = jit.shape_key(mg.spec_for("MultiGrid-RedBlueDoors-8x8-v0", agents=2), 4096)
    code = jit.compile_shape(key)
    p = tmp_path / "k.co"
    p.write_bytes(code)
    notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(p)], text=True)
    assert ".name:           mgx_jit_step\n" in notes and "mgx_jit_step_ar" in notes
    assert notes.count(".private_segment_fixed_size: 0") == 2 and notes.count(".vgpr_spill_count: 0") == 2


CASES = [("MultiGrid-Empty-8x8-v0", dict(agents=2), 4096), ("MultiGrid-RedBlueDoors-8x8-v0", dict(agents=2), 4096),
         ("MultiGrid-Empty-Random-6x6-v0", dict(agents=3), 2048), ("MultiGrid-LockedHallway-4Rooms-v0", dict(agents=2), 1000)]


@pytest.mark.gpu
@pytest.mark.skipif(not jit.hiprtc_available(), reason="no libhiprtc")
@pytest.mark.parametrize("env_id,kw,B", CASES, ids=[c[0] for c in CASES])
def test_runtime_compiled_shape_equals_oracle_and_generic_kernel(env_id, kw, B):
    dev = torch.device("cuda", 0)
    spec = mg.spec_for(env_id, **kw)
    st = util.random_state(spec, B, seed=21)
    ref = dict(grid=st["grid"].copy(), agents=st["agents"].copy(), rng=st["rng"].copy(), step_count=st["step_count"].copy(),
               aux=None if st["target"] is None else st["target"].copy())
    generic = BatchedMultiGridEnv(spec, B, dev)
    generic.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    outs = []
    acts = [torch.from_numpy(util.random_actions(B, spec.num_agents, seed=100 + t)).to(dev) for t in range(10)]
    for a in acts:                                                   # (before anything is registered: the generic instantiation)
        outs.append([x.clone() for x in generic.step(a)])
    assert generic.backend.launch_info(B)["fixed_shape"] == 0
    fast = BatchedMultiGridEnv(spec, B, dev, specialise=True)
    assert fast.shape_kernel in ("compiled", "registered")
    assert fast.backend.launch_info(B)["fixed_shape"] == 100         # MGX_SHAPE_RUNTIME_COMPILED
    fast.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    for t, a in enumerate(acts):
        got = fast.step(a)
        want = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], a.cpu().numpy(), ref["aux"], nthreads=8)
        for k, (g, w, o) in enumerate(zip(got, want, outs[t])):
            assert g.cpu().numpy().tobytes() == w.tobytes(), f"{env_id} step {t} output {k} vs oracle"
            assert torch.equal(g, o), f"{env_id} step {t} output {k} vs generic kernel"
    assert util.grid3(fast.cells).tobytes() == ref["grid"].tobytes() and torch.equal(fast.cells, generic.cells)
    assert torch.equal(fast.agents, generic.agents) and torch.equal(fast.rng, generic.rng)
    fast.check_errors()
    assert BatchedMultiGridEnv(spec, B, dev, specialise=True).shape_kernel == "registered"
    # another batch of the same spec in the throughput regime is left alone
    assert jit.ensure_shape(spec, 1 << 18, dev) in ("not-latency", "built-in")


@pytest.mark.gpu
@pytest.mark.skipif(not jit.hiprtc_available(), reason="no libhiprtc")
def test_a_code_object_built_for_another_library_is_refused():
    spec = mg.spec_for("MultiGrid-Empty-5x5-v0", agents=2)
    key = jit.shape_key(spec, 512)
    code = jit.code_object_for(key)
    bad = jit.MgxShapeKey.from_buffer_copy(key)
    bad.kernel_args_bytes += 8
    buf = C.create_string_buffer(code, len(code))
    assert jit._bind().mgx_shape_register(C.byref(bad), buf, len(code)) == _lib.ERR_INVALID_ARGUMENT
    junk = C.create_string_buffer(b"not an ELF" * 10, 100)
    assert jit._bind().mgx_shape_register(C.byref(key), junk, 100) == _lib.ERR_LAUNCH
    assert _lib.lib().mgx_last_hip_error() != 0                     # hipErrorInvalidImage, reported through the library ...
    # ... and NOT left behind as HIP's sticky per-thread error: torch reads that after its own launches, and the caller's next op
    # would raise "device kernel image is invalid" for a failure that was ours and had already been reported (round 4: the full
    # GPU suite found it -- this test, then a torch.zeros in another test)
    x = torch.zeros(1 << 16, dtype=torch.int16, device="cuda:0")
    x.add_(3)
    torch.cuda.synchronize()
    assert int(x.sum()) == 3 << 16
    env = BatchedMultiGridEnv(spec, 512, torch.device("cuda", 0))    # (what failed in the suite: an env made after the refusal)
    st = util.random_state(spec, 512, seed=2)
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    env.step(torch.from_numpy(util.random_actions(512, 2, seed=0)).to("cuda:0"))
    env.check_errors()
