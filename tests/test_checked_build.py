"""Sanitizer-style debug builds:
* device: lib/libmgx_chk.so (-DMGX_BOUNDS_CHECK=1) asserts every computed LDS address of the fused kernel inside its
  wavefront's slice; the randomised soak (random specs / shapes / states, step + rollout + one-hot vs the oracle) must run
  on it with zero violations -- and, of course, still bit-exact;
* host: the integer rules (mgx_rules.h via the host shim) and the oracle run under AddressSanitizer + UBSan on CPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_soak_on_the_bounds_checked_build():
    from multigrid_amd import build
    lib = build.LIB_CHK
    assert os.path.exists(lib), "libmgx_chk.so is missing: __graft_entry__.build() makes it"
    env = dict(os.environ, MGX_LIBMGX=lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "12", "777"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert "bounds check: 0 LDS accesses" in out.stdout, out.stdout[-500:]


@pytest.mark.gpu
def test_generation_soak_on_the_checked_build():
    """The group-cooperative candidate generation (mgx_layout_gen.h: place_group) falls back to the serial place_obj loop when a lane's
    bounded draw would re-sample -- one draw in ~10^9 in the product.  In the checked build one draw in 32 claims it, so the fallback
    runs in about every tenth place_obj call: the staged generation must still equal the unstaged one, env for env, bit for bit."""
    from multigrid_amd import build
    env = dict(os.environ, MGX_LIBMGX=build.LIB_CHK)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_generate.py"), "15", "4242"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_checked_build_exports_the_counter_and_product_does_not():
    """No GPU needed: the symbol tables."""
    import ctypes
    from multigrid_amd import build
    chk = ctypes.CDLL(build.build_checked_lib())
    assert hasattr(chk, "mgx_debug_bounds_violations")
    prod = ctypes.CDLL(build.build_lib())
    for sym in ("mgx_debug_bounds_violations", "mgx_debug_skip_phases", "mgx_debug_set_envs_per_wavefront"):
        assert not hasattr(prod, sym), sym


def test_rules_and_oracle_under_asan_ubsan(tmp_path):
    """g++ -fsanitize=address,undefined build of the host shim (mgx_rules.h) and of the oracle, driven through a few of
    the rules-vs-oracle cases in a subprocess (the sanitizer runtime has to be preloaded into Python)."""
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    code = (
        "import numpy as np\n"
        "from tests import hostshim, util\n"
        "from oracle import binding as ob\n"
        "from multigrid_amd import EnvSpec\n"
        "for spec in (EnvSpec(16, 16, 4, 7, max_steps=1024), EnvSpec(9, 7, 3, 5, max_steps=50, allow_agent_overlap=False),\n"
        "             EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind='blockedunlockpickup'),\n"
        "             EnvSpec(30, 30, 16, 9, max_steps=60), EnvSpec(24, 24, 2, 15, max_steps=40)):\n"
        "    st = util.random_state(spec, 8, seed=3)\n"
        "    ref = {k: v.copy() for k, v in st.items()}\n"
        "    for t in range(6):\n"
        "        act = util.random_actions(8, spec.num_agents, seed=t)\n"
        "        want = ob.step_batch(spec.as_dict(), ref['grid'], ref['agents'], ref['rng'], ref['step_count'], act, ref['target'])\n"
        "        for b in range(8):\n"
        "            out = hostshim.step_env(spec, st['grid'][b], st['agents'][b], act[b], st['rng'][b], st['step_count'][b], st['target'][b], bool(t & 1))\n"
        "            st['step_count'][b] = out['step_count']\n"
        "            assert (out['obs'] == want[0][b]).all()\n"
        "print('sanitized ok')\n")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", MGX_SANITIZE="1",
               MGX_SANITIZE_DIR=str(tmp_path), PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0 and "sanitized ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
