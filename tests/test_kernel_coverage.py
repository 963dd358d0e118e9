"""Launched-vs-shipped: every gfx950 kernel symbol lib/libmgx.so carries must be on the committed list of kernels
that a run of `python -m pytest tests -m gpu` LAUNCHED on an MI355X (profiles/kernels_launched.txt, made by `tools/kernel_coverage.py
run` under rocprofv3 --kernel-trace), or on the short, argued list of exceptions (profiles/kernels_unlaunched_ok.txt).  CPU-only: the
symbol table is read from the library's code objects.  A new template flag that doubles part of the matrix fails here until the GPU
suite reaches the new instantiations (tests/test_instantiations.py walks the matrix) and the list is regenerated."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_coverage as kc  # noqa: E402


def test_every_shipped_kernel_was_launched_by_the_gpu_suite():
    from multigrid_amd import build
    build.build_lib()
    ship, got, ok = kc.shipped(), kc.read_list(kc.LAUNCHED), kc.read_list(kc.UNLAUNCHED_OK)
    assert len(got) > 300, "profiles/kernels_launched.txt is missing or truncated"
    missing = sorted(n for n in ship if n not in got and n not in ok)
    assert not missing, (f"{len(missing)} kernel(s) in libmgx.so were never launched by the GPU suite (add a case to "
                         f"tests/test_instantiations.py, re-run tools/kernel_coverage.py run, or drop the instantiation): {missing[:12]}")
    assert len(ok) <= 12, "the exceptions are meant to stay a short, argued list"
