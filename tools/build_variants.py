#!/usr/bin/env python3
"""Build experiment variants of the library into altlib/<name>.so:  python tools/build_variants.py name=DEF1,DEF2 ...
(run one with MGX_LIBMGX=altlib/<name>.so; bench.py marks such lines invalid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import build  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, "altlib"), exist_ok=True)
for arg in sys.argv[1:]:
    name, _, defs = arg.partition("=")
    out = os.path.join(ROOT, "altlib", name + ".so")
    build.build_lib(lib=out, defines=tuple(d for d in defs.split(",") if d))
    print("built", out)
