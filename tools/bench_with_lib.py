#!/usr/bin/env python3
"""bench.py on an experimental build of the library (SS_TOOL_LIB=tools/_build/lib_<variant>.so; unset: the product library):
same-box A/B of a kernel change on the whole step.  usage: [SS_TOOL_LIB=...] python tools/bench_with_lib.py [bench.py args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semstereo_amd as sa  # noqa: E402

if os.environ.get("SS_TOOL_LIB"):
    sa._lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
