#!/usr/bin/env python3
"""Upper bounds for PixArt-Sigma's non-GEMM families: the bench with one family's launches skipped (results are WRONG; timing
only).  PIXART_SKIP = colsum | none.  Diagnostic only."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yat_amd import ops

skip = os.environ.get("PIXART_SKIP", "none")
if skip == "colsum":
    ops.colsum = lambda *a, **k: None
sys.argv = [sys.argv[0]] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_pixart.py"), run_name="__main__")
