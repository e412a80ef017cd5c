#!/usr/bin/env python3
"""Gauss-Seidel (multicolour) on the power-law matrix only: for kernel traces.  N rows from argv[1]."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen, solver as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
order = capi.GS_MULTICOLOR if (len(sys.argv) < 3 or sys.argv[2] == "mc") else capi.GS_LEVEL
rp, col, val = (matgen.laplace3d_7pt(int(os.environ["LAP"])) if os.environ.get("LAP") else
                matgen.convdiff3d_27pt(int(os.environ["CD27"])) if os.environ.get("CD27") else matgen.powerlaw(n))
n = len(rp) - 1
args = S.Args("gauss-seidel", S.LoopParams(max_iters=10**9, tol=0.0))
sol = S.Solver("gauss-seidel", device=0, gs_order=order)
sol.handle.set_option("fused_residual", int(os.environ.get("FUSED", "1")))
sol.handle.set_option("sweep_ell_levels", int(os.environ.get("LEVELL", "1")))
sol.handle.set_option("sweep_stream", int(os.environ.get("SWEEPSTREAM", "1")))
S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
h = sol.handle
for it in range(2): h.iterate(it)
h.synchronize(); t0 = time.perf_counter()
for it in range(2, 7): h.iterate(it)
h.synchronize(); print("ms/trip %.3f colours %d" % ((time.perf_counter() - t0) / 5 * 1e3, sol.n_colors), flush=True)
