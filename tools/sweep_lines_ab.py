#!/usr/bin/env python3
"""Natural-order Gauss-Seidel trip (the reference's default sweep order) with the line-pipelined sweep (option sweep_lines 1,
csrc/its_lines.hip) against the level schedule (0).  usage: sweep_lines_ab.py <lap3d N | lap2d N> ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen, solver as S
TRIPS = int(os.environ.get("TRIPS", "20"))
argv = sys.argv[1:] or ["lap3d", "256"]
for kind, N in zip(argv[0::2], argv[1::2]):
    N = int(N)
    rp, col, val = {"lap3d": matgen.laplace3d_7pt, "lap2d": matgen.laplace2d_5pt}[kind](N)
    n = len(rp) - 1
    for lines in (1, 0):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_LEVEL)
        sol.handle.set_option("sweep_lines", 2 * lines)
        t0 = time.perf_counter()
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
        setup = time.perf_counter() - t0
        h = sol.handle
        for it in range(3): h.iterate(it)
        h.synchronize(); t0 = time.perf_counter()
        for it in range(3, 3 + TRIPS): r = h.iterate(it)
        h.synchronize()
        print("%s %d natural order, sweep_lines %d: %.4f ms/trip (set-up %.2f s, residual %.6e, %s)" % (kind, N, lines,
              (time.perf_counter() - t0) / TRIPS * 1e3, setup, r, h.sweep_info()), flush=True)
        h.close()
